// Blocked right-looking Cholesky / static-pivot LDL^T on the device: the
// `factorize!` of the AbstractLinearSolver contract (reference
// src/LinearSolvers/lapack_common.jl:54-66), replacing `transfer_matrix!` +
// LAPACK dpotrf('L') / dsytrf('L') (reference src/LinearSolvers/lapack.jl:145-148,
// 164-167) and their rocSOLVER twins (reference
// lib/MadNLPGPU/ext/MadNLPGPUAMDGPUExt/rocsolver.jl:47-229).
//
// Structure (two-level blocking, everything in HBM, column-major, lower):
//   for each outer panel of NBO columns                      [panel stream, high priority]
//     for each inner block of NBI = 64 columns inside it
//        panel64_kernel  : every workgroup re-factors the 64x64 diagonal block in
//                          registers/LDS (redundantly) and carries two 64-row tiles of the
//                          panel below it through the same eliminations, which IS the
//                          triangular solve X = A L^-T (D^-1): no separate TRSM launch,
//                          no inverse on the critical path.
//        gemm_nt mode 2  : update the remaining columns of the outer panel (K = 64)
//   trailing update with K = NBO on the fp64 MFMA tile kernel  [update stream]
//        (a) the columns of the next outer panel first, then (b) the rest, so that the
//        next panel is factored (look-ahead) while (b) keeps the matrix cores busy.
//   linv64_kernel (batched, once): inv(L_jj) of every diagonal block for the solves.
// A device-side `info` word makes every later kernel a no-op once a pivot fails
// (LAPACK stops at the failing column; we cannot stop the host without a sync).
#include <atomic>
#include <type_traits>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdlib>

#include "ls.h"
#include "leaf64.h"

namespace mnk {

// ---------------------------------------------------------------------------------------
// Split panel step (the default): potrf64_kernel + trsm64_mfma_kernel replace the fused elimination
// kernel below.  The fused kernel carries every 64-row tile of the panel through the 64 sequential
// pivots of the diagonal block on the VALU (23-40 us per block column, 0.8 TFLOP/s); here ONE
// workgroup factors the 64x64 diagonal block (the unavoidable pivot chain) and also inverts its four
// 16x16 diagonal sub-blocks, and the triangular solve of all rows below runs as block substitution
// on the fp64 matrix cores: X^T[cb] = inv(L[cb,cb]) (B^T[cb] - sum_{ib<cb} L[cb,ib] X^T[ib]), every
// product a chain of v_mfma_f64_16x16x4 on a 16-row strip that never leaves the wave's registers.
// (Block substitution with 16x16 inverses: backward error indistinguishable from row-by-row
// substitution on the condensed KKT systems -- tools/emul_block_trsm.py, 2.1e-16 either way.)
// ---------------------------------------------------------------------------------------
// X = B L_jj^-T (Cholesky) / V = B L_jj^-T, X = V D^-1 (LDL) for every row below the diagonal block.
// Wave = NS strips of 16 rows; lane (l15, l4): accumulator register r of column block cb holds
// X[row0 + l15][16 cb + l4 + 4 r] (the C^T layout of gemm_f64.hip, so register r of X^T[ib] IS the
// B operand of k-step r in the next product and nothing is ever shuffled).
template <bool LDL, int NS>
__device__ __forceinline__ void trsm64_mfma_body(double* __restrict__ F, int64_t ld, int64_t j0, int64_t Np,
                                                 const double* __restrict__ Dblk, const double* __restrict__ inv16,
                                                 const double* __restrict__ dinv, double* __restrict__ W, int64_t ldw,
                                                 int64_t wcol, const int* __restrict__ info,
                                                 unsigned long long* __restrict__ vmax) {
    if (*info != 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t r0 = j0 + 64 + ((int64_t)blockIdx.x * 4 + wave) * (16 * NS);
    if (r0 >= Np) return;
    // A operands: lane (l15, l4) of k-step s holds M[i = l15][k = 4 s + l4]
    double Ln[6][4];  // -L_jj[cb, ib] for (cb, ib) = (1,0) (2,0) (2,1) (3,0) (3,1) (3,2)
    double Iv[4][4];  // inv(L_jj[cb, cb])
    {
        int p = 0;
#pragma unroll
        for (int cb = 1; cb < 4; ++cb)
#pragma unroll
            for (int ib = 0; ib < cb; ++ib, ++p)
#pragma unroll
                for (int s = 0; s < 4; ++s) Ln[p][s] = -Dblk[(16 * cb + l15) + 64 * (16 * ib + 4 * s + l4)];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int s = 0; s < 4; ++s) Iv[cb][s] = inv16[cb * 256 + l15 + 16 * (4 * s + l4)];
    }
    v4d X[NS][4];
#pragma unroll
    for (int ns = 0; ns < NS; ++ns)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) X[ns][cb][r] = F[(r0 + 16 * ns + l15) + (j0 + 16 * cb + l4 + 4 * r) * ld];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) {
            v4d t = X[ns][cb];
#pragma unroll
            for (int ib = 0; ib < cb; ++ib) {
                const int p = cb * (cb - 1) / 2 + ib;
#pragma unroll
                for (int s = 0; s < 4; ++s) t = __builtin_amdgcn_mfma_f64_16x16x4f64(Ln[p][s], X[ns][ib][s], t, 0, 0, 0);
            }
            v4d x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; ++s) x = __builtin_amdgcn_mfma_f64_16x16x4f64(Iv[cb][s], t[s], x, 0, 0, 0);
            X[ns][cb] = x;
        }
    }
    double vm = 0.0;
#pragma unroll
    for (int ns = 0; ns < NS; ++ns)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * cb + l4 + 4 * r;
                const int64_t row = r0 + 16 * ns + l15;
                if (LDL) {
                    W[row + (wcol + c) * ldw] = X[ns][cb][r];
                    F[row + (j0 + c) * ld] = X[ns][cb][r] * dinv[j0 + c];
                    vm = fmax(vm, fabs(X[ns][cb][r]));
                } else {
                    F[row + (j0 + c) * ld] = X[ns][cb][r];
                }
            }
    if (LDL) growth_fold(vmax, vm);
}

template <bool LDL, int NS>
__global__ __launch_bounds__(256) void trsm64_mfma_kernel(double* __restrict__ F, int64_t ld, int64_t j0, int64_t Np,
                                                           const double* __restrict__ Dblk,
                                                           const double* __restrict__ inv16,
                                                           const double* __restrict__ dinv, double* __restrict__ W,
                                                           int64_t ldw, int64_t wcol, int* __restrict__ info,
                                                           unsigned long long* __restrict__ vmax = nullptr) {
    trsm64_mfma_body<LDL, NS>(F, ld, j0, Np, Dblk, inv16, dinv, W, ldw, wcol, info, vmax);
}

// The same block substitution for the rows of SEVERAL independent right-hand-side blocks against their own factors in one
// launch (blockIdx.y: which one; the Schur stage's per-scenario sweeps, schur.hip): rows X_i (nrows x ., leading dimension
// ldr) <- X_i[:, j0 : j0 + 64] L_i,jj^-T (LDL: V_i gets that, X_i gets it times D^-1).
template <bool LDL>
__global__ __launch_bounds__(256) void trsm64_mfma_batch_kernel(const TrsmBatchRec* __restrict__ recs, int64_t ldr, int64_t j0,
                                                                 int64_t nrows) {
    const TrsmBatchRec rec = recs[blockIdx.y];
    // (the body addresses rows j0 + 64 ... of ONE matrix: shift the row blocks so that its first such row is their row 0)
    trsm64_mfma_body<LDL, 1>(rec.X - (j0 + NBI), ldr, j0, j0 + NBI + nrows, rec.dblk + (j0 / NBI) * 4096,
                             rec.inv16 + (j0 / NBI) * 1024, rec.dinv, rec.V != nullptr ? rec.V - (j0 + NBI) : nullptr, ldr, j0,
                             rec.info, nullptr);
}

#ifndef MNK_LEAF_V
#define MNK_LEAF_V 1   // the pivot leaf: 1 = potrf64v_core (round 6: indicator sums, pivot tests behind the chain, 4x4x4 block solves, third-order
                       // reciprocals), 0 = potrf64w_core (rounds 2-5) -- A/B builds (tools/leaf_ab.sh)
#endif
template <bool LDL>
__device__ __forceinline__ void potrf64w_body(const double* __restrict__ F, int64_t ld, int64_t j0,
                                              double* __restrict__ Dout, double* __restrict__ inv16,
                                              double* __restrict__ dvec, double* __restrict__ dinv,
                                              int* __restrict__ info, double pivot_tol, double* Lsh, double* Ish,
                                              unsigned long long* __restrict__ vmax = nullptr) {
    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, l4 = lane >> 4;
    v4d Lt[4][4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int b = 0; b <= cb; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double v = F[(j0 + 16 * cb + l15) + (j0 + 16 * b + l4 + 4 * r) * ld];
                // 'L' storage: the strict upper triangle of the block may hold anything (NaN included)
                Lt[cb][b][r] = (cb == b && l15 < l4 + 4 * r) ? 0.0 : v;
            }
#if MNK_LEAF_V
    potrf64v_core<LDL>(Lt, j0, Dout, inv16, dvec, dinv, info, pivot_tol, Lsh, Ish, vmax);
#else
    potrf64w_core<LDL>(Lt, j0, Dout, inv16, dvec, dinv, info, pivot_tol, Lsh, Ish, vmax);
#endif
}

template <bool LDL>
__global__ __launch_bounds__(64) void potrf64w_kernel(const double* __restrict__ F, int64_t ld, int64_t j0,
                                                       double* __restrict__ Dout, double* __restrict__ inv16,
                                                       double* __restrict__ dvec, double* __restrict__ dinv,
                                                       int* __restrict__ info, double pivot_tol,
                                                       unsigned long long* __restrict__ vmax) {
    if (*info != 0) return;
    potrf64w_body<LDL>(F, ld, j0, Dout, inv16, dvec, dinv, info, pivot_tol, nullptr, nullptr, vmax);
}

// ---------------------------------------------------------------------------------------
// Persistent panel kernel (panel_algo = 4): ONE launch factors nb <= NB 64-column blocks of a panel, every row of it.
// Workgroup t owns the 64-row strip t of the panel (wave w: 16 rows, its nb x 4 column blocks of 16 stay in
// registers, C^T layout as in trsm64_mfma_kernel).  The strips 0..nb-1 hold the diagonal blocks.  Right-looking, per
// column block j:   strip j: wave 0 factors the diagonal block (potrf64w_core) and publishes it;
//                   strip t > j: waits for it, X = T L_jj^-T, stores V / L, then T[t, c] -= V[t, j] L[c, j]^T for the
//                   later column blocks c <= t, where L[c, j] is what strip c published in ITS step j.
// prog[c] = 16 * epoch + (number of column blocks strip c has completed and published); release/acquire at agent scope
// (producer: stores -> release fence -> drained flag store; consumer: relaxed polls, one acquire).  Only the nb diagonal strips are ever waited for, and a diagonal strip only waits for lower-numbered
// ones, so the kernel needs no co-residency beyond "lower block ids are dispatched no later than higher ones"; every
// wait is bounded (info = -7 instead of a hang).  Strips that are dispatched late find every flag set.
// The critical chain per 64 columns is potrf (one wave) -> flag hop -> strip j+1: triangular solve of 64 rows, K = 64
// update of its diagonal block from LDS, exchange to wave 0 -> potrf: no kernel boundaries and no idle launches.
// ---------------------------------------------------------------------------------------
constexpr long PP_SPIN_LIMIT = 1L << 20;  // ~0.5 s
#ifndef MNK_LEAF_WAVES
#define MNK_LEAF_WAVES 1   // waves that factor a 64x64 diagonal block of the pivot chain.  1: potrf64w_core on wave 0 (the shipped leaf).  4: potrf64q_core,
                           // every wave its own block row -- built and measured in round 5: the same bits, 0.89 vs 0.86 ms at N = 2048 (tools/leaf_ab.sh,
                           // profiles/r05_leaf_lab.txt): the scalar 4x4 factorizations, 57 % of the leaf, are repeated by every wave and the
                           // exchanges through LDS cost what the split of the MFMAs saves.  Diagnostic builds only.
#endif
#ifndef MNK_DIAG_RACY_PUB
#define MNK_DIAG_RACY_PUB 0
#endif
#ifndef MNK_DIAG_STEP_TRACE
#define MNK_DIAG_STEP_TRACE 0
#endif
#ifndef MNK_DIAG_NO_EARLY
#define MNK_DIAG_NO_EARLY 0   // (-DMNK_DIAG_NO_EARLY=1: a diagnostic build without the chain's early diagonal update)
#endif
#ifndef MNK_TILE_DMA
#define MNK_TILE_DMA 1   // (0: the staging tiles of the chain strips go through registers, rounds 2-5)
#endif
// A staging tile that arrives by LDS-DMA lies k-column-major: request g' = 2 g + odd of 32 (one global_load_lds_dwordx4 = 64 lanes x 16
// bytes = rows 0..63 of the k-columns 4 g + odd and 4 g + odd + 2) at byte 2176 g + 1152 odd -- the two k-columns a ds_read_b64 lane group
// (lanes 0-31: l4 = 0, 1; lanes 32-63: l4 = 2, 3) reads are 1152 = 128 (mod 256) bytes apart: different bank halves.
constexpr int KT_G = 2176, KT_ODD = 1152, KT_BYTES = 16 * KT_G;
constexpr int KT_NULL = 2 * KT_BYTES;   // 1 KB per wave behind the staging tiles: where a step that has no next tile sends its requests (no branch in the loop)
constexpr int PP_STAGE_BYTES = MNK_TILE_DMA ? 2 * KT_BYTES + 4096 : 2 * 4096 * 8;
constexpr int PP_LDS_BYTES = PP_STAGE_BYTES + 4096 * 8;  // two staging tiles (the first doubles as the exchange buffer) + own tile
constexpr int PC_LDS_BYTES = PP_STAGE_BYTES + 2 * 4096 * 8;  // pivot-chain kernels: + the strip's NEXT diagonal block (see pp_strip, EARLY)

// acc[cb2] += sum over ib, s of tile[(4 cb2 + ib) * 64 + lane][s] * B[ib][s] for the 16x16 blocks cb2 < ncb2 (wave-uniform) of one 64x64 tile
// in LDS (the chain strips' tile step: 64 products per wave).  The callers negate the B operand once per block column instead of
// every A operand on its way from LDS (round 5: two v_xor per LDS read), and the read of block q + 1 is requested before the products
// of block q.  Measured in the schedule (tools/chain_steps2.py, profiles/r06_chain_steps.txt): 1.79 us per tile step against the matrix
// pipe's 1.71; the same loop as ONE inline-assembly block with every A operand requested six products ahead (ds_read_b64 into a ring
// of eight registers) was 1.85 -- the LDS latency is not what a tile step waits for; its other 0.7 us are the 16 global loads of the
// next tile (0.38), the arrival of this one + its way into LDS (0.26) and the barrier (0.05).
template <bool FULL>
__device__ __forceinline__ void tile_mac(const v4d* __restrict__ tile, const int lane, const int ncb2, const v4d* __restrict__ B, v4d* __restrict__ acc) {
    v4d a = tile[lane];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int cb2 = q >> 2, ib = q & 3;
        if (!FULL && ib == 0 && cb2 >= ncb2) break;
        v4d an = a;
        if (q < 15) an = tile[(q + 1) * 64 + lane];
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[cb2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], B[ib][s], acc[cb2], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one LDS read ...
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // ... then this block's four products
        a = an;
    }
}
// (a diagonal strip's own block: lower triangle only -- the 16x16 blocks cb2 <= w)
__device__ __forceinline__ void tile_mac(const v4d* __restrict__ tile, const int lane, const bool own_block, const int w, const v4d* __restrict__ B, v4d* __restrict__ acc) {
    if (own_block) tile_mac<false>(tile, lane, w + 1, B, acc);
    else tile_mac<true>(tile, lane, 4, B, acc);
}
// The same sum (same products, same order, same bits) over a staging tile in the LDS-DMA layout; `next(q)`, q = 0..7, issues this wave's
// q-th request of the NEXT tile -- spread over the products, so that the texture addresser works under the matrix pipe and the
// requests cost the step nothing (tools/hip/tilestep_lab.hip, profiles/r06_tilestep_lab.txt, one per two blocks of four products: 2.42 ->
// 1.96 us per tile step, the loop's floor without any global load being 1.95; all eight requests in front of the products: 2.22).
// Shipped: one per block over the FIRST HALF of the step -- the last request then has ~1 us of products behind it instead of 0.25, and
// the wait at the next step's barrier is that much shorter (in the chain: N = 2048 0.778 -> 0.757 ms, 4096 1.532 -> 1.488; two per
// block over the first quarter: the same).
template <bool FULL, class Next>
__device__ __forceinline__ void tile_mac_k(const char* __restrict__ tile, const int l15, const int l4, const int ncb2, const v4d* __restrict__ B,
                                           v4d* __restrict__ acc, Next&& next) {
    const char* tb = tile + 8 * l15 + KT_ODD * (l4 & 1) + 512 * (l4 >> 1);
    double a[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) a[s] = *reinterpret_cast<const double*>(tb + KT_G * s);
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // (the first block's reads; from here on: the next block's reads, this block's products)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int cb2 = q >> 2, ib = q & 3;
        if (!FULL && ib == 0 && cb2 >= ncb2) break;
        double an[4] = {a[0], a[1], a[2], a[3]};
        if (q < 15) {
#pragma unroll
            for (int s = 0; s < 4; ++s) an[s] = *reinterpret_cast<const double*>(tb + KT_G * (4 * ((q + 1) & 3) + s) + 128 * ((q + 1) >> 2));
        }
        if (q < 8) next(q);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[cb2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], B[ib][s], acc[cb2], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // the next block's four LDS reads ...
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // ... then this block's four products
#pragma unroll
        for (int s = 0; s < 4; ++s) a[s] = an[s];
    }
    if (!FULL) {
#pragma unroll
        for (int q2 = 0; q2 < 8; ++q2)
            if (q2 >= 4 * ncb2) next(q2);
    }
}
__device__ __forceinline__ void pin(v4d& x) {   // (the value as it is, in registers: a negation is not to be re-done at every use)
    asm volatile("" : "+v"(x));
}

// Wave-uniform bounded wait for prog[c] >= target.  `seen` caches the last values read: one acquire covers everything
// that was published before ANY flag value read ahead of it, so every successful wait refreshes all nb entries and
// later waits that are already satisfied cost nothing (no further cache invalidation).
template <int NB, bool DBG = false>
__device__ __forceinline__ void pp_wait(const int* prog, int c, int nb, int target, int (&seen)[NB], int* info, long limit = PP_SPIN_LIMIT,
                                        int* dbg = nullptr, int dbg_js = 0, int dbg_t = 0) {
    if (seen[c] >= target) return;
    long spins = 0;
    while (__hip_atomic_load(prog + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 1023) == 0) {
            if (DBG && dbg != nullptr && threadIdx.x == 0 && (spins & 0xfffff) == 0) {   // (a long wait: say what for)
                dbg[0] = 4; dbg[1] = dbg_js; dbg[2] = dbg_t; dbg[3] = c; dbg[4] = target;
                dbg[5] = __hip_atomic_load(prog + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); dbg[6] = (int)(spins >> 20);
            }
            // a failed factorization (or a dependency that never arrives) must not hang: carry on with whatever is
            // there -- info != 0 makes every result of this factorization void
            if (__hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
            if (spins > limit) {
                if (atomicCAS(info, 0, -7) == 0) info[1] = 4;   // (site 4: a strip waiting for a diagonal block of the chain)
                break;
            }
        }
    }
    int v[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) v[i] = i < nb ? __hip_atomic_load(prog + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#pragma unroll
    for (int i = 0; i < NB; ++i) seen[i] = v[i];
}


// One strip t of one persistent panel step (the body of ppanel_kernel / pchain_kernel): every thread of the workgroup calls
// it with the same arguments; waves may return at different times (callers that go on synchronize first).
//
// EARLY (the pivot-chain kernels, whose workgroups keep their ROWS from one strip-column to the next: strip t + 4 of Js is
// strip t of Js + 1): a strip 4..7 is a diagonal strip of the next strip-column, and the 256 columns it finishes here are
// exactly what its next diagonal block is still missing.  Instead of publishing them, having the next strip-column's
// workgroup wait for the flag, read them back and apply them in its prologue (6 us of hop + 14 us of K = 256 product in
// front of EVERY strip-column's first diagonal block: 20 of the chain's 95 us per 256 columns), the strip sums the four
// products Z = - sum_j V_j L_j^T for its own next diagonal block right after each substitution -- the operands are in
// registers (V) and in LDS (`own`: L) at that moment -- in an accumulator that starts from zero (`xn`, LDS: the band tile
// T' the bulk kernel accumulates over the older columns is not final yet at that time), and the block becomes T' + Z:
// at the start of the last step if T' is final by then (loaded while the strip waits for the last diagonal block; state
// 2), else in the next strip-column behind the usual wait for the band tile (state 1).  The next strip-column's strip 0
// then starts with its diagonal step, behind the same hand-over as any other block of the chain.  ALWAYS taken by the
// strips 4..7 (the sum order T' + (((0 - c0) - c1) - c2) - c3 differs from the prologue's ((T' - c0) - c1) ...: one
// path per strip, the same bits in every run, and both T' + Z paths add the same two numbers).
template <bool LDL, int NB, bool WT = false, bool EARLY = false>
__device__ __forceinline__ void pp_strip(const int t, double* __restrict__ F, int64_t ld, int64_t p0, int nb, int64_t Np,
                                         double* __restrict__ dblk0, double* __restrict__ inv0, double* __restrict__ dvec,
                                         double* __restrict__ dinv, double* __restrict__ W, int64_t ldw, int64_t wcol0,
                                         int* __restrict__ info, double pivot_tol, int* __restrict__ prog, int epoch16,
                                         int dbg_missing, const double* __restrict__ Vp, int64_t ldv, int Kp, PpDag dag,
                                         char* pp_smem, int* s_go, int* xn_have = nullptr) {
    v4d* stage = reinterpret_cast<v4d*>(pp_smem);          // [2][1024] v4d
    v4d* own = reinterpret_cast<v4d*>(pp_smem + PP_STAGE_BYTES);  // [1024] v4d
    v4d* xn = own + 1024;   // [1024] v4d (EARLY only: PC_LDS_BYTES)
    const int tid = threadIdx.x;
    const int64_t R = p0 + 64 * (int64_t)t;
    // this strip's own diagonal block was brought up to date by the previous strip-column's steps
    const int xn_state = EARLY && xn_have != nullptr ? *xn_have : 0;   // (rewritten behind the barrier below)
    const bool use_xn = xn_state != 0;
    // ... and whether this strip does the same for the next strip-column
    const bool make_xn = EARLY && !MNK_DIAG_NO_EARLY && xn_have != nullptr && dag.front != nullptr && t >= 4 && t < 8 && nb == 4 && R < Np && t != dbg_missing;
    if (tid == 0) *s_go = (__hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) ? 1 : 0;
    __syncthreads();
    const int go_bits = *s_go;
    if (EARLY && xn_have != nullptr && tid == 0) *xn_have = make_xn ? 1 : 0;   // (read again only behind the caller's barrier)
    if (!(go_bits & 1)) return;
    const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    if (R >= Np || t == dbg_missing) return;
    const int64_t r0 = R + 16 * w;
    const int jmax = t < nb - 1 ? t : nb - 1;
    const bool diag_strip = t < nb;
    // Persistent chain: a strip may wait for a diagonal block whose strip is itself waiting for the BULK kernel -- with the
    // schedule's own (long) bound.  The short bound of the launch-per-panel schedule expired there when the bulk kernel's
    // very first launch in a process took longer than ~0.5 s to start (code upload): info = -7, and the solver stayed on
    // schedule 1 for good -- 12.5 instead of 9.3 ms per factorize!, seen in 2 of ~150 bench processes.
    const long pp_limit = dag.front != nullptr ? dag.spin_limit : PP_SPIN_LIMIT;
    int seen[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) seen[c] = 0;
    const int tabs = (int)(p0 >> 6) + t;  // this strip's 64-row block
    unsigned long long* ptr_tr = dag.trace != nullptr && tid == 0 ? dag.trace + 8 * t : nullptr;
    if (ptr_tr) ptr_tr[0] = wall_clock64();
#if MNK_DIAG_STEP_TRACE   // (diagnostic build: 16 more stamps per strip -- per step: block seen / substitution done / rows published / updates done;
                          // second half of the chain's trace region, single-phase schedules of <= 64 strip-columns; tools/chain_steps2.py)
    unsigned long long* tr2 = ptr_tr != nullptr && (p0 >> 8) < 64 ? ptr_tr + 2048 * 8 + ((p0 >> 8) * 8 * (int64_t)gridDim.x + 8 * t) : nullptr;
#define MNK_TR2(slot) do { if (tr2) tr2[slot] = wall_clock64(); } while (0)
    // ... and eight sums over the prologue's tile steps, in shader cycles: {steps, wait for the tile's loads + LDS write, barrier, issue of the
    // next loads, the 64 products, total}
    unsigned long long* tr3 = ptr_tr != nullptr && (p0 >> 8) < 64 ? ptr_tr + 3072 * 8 : nullptr;
    unsigned long long pr_n = 0, pr_vm = 0, pr_bar = 0, pr_ld = 0, pr_mac = 0;
#define MNK_PCLK() __builtin_readcyclecounter()
#else
#define MNK_TR2(slot) do { } while (0)
#endif
    if (EARLY && make_xn) {
#pragma unroll
        for (int cb2 = 0; cb2 < 4; ++cb2)
            if (cb2 <= w) xn[(w * 4 + cb2) * 64 + lane] = v4d{0.0, 0.0, 0.0, 0.0};   // (wave w: its 16 rows, column blocks cb2 <= w)
    }
    // (strip 0 with its diagonal block complete depends on nothing but that block)
    if (dag.front != nullptr && !(xn_state == 2 && t == 0) && (dag.af_tilecol >= 0 || (dag.need_front > 0 && t >= dag.front_from))) {
        // the strip's tiles as the bulk kernel leaves them (accumulated over the older columns), and -- rows below the
        // previous launch's band -- its rows of the previous strip-column, finalized by the bulk kernel
        if (tid == 0) {
            auto wait_ge = [&](const int* word, int target) {
                long spins = 0;
                while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    __builtin_amdgcn_s_sleep(2);
                    if ((++spins & 255) == 0) {
                        if (LDL && dag.dbg != nullptr && (spins & 0xfffff) == 0) {   // (a long wait: say what for; the LDL^T build only --
                                                                                     // the Cholesky kernel has no register to spare)
                            int* d8 = dag.dbg + 8 * t;
                            const bool is_front = word < dag.af;   // (the words are laid out front | af | tprog)
                            d8[0] = 5; d8[1] = (int)(p0 >> 8); d8[2] = t;
                            d8[3] = is_front ? (int)(word - dag.front) : -1 - (int)(word - dag.af);
                            d8[4] = target; d8[5] = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); d8[6] = (int)(spins >> 20);
                            d8[7] = (int)((__builtin_amdgcn_s_getreg(63492) & 0xffffu) | ((__builtin_amdgcn_s_getreg(63508) & 15u) << 16));   // (which CU: HW_ID, XCC_ID)
                        }
                        if (__hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                        if (spins > dag.spin_limit) {
                            if (atomicCAS(info, 0, -7) == 0) info[1] = 5;   // (site 5: a chain strip waiting for the bulk kernel: rows / band tiles)
                            break;
                        }
                    }
                }
            };
            if (dag.need_front > 0 && t >= dag.front_from) wait_ge(dag.front + tabs, dag.need_front);
            // persistent chain: the prologue also reads the rows of this strip-column's diagonal strips in the previous
            // strip-column (strips 4..7 there; nothing but these counters orders the strip-columns)
            if (dag.need_front > 0 && dag.front_from == 0)
                for (int c = 0; c < nb && c <= t; ++c) wait_ge(dag.front + (int)(p0 >> 6) + c, dag.need_front);
            if (dag.trace != nullptr) dag.trace[8 * t + 5] = wall_clock64();
            if (dag.af_tilecol >= 0) {
                wait_ge(dag.af + (int64_t)(tabs >> 1) * dag.ntile + dag.af_tilecol, 1);
                if (t >= 2 && nb > 2) wait_ge(dag.af + (int64_t)(tabs >> 1) * dag.ntile + dag.af_tilecol + 1, 1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    if (ptr_tr) ptr_tr[1] = wall_clock64();

    v4d X[4 * NB];
#pragma unroll
    for (int g = 0; g < 4 * NB; ++g)
        if (g < 4 * (jmax + 1)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) X[g][r] = F[(r0 + l15) + (p0 + 16 * g + l4 + 4 * r) * ld];
        }

    // ---- optional left-looking prologue: apply the Kp columns just before this launch (the previous outer panel, or
    // the previous launch of this panel) to the strip, instead of a separate update kernel in front of the launch:
    // T[t, c] -= V[t, p0-Kp : p0] L[p0 + 64 c .., p0-Kp : p0]^T.  Used where the panel chain is the critical path (few
    // rows left): the first diagonal block is ready after Kp/64 K = 64 products of one workgroup, with no kernel
    // boundary and no cross-stream hand-over in front of the pivot chain.
    const int ncb_pro = use_xn ? t : jmax + 1;   // (use_xn: the strip's own diagonal block -- column block t -- needs no prologue)
    if (Kp > 0 && ncb_pro > 0) {
        const int nch = Kp >> 6, ncb = ncb_pro;
#if MNK_TILE_DMA
        v4d Bv[4], Bn[4];
        // this wave's eight requests of a tile: k-columns 16 w + 4 (q >> 1) + (q & 1) + 2 (lane >> 5), rows 2 (lane & 31), + 1
        const double* dsrc = F + (p0 + 2 * (lane & 31)) + (p0 - Kp + 16 * w + 2 * (lane >> 5)) * ld;
        char* const dbase = pp_smem + 4 * KT_G * w;
        char* const dnull = pp_smem + KT_NULL + 1024 * w;
        auto tile_dma = [&](int kc, int c, int buf, int q, bool real = true) {
            char* dst = dbase + buf * KT_BYTES + KT_G * (q >> 1) + KT_ODD * (q & 1);
            if (!real) dst = dnull;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dsrc + 64 * c + (64 * (int64_t)kc + 4 * (q >> 1) + (q & 1)) * ld),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        };
        auto b_load = [&](int kc, v4d (&B)[4]) {
            const double* src = Vp + (r0 + l15) + (64 * (int64_t)kc + l4) * ldv;
#pragma unroll
            for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                for (int s = 0; s < 4; ++s) B[ib][s] = src[(16 * ib + 4 * s) * ldv];
        };
        int it = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) tile_dma(0, 0, 0, q);
        b_load(0, Bn);
        for (int kc = 0; kc < nch; ++kc) {
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) { Bv[ib] = -Bn[ib]; pin(Bv[ib]); }   // (T -= V L^T: the sign goes into the operand that is copied anyway)
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                if (c >= ncb) break;
                const int buf = it & 1;
                ++it;
#if MNK_DIAG_STEP_TRACE
                const unsigned long long k0 = MNK_PCLK();
#endif
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part of the tile is in LDS ...
#if MNK_DIAG_STEP_TRACE
                const unsigned long long k1 = MNK_PCLK();
#endif
                __syncthreads();                                    // ... and everybody's; the other buffer is free
#if MNK_DIAG_STEP_TRACE
                const unsigned long long k2c = MNK_PCLK();
#endif
                // the next k-chunk's V behind the barrier: a step away from the wait above, and from the one that takes it
                if (c == 0 && kc + 1 < nch) b_load(kc + 1, Bn);
                int c2 = c + 1, k2 = kc;
                if (c2 >= ncb) { c2 = 0; k2 = kc + 1; }
                const bool more = k2 < nch;
                if (!more) { k2 = kc; c2 = c; }   // (the last step asks for its own tile again, into the null region)
                auto next = [&](int q) { tile_dma(k2, c2, buf ^ 1, q, more); };
#if MNK_DIAG_STEP_TRACE
                const unsigned long long k3 = MNK_PCLK();
#endif
                if (c == t) tile_mac_k<false>(pp_smem + buf * KT_BYTES, l15, l4, w + 1, Bv, &X[4 * c], next);
                else tile_mac_k<true>(pp_smem + buf * KT_BYTES, l15, l4, 4, Bv, &X[4 * c], next);
#if MNK_DIAG_STEP_TRACE
                const unsigned long long k4 = MNK_PCLK();
                pr_n += 1; pr_vm += k1 - k0; pr_bar += k2c - k1; pr_ld += k3 - k2c; pr_mac += k4 - k3;
#endif
            }
        }
        __syncthreads();
#else
        v4d pre[4], Bv[4], Bn[4];
        auto tile_load = [&](int kc, int c) {
            const double* src = F + (p0 + 64 * (int64_t)c + lane) + (p0 - Kp + 64 * (int64_t)kc + w) * ld;
#pragma unroll
            for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                for (int s = 0; s < 4; ++s) pre[ib][s] = src[(16 * ib + 4 * s) * ld];
        };
        auto b_load = [&](int kc, v4d (&B)[4]) {
            const double* src = Vp + (r0 + l15) + (64 * (int64_t)kc + l4) * ldv;
#pragma unroll
            for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                for (int s = 0; s < 4; ++s) B[ib][s] = src[(16 * ib + 4 * s) * ldv];
        };
        int it = 0;
        tile_load(0, 0);
        b_load(0, Bn);
        for (int kc = 0; kc < nch; ++kc) {
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) { Bv[ib] = -Bn[ib]; pin(Bv[ib]); }   // (T -= V L^T: the sign goes into the operand that is copied anyway)
            if (kc + 1 < nch) b_load(kc + 1, Bn);
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                if (c >= ncb) break;
                v4d* tile = stage + (it & 1) * 1024;
                ++it;
#if MNK_DIAG_STEP_TRACE
                const unsigned long long k0 = MNK_PCLK();
#endif
#pragma unroll
                for (int ib = 0; ib < 4; ++ib) tile[((lane >> 4) * 4 + ib) * 64 + (lane & 15) + 16 * w] = pre[ib];
#if MNK_DIAG_STEP_TRACE
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const unsigned long long k1 = MNK_PCLK();
#endif
                __syncthreads();
#if MNK_DIAG_STEP_TRACE
                const unsigned long long k2c = MNK_PCLK();
#endif
                {
                    int c2 = c + 1, k2 = kc;
                    if (c2 >= ncb) { c2 = 0; k2 = kc + 1; }
                    if (k2 < nch) tile_load(k2, c2);
                }
#if MNK_DIAG_STEP_TRACE
                const unsigned long long k3 = MNK_PCLK();
#endif
                tile_mac(tile, lane, c == t, w, Bv, &X[4 * c]);
#if MNK_DIAG_STEP_TRACE
                const unsigned long long k4 = MNK_PCLK();
                pr_n += 1; pr_vm += k1 - k0; pr_bar += k2c - k1; pr_ld += k3 - k2c; pr_mac += k4 - k3;
#endif
            }
        }
        __syncthreads();
#endif
    }
    if (EARLY && use_xn) {
#pragma unroll
        for (int c = 0; c < NB; ++c)
            if (c == t) {
#pragma unroll
                for (int cb2 = 0; cb2 < 4; ++cb2)
                    if (cb2 <= w) {
                        const v4d z = xn[(w * 4 + cb2) * 64 + lane];
                        if (xn_state == 2) X[4 * c + cb2] = z;   // (T' + Z, summed by the previous strip-column's last step)
                        else X[4 * c + cb2] += z;
                    }
            }
    }
    if (ptr_tr) ptr_tr[4] = wall_clock64();
#if MNK_DIAG_STEP_TRACE
    if (tr3) { tr3[0] = pr_n; tr3[1] = pr_vm; tr3[2] = pr_bar; tr3[3] = pr_ld; tr3[4] = pr_mac; }
#endif

    // One step per column block.  `j` is a compile-time constant (generic lambda over integral_constant), so every
    // index into X is static from the start and the strip stays in registers; returns true when the strip is done.
    auto step = [&](auto Jc) __attribute__((always_inline)) -> bool {
        constexpr int j = decltype(Jc)::value;
        if (j > jmax) return true;
        bool tn_now = false;   // EARLY: the band tile T' of the next diagonal block is final and parked in the staging tile
        if (EARLY && NB == 4 && j == 3 && make_xn) {
            // last step of a strip 4..7 (no update loop: the staging tiles are free).  T' is read while the strip waits for the
            // last diagonal block -- if the bulk kernel is done with it (never waited for here).
            if (tid == 0) {
                const int jsn = (int)(p0 >> 8) + 1;
                bool ready = true;
                if (2 * jsn - 2 > 0) {
                    ready = __hip_atomic_load(dag.af + (int64_t)(tabs >> 1) * dag.ntile + 2 * jsn + ((t - 4) >> 1), __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT) >= 1;
                    if (ready) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                *s_go = ready ? 3 : 1;
            }
            __syncthreads();
            tn_now = (*s_go & 2) != 0;
            if (tn_now) {
#pragma unroll
                for (int cb2 = 0; cb2 < 4; ++cb2)
                    if (cb2 <= w) {
                        v4d v;
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = F[(r0 + l15) + (p0 + 256 + 64 * (int64_t)(t - 4) + 16 * cb2 + l4 + 4 * r) * ld];
                        stage[(w * 4 + cb2) * 64 + lane] = v;
                    }
            }
        } else if (j > 0) __syncthreads();  // the LDS tiles of the previous step are free
        if (j == t && MNK_LEAF_WAVES == 4) {
            // ---- diagonal step, four waves: every wave factors its own block row of the 64x64 block (potrf64q_core); the
            // stores of all four are complete before one lane publishes the block
            const int64_t jb = (p0 >> 6) + j;
            v4d Lq[4];
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) Lq[b][r] = (b == w && l15 < l4 + 4 * r) ? 0.0 : X[4 * j + b][r];
            potrf64q_core<LDL, WT>(Lq, w, p0 + 64 * j, dblk0 + jb * 4096, inv0 + jb * 1024, dvec, dinv, info, pivot_tol,
                                   reinterpret_cast<double*>(stage), dag.vmax);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                if (!WT) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(prog + j, epoch16 + j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // (its last step: the strip's rows are final through the tile column of block j)
                if (dag.front != nullptr)
                    __hip_atomic_store(dag.front + tabs, (int)(p0 >> 7) + (j >> 1) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (dag.trace != nullptr) dag.trace[8 * t + 2] = wall_clock64();
            }
            return true;
        }
        if (j == t) {
            // ---- diagonal step, one wave (MNK_LEAF_WAVES = 1: rounds 2-4): hand the updated 64x64 block to wave 0 (same
            // lane mapping), factor, publish
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (b <= w) stage[(w * (w + 1) / 2 + b) * 64 + lane] = X[4 * j + b];
            __syncthreads();
            if (w != 0) return true;
            v4d Lt[4][4];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int b = 0; b <= cb; ++b) {
                    const v4d v = stage[(cb * (cb + 1) / 2 + b) * 64 + lane];
#pragma unroll
                    for (int r = 0; r < 4; ++r) Lt[cb][b][r] = (cb == b && l15 < l4 + 4 * r) ? 0.0 : v[r];
                }
            const int64_t jb = (p0 >> 6) + j;
            if (ptr_tr) ptr_tr[7] = wall_clock64();   // (trace: the leaf starts)
#if MNK_LEAF_V
            potrf64v_core<LDL, WT>(Lt, p0 + 64 * j, dblk0 + jb * 4096, inv0 + jb * 1024, dvec, dinv, info, pivot_tol, nullptr,
                                   nullptr, dag.vmax);
#else
            potrf64w_core<LDL, WT>(Lt, p0 + 64 * j, dblk0 + jb * 4096, inv0 + jb * 1024, dvec, dinv, info, pivot_tol, nullptr,
                                   nullptr, dag.vmax);
#endif
            if (!WT) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) {
                __hip_atomic_store(prog + j, epoch16 + j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // (its last step: the strip's rows are final through the tile column of block j)
                if (dag.front != nullptr)
                    __hip_atomic_store(dag.front + tabs, (int)(p0 >> 7) + (j >> 1) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (dag.trace != nullptr) dag.trace[8 * t + 2] = wall_clock64();
            }
            return true;
        }
        // ---- wait for the diagonal block j, X = T L_jj^-T
        pp_wait<NB, LDL>(prog, j, nb, epoch16 + j + 1, seen, info, pp_limit, dag.dbg != nullptr ? dag.dbg + 8 * t : nullptr, (int)(p0 >> 8), t);
        if (ptr_tr && j == t - 1) ptr_tr[6] = wall_clock64();   // (trace, tools/chain_steps.py: the block in front of this strip's own seen)
        MNK_TR2(4 * j + 0);
        const int64_t jb = (p0 >> 6) + j;
        const double* Dblk = dblk0 + jb * 4096;
        const double* Iv16 = inv0 + jb * 1024;
        // D^-1 of the block's 64 pivots for X = V D^-1 below: requested here, together with the block itself, so that the
        // latency is covered by the triangular solve instead of sitting between it and the stores
        double dsc[4][4];
        if (LDL) {
#pragma unroll
            for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                for (int r = 0; r < 4; ++r) dsc[ib][r] = dinv[p0 + 64 * j + 16 * ib + l4 + 4 * r];
        }
        {
            double Ln[6][4], Iv[4][4];
            int p = 0;
#pragma unroll
            for (int cb = 1; cb < 4; ++cb)
#pragma unroll
                for (int ib = 0; ib < cb; ++ib, ++p)
#pragma unroll
                    for (int s = 0; s < 4; ++s) Ln[p][s] = -Dblk[(16 * cb + l15) + 64 * (16 * ib + 4 * s + l4)];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int s = 0; s < 4; ++s) Iv[cb][s] = Iv16[cb * 256 + l15 + 16 * (4 * s + l4)];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                v4d tt = X[4 * j + cb];
#pragma unroll
                for (int ib = 0; ib < cb; ++ib) {
                    const int q = cb * (cb - 1) / 2 + ib;
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        tt = __builtin_amdgcn_mfma_f64_16x16x4f64(Ln[q][s], X[4 * j + ib][s], tt, 0, 0, 0);
                }
                v4d x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < 4; ++s) x = __builtin_amdgcn_mfma_f64_16x16x4f64(Iv[cb][s], tt[s], x, 0, 0, 0);
                X[4 * j + cb] = x;
            }
        }
        MNK_TR2(4 * j + 1);
        // ---- store V (LDL: to the W panel) and L; a diagonal strip also keeps its L rows in LDS and publishes
        double vm = 0.0;
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
            v4d lv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 64 * j + 16 * ib + l4 + 4 * r;
                const int64_t row = r0 + l15;
                const double v = X[4 * j + ib][r];
                if (LDL) {
                    lv[r] = v * dsc[ib][r];
                    vm = fmax(vm, fabs(v));
                    if (W != nullptr) put<WT>(W + row + (wcol0 + c) * ldw, v);
                } else {
                    lv[r] = v;
                }
                put<WT>(F + row + (p0 + c) * ld, lv[r]);
            }
            if (diag_strip || (EARLY && make_xn)) own[(w * 4 + ib) * 64 + lane] = lv;
        }
        if (LDL) growth_fold(dag.vmax, vm);
        // task-DAG schedule: the strip's rows are final through a whole tile column after every second block
        const bool pub_front = dag.front != nullptr && ((j & 1) != 0 || j == jmax);
        if (diag_strip || pub_front) {
#if MNK_DIAG_RACY_PUB   // (timing only, results void: what the stores' completion in front of the publication costs the chain)
            if (!diag_strip) {
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#if MNK_DIAG_RACY_PUB
            }
#endif
            if (tid == 0) {
                if (!WT) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (diag_strip) __hip_atomic_store(prog + t, epoch16 + j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (pub_front)
                    __hip_atomic_store(dag.front + tabs, (int)(p0 >> 7) + (j >> 1) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (pub_front && dag.trace != nullptr) dag.trace[8 * t + (j == jmax ? 2 : 3)] = wall_clock64();
            }
        }
        MNK_TR2(4 * j + 2);
        // ---- T[t, c] -= V[t, j] L[c, j]^T for the later column blocks (software-pipelined through LDS)
#if MNK_TILE_DMA
        // (the tile of column block c -- rows of the diagonal strip c, k-columns of block j -- goes straight into the staging tile c & 1: the
        // first one at once, the following ones under the products of the one before)
        const double* dsrc = F + (p0 + 2 * (lane & 31)) + (p0 + 64 * j + 16 * w + 2 * (lane >> 5)) * ld;
        char* const dbase = pp_smem + 4 * KT_G * w;
        char* const dnull = pp_smem + KT_NULL + 1024 * w;
        auto tile_dma = [&](int c, int q, bool real = true) {
            char* dst = dbase + (c & 1) * KT_BYTES + KT_G * (q >> 1) + KT_ODD * (q & 1);
            if (!real) dst = dnull;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dsrc + 64 * c + (4 * (q >> 1) + (q & 1)) * ld),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        };
        auto await = [&](int c) {
            pp_wait<NB, LDL>(prog, c, nb, epoch16 + j + 1, seen, info, pp_limit, dag.dbg != nullptr ? dag.dbg + 8 * t : nullptr, (int)(p0 >> 8), t);
        };
        if (j + 1 <= jmax && j + 1 != t) {
            await(j + 1 < NB ? j + 1 : 0);
#pragma unroll
            for (int q = 0; q < 8; ++q) tile_dma(j + 1 < NB ? j + 1 : 0, q);
        }
        // the block column's V (stored above, final) is the B operand of every product below: its sign is flipped once, in place
        // (X[4 j ..] is not read again as what it was: later steps and strip-columns take the stored values)
        v4d* const Bj = &X[4 * j];
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) { Bj[ib] = -Bj[ib]; pin(Bj[ib]); }
#pragma unroll
        for (int c = j + 1; c < NB; ++c) {
            if (c > jmax) break;
            if (c != t) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const bool more = c + 1 < NB && c + 1 <= jmax && c + 1 != t;
            if (more) await(c + 1 < NB ? c + 1 : 0);
            // (no next tile: the same one again, into the null region -- no branch in the loop)
            auto next = [&](int q) { tile_dma(more ? (c + 1 < NB ? c + 1 : 0) : c, q, more); };
            if (c == t) tile_mac<false>(own, lane, w + 1, Bj, &X[4 * c]);   // (the block behind a strip's own one is past jmax: no next tile)
            else tile_mac_k<true>(pp_smem + (c & 1) * KT_BYTES, l15, l4, 4, Bj, &X[4 * c], next);
        }
#else
        v4d pre[4];
        auto prefetch = [&](int c) {
            pp_wait<NB, LDL>(prog, c, nb, epoch16 + j + 1, seen, info, pp_limit, dag.dbg != nullptr ? dag.dbg + 8 * t : nullptr, (int)(p0 >> 8), t);
            const double* src = F + (p0 + 64 * (int64_t)c + lane) + (p0 + 64 * j + w) * ld;
#pragma unroll
            for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                for (int s = 0; s < 4; ++s) pre[ib][s] = src[(16 * ib + 4 * s) * ld];
        };
        if (j + 1 <= jmax && j + 1 != t) prefetch(j + 1 < NB ? j + 1 : 0);
        // the block column's V (stored above, final) is the B operand of every product below: its sign is flipped once, in place
        // (X[4 j ..] is not read again as what it was: later steps and strip-columns take the stored values)
        v4d* const Bj = &X[4 * j];
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) { Bj[ib] = -Bj[ib]; pin(Bj[ib]); }
#pragma unroll
        for (int c = j + 1; c < NB; ++c) {
            if (c > jmax) break;
            v4d* tile = c == t ? own : stage + (c & 1) * 1024;
            if (c != t) {
#pragma unroll
                for (int ib = 0; ib < 4; ++ib) tile[((lane >> 4) * 4 + ib) * 64 + (lane & 15) + 16 * w] = pre[ib];
            }
            __syncthreads();
            if (c + 1 < NB && c + 1 <= jmax && c + 1 != t) prefetch(c + 1 < NB ? c + 1 : 0);
            tile_mac(tile, lane, c == t, w, Bj, &X[4 * c]);
        }
#endif
        if (EARLY && make_xn) {
            // ---- the same product for the strip's diagonal block of the NEXT strip-column (its k-chunk j)
            __syncthreads();   // `own` is complete
            v4d acc[4];
#pragma unroll
            for (int cb2 = 0; cb2 < 4; ++cb2)
                if (cb2 <= w) acc[cb2] = xn[(w * 4 + cb2) * 64 + lane];
            tile_mac<false>(own, lane, w + 1, Bj, acc);
#pragma unroll
            for (int cb2 = 0; cb2 < 4; ++cb2)
                if (cb2 <= w) {
                    if (tn_now) acc[cb2] = stage[(w * 4 + cb2) * 64 + lane] + acc[cb2];
                    xn[(w * 4 + cb2) * 64 + lane] = acc[cb2];
                }
            if (tn_now && tid == 0) *xn_have = 2;
        }
        MNK_TR2(4 * j + 3);
        return false;
    };
    if (step(std::integral_constant<int, 0>{})) return;
    if (NB > 1 && step(std::integral_constant<int, (NB > 1 ? 1 : 0)>{})) return;
    if (NB > 2 && step(std::integral_constant<int, (NB > 2 ? 2 : 0)>{})) return;
    if (NB > 3 && step(std::integral_constant<int, (NB > 3 ? 3 : 0)>{})) return;
    if (NB > 4 && step(std::integral_constant<int, (NB > 4 ? 4 : 0)>{})) return;
    if (NB > 5 && step(std::integral_constant<int, (NB > 5 ? 5 : 0)>{})) return;
    if (NB > 6 && step(std::integral_constant<int, (NB > 6 ? 6 : 0)>{})) return;
    if (NB > 7 && step(std::integral_constant<int, (NB > 7 ? 7 : 0)>{})) return;
}

template <bool LDL, int NB>
__global__ __launch_bounds__(256) void ppanel_kernel(double* __restrict__ F, int64_t ld, int64_t p0, int nb, int64_t Np,
                                                      double* __restrict__ dblk0, double* __restrict__ inv0,
                                                      double* __restrict__ dvec, double* __restrict__ dinv,
                                                      double* __restrict__ W, int64_t ldw, int64_t wcol0,
                                                      int* __restrict__ info, double pivot_tol, int* __restrict__ prog,
                                                      int epoch16, int dbg_missing, const double* __restrict__ Vp,
                                                      int64_t ldv, int Kp, PpDag dag) {
    extern __shared__ __attribute__((aligned(128))) char pp_smem[];
    __shared__ int s_go;
    pp_strip<LDL, NB>((int)blockIdx.x, F, ld, p0, nb, Np, dblk0, inv0, dvec, dinv, W, ldw, wcol0, info, pivot_tol, prog, epoch16,
                      dbg_missing, Vp, ldv, Kp, dag, pp_smem, &s_go);
}

// Persistent pivot chain of the task-DAG schedule (dag.hip): ONE launch for the whole factorization.  Workgroup t owns strip
// t of the band of EVERY strip-column Js = 0, 1, ... (the band: `gridDim.x` strips of 64 rows from the strip-column's first
// row; strips 0..3 are its diagonal strips, strip t + 4 of one strip-column is strip t of the next).  A strip waits for its
// rows of the older columns (front[]: published by strip t + 4 of the previous strip-column, or by the bulk kernel for the
// rows that enter the band), for its band tiles (af[]) and for the diagonal blocks (prog[]); nothing else orders the
// strip-columns, so the diagonal strips of Js + 1 start the moment strips 4..7 of Js are done, however long the strips
// further down wait for the bulk kernel.  All workgroups must be resident (one per CU of the chain's partition); every
// workgroup walks its strips in order and every wait is on a strip of an earlier strip-column or of the same one with a
// smaller index, so residency implies progress.
template <bool LDL>
__global__ __launch_bounds__(256) void pchain_kernel(double* __restrict__ F, int64_t ld, int64_t Np, double* __restrict__ dblk0,
                                                      double* __restrict__ inv0, double* __restrict__ dvec,
                                                      double* __restrict__ dinv, double* __restrict__ V, int* __restrict__ info,
                                                      double pivot_tol, int* __restrict__ flag_p, int epoch16, int dbg_missing,
                                                      PpDag dag, int js_begin, int js_end) {
    extern __shared__ __attribute__((aligned(128))) char pp_smem[];
    __shared__ int s_go;
    __shared__ int s_xn;
    // Workgroup g keeps its ROWS from one strip-column to the next (row block 4 js_begin + g, then + gridDim.x, ...): in
    // strip-column Js it is strip (g - 4 (Js - js_begin)) mod gridDim.x of the band, the rows of a diagonal strip are still in
    // its registers / LDS when they become the next strip-column's diagonal strip (pp_strip, EARLY), and the workgroups that
    // are done with their diagonal blocks take the rows that enter the band.
    const int g = blockIdx.x, G = gridDim.x;
    if (threadIdx.x == 0) s_xn = 0;
    __syncthreads();
    for (int64_t Js = js_begin, p0 = 256 * (int64_t)js_begin; Js < js_end && p0 < Np; p0 += 256, ++Js) {
        int t = (int)((g - 4 * (Js - js_begin)) % G);
        if (t < 0) t += G;
        if (p0 + 64 * (int64_t)t >= Np) continue;   // (no such rows; s_xn is 0: only a strip with rows sets it, and the next one consumes it)
        const int nb = (int)((Np - p0) / 64 < 4 ? (Np - p0) / 64 : 4);
        PpDag d = dag;
        d.need_front = Js > 0 ? (int)(2 * Js) : 0;
        d.front_from = 0;
        d.af_tilecol = 2 * Js - 2 > 0 ? (int)(2 * Js) : -1;
        if (dag.trace != nullptr) d.trace = dag.trace + (Js - js_begin) * 8 * (int64_t)gridDim.x;
        const double* Vp = Js > 0 ? (LDL ? V : F) + (p0 - 256) * ld : nullptr;
        pp_strip<LDL, 4, true, true>(t, F, ld, p0, nb, Np, dblk0, inv0, dvec, dinv, LDL ? V : nullptr, LDL ? ld : 0, p0, info, pivot_tol,
                         flag_p + p0 / 64, epoch16, dbg_missing, Vp, ld, Js > 0 ? 256 : 0, d, pp_smem, &s_go, &s_xn);
        __syncthreads();  // (waves leave a strip at different times; its LDS tiles and s_go are reused)
    }
}

// The pivot chains of SEVERAL small systems in one launch (batches of small factorizations: the dense blocks of the Schur
// stage, C2-size systems): workgroups [strips i, strips (i + 1)) are the strips of system i, every row of it in the chain's
// band -- a system of a few hundred rows is nothing but its chain (the previous strip-column is applied by the strips'
// prologue), larger ones get their band tiles accumulated by the batch's bulk kernel.  All workgroups must be resident
// (one per CU of the launch stream's mask).  The record of a system is read through the constant address space.
struct PcSys {
    double* F;
    int64_t ld;
    int64_t Np;
    double* dblk0;
    double* inv0;
    double* dvec;
    double* dinv;
    double* V;
    int* info;
    double pivot_tol;
    int* flag_p;
    int epoch16;
    int* front;
    const int* af;
    int ntile;
    long spin_limit;
    unsigned long long* vmax;
};
typedef const PcSys __attribute__((address_space(4))) * PcSysP;

template <bool LDL>
__global__ __launch_bounds__(256) void pchain_multi_kernel(const PcSys* __restrict__ sys, int strips) {
    extern __shared__ __attribute__((aligned(128))) char pp_smem[];
    __shared__ int s_go;
    __shared__ int s_xn;
    const int isys = (int)blockIdx.x / strips, g = (int)blockIdx.x % strips;   // (workgroup g keeps row block g: strip g - 4 Js)
    if (threadIdx.x == 0) s_xn = 0;
    __syncthreads();
    const PcSysP y = (PcSysP)(uintptr_t)(sys + isys);
    double* F = y->F;
    const int64_t ld = y->ld, Np = y->Np;
    double* V = y->V;
    const PpDag dag{y->front, y->af, y->ntile, 0, 0, -1, y->spin_limit, nullptr, y->vmax};
    for (int64_t Js = 0, p0 = 0; 4 * Js <= g; p0 += 256, ++Js) {
        const int t = g - 4 * (int)Js;
        const int nb = (int)((Np - p0) / 64 < 4 ? (Np - p0) / 64 : 4);
        PpDag d = dag;
        d.need_front = Js > 0 ? (int)(2 * Js) : 0;
        d.front_from = 0;
        d.af_tilecol = 2 * Js - 2 > 0 ? (int)(2 * Js) : -1;
        const double* Vp = Js > 0 ? (LDL ? V : F) + (p0 - 256) * ld : nullptr;
        pp_strip<LDL, 4, true, true>(t, F, ld, p0, nb, Np, y->dblk0, y->inv0, y->dvec, y->dinv, LDL ? V : nullptr, LDL ? ld : 0, p0, y->info,
                               y->pivot_tol, y->flag_p + p0 / 64, y->epoch16, -1, Vp, ld, Js > 0 ? 256 : 0, d, pp_smem, &s_go, &s_xn);
        __syncthreads();  // (waves leave a strip at different times; its LDS tiles and s_go are reused)
    }
}

__global__ void pc_set_sys_kernel(PcSys rec, PcSys* __restrict__ dst) { *dst = rec; }

// inv(L_jj) of every 64x64 diagonal block (unit diagonal for LDL), for the triangular solves.
// One workgroup of 4 waves per block: thread (c, p) solves L x = e_c for the rows r = 4i + p by column-oriented
// substitution; the owner of row k publishes x_k through LDS (one barrier per pivot, double-buffered), so every
// thread carries 16 of the 64 rows (13 us instead of 52 with one wave per block).
template <bool LDL>
__global__ __launch_bounds__(256) void linv64_kernel(double* __restrict__ F, int64_t ld,
                                                      const double* __restrict__ Dblk,
                                                      double* __restrict__ Linv, const int* __restrict__ info, int blk0,
                                                      const SmallSysRec* __restrict__ recs = nullptr) {
    __shared__ double Lt[64 * 64];  // Lt[k*64 + r] = L[r][k]
    __shared__ double rd[64];
    __shared__ double xk[2][64];
    if (recs != nullptr) {   // (a batch of small systems: blockIdx.z = system)
        const SmallSysRec r = recs[blockIdx.z];
        F = r.F; ld = r.ld; Dblk = r.dblk; Linv = r.linv; info = r.info;
    }
    if (*info != 0) return;
    const int64_t blk = (int64_t)blockIdx.x + blk0;  // (the blocks may be inverted in two launches: see mnk_ls_invert_blocks)
    const int64_t j0 = blk * 64;
    const int c = threadIdx.x & 63;
    const int p = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const double* A = Dblk + blk * 4096;  // factored diagonal block, column-major 64x64
    double* Fd = F + j0 + j0 * ld;
    // stage the block: wave p takes the columns q = 4i + p; lane = row
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int q = 4 * i + p, lane = c;
        const double v = lane >= q ? A[lane + 64 * q] : 0.0;
        if (lane >= q) Fd[lane + (int64_t)q * ld] = v;  // put the block back into the factor (lower part)
        Lt[q * 64 + lane] = lane > q ? v : (lane == q ? (LDL ? 1.0 : v) : 0.0);
        if (lane == q) rd[q] = LDL ? 1.0 : 1.0 / v;
    }
    __syncthreads();
    double s[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = (4 * i + p == c) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < 64; ++k) {
        if ((k & 3) == p) {  // wave-uniform: this wave owns row k
            const double x = s[k >> 2] * rd[k];
            s[k >> 2] = x;
            xk[k & 1][c] = x;
        }
        __syncthreads();
        const double x = xk[k & 1][c];
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (4 * i + 3 > k) {  // compile-time: some row of this group may lie below k
                const int r = 4 * i + p;
                if (r > k) s[i] -= Lt[k * 64 + r] * x;
            }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) Lt[c * 64 + 4 * i + p] = s[i];  // Lt[c*64 + r] = inv(L)[r][c]
    __syncthreads();
    double* out = Linv + blk * 4096;  // column-major 64x64: out[r + 64 c]
#pragma unroll 4
    for (int i = 0; i < 16; ++i) out[c + 64 * (4 * i + p)] = Lt[c + 64 * (4 * i + p)];
}

// The host reads the inertia counters and the info word from pinned, device-mapped memory that this one-thread kernel
// fills with system-scope stores: no copy engine and no staging copy between the last kernel and the host.
// (Bunch-Kaufman tier only since round 4: the static tiers publish through finish_info_kernel.)
__global__ void publish_info_kernel(const unsigned long long* __restrict__ inertia, const int* __restrict__ info,
                                    unsigned long long* __restrict__ host_words) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        for (int i = 0; i < 3; ++i) __hip_atomic_store(host_words + i, inertia[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        for (int i = 0; i < 3; ++i) __hip_atomic_store(host_words + 4 + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(host_words + 3, (unsigned long long)(long long)*info, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Inertia, growth-guard words and `info` of a static-pivot factorization in ONE launch of ONE workgroup that is queued
// right behind the factorization (round 3: a memset, inertia_kernel and publish_info_kernel, enqueued when the inertia was
// asked for): signs of D over the first N pivots (reference inertia rule for an unpivoted LDL^T: the signs of D), max|d_k|
// folded with the max|v_ik| the factorization's kernels recorded, sign changes along the pivot sequence, max|a_ij| of the
// transfer -- stored to pinned host words with system scope, `info` last with release semantics.  Cholesky: `info` only
// (N = 0).  The host then needs nothing but the stream synchronization it does anyway.
__global__ __launch_bounds__(1024) void finish_info_kernel(const double* __restrict__ dvec, int64_t N, const int* __restrict__ info,
                                                            unsigned long long* __restrict__ host_words,
                                                            const unsigned long long* __restrict__ amax,
                                                            const SmallSysRec* __restrict__ recs = nullptr) {
    __shared__ unsigned long long red[16][4];
    __shared__ double redm[16];
    if (recs != nullptr) {   // (a batch of small systems: blockIdx.x = system)
        const SmallSysRec r = recs[blockIdx.x];
        dvec = r.dvec; N = r.ninertia; info = r.info; host_words = r.pin_dev; amax = r.amax;
    }
    unsigned long long pos = 0, zer = 0, neg = 0, chg = 0;
    double amx = 0.0;
    // (an early rejection, info = -9: D is valid through pivot info[1] only -- what lies behind is the previous factorization's)
    if (N > 0 && *info == -9) N = info[1] + 1 < N ? info[1] + 1 : N;
    for (int64_t k = threadIdx.x; k < N; k += blockDim.x) {
        const double d = dvec[k];
        if (d > 0.0) ++pos;
        else if (d < 0.0) ++neg;
        else ++zer;
        const double a = fabs(d) <= DBL_MAX ? fabs(d) : __longlong_as_double(0x7ff0000000000000LL);  // NaN / Inf pivot -> Inf
        amx = fmax(amx, a);
        if (k + 1 < N && ((d > 0.0) != (dvec[k + 1] > 0.0))) ++chg;   // <= 1 in total: "positive pivots, then negative ones"
    }
    for (int off = 32; off > 0; off >>= 1) {
        pos += __shfl_xor((long long)pos, off);
        zer += __shfl_xor((long long)zer, off);
        neg += __shfl_xor((long long)neg, off);
        chg += __shfl_xor((long long)chg, off);
        amx = fmax(amx, __shfl_xor(amx, off));
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[w][0] = pos; red[w][1] = zer; red[w][2] = neg; red[w][3] = chg; redm[w] = amx; }
    __syncthreads();
    // max|a_ij| of the transfer: the maximum over the slots its kernels folded into (one wave)
    unsigned long long am = 0ull;
    if (amax != nullptr && threadIdx.x < 64) {
        for (int i = threadIdx.x; i < AMAX_SLOTS; i += 64) am = std::max(am, amax[AMAX_SLOT0 + AMAX_STRIDE * i]);
        for (int off = 32; off > 0; off >>= 1) am = std::max(am, (unsigned long long)__shfl_xor((long long)am, off));
        am = std::max(am, amax[0]);
    }
    if (threadIdx.x == 0) {
        const int nw = (int)(blockDim.x >> 6);
        unsigned long long t[4] = {0, 0, 0, 0};
        double m = 0.0;
        for (int i = 0; i < nw; ++i) {
            for (int j = 0; j < 4; ++j) t[j] += red[i][j];
            m = fmax(m, redm[i]);
        }
        for (int i = 0; i < 3; ++i) __hip_atomic_store(host_words + i, t[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        unsigned long long gw = 0ull;
        if (amax != nullptr) gw = std::max(amax[1], (unsigned long long)__double_as_longlong(m));   // (bit patterns of non-negative doubles order like the values)
        __hip_atomic_store(host_words + 4, am, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(host_words + 5, gw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(host_words + 6, amax != nullptr ? t[3] : 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(host_words + 7, (unsigned long long)(long long)info[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // which bounded wait expired (info = -7)
        __hip_atomic_store(host_words + 3, (unsigned long long)(long long)*info, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace mnk

using namespace mnk;

// the word the kernels fold max|V| into (zeroed with max|a_ij| when the matrix was transferred), or NULL: guard off
unsigned long long* mnk_ls_growth_word(mnk_ls* ls) {
    return (ls->algo == MNK_LDL && ls->bk_requested && ls->bk_fallback && ls->amax_dev.p) ? ls->amax_dev.p + 1 : nullptr;
}

// panel_algo = 4: persistent panel launches (ppanel_kernel) of 4 blocks, recursive updates between them
static int factor_outer_panel_pp(mnk_ls* ls, hipStream_t s, int64_t ko, int64_t kend, double* wbase,
                                 hipEvent_t rest_ready, int64_t rest_from, const double* Vfirst, int64_t ldvfirst,
                                 int64_t Kfirst) {
    const int64_t Np = ls->Np, ld = ls->ld;
    const bool ldl = ls->algo == MNK_LDL;
    double* F = ls->fact.p;
    {   // 96 KB of dynamic LDS: the attribute belongs to the (kernel, device) pair
        static std::atomic<uint64_t> attr_devs{0};
        int dev = 0;
        MNK_HIP(hipGetDevice(&dev));
        if (!(attr_devs.load(std::memory_order_relaxed) >> (dev & 63) & 1)) {
            MNK_HIP(hipFuncSetAttribute((const void*)ppanel_kernel<true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES));
            MNK_HIP(hipFuncSetAttribute((const void*)ppanel_kernel<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES));
            attr_devs.fetch_or(1ull << (dev & 63), std::memory_order_relaxed);
        }
    }
    constexpr int NBs = 4;  // 64-column blocks per persistent launch (8 was built and measured slower: 197 + 256 registers)
    const int64_t ws = (int64_t)NBI * NBs;
    const int epoch16 = ls->epoch * 16;
    bool waited = rest_ready == nullptr;
    // columns in front of the next launch that it applies itself (ppanel_kernel's prologue)
    const double* Vp = Vfirst;
    int64_t ldv = ldvfirst;
    int Kp = (int)Kfirst;
    const bool fuse_mid = ls->pp_fuse_rows > 0 && Np - ko <= ls->pp_fuse_rows;
    for (int64_t p = ko; p < kend; p += ws) {
        const int nbk = (int)std::min<int64_t>(NBs, (kend - p) / NBI);
        if (!waited && p + NBI * nbk > ko + rest_from) {
            MNK_HIP(hipStreamWaitEvent(s, rest_ready, 0));
            waited = true;
        }
        const unsigned grid = (unsigned)((Np - p) / NBI);
    unsigned long long* vmaxw = mnk_ls_growth_word(ls);
#define MNK_PP(LD, NBT)                                                                                              \
    hipLaunchKernelGGL((ppanel_kernel<LD, NBT>), dim3(grid), dim3(256), PP_LDS_BYTES, s, F, ld, p, nbk, Np, ls->dblk.p, \
                       ls->inv16.p, ls->dvec.p, ls->dinv.p, LD ? wbase : (double*)nullptr, LD ? ls->ldw : (int64_t)0,  \
                       p - ko, ls->info_dev.p, ls->pivot_tol, ls->flag_p.p + p / NBI, epoch16, ls->debug_pp_missing, Vp,  \
                       ldv, Kp, PpDag{nullptr, nullptr, 0, 0, 4, -1, 0, nullptr, vmaxw})
        if (ldl) MNK_PP(true, 4);
        else MNK_PP(false, 4);
#undef MNK_PP
        Vp = nullptr;
        Kp = 0;
        const int64_t p1 = p + NBI * nbk;
        if (p1 >= kend) break;
        const int64_t jj = (p1 - ko) / ws;    // launches of this outer panel that are finished
        const int64_t w = ws * (jj & -jj);     // columns whose contribution is applied now
        const int64_t pb = p1 - w;
        const int64_t ncols = std::min<int64_t>(w, kend - p1);
        if (fuse_mid && w == ws && ncols <= ws) {  // exactly the next launch's columns: it applies them itself
            Vp = ldl ? wbase + (pb - ko) * ls->ldw : F + pb * ld;
            ldv = ldl ? ls->ldw : ld;
            Kp = (int)w;
            continue;
        }
        if (!waited && p1 + ncols > ko + rest_from) {
            MNK_HIP(hipStreamWaitEvent(s, rest_ready, 0));
            waited = true;
        }
        const double* Wp = ldl ? wbase + p1 + (pb - ko) * ls->ldw : F + p1 + pb * ld;
        int rc;
        if (gemm_nt_lower_tiles(Np - p1, ncols) < ls->small_tiles_mid)
            rc = launch_gemm_nt_lower_small(s, Np - p1, ncols, w, Wp, ldl ? ls->ldw : ld, F + p1 + pb * ld, ld,
                                            F + p1 + p1 * ld, ld, ls->info_dev.p);
        else
            rc = launch_gemm_nt(s, 2, Np - p1, ncols, w, Wp, ldl ? ls->ldw : ld, F + p1 + pb * ld, ld, F + p1 + p1 * ld,
                                ld, nullptr, nullptr, 0, ls->info_dev.p);
        if (rc) return rc;
    }
    MNK_HIP(hipGetLastError());
    return 0;
}

// (Vfirst, ldvfirst, Kfirst: panel_algo 4 only -- the Kfirst columns in front of the panel are applied to its first
// 256 columns by the first persistent launch itself; V[row, k] = Vfirst[row + k * ldvfirst])
static int factor_outer_panel(mnk_ls* ls, hipStream_t s, int64_t ko, int64_t kend, double* wbase,
                              hipEvent_t rest_ready = nullptr, int64_t rest_from = 256, const double* Vfirst = nullptr,
                              int64_t ldvfirst = 0, int64_t Kfirst = 0) {
    if (ls->algo_now == 4)
        return factor_outer_panel_pp(ls, s, ko, kend, wbase, rest_ready, rest_from, Vfirst, ldvfirst, Kfirst);
    const int64_t Np = ls->Np, ld = ls->ld;
    const bool ldl = ls->algo == MNK_LDL;
    double* F = ls->fact.p;
    bool waited = rest_ready == nullptr;
    for (int64_t j = ko; j < kend; j += NBI) {
        double* dblk = ls->dblk.p + (j / NBI) * 4096;
        double* inv16 = ls->inv16.p + (j / NBI) * 1024;
        if (ldl)
            hipLaunchKernelGGL(potrf64w_kernel<true>, dim3(1), dim3(64), 0, s, F, ld, j, dblk, inv16, ls->dvec.p, ls->dinv.p,
                               ls->info_dev.p, ls->pivot_tol, mnk_ls_growth_word(ls));
        else
            hipLaunchKernelGGL(potrf64w_kernel<false>, dim3(1), dim3(64), 0, s, F, ld, j, dblk, inv16, ls->dvec.p, ls->dinv.p,
                               ls->info_dev.p, ls->pivot_tol, (unsigned long long*)nullptr);
        const int64_t M = Np - j - NBI;
        if (M <= 0) break;
        // 16-row strips per wave: one while the panel is short (more workgroups, shortest chain), two beyond
        const bool two = M / 64 > 2 * (int64_t)ls->ctx->num_cu;
        const unsigned grid = (unsigned)((M / (two ? 32 : 16) + 3) / 4);
#define MNK_TRSM(LD, NS)                                                                                          \
    hipLaunchKernelGGL((trsm64_mfma_kernel<LD, NS>), dim3(grid), dim3(256), 0, s, F, ld, j, Np, dblk, inv16,    \
                       ls->dinv.p, LD ? wbase : (double*)nullptr, LD ? ls->ldw : (int64_t)0, j - ko, ls->info_dev.p,  \
                       mnk_ls_growth_word(ls))
        if (ldl) { if (two) MNK_TRSM(true, 2); else MNK_TRSM(true, 1); }
        else { if (two) MNK_TRSM(false, 2); else MNK_TRSM(false, 1); }
#undef MNK_TRSM
        const int64_t p1 = j + NBI;
        if (p1 >= kend) break;
        const int64_t jj = (p1 - ko) / NBI;            // blocks of this outer panel that are finished
        const int64_t w = NBI * (jj & -jj);             // columns whose contribution is applied now
        const int64_t p0 = p1 - w;
        const int64_t ncols = std::min<int64_t>(w, kend - p1);
        if (!waited && p1 + ncols > ko + rest_from) {  // first kernel that touches columns the caller delivers late
            MNK_HIP(hipStreamWaitEvent(s, rest_ready, 0));
            waited = true;
        }
        const double* Wp = ldl ? wbase + p1 + (p0 - ko) * ls->ldw : F + p1 + p0 * ld;
        int rc;
        if (gemm_nt_lower_tiles(Np - p1, ncols) < ls->small_tiles_mid)
            rc = launch_gemm_nt_lower_small(s, Np - p1, ncols, w, Wp, ldl ? ls->ldw : ld, F + p1 + p0 * ld, ld,
                                            F + p1 + p1 * ld, ld, ls->info_dev.p);
        else
            rc = launch_gemm_nt(s, 2, Np - p1, ncols, w, Wp, ldl ? ls->ldw : ld, F + p1 + p0 * ld, ld,
                                F + p1 + p1 * ld, ld, nullptr, nullptr, 0, ls->info_dev.p);
        if (rc) return rc;
    }
    MNK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------
// factorization driver (host orchestration; asynchronous)
// ---------------------------------------------------------------------------------------
// Small systems are latency-bound end to end: one outer panel (middle-level right-looking updates over all
// remaining columns, no outer trailing update, no stream hand-overs) beats the look-ahead schedule up to
// N ~ 4.5k (measured: N = 1024 -11 %, 2048 -12 %, 3072 -8 %, 4096 -3 %; N = 6144 +3 %).
int64_t mnk_ls_effective_nbo(const mnk_ls* ls) {
    if (ls->single_rows > 0 && ls->Np <= ls->single_rows) return ls->Np;
    // outer_block = 0 ("by size", the default since round 6): 512 columns per trailing update, 1024 from 32 768 rows on -- the update
    // reads and writes the whole trailing triangle once per outer panel (profiles/r06_config_C4_pmc_traffic.md: 8.3 TB at the
    // L2 -> fabric interface per factorization at N = 85 568, 3.3 TB of it the triangle itself), so twice the width halves that part;
    // measured on one box at N = 85 568: 512 -> 3.104 s, 768 -> 3.048, 1024 -> 3.035, 2048 -> 3.043 (0.856 -> 0.875 of the fp64 peak).
    // Below ~3e4 rows the wider panel's own (slower) work costs more than the triangle's traffic saves (r02_knob_sweeps.md: C3).
    if (ls->nbo_auto) return ls->Np >= 32768 ? 1024 : 512;
    return ls->nbo;
}

// inv(L_jj) of the 64x64 diagonal blocks and the explicit inverses of the 256x256 diagonal triangles (what the solves
// use) for the strip-columns [sc0, sc1) of 256 columns.  (The Bunch-Kaufman tier always inverts unit-lower blocks.)
double mnk_host_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static std::atomic<long long> g_stall_us{0};
void mnk_add_process_stall_ms(double ms) { g_stall_us.fetch_add((long long)(ms * 1e3), std::memory_order_relaxed); }
double mnk_process_stall_ms() { return (double)g_stall_us.load(std::memory_order_relaxed) * 1e-3; }

int mnk_ls_invert_blocks(mnk_ls* ls, hipStream_t s, int64_t sc0, int64_t sc1) {
    if (sc1 <= sc0) return 0;
    const int64_t b0 = 4 * sc0, b1 = std::min<int64_t>(4 * sc1, ls->Np / NBI);
    const bool ldl = ls->algo == MNK_LDL || ls->bk_active;
    if (ldl)
        hipLaunchKernelGGL(linv64_kernel<true>, dim3((unsigned)(b1 - b0)), dim3(256), 0, s, ls->fact.p, ls->ld, ls->dblk.p,
                           ls->linv.p, ls->info_dev.p, (int)b0);
    else
        hipLaunchKernelGGL(linv64_kernel<false>, dim3((unsigned)(b1 - b0)), dim3(256), 0, s, ls->fact.p, ls->ld, ls->dblk.p,
                           ls->linv.p, ls->info_dev.p, (int)b0);
    MNK_HIP(hipGetLastError());
    return mnk_ls_build_inverses(ls, s, sc0, sc1);
}

// The persistent pivot chain of the task-DAG schedule (dag.hip's host driver launches it beside the bulk kernel).
int mnk_launch_pchain(mnk_ls* ls, hipStream_t sp, const mnk::PpDag& dag, int js_begin, int js_end, unsigned strips) {
    {   // 96 KB of dynamic LDS: the attribute belongs to the (kernel, device) pair
        static std::atomic<uint64_t> attr_devs{0};
        int dev = 0;
        MNK_HIP(hipGetDevice(&dev));
        if (!(attr_devs.load(std::memory_order_relaxed) >> (dev & 63) & 1)) {
            MNK_HIP(hipFuncSetAttribute((const void*)pchain_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, PC_LDS_BYTES));
            MNK_HIP(hipFuncSetAttribute((const void*)pchain_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, PC_LDS_BYTES));
            attr_devs.fetch_or(1ull << (dev & 63), std::memory_order_relaxed);
        }
    }
    const bool ldl = ls->algo == MNK_LDL;
    const int epoch16 = ls->epoch * 16;
    double* V = ldl ? ls->vfull.p : nullptr;
    if (ldl)
        hipLaunchKernelGGL(pchain_kernel<true>, dim3(strips), dim3(256), PC_LDS_BYTES, sp, ls->fact.p, ls->ld, ls->Np, ls->dblk.p,
                           ls->inv16.p, ls->dvec.p, ls->dinv.p, V, ls->info_dev.p, ls->pivot_tol, ls->flag_p.p, epoch16,
                           ls->debug_pp_missing, dag, js_begin, js_end);
    else
        hipLaunchKernelGGL(pchain_kernel<false>, dim3(strips), dim3(256), PC_LDS_BYTES, sp, ls->fact.p, ls->ld, ls->Np, ls->dblk.p,
                           ls->inv16.p, ls->dvec.p, ls->dinv.p, V, ls->info_dev.p, ls->pivot_tol, ls->flag_p.p, epoch16,
                           ls->debug_pp_missing, dag, js_begin, js_end);
    MNK_HIP(hipGetLastError());
    return 0;
}

static int run_factorization_body(mnk_ls* ls);

// The chains of the `n` systems v[0..n) (same order and algorithm, every row in the band: `strips` = Np / 64 workgroups each)
// in ONE launch on `sp` (its CU mask must hold n * strips CUs); `table`: device memory for n records (>= n * mnk_pchain_sys_bytes()).
size_t mnk_pchain_sys_bytes() { return sizeof(mnk::PcSys); }
void mnk_pchain_fill_sys(mnk_ls* ls, void* rec_host, int* front, const int* af) {
    const bool ldl = ls->algo == MNK_LDL;
    *static_cast<mnk::PcSys*>(rec_host) = mnk::PcSys{ls->fact.p, ls->ld, ls->Np, ls->dblk.p, ls->inv16.p, ls->dvec.p, ls->dinv.p,
                                                       ldl ? ls->vfull.p : nullptr, ls->info_dev.p, ls->pivot_tol, ls->flag_p.p,
                                                       ls->epoch * 16, front, af, (int)(ls->Np / 128), mnk_ls_dag_spin_limit(ls),
                                                       mnk_ls_growth_word(ls)};
}

// (fill_stream == nullptr: the table has been uploaded by the caller -- mnk_pchain_fill_sys)
int mnk_launch_pchain_multi(mnk_ls* const* v, int n, hipStream_t sp, hipStream_t fill_stream, void* table, int* const* front,
                            const int* const* af) {
    {   // 96 KB of dynamic LDS: the attribute belongs to the (kernel, device) pair
        static std::atomic<uint64_t> attr_devs{0};
        int dev = 0;
        MNK_HIP(hipGetDevice(&dev));
        if (!(attr_devs.load(std::memory_order_relaxed) >> (dev & 63) & 1)) {
            MNK_HIP(hipFuncSetAttribute((const void*)pchain_multi_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, PC_LDS_BYTES));
            MNK_HIP(hipFuncSetAttribute((const void*)pchain_multi_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, PC_LDS_BYTES));
            attr_devs.fetch_or(1ull << (dev & 63), std::memory_order_relaxed);
        }
    }
    mnk::PcSys* dst = static_cast<mnk::PcSys*>(table);
    const bool ldl = v[0]->algo == MNK_LDL;
    const int strips = (int)(v[0]->Np / NBI);
    for (int i = 0; i < n && fill_stream != nullptr; ++i) {
        mnk_ls* ls = v[i];
        mnk::PcSys rec{ls->fact.p, ls->ld, ls->Np, ls->dblk.p, ls->inv16.p, ls->dvec.p, ls->dinv.p, ldl ? ls->vfull.p : nullptr,
                       ls->info_dev.p, ls->pivot_tol, ls->flag_p.p, ls->epoch * 16, front[i], af[i], (int)(ls->Np / 128),
                       mnk_ls_dag_spin_limit(ls), mnk_ls_growth_word(ls)};
        hipLaunchKernelGGL(pc_set_sys_kernel, dim3(1), dim3(1), 0, fill_stream, rec, dst + i);
    }
    MNK_HIP(hipGetLastError());
    if (ldl)
        hipLaunchKernelGGL(pchain_multi_kernel<true>, dim3((unsigned)(n * strips)), dim3(256), PC_LDS_BYTES, sp, dst, strips);
    else
        hipLaunchKernelGGL(pchain_multi_kernel<false>, dim3((unsigned)(n * strips)), dim3(256), PC_LDS_BYTES, sp, dst, strips);
    MNK_HIP(hipGetLastError());
    return 0;
}

int mnk_ls_run_factorization(mnk_ls* ls) {
    mnk_ctx* ctx = ls->ctx;
    hipStream_t s = ctx->stream;
    const int64_t Np = ls->Np;
    if (ls->panel_algo >= 4 && !ls->flag_p.p) {
        if (ls->flag_p.alloc(Np / NBI + 1)) return -2;
        MNK_HIP(hipMemsetAsync(ls->flag_p.p, 0, (Np / NBI + 1) * sizeof(int), s));
    }
    ++ls->epoch;  // the hand-off flags of this factorization carry this value (never reset)
    ls->inv_done = 0;
    // (a verdict belongs to ONE factorization -- a new one with rejection disarmed, followed by a solve that never
    // fetches the info word, must not meet the previous factorization's "rejected early" and refactor a valid factor)
    ls->factor_invalid = false;
    ls->t_fact_launch_ms = mnk_host_ms();
    {   // info[2]: the leaf stops the factorization at the first pivot that is not positive (leaf64.h: early rejection)
        // (also inside a factorization batch: a member that stops early leaves the others alone -- dag.hip, the workgroup-uniform
        // `dead` of the batch kernel; tests/test_hip_round5.py::test_a_batch_in_which_some_instances_are_indefinite)
        const int want = (ls->early_reject && ls->accept_only_pd && ls->algo == MNK_LDL && ls->src_persistent) ? 1 : 0;
        if (want != ls->reject_on_device) {
            MNK_HIP(hipMemsetAsync(ls->info_dev.p + 2, want, sizeof(int), s));
            ls->reject_on_device = want;
        }
    }
    // The persistent kernels keep waiting workgroups resident.  Two of them from different contexts on the same CUs can
    // starve each other's diagonal strips (per-XCD dispatch order), so the persistent operations of one process take turns
    // on the device (the arbiter below); a wait that expires anyway (another PROCESS) falls back (mnk_ls_fetch_info) and the
    // persistent schedule is tried again 16, then 64, 256, ... factorizations later.
    ls->algo_now = ls->panel_algo;
    // (inside an open batch small systems take the task-DAG schedule too: their pivot chains run side by side, dag.hip)
    const int64_t min_rows = mnk_batch_active() ? std::min<int64_t>(ls->dag_min_rows, 256) : ls->dag_min_rows;
    if (ls->algo_now == 5 && (ctx->dag_cus < ls->dag_band || !ls->lookahead || Np < min_rows || Np > ls->dag_max_rows)) ls->algo_now = 4;
    if (ls->algo_now == 5 && ctx->sp_dag == nullptr && mnk_ctx_ensure_dag(ctx) != 0) { (void)hipGetLastError(); ls->algo_now = 4; }   // (released while idle)
    if (ls->algo_now == 5 && ctx->sp_dag == nullptr) ls->algo_now = 4;
    ++ls->fact_count;
    if (ls->pp_blocked && ls->fact_count >= ls->pp_retry_at) ls->pp_blocked = false;   // (a time-out may have been transient)
    if (ls->algo_now >= 4 && ls->pp_blocked) ls->algo_now = 1;
    // the task-DAG schedule's buffers (task list, progress words, V = L D): a device that cannot hold them keeps schedule 4
    if (ls->algo_now == 5 && mnk_ls_dag_prepare(ls) != 0) ls->algo_now = 4;
    if (ls->algo_now == 5 && ls->dag_js2 < (Np + 255) / 256) {   // (the deep-band stream pair: made with the first such factorization)
        if (mnk_ctx_ensure_dag2(ctx) != 0) (void)hipGetLastError();
        if (ctx->sp_dag2 == nullptr) ls->algo_now = 4;
    }
    // Persistent schedules of different contexts take turns on the device (common.h: mnk_persist_begin); round 3 sent every
    // solver to schedule 1 as soon as a second context was alive (12.4 instead of 9.3 ms at C3).
    // a batch of independent factorizations (dag.hip: mnk_factorize_batch_begin / _end) launches them together later
    if (ls->algo_now == 5 && mnk_batch_defer(ls)) return 0;
    return mnk_ls_run_factorization_now(ls);
}

int mnk_ls_run_factorization_now(mnk_ls* ls) {
    mnk_ctx* ctx = ls->ctx;
    hipStream_t s = ctx->stream;
    const bool persistent = ls->algo_now >= 4;
    if (persistent) {
        int rc0 = mnk_persist_begin(ctx, s);
        if (rc0) return rc0;
    }
    const int rc_run = run_factorization_body(ls);
    return persistent ? mnk_persist_end(ctx, s, rc_run) : rc_run;
}

static int run_factorization_body(mnk_ls* ls) {
    mnk_ctx* ctx = ls->ctx;
    hipStream_t s = ctx->stream;
    const int64_t Np = ls->Np, ld = ls->ld;
    const bool ldl = ls->algo == MNK_LDL;
    double* F = ls->fact.p;
    const int64_t NBO = mnk_ls_effective_nbo(ls);
    if (ls->algo_now != 5) MNK_HIP(hipMemsetAsync(ls->info_dev.p, 0, 2 * sizeof(int), s));   // (the task-DAG driver resets it with its flags)
    // Outer panel boundaries.  Once the remaining matrix is small the factorization is bound by the panel
    // chain, not by the update: narrower outer panels (tail_nbo) then drop the middle-level update and halve
    // the depth of the (a) piece the chain waits for (measured: 355 -> ~300 us per 512 columns of the tail).
    std::vector<int64_t> bnd{0};
    for (int64_t pos = 0; pos < Np;) {
        const bool tail = ls->lookahead && NBO < Np && ls->tail_rows > 0 && Np - pos <= ls->tail_rows && ls->tail_nbo < NBO;
        pos = std::min<int64_t>(pos + (tail ? ls->tail_nbo : NBO), Np);
        bnd.push_back(pos);
    }
    const int64_t npanel = (int64_t)bnd.size() - 1;
    const bool la = ls->lookahead && npanel > 1;

    if (ls->algo_now == 5) {
        int rc = mnk_ls_run_factorization_dag(ls);
        if (rc) return rc;
    } else if (!la) {
        for (int64_t ko = 0; ko < Np; ko += NBO) {
            const int64_t kend = std::min<int64_t>(ko + NBO, Np);
            int rc = factor_outer_panel(ls, s, ko, kend, ls->wbuf[0].p);
            if (rc) return rc;
            const int64_t Mt = Np - kend;
            if (Mt > 0) {
                const double* Wsrc = ldl ? ls->wbuf[0].p + kend : F + kend + ko * ld;
                rc = launch_gemm_nt(s, 2, Mt, Mt, kend - ko, Wsrc, ldl ? ls->ldw : ld, F + kend + ko * ld, ld,
                                    F + kend + kend * ld, ld, nullptr, nullptr, 0, ls->info_dev.p);
                if (rc) return rc;
            }
        }
    } else {
        // look-ahead: panel stream sp (high priority) factors panel k+1 while the update
        // stream su applies panel k to the rest of the trailing matrix.
        // One CU partition for every size: a quarter of the CUs for the panel stream.  (A second pair with an
        // eighth, for "update-bound" sizes, measured slower at every N once the panel stream's CUs join the
        // trailing update: N = 30 000 156 vs 178 ms, N = 11 192 +0.5 ms per step taken on it.)
        { int rc_s = mnk_ctx_ensure_panel_streams(ctx); if (rc_s) return rc_s; }   // (made with the first factorization that takes this schedule)
        hipStream_t sp = ctx->sp, su = ctx->su;
        if ((int64_t)ctx->ev_panel.size() < npanel + 1) {
            const size_t old = ctx->ev_panel.size();
            ctx->ev_panel.resize(npanel + 1);
            ctx->ev_next.resize(npanel + 1);
            ctx->ev_next2.resize(npanel + 1);
            ctx->ev_bdone.resize(npanel + 1);
            for (size_t e = old; e < ctx->ev_panel.size(); ++e) {
                MNK_HIP(hipEventCreateWithFlags(&ctx->ev_panel[e], hipEventDisableTiming));
                MNK_HIP(hipEventCreateWithFlags(&ctx->ev_next[e], hipEventDisableTiming));
                MNK_HIP(hipEventCreateWithFlags(&ctx->ev_next2[e], hipEventDisableTiming));
                MNK_HIP(hipEventCreateWithFlags(&ctx->ev_bdone[e], hipEventDisableTiming));
            }
        }
        // Work sharing: while the trailing update is the longer leg, the panel stream's CUs would
        // idle once panel k+1 is factored; (b) is then launched as a tile queue on both streams
        // (update stream right after (a), panel stream after the panel) and the two launches drain
        // one counter.  Estimated leg lengths only decide whether the second launch is worth it.
        const int pcus = ctx->panel_cus;
        const int ucus = pcus > 0 ? ctx->num_cu - pcus : ctx->num_cu;
        const bool share = ls->share != 0;
        if (share) {
            if (ls->tile_ctr.n < 8 * ((size_t)npanel + 1)) {
                int rc0 = ls->tile_ctr.alloc(8 * ((size_t)npanel + 1));
                if (rc0) return rc0;
            }
            MNK_HIP(hipMemsetAsync(ls->tile_ctr.p, 0, 8 * ((size_t)npanel + 1) * sizeof(int), s));
        }
        // Panel 0 has nothing to overlap with: it runs on the caller's stream, i.e. on the whole chip (its
        // triangular solves and inner updates are throughput-bound at this height), before the fork.
        int rc = 0;
        if (ls->panel0_whole) rc = factor_outer_panel(ls, s, 0, bnd[1], ls->wbuf[0].p);
        if (rc) return rc;
        MNK_HIP(hipEventRecord(ctx->ev_a, s));
        MNK_HIP(hipStreamWaitEvent(sp, ctx->ev_a, 0));
        MNK_HIP(hipStreamWaitEvent(su, ctx->ev_a, 0));
        if (!ls->panel0_whole) rc = factor_outer_panel(ls, sp, 0, bnd[1], ls->wbuf[0].p);
        if (rc) return rc;
        MNK_HIP(hipEventRecord(ctx->ev_panel[0], sp));
        for (int64_t k = 0; k + 1 < npanel; ++k) {
            const int64_t ko = bnd[k], kend = bnd[k + 1];
            const int64_t Mt = Np - kend;
            if (Mt <= 0) break;
            const int64_t nnext = bnd[k + 2] - kend;
            double* wk = ls->wbuf[k & 1].p;
            const double* Wsrc = ldl ? wk + kend : F + kend + ko * ld;
            const int64_t ldws = ldl ? ls->ldw : ld;
            const int64_t Kw = kend - ko;
            // ev_panel[k] is recorded after panel k and after the panel stream's share of (b)_{k-1}
            MNK_HIP(hipStreamWaitEvent(su, ctx->ev_panel[k], 0));
            // (a) columns of the next outer panel, delivered in two pieces: its first 256-column middle
            // panel (the panel stream starts on it at once), then the remaining columns (needed only when
            // that middle panel is finished)
            auto update_a = [&](hipStream_t st, int64_t c0, int64_t c1) -> int {
                const int64_t Mp = Mt - c0, Nn = c1 - c0;
                double* Cp = F + (kend + c0) + (kend + c0) * ld;
                if (gemm_nt_lower_tiles(Mp, Nn) < ls->small_tiles)
                    return launch_gemm_nt_lower_small(st, Mp, Nn, Kw, Wsrc + c0, ldws, F + kend + c0 + ko * ld, ld, Cp,
                                                      ld, ls->info_dev.p);
                return launch_gemm_nt(st, 2, Mp, Nn, Kw, Wsrc + c0, ldws, F + kend + c0 + ko * ld, ld, Cp, ld, nullptr,
                                      nullptr, 0, ls->info_dev.p);
            };
            // split_a == 2: the first 64 columns of the next panel are updated on the PANEL stream itself, right
            // behind panel k (no cross-stream hand-over before the next pivot block starts); the update stream
            // delivers the other columns while that block is being factored.  split_a == 1: first 256 columns
            // on the update stream, then the rest.
            // (the 256-column panel step needs the first 256 columns at once: they come from the update stream)
            // Few rows left (the pivot chain is the critical path): the first persistent launch of the next panel applies
            // panel k to its own 256 columns itself (ppanel_kernel's prologue) -- no (a) kernel and no cross-stream
            // hand-over in front of the chain; the update stream delivers the other columns of the panel meanwhile.
            const bool fuse_a = ls->algo_now == 4 && ls->pp_fuse_rows > 0 && Mt <= ls->pp_fuse_rows && Kw <= 512;
            const bool own_first = !fuse_a && ls->split_a == 2 && nnext > NBI;
            const int64_t n1 = own_first ? std::min<int64_t>(ls->own_cols, nnext - NBI) : std::min<int64_t>(256, nnext);
            const bool split_a = (ls->split_a || fuse_a) && nnext > n1;
            if (fuse_a) {
                if (k > 0) MNK_HIP(hipStreamWaitEvent(sp, ctx->ev_bdone[k - 1], 0));  // (b)_{k-1} touched these columns
                if (split_a) {
                    rc = update_a(su, n1, nnext);
                    if (rc) return rc;
                    MNK_HIP(hipEventRecord(ctx->ev_next2[k], su));
                }
            } else if (own_first) {
                if (k > 0) MNK_HIP(hipStreamWaitEvent(sp, ctx->ev_bdone[k - 1], 0));  // (b)_{k-1} touched these columns
                rc = update_a(sp, 0, n1);
                if (rc) return rc;
                rc = update_a(su, n1, nnext);
                if (rc) return rc;
                MNK_HIP(hipEventRecord(ctx->ev_next2[k], su));
            } else {
                rc = update_a(su, 0, split_a ? n1 : nnext);
                if (rc) return rc;
                MNK_HIP(hipEventRecord(ctx->ev_next[k], su));
                if (split_a) {
                    rc = update_a(su, n1, nnext);
                    if (rc) return rc;
                    MNK_HIP(hipEventRecord(ctx->ev_next2[k], su));
                }
            }
            // (b) the rest of the trailing matrix
            const int64_t Mb = Mt - nnext;
            const int64_t Nb = Mb;
            bool shared_b = false;
            if (Mb > 0 && Nb > 0) {
                const int ntiles = gemm_nt_lower_tiles(Mb, Nb);
                // CU-us per 128x128xKw tile at ~80 % of the MFMA rate; ~55 us of panel stream per 64 columns
                const double t_upd = ntiles * (68.0 * (double)Kw / 512.0) / ucus;
                const double t_pan = 55.0 * (double)nnext / 64.0 * (pcus > 0 ? 64.0 / pcus : 1.0);
                shared_b = share && pcus > 0 && (t_upd > t_pan || ls->share == 2);
                if (shared_b)
                    rc = launch_gemm_nt_queue(su, Mb, Nb, Kw, Wsrc + nnext, ldws, F + kend + nnext + ko * ld, ld,
                                              F + (kend + nnext) + (kend + nnext) * ld, ld, ls->tile_ctr.p + 8 * k,
                                              ucus, ls->info_dev.p);
                else
                    rc = launch_gemm_nt(su, 2, Mb, Nb, Kw, Wsrc + nnext, ldws, F + kend + nnext + ko * ld, ld,
                                        F + (kend + nnext) + (kend + nnext) * ld, ld, nullptr, nullptr, 0,
                                        ls->info_dev.p);
                if (rc) return rc;
            }
            MNK_HIP(hipEventRecord(ctx->ev_bdone[k], su));
            // panel k+1 on the panel stream, as soon as (a) is done
            if (!own_first && !fuse_a) MNK_HIP(hipStreamWaitEvent(sp, ctx->ev_next[k], 0));
            rc = factor_outer_panel(ls, sp, kend, kend + nnext, ls->wbuf[(k + 1) & 1].p,
                                    split_a ? ctx->ev_next2[k] : nullptr, own_first ? n1 : 256,
                                    fuse_a ? (ldl ? wk : F + ko * ld) : nullptr, ldws, fuse_a ? Kw : 0);
            if (rc) return rc;
            if (shared_b) {
                rc = launch_gemm_nt_queue(sp, Mb, Nb, Kw, Wsrc + nnext, ldws, F + kend + nnext + ko * ld, ld,
                                          F + (kend + nnext) + (kend + nnext) * ld, ld, ls->tile_ctr.p + 8 * k,
                                          pcus, ls->info_dev.p);
                if (rc) return rc;
            }
            MNK_HIP(hipEventRecord(ctx->ev_panel[k + 1], sp));
        }
        MNK_HIP(hipEventRecord(ctx->ev_a, sp));
        MNK_HIP(hipEventRecord(ctx->ev_b, su));
        MNK_HIP(hipStreamWaitEvent(s, ctx->ev_a, 0));
        MNK_HIP(hipStreamWaitEvent(s, ctx->ev_b, 0));
    }
    // inverses of the diagonal blocks for the solves (batched, off the critical path of the panels); the task-DAG schedule
    // has already inverted the strip-columns below ls->inv_done on its bulk stream, under the chain's last steps
    {
        int rc = mnk_ls_invert_blocks(ls, s, ls->inv_done, (Np + 255) / 256);
        if (rc) return rc;
    }
    ls->factorized = true;
    ls->info_valid = false;
    ls->bk_active = false;
    {   // inertia / growth words / info go to the pinned host words right behind the factorization (mnk_ls_fetch_info only waits)
        int rc = mnk_ls_launch_finish_info(ls, s);
        if (rc) return rc;
    }
    return mnk_ls_prefill_spare(ls);
}

void mnk_ls_fill_small_rec(mnk_ls* ls, mnk::SmallSysRec* rec) {
    const bool lmode = ls->algo == MNK_LDL;
    const int64_t ntile = ls->Np / 128;
    rec->flags = ls->dag_flags.p;
    rec->nflags = 2 + ls->Np / NBI + 2 * ntile * ntile;
    rec->info = ls->info_dev.p;
    rec->F = ls->fact.p; rec->ld = ls->ld; rec->dblk = ls->dblk.p; rec->linv = ls->linv.p;
    rec->linv256 = ls->linv256.p; rec->linv256t = ls->linv256t.p; rec->Np = ls->Np;
    rec->dvec = ls->dvec.p; rec->ninertia = lmode ? ls->N : 0; rec->pin_dev = ls->pin_dev;
    rec->amax = mnk_ls_growth_word(ls) != nullptr ? ls->amax_dev.p : nullptr;
}

int mnk_ls_invert_blocks_batch(hipStream_t s, bool ldl, const mnk::SmallSysRec* recs_dev, int n, int64_t Np) {
    const unsigned nb = (unsigned)(Np / NBI);
    if (ldl) hipLaunchKernelGGL(linv64_kernel<true>, dim3(nb, 1, (unsigned)n), dim3(256), 0, s, (double*)nullptr, (int64_t)0, (const double*)nullptr,
                                (double*)nullptr, (const int*)nullptr, 0, recs_dev);
    else hipLaunchKernelGGL(linv64_kernel<false>, dim3(nb, 1, (unsigned)n), dim3(256), 0, s, (double*)nullptr, (int64_t)0, (const double*)nullptr,
                            (double*)nullptr, (const int*)nullptr, 0, recs_dev);
    MNK_HIP(hipGetLastError());
    return mnk_ls_build_inverses_batch(s, recs_dev, n, Np);
}

int mnk_ls_finish_info_batch(hipStream_t s, const mnk::SmallSysRec* recs_dev, int n, int threads) {
    hipLaunchKernelGGL(finish_info_kernel, dim3((unsigned)n), dim3(threads), 0, s, (const double*)nullptr, (int64_t)0, (const int*)nullptr,
                       (unsigned long long*)nullptr, (const unsigned long long*)nullptr, recs_dev);
    MNK_HIP(hipGetLastError());
    return 0;
}

int mnk_ls_launch_finish_info(mnk_ls* ls, hipStream_t s) {
    const bool lmode = ls->algo == MNK_LDL;
    const int threads = lmode ? (ls->N >= 4096 ? 1024 : 256) : 64;
    hipLaunchKernelGGL(finish_info_kernel, dim3(1), dim3(threads), 0, s, ls->dvec.p, lmode ? ls->N : (int64_t)0, ls->info_dev.p, ls->pin_dev,
                       mnk_ls_growth_word(ls) != nullptr ? ls->amax_dev.p : (const unsigned long long*)nullptr);
    MNK_HIP(hipGetLastError());
    // mnk_ls_fetch_info waits for THIS point, not for the stream: whatever the caller has queued behind the factorization
    // (the solves of other instances of a batch, the next assembly) does not delay the inertia
    if (!ls->ev_info) MNK_HIP(hipEventCreateWithFlags(&ls->ev_info, hipEventDisableTiming));
    MNK_HIP(hipEventRecord(ls->ev_info, s));
    ls->ev_info_recorded = true;
    return 0;
}

int mnk_ls_right_trsm_rows(mnk_ls* ls, hipStream_t s, int64_t j0, double* Xrows, double* Vrows, int64_t ldr, int64_t nrows) {
    MNK_REQUIRE(ls->factorized && !ls->bk_active && nrows % 16 == 0 && j0 % NBI == 0 && j0 < ls->Np,
                "mnk_ls_right_trsm_rows: needs a static-pivot factor, 16-row multiples and a block-aligned column");
    // the kernel addresses rows j0 + 64 ... of ONE matrix: shift the row blocks so that its row r0 is their row 0
    double* Fp = Xrows - (j0 + NBI);
    double* Wp = Vrows ? Vrows - (j0 + NBI) : nullptr;
    const unsigned grid = (unsigned)((nrows / 16 + 3) / 4);
    const double* dblk = ls->dblk.p + (j0 / NBI) * 4096;
    const double* inv16 = ls->inv16.p + (j0 / NBI) * 1024;
    if (ls->algo == MNK_LDL)
        hipLaunchKernelGGL((trsm64_mfma_kernel<true, 1>), dim3(grid), dim3(256), 0, s, Fp, ldr, j0, j0 + NBI + nrows, dblk,
                           inv16, ls->dinv.p, Wp, ldr, (int64_t)j0, ls->info_dev.p);
    else
        hipLaunchKernelGGL((trsm64_mfma_kernel<false, 1>), dim3(grid), dim3(256), 0, s, Fp, ldr, j0, j0 + NBI + nrows, dblk,
                           inv16, ls->dinv.p, (double*)nullptr, (int64_t)0, (int64_t)0, ls->info_dev.p);
    MNK_HIP(hipGetLastError());
    return 0;
}

int mnk_launch_trsm64_batch(hipStream_t s, bool ldl, const mnk::TrsmBatchRec* recs_dev, int nbatch, int64_t j0, int64_t nrows,
                            int64_t ldr) {
    if (nbatch <= 0 || nrows <= 0) return 0;
    MNK_REQUIRE(nrows % 16 == 0 && j0 % NBI == 0, "mnk_launch_trsm64_batch: 16-row multiples and a block-aligned column");
    const dim3 grid((unsigned)((nrows / 16 + 3) / 4), (unsigned)nbatch);
    if (ldl) hipLaunchKernelGGL(trsm64_mfma_batch_kernel<true>, grid, dim3(256), 0, s, recs_dev, ldr, j0, nrows);
    else hipLaunchKernelGGL(trsm64_mfma_batch_kernel<false>, grid, dim3(256), 0, s, recs_dev, ldr, j0, nrows);
    MNK_HIP(hipGetLastError());
    return 0;
}

// Tier 2 of BUNCHKAUFMAN: fetch the matrix again, factor it with pivoting (bk.hip), rebuild the inverses the solves use.
static int bk_fallback(mnk_ls* ls) {
    hipStream_t s = ls->ctx->stream;
    int rc = ls->retransfer();
    if (rc) return rc;
    rc = mnk_ls_run_bunchkaufman(ls);
    if (rc) return rc;
    rc = mnk_ls_invert_blocks(ls, s, 0, (ls->Np + 255) / 256);
    if (rc) return rc;
    ++ls->bk_count;
    return mnk_ls_prefill_spare(ls);   // (the second transfer swapped the factor buffers again: the spare one holds the discarded static factor)
}

int mnk_ls_fetch_info(mnk_ls* ls) {
    { int rc_d = mnk_ls_sync_deferred(ls); if (rc_d) return rc_d; }
    if (ls->info_valid) return 0;
    hipStream_t s = ls->ctx->stream;
    if (ls->bk_active) {
        MNK_HIP(hipMemsetAsync(ls->inertia_dev.p, 0, 3 * sizeof(unsigned long long), s));
        // reference rule for a Bunch-Kaufman factor (src/LinearSolvers/lapack.jl:240-268): numzero = info > 0,
        // numneg from the 1x1 / 2x2 blocks of D (-1 if a block is exactly singular), numpos the rest
        int rc = mnk_ls_bk_inertia(ls, ls->inertia_dev.p);
        if (rc) return rc;
        hipLaunchKernelGGL(publish_info_kernel, dim3(1), dim3(64), 0, s, ls->inertia_dev.p, ls->info_dev.p, ls->pin_dev);
        MNK_HIP(hipGetLastError());
        MNK_HIP(mnk::stream_wait(s));
        volatile unsigned long long* pw = ls->pin;
        unsigned long long h[3] = {pw[0], pw[1], pw[2]};
        int hinfo = (int)(long long)pw[3];
        ls->info = hinfo;
        ls->nneg = h[1] > 0 ? -1 : (int64_t)h[0];
        ls->nzero = hinfo > 0 ? 1 : 0;
        ls->npos = ls->N - ls->nneg - ls->nzero;
        ls->info_valid = true;
        return 0;
    }
    // growth guard of the static-pivot tier (BUNCHKAUFMAN): max|a_ij| was recorded when the matrix was transferred
    const bool guard = ls->algo == MNK_LDL && ls->bk_requested && ls->bk_fallback && ls->retransfer && ls->amax_dev.p != nullptr;
    // (finish_info_kernel, queued behind the factorization, has stored everything in the pinned words)
    if (ls->ev_info_recorded) MNK_HIP(hipEventSynchronize(ls->ev_info));
    else MNK_HIP(mnk::stream_wait(s));
    volatile unsigned long long* pw = ls->pin;
    unsigned long long h[3] = {pw[0], pw[1], pw[2]};
    int hinfo = (int)(long long)pw[3];
    if (guard) {
        double am, dm;
        const unsigned long long wa = pw[4], wd = pw[5];
        memcpy(&am, &wa, sizeof am);
        memcpy(&dm, &wd, sizeof dm);
        ls->last_growth = am > 0.0 ? dm / am : (dm > 0.0 ? HUGE_VAL : 0.0);
        ls->last_sign_changes = (int64_t)pw[6];
    }
    if (hinfo == -7) ls->last_timeout_site = (int)(long long)pw[7];
    if (hinfo == -7 && ls->dag_debug && ls->dag_flags.p && ls->dag_dbg.p) {
        // diagnostics: the schedule's progress words and what the chain's strips were waiting for, as the time-out left them
        MNK_HIP(mnk::stream_wait(s));
        const int64_t ntile = ls->Np / 128;
        ls->dbg_flags.resize((size_t)(2 + ls->Np / NBI + 2 * ntile * ntile));
        ls->dbg_chain.resize(ls->dag_dbg.n);
        MNK_HIP(mnk::d2h_copy(ls->dbg_flags.data(), ls->dag_flags.p, ls->dbg_flags.size() * sizeof(int), s));
        MNK_HIP(mnk::d2h_copy(ls->dbg_chain.data(), ls->dag_dbg.p, ls->dbg_chain.size() * sizeof(int), s));
    }
    if (hinfo != 0 && ls->spare_by_dag) {
        // The spare factor buffer was to be zeroed by the DAG_FILL tasks of this factorization's queue -- which its
        // workgroups dropped when `info` became non-zero.  Found as an intermittent wrong matrix behind a time-out: the redo
        // below swapped the half-zeroed buffer (the previous factor) in and scattered the sparse entries over it (1 in ~40
        // runs of tests/test_hip_c5.py once the waits' bound had come down from 3 s to 0.3 s; tools/c5_loop.py).
        ls->spare_zeroed = false;
        ls->spare_by_dag = false;
    }
    if (hinfo == -7 && ls->algo_now >= 4 && ls->retransfer) {
        // the persistent panel kernel gave up on a dependency (CUs shared with another process' persistent kernels):
        // factor again with one launch per panel piece, and stay there for a while (16, 64, 256, ... factorizations)
        ls->pp_blocked = true;
        ls->pp_retry_at = ls->fact_count + ls->pp_backoff + 1;   // (+1: the redo below counts)
        ls->pp_backoff *= 4;
        ++ls->pp_fallbacks;
        {   // what the expired wait cost: from the launch of that factorization to this moment (the redo below comes on top)
            const double lost = std::max(0.0, mnk_host_ms() - ls->t_fact_launch_ms);
            ls->stall_ms_total += lost;
            mnk_add_process_stall_ms(lost);
        }
        int rc = ls->retransfer();
        if (rc) return rc;
        rc = mnk_ls_run_factorization(ls);
        if (rc) return rc;
        return mnk_ls_fetch_info(ls);
    }
    ls->factor_invalid = false;
    if (hinfo == -9) {
        // EARLY REJECTION (option early_reject with accept_only_pd; leaf64.h): the static-pivot LDL^T met a pivot that is not
        // positive and every kernel behind that diagonal block dropped its work.  The matrix is not positive definite -- all the
        // caller asked.  Reported: the signs of the pivots that were computed, everything behind them counted as negative
        // (num_neg >= 1 is the statement; the reference's Cholesky path reports a failed factorization as (0, n, 0) in the
        // same spirit, lapack_common.jl:96-98).  The factor is not usable: solve / get_factor refuse it.
        ls->info = 1;
        ls->npos = (int64_t)h[0];
        ls->nzero = (int64_t)h[1];
        ls->nneg = ls->N - ls->npos - ls->nzero;
        ls->factor_invalid = true;
        ++ls->early_rejects;
        ls->early_reject_col = (int64_t)(long long)pw[7];
        ls->info_valid = true;
        // (history of the leading-block probe, ls.h)
        ls->probe_last_rejected = true;
        if (2 * ls->early_reject_col < ls->N) { ls->probe_hint_col = ls->early_reject_col; ls->probe_since_hint = 0; }
        else if (ls->probe_since_hint < (1 << 20)) ++ls->probe_since_hint;
        return 0;
    }
    if (hinfo < 0) {
        // a bounded device-side wait expired and there is no way to redo the factorization (no source to transfer again)
        set_error("factorize!: a device-side hand-off timed out (info = %d) and the matrix cannot be transferred again; the "
                  "factor is invalid -- set panel_algo = 1", hinfo);
        return -4;
    }
    ls->info = hinfo;
    if (ls->algo == MNK_LDL) {
        ls->npos = (int64_t)h[0];
        ls->nzero = (int64_t)h[1];
        ls->nneg = (int64_t)h[2];
        if (ls->nzero > 0 && ls->info == 0) ls->info = 1;  // LAPACK-style "singular D" signal
        // (pivots that are all positive and then all negative: a positive definite leading block and a negative definite Schur
        // complement, the quasi-definite structure of the KKT systems, whose growth |J|^2 / lambda_min(H) is in the data)
        const double gtol = ls->last_sign_changes <= 1 ? ls->bk_growth_tol_qd : ls->bk_growth_tol;
        // accept_only_pd: the caller's inertia test accepts positive definite matrices only (the condensed KKT systems
        // without equality rows, reference src/KKT/Sparse/condensed.jl:138-140).  A zero or negative pivot of the unpivoted
        // LDL^T PROVES that the matrix is not positive definite -- the pivoted tier could only confirm a rejection, at
        // 0.47 s for N = 11 192 -- so the static tier's counts are reported as they are.
        const bool verdict_final = ls->accept_only_pd && (ls->nzero > 0 || ls->nneg > 0);
        if (!verdict_final && (ls->nzero > 0 || (guard && !(ls->last_growth <= gtol))) && ls->bk_requested && ls->bk_fallback && ls->retransfer) {
            // the static-pivot factorization broke down (a zero pivot) or grew (tiny pivots: the Schur complements left the
            // scale of the matrix) on a matrix that is not quasi-definite in the given order: BUNCHKAUFMAN means dsytrf
            // semantics, so factor it again with 1x1 / 2x2 pivoting
            int rc = bk_fallback(ls);
            if (rc) return rc;
            return mnk_ls_fetch_info(ls);
        }
    } else {
        // inertia_cholesky, reference src/LinearSolvers/lapack_common.jl:96-98
        if (hinfo == 0) { ls->npos = ls->N; ls->nzero = 0; ls->nneg = 0; }
        else { ls->npos = 0; ls->nzero = ls->N; ls->nneg = 0; }
    }
    ls->probe_last_rejected = false;   // (history of the leading-block probe: a verdict that is not an early rejection)
    if (ls->probe_since_hint < (1 << 20)) ++ls->probe_since_hint;
    ls->info_valid = true;
    return 0;
}

