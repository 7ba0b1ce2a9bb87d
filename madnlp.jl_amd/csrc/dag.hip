// Task-DAG schedule of the blocked factorization (panel_algo = 5): the bulk of `factorize!` (reference
// src/LinearSolvers/lapack_common.jl:54-66 -> dsytrf / dpotrf, src/LinearSolvers/lapack.jl:145-148,164-167) as ONE
// persistent, LEFT-LOOKING tile kernel that runs beside the pivot chain instead of one trailing-update launch per outer
// panel.
//
// Units: block = 64 columns, tile = 128 x 128, strip-column = 256 columns (one persistent panel launch of factor.hip).
//   * chain (factor.hip, ppanel_kernel, panel stream, a few CUs): for every strip-column Js the eight 64-row strips of the
//     BAND -- its four diagonal strips and the four below them, which are the diagonal strips of Js + 1.  A strip applies
//     the previous strip-column itself (left-looking prologue, K = 256) and then runs the right-looking pivot chain.
//   * bulk (this file, update stream, all other CUs): every 128 x 128 tile (I, J) below the band receives
//     C(I, J) -= sum_{k < 128 J} L(I, k) d_k L(J, k)^T  left-looking, in CHUNKS of tile columns of k: one task = one chunk
//     accumulated in registers (the k-loop of the MFMA tile kernel) and applied to the tile in memory; the task of the
//     last tile column also finalizes the tile in place, X = C L_JJ^-T (D^-1) against the two diagonal blocks of tile
//     column J, and publishes the rows.  The band's own tiles are accumulated the same way up to the columns the chain's
//     prologue covers (BANDACC tasks).  (One task per tile over the whole depth -- the tile written once -- was built
//     first: a task that has caught up with the pivot chain then holds its workgroup slot while it advances one tile
//     column per chain step, and the slots doing catch-up work were too few: 63 % of the slot-time computing.)
//   * order: tasks are drawn from ONE queue sorted by the chain position at which they become ready (the last tile
//     column they read), band tiles and rows next to the band first: a drawn task can start at once or nearly so, and a
//     workgroup that holds a task only ever waits for tasks drawn before it, so any number of resident workgroups makes
//     progress.
//   * progress: front[t] = number of leading 128-column tile columns for which the 64-row strip t of L is final
//     (monotone; zeroed per factorization); af[] = "band tile accumulated"; tprog[I, J] = chunks applied to tile (I, J).
//     Producer: stores -> barrier -> one lane's
//     agent-scope release -> drained relaxed store; consumer: relaxed polls by one wave, one agent-scope acquire, barrier.
//     Every wait is bounded (info = -7 -> the host redoes the factorization with the launch-per-piece schedule).
// There is no trailing-update launch per outer panel any more and no barrier between the updates of different panels: the
// look-ahead is as deep as the queue -- the pivot chain never waits for a bulk update of an older panel.
// LDL^T: V = L D is kept in a second N x N array (written once per element, next to L): scaling the B operand's k-columns
// by d_k inside the k-loop instead was built first and cost 16 % of the loop (fp64 VALU work between the fp64 MFMAs).
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <climits>
#include <vector>

#include "gemm_tile.h"
#include "ls.h"

namespace mnk {

// What a bulk task needs to know about the factorization it belongs to.  One launch may serve SEVERAL independent
// factorizations of the same order (mnk_factorize_batch_*: the task lists of the instances are merged into one queue, each
// task carries its instance index), so these live in a per-instance record instead of the kernel's argument block.
// Diagonal tiles are accumulated in SUBTRACT order (gemm_tile.h: gemm_nt_load_neg_lower): C - t_1 - t_2 ... with the tile in
// the accumulators from the first chunk on, instead of a sum of products that grows from zero and is subtracted at the end.
// Both are backward stable; the difference shows on condensed KKT matrices whose pivots are the difference of numbers of
// size 1e13 that agree to 15 digits (AC-OPF case1354): the pivot chain saw -0.031 where the exact
// value is +0.0195, the interior-point run needed 10 Richardson steps per solve and left LAPACK's trajectory at iteration 9;
// with the subtract order it follows it to three digits through all 20 iterations (47 -> 30 back-solves).  Cost: a tile's
// chunks run one after the other (+0.5 % at N = 11 192, tools/diag_sub_ab.sh).  0 = the old order (A/B builds).
#ifndef MNK_DAG_DIAG_SUB
#define MNK_DAG_DIAG_SUB 1
#endif
struct DagInst {
    double* F;
    int64_t ld;
    double* V;            // LDL^T: V = L D, same layout as F (the B operand of the updates); Cholesky: nullptr (V = L)
    const double* dinv;   // 1 / d_k
    const double* dblk;   // factored 64x64 diagonal blocks
    const double* inv16;  // inverses of their 16x16 diagonal sub-blocks
    int* front;
    int* af;
    int* tprog;           // [I * ntile + J]: chunks of tile (I, J) that are in memory
    int* info;
    unsigned long long* vmax;   // growth monitor of the static-pivot LDL^T (factor.hip growth_fold): receives max|V|, or NULL
    double* zfill;        // DAG_FILL tasks: the buffer that becomes the NEXT factorization's zeroed factor buffer (or NULL)
    int64_t N;            // order of the matrix (rows / columns N .. Np - 1 are padding: unit diagonal)
};

// (read through the constant address space: the fields are uniform and never written while a bulk kernel runs, so every use
// is a scalar load that can be repeated instead of a value that must stay live -- a by-value copy of the record in the task
// loop cost 88 spilled registers and a scratch access inside the K-loop)
typedef const DagInst __attribute__((address_space(4))) * DagInstP;

struct DagArgs {       // (the batch kernel's; a single factorization: DagArgs1 / dag_bulk_kernel1 below)
    const DagInst* insts; // the instance records in device memory (written by dag_reset_kernel), indexed by the task's instance field
    const int4* tasks;    // (flags | chunk index << 8, I | instance << 16, J, kbeg | kend << 16)
    int ntasks;
    int ntile;
    int* qctr;
    long spin_limit;
    unsigned long long* trace;  // diagnostics (option dag_trace): 8 time stamps per task
    unsigned long long* wgstat; // diagnostics: per workgroup {first grab, exit, ticks waited, tasks, ticks in finalize}
    int fake_share;             // DIAGNOSTIC (env MNK_DAG_FAKE_SHARE, wrong results): every chunk reads the operand rows of tile rows 0..n-1
};

#ifndef MNK_DIAG_BULK_DBG
#define MNK_DIAG_BULK_DBG 0   // DIAGNOSTIC build (tools/stall_hunt.py): every workgroup of dag_bulk_kernel1 records its task and its long waits
#endif
constexpr int DAG_BANDACC = 1, DAG_FINAL = 2, DAG_FIRST = 4, DAG_FILL = 8;  // task flags

// Wave 0 waits until min(front[s0..s3]) > c and returns that minimum (clamped to kend): tile columns [c, ret) are final
// for all four strips.  -1: the factorization failed elsewhere or the wait expired.  Ends with an acquire + barrier.
template <class IP>
__device__ __forceinline__ int dag_wait_front(IP a, long spin_limit, int s0, int s1, int s2, int s3, int c, int kend, int* s_val, int* d8 = nullptr) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const int idx = lane == 0 ? s0 : (lane == 1 ? s1 : (lane == 2 ? s2 : s3));
        long spins = 0;
        int r;
        for (;;) {
            int f = lane < 4 ? __hip_atomic_load(a->front + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : INT_MAX;
#if MNK_DIAG_BULK_DBG
            const int f_raw = f;
#endif
            f = min(f, __shfl_xor(f, 1));
            f = min(f, __shfl_xor(f, 2));
            r = __builtin_amdgcn_readfirstlane(f);
            if (r > c) break;
            __builtin_amdgcn_s_sleep(4);
            if ((++spins & 255) == 0) {
#if MNK_DIAG_BULK_DBG
                if (d8 != nullptr && (spins & 0x3ffff) == 0) {
                    if (lane < 4) d8[8 + lane] = f_raw;   // (the four words as this workgroup sees them)
                    if (lane == 0) { d8[1] = 2; d8[2] = s0 | (s2 << 16); d8[3] = c; d8[4] = r; d8[5] = (int)(spins >> 18); }
                }
#endif
                if (__hip_atomic_load(a->info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
#if MNK_DIAG_BULK_DBG
                    if (d8 != nullptr && lane == 0) { const unsigned long long now = wall_clock64(); d8[1] |= 16; d8[14] = (int)(unsigned)now; d8[15] = (int)(unsigned)(now >> 32); }
#endif
                    r = -1;
                    break;
                }
                if (spins > spin_limit) {
                    if (lane == 0 && atomicCAS(a->info, 0, -7) == 0) a->info[1] = 1;   // (site 1: a bulk task waiting for operand rows)
                    r = -1;
                    break;
                }
            }
        }
        if (r > kend) r = kend;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (lane == 0) *s_val = r;
    }
    __syncthreads();
    const int r = *s_val;
    __syncthreads();
    return r;
}

// Wave 0 waits until *w0 >= t0 and *w1 >= t1 (progress words of the pivot chain); false: failed / expired.  Acquire + barrier.
template <class IP>
__device__ __forceinline__ bool dag_wait_words(IP a, long spin_limit, const int* w0, int t0, const int* w1, int t1, int* s_val, int* d8 = nullptr) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        long spins = 0;
        int ok = 1;
        for (;;) {
            const int v = lane < 2 ? __hip_atomic_load(lane == 0 ? w0 : w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : INT_MAX;
            if (__all(v >= (lane == 0 ? t0 : (lane == 1 ? t1 : INT_MIN)))) break;
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 255) == 0) {
#if MNK_DIAG_BULK_DBG
                if (d8 != nullptr && (spins & 0x3ffff) == 0 && lane == 0) { d8[1] = 3; d8[2] = 0; d8[3] = t0; d8[4] = v; d8[5] = (int)(spins >> 18); }
#endif
                if (__hip_atomic_load(a->info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
#if MNK_DIAG_BULK_DBG
                    if (d8 != nullptr && lane == 0) { const unsigned long long now = wall_clock64(); d8[1] |= 16; d8[14] = (int)(unsigned)now; d8[15] = (int)(unsigned)(now >> 32); }
#endif
                    ok = 0;
                    break;
                }
                if (spins > spin_limit) {
                    if (lane == 0 && atomicCAS(a->info, 0, -7) == 0) a->info[1] = 2;   // (site 2: chunk order of a tile)
                    ok = 0;
                    break;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (lane == 0) *s_val = ok;
    }
    __syncthreads();
    const int ok = *s_val;
    __syncthreads();
    return ok != 0;
}

// In-place finalization of a bulk tile against the two factored diagonal blocks (ja, jb = ja + 1) of its tile column:
// the block substitution of trsm64_mfma_kernel (factor.hip) for the 64 columns of block ja, the K = 64 update of the
// columns of block jb with L(jb, ja), the substitution for block jb.  Wave w owns the 16-row strips 2w, 2w + 1 of the
// tile (C^T layout: register r of column block cb at lane (l15, l4) is C[row + l15][16 cb + l4 + 4 r]) and runs their two
// independent MFMA chains interleaved.  The operands every wave needs -- the strictly lower 16x16 blocks and the 16x16
// inverses of a diagonal block, then L(jb, ja) -- are staged ONCE per workgroup through the k-loop's LDS tiles in MFMA
// A-operand order (one v4 per lane and block: M[l15][l4 + 4 s], s = 0..3): fed straight from global memory the chains
// waited for one L2 round trip per MFMA (90 us per tile measured, on the critical path of every row of tiles; ~30 us
// staged).  (Starting the block-ja part before D_jb is published, with all operands prefetched into registers, was built
// and measured slower: the 168-register budget of the k-loop forces the strips to run one after the other.)
// X: the tile in that layout, X[cb][ns] = column block cb (0..7) of strip ns -- the accumulators of gemm_nt_mainloop3<2, 8>.
template <bool LDL, class IP>
__device__ __forceinline__ void dag_finalize_tile(IP a, int64_t row0, int64_t col0, int ja, char* smem_raw, int tid,
                                                  v4f64 (&X)[8][2], unsigned long long* tr) {
    // trace: six 16-bit stage times (ticks of 10 ns since entry) packed into tr[6] (stages 1..4) and tr[7] (5, 6)
    const unsigned long long t_in = tr ? wall_clock64() : 0;
    auto stamp = [&](int stage) {
        if (tr) {
            const unsigned long long d = (wall_clock64() - t_in) & 0xffff;
            if (stage < 4) tr[6] = (stage == 0 ? 0 : tr[6]) | d << (16 * stage);
            else tr[7] = (stage == 4 ? 0 : tr[7]) | d << (16 * (stage - 4));
        }
    };
    v4f64* S = reinterpret_cast<v4f64*>(smem_raw);  // up to 16 blocks x 64 lanes (32 KB of the 36 KB)
    double* F = a->F;
    const int64_t ld = a->ld;
    const int lane = tid & 63, w = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    double* Fs = F + (row0 + 32 * w + l15) + (col0 + l4) * ld;  // this lane's element (row of strip 0, column l4)
    double* Vs = LDL ? a->V + (row0 + 32 * w + l15) + (col0 + l4) * ld : nullptr;
    double vm = 0.0;  // max|V| of this wave's strips (growth monitor)

    // blocks 0..5: -L_kk[cb, ib] for (cb, ib) = (1,0) (2,0) (2,1) (3,0) (3,1) (3,2); blocks 6..9: inv(L_kk[cb, cb])
    auto fill_diag = [&](int jk) {
        const double* __restrict__ Dk = a->dblk + (int64_t)jk * 4096;
        const double* __restrict__ Iv = a->inv16 + (int64_t)jk * 1024;
        for (int slot = tid; slot < 640; slot += 256) {
            const int q = slot >> 6, i = slot & 15, k4 = (slot >> 4) & 3;
            v4f64 v;
            if (q < 6) {
                const int cb = q == 0 ? 1 : (q < 3 ? 2 : 3), ib = q == 0 ? 0 : (q < 3 ? q - 1 : q - 3);
                const double* src = Dk + (16 * cb + i) + 64 * (16 * ib + k4);
#pragma unroll
                for (int sx = 0; sx < 4; ++sx) v[sx] = -src[64 * 4 * sx];
            } else {
                const double* src = Iv + (q - 6) * 256 + i + 16 * k4;
#pragma unroll
                for (int sx = 0; sx < 4; ++sx) v[sx] = src[16 * 4 * sx];
            }
            S[slot] = v;
        }
        // D^-1 of the block's 64 columns rides along (LDL^T): read back four at a time in front of the stores -- a global
        // load behind a possibly aliasing store would wait for it, and 16 preloaded values do not fit the registers
        if (LDL && tid < 64) reinterpret_cast<double*>(S + 640)[tid] = a->dinv[(int64_t)64 * jk + tid];
    };
    // X <- X L_kk^-T for both strips; stores L (LDL^T: V D^-1, and V next to it), keeps V in X
    auto trsm = [&](auto half, int coff) {
        constexpr int h4 = 4 * decltype(half)::value;
        const double* Sd = reinterpret_cast<const double*>(S + 640) + l4;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            v4f64 t0 = X[h4 + cb][0], t1 = X[h4 + cb][1];
#pragma unroll
            for (int ib = 0; ib < cb; ++ib) {
                const v4f64 aop = S[(cb * (cb - 1) / 2 + ib) * 64 + lane];
#pragma unroll
                for (int sx = 0; sx < 4; ++sx) {
                    t0 = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[sx], X[h4 + ib][0][sx], t0, 0, 0, 0);
                    t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[sx], X[h4 + ib][1][sx], t1, 0, 0, 0);
                }
            }
            const v4f64 iv = S[(6 + cb) * 64 + lane];
            v4f64 x0 = {0.0, 0.0, 0.0, 0.0}, x1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int sx = 0; sx < 4; ++sx) {
                x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(iv[sx], t0[sx], x0, 0, 0, 0);
                x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(iv[sx], t1[sx], x1, 0, 0, 0);
            }
            X[h4 + cb][0] = x0;
            X[h4 + cb][1] = x1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (LDL) vm = fmax(vm, fmax(fabs(x0[r]), fabs(x1[r])));
                const int64_t cc = (int64_t)(coff + 16 * cb + 4 * r) * ld;
                const double di = LDL ? Sd[16 * cb + 4 * r] : 1.0;
                Fs[cc] = LDL ? x0[r] * di : x0[r];
                Fs[cc + 16] = LDL ? x1[r] * di : x1[r];
                if (LDL) {
                    Vs[cc] = x0[r];
                    Vs[cc + 16] = x1[r];
                }
            }
            __builtin_amdgcn_sched_barrier(0);  // (the whole tile is live in registers: nothing of the next block may be hoisted)
        }
    };
    fill_diag(ja);
    __syncthreads();
    stamp(0);
    trsm(std::integral_constant<int, 0>(), 0);
    __syncthreads();
    stamp(1);
    {   // blocks (cb2, ib): -L(jb, ja)[16 cb2 + i][16 ib + k]
        const double* __restrict__ Lba = F + (int64_t)64 * (ja + 1) + (int64_t)64 * ja * ld;
        for (int slot = tid; slot < 1024; slot += 256) {
            const int q = slot >> 6, i = slot & 15, k4 = (slot >> 4) & 3;
            const double* src = Lba + (16 * (q >> 2) + i) + (int64_t)(16 * (q & 3) + k4) * ld;
            v4f64 v;
#pragma unroll
            for (int sx = 0; sx < 4; ++sx) v[sx] = -src[(int64_t)(4 * sx) * ld];
            S[slot] = v;
        }
    }
    __syncthreads();
    stamp(2);
#pragma unroll
    for (int cb2 = 0; cb2 < 4; ++cb2)
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
            const v4f64 aop = S[(cb2 * 4 + ib) * 64 + lane];
#pragma unroll
            for (int sx = 0; sx < 4; ++sx) {
                X[4 + cb2][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[sx], X[ib][0][sx], X[4 + cb2][0], 0, 0, 0);
                X[4 + cb2][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[sx], X[ib][1][sx], X[4 + cb2][1], 0, 0, 0);
            }
            if (ib == 3) __builtin_amdgcn_sched_barrier(0);
        }
    __syncthreads();
    stamp(3);
    fill_diag(ja + 1);
    __syncthreads();
    stamp(4);
    trsm(std::integral_constant<int, 1>(), 64);
    stamp(5);
    if (LDL && a->vmax != nullptr) {
        if (!(vm <= DBL_MAX)) vm = __longlong_as_double(0x7ff0000000000000LL);
        for (int off = 32; off > 0; off >>= 1) vm = fmax(vm, __shfl_xor(vm, off));
        if (lane == 0 && vm > 0.0) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(vm);
            if (bits > __hip_atomic_load(a->vmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a->vmax, bits);
        }
    }
}

// Gate on the bulk stream of a small system (every row a strip of the chain): returns when `*word >= target` -- the chain's
// last diagonal strip has published its rows -- so that the inverses queued behind it start within microseconds of the
// chain's end instead of after the event hand-over between the streams (45 us of a 1 ms factorize! at N = 2048; with a
// shallow band the bulk stream is busy until the end and the same gate measured slower).  Bounded like every device-side
// wait; a failed factorization opens it at once.
__global__ void dag_gate_kernel(const int* __restrict__ word, int target, int* __restrict__ info, long spin_limit) {
    long spins = 0;
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(8);
        if ((++spins & 255) == 0) {
            if (__hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
            if (spins > spin_limit) {
                if (atomicCAS(info, 0, -7) == 0) info[1] = 3;   // (site 3: a gate on the bulk stream)
                break;
            }
        }
    }
}

__global__ __launch_bounds__(256) void dag_reset_kernel(int* __restrict__ flags, int64_t n, int* __restrict__ info, DagInst rec,
                                                        DagInst* __restrict__ rec_dst, const SmallSysRec* __restrict__ recs = nullptr) {
    if (recs != nullptr) {   // (a batch of small systems: blockIdx.y = system; the instance records are uploaded by the host)
        const SmallSysRec r = recs[blockIdx.y];
        flags = r.flags; n = r.nflags; info = r.info;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && rec_dst != nullptr) *rec_dst = rec;   // the instance record the bulk kernel reads
    int4* f4 = reinterpret_cast<int4*>(flags);   // (hipMalloc alignment; the tail is done word by word)
    const int64_t n4 = n / 4, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) f4[i] = make_int4(0, 0, 0, 0);
    if (blockIdx.x == 0) {
        if (threadIdx.x < (unsigned)(n - 4 * n4)) flags[4 * n4 + threadIdx.x] = 0;
        if (threadIdx.x == 0) { info[0] = 0; info[1] = 0; }
    }
}

// Zero one 128 x 128 tile (I >= J) of the buffer that the NEXT factorization scatters its sparse matrix into, with the unit
// diagonal of the padding rows: the background zero-fill of ls.hip (fill_lower_kernel on a side stream, where it ran beside
// the solves and cost the first of them ~60 us of HBM contention) as tasks of the bulk queue -- 128 KB of stores per task,
// a few microseconds of a slot, absorbed where the matrix cores are the bound.
template <class IP>
__device__ __forceinline__ void dag_fill_tile(IP in, int I, int J, int tid) {
    double* Z = in->zfill;
    const int64_t ld = in->ld;
    // thread: two consecutive rows (a double2) of 4 columns per pass; 64 threads cover 128 rows, 4 groups x 8 passes the 128 columns... 
    const int r2 = (tid & 63) * 2, cg = tid >> 6;
#pragma unroll 4
    for (int c = cg; c < 128; c += 4) {
        const int64_t col = (int64_t)128 * J + c, row = (int64_t)128 * I + r2;
        double2 v = make_double2(0.0, 0.0);
        if (I == J && col >= in->N) {
            if (row == col) v.x = 1.0;
            if (row + 1 == col) v.y = 1.0;
        }
        *reinterpret_cast<double2*>(Z + row + col * ld) = v;
    }
}

template <bool LDL>
__global__ __launch_bounds__(256, 3) void dag_bulk_kernel(DagArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ int s_val;
    // per-workgroup statistics of the trace option live in LDS (thread 0 only): no registers across the task loop
    __shared__ unsigned long long s_stat[4];  // first grab, ticks waited, tasks, ticks in the finalization
    if (threadIdx.x < 4) s_stat[threadIdx.x] = 0;
    __syncthreads();
    for (;;) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));  // (keeps the thread-id arithmetic out of the task loop's live ranges, as in gemm_nt_tile)
        if (tid == 0) s_val = atomicAdd(a.qctr, 1);
        __syncthreads();
        const int t = s_val;
        __syncthreads();
        if (t >= a.ntasks) {
            if (a.wgstat != nullptr && tid == 0) {
                unsigned long long* w = a.wgstat + (int64_t)blockIdx.x * 8;
                w[0] = s_stat[0]; w[1] = wall_clock64(); w[2] = s_stat[1]; w[3] = s_stat[2]; w[4] = s_stat[3];
            }
            return;
        }
        if (a.wgstat != nullptr && tid == 0) { if (s_stat[0] == 0) s_stat[0] = wall_clock64(); ++s_stat[2]; }
        const int4 tk = a.tasks[t];
        const int flags = __builtin_amdgcn_readfirstlane(tk.x) & 255, q = __builtin_amdgcn_readfirstlane(tk.x) >> 8;
        const int I = __builtin_amdgcn_readfirstlane(tk.y) & 0xffff, inst = __builtin_amdgcn_readfirstlane(tk.y) >> 16;
        const int J = __builtin_amdgcn_readfirstlane(tk.z);
        const int kbeg = __builtin_amdgcn_readfirstlane(tk.w) & 0xffff, kend = __builtin_amdgcn_readfirstlane(tk.w) >> 16;
        const int64_t row0 = (int64_t)128 * I, col0 = (int64_t)128 * J;
        unsigned long long* tr = a.trace != nullptr && tid == 0 ? a.trace + (int64_t)t * 8 : nullptr;
        if (tr) { tr[0] = wall_clock64(); tr[6] = 0; tr[7] = 0; }
        // `in`: the factorization this task belongs to -- the task's instance record, read through the constant address space
        // (uniform: scalar loads)
        auto do_task = [&](auto in) __attribute__((always_inline)) {
        if (flags & DAG_FILL) {   // (no dependence on anything: not even on the instance's factorization being alive)
            if (in->zfill != nullptr) dag_fill_tile(in, I, J, tid);
            return;
        }

        // One task; false: the instance's factorization has failed (a non-positive Cholesky pivot, an expired wait) -- the
        // task is dropped, nothing is published, and every other task of that instance will be dropped the same way (their
        // waits see `info`); tasks of OTHER instances of a batch are not affected.
        // (`dead` must be the same for every wave of the workgroup: ONE thread looks and the answer goes through LDS.  Round 4
        // let every thread look for itself -- a member of a batch that died between two waves' looks sent some waves of the
        // workgroup into the task and the others on to the next one, sharing the LDS tiles and s_val: wrong tiles in OTHER
        // members, spurious breakdowns, once a memory fault; found in round 5, tools/dbg_batch_reject.py.)
        if (tid == 0) s_val = __hip_atomic_load(in->info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ? 1 : 0;
        __syncthreads();
        const bool dead = s_val != 0;
        __syncthreads();
        auto run_task = [&]() __attribute__((always_inline)) -> bool {
        if (dead) return false;
        // k-tiles [0, limit) of this chunk have final operands; the gate blocks at the first k-tile of a tile column that is
        // not final yet (a rare event: the queue is sorted by readiness)
        int limit = 0;
        auto gate = [&](int kt) -> bool {
            if (kt < limit) return true;
            const unsigned long long w0 = tr ? wall_clock64() : 0;
            const int r = dag_wait_front(in, a.spin_limit, 2 * I, 2 * I + 1, 2 * J, 2 * J + 1, kbeg + (kt >> 4), kend, &s_val);
            if (tr) { const unsigned long long w1 = wall_clock64(); tr[2] = w1; tr[6] += 1; tr[7] += w1 - w0; s_stat[1] += w1 - w0; }
            if (r < 0) return false;
            limit = (__builtin_amdgcn_readfirstlane(r) - kbeg) * 16;
            return true;
        };
        const double* Ak = in->F + (a.fake_share > 0 ? (int64_t)128 * (kend + I % a.fake_share) : row0) + (int64_t)128 * kbeg * in->ld;
        const double* Bk = (LDL ? in->V : in->F) + (a.fake_share > 0 ? (int64_t)128 * (kend + J % a.fake_share) : col0) + (int64_t)128 * kbeg * in->ld;
        auto wait_chunk_order = [&]() -> bool {   // the chunks of one tile are applied in order
            if (flags & DAG_FIRST) return true;
            const int* word = in->tprog + (int64_t)I * a.ntile + J;
            const unsigned long long w0 = tr ? wall_clock64() : 0;
            if (!dag_wait_words(in, a.spin_limit, word, q, word, q, &s_val)) return false;
            if (tr) s_stat[1] += wall_clock64() - w0;
            return true;
        };
        if ((flags & DAG_FINAL) && !(flags & DAG_BANDACC)) {
            // Tile-closing task: on the critical path of its row of tiles (and through it of the band the row enters).  The
            // last tile column is multiplied with every wave owning 32 full rows -- the register layout of the
            // finalization -- so the tile goes from the accumulators through the two substitutions to memory ONCE:
            // X = C - acc without a store, no reload (the round trip through memory was 14 + ~5 us of the task's ~78).
            __builtin_amdgcn_s_setprio(3);
            // the tile as the earlier chunks left it goes into the accumulators BEFORE the wait for the row's previous tile
            // column: its load is off the critical path; the K-loop then subtracts (NEG)
            if (!wait_chunk_order()) return false;
            v4f64 X[8][2];
            {
                const int lane = tid & 63, w = tid >> 6;
                const double* Cs = in->F + (row0 + 32 * w + (lane & 15)) + (col0 + (lane >> 4)) * in->ld;
#pragma unroll
                for (int cb = 0; cb < 8; ++cb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double* cp = Cs + (int64_t)(16 * cb + 4 * r) * in->ld;
                        X[cb][0][r] = cp[0];
                        X[cb][1][r] = cp[16];
                    }
            }
            if (kend > kbeg && !gate(0)) return false;   // one tile column: its operands are final at once
            (void)gemm_nt_mainloop3<2, 8, true>(X, Ak, in->ld, Bk, in->ld, (kend - kbeg) * 16, smem_raw, tid);
            // (new live ranges: the register pressure of the finalization below must not push the accumulators of the
            // K-loop above into scratch)
#pragma unroll
            for (int i = 0; i < 8; ++i) { asm volatile("" : "+v"(X[i][0])); asm volatile("" : "+v"(X[i][1])); }
            if (tr) tr[1] = tr[3] = wall_clock64();  // K-loop done = tile applied
            // the diagonal blocks of tile column J and L(2J + 1, 2J)
            if (dag_wait_front(in, a.spin_limit, 2 * J, 2 * J + 1, 2 * J, 2 * J + 1, J, J + 1, &s_val) < 0) return false;
            if (tr) { tr[4] = wall_clock64(); s_stat[1] += tr[4] - tr[3]; }
            dag_finalize_tile<LDL>(in, row0, col0, 2 * J, smem_raw, tid, X, tr);
            if (tr) s_stat[3] += wall_clock64() - tr[4];
        } else if (!(MNK_DAG_DIAG_SUB && I == J && kend > kbeg)) {
            v4f64 acc[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = v4f64{0.0, 0.0, 0.0, 0.0};
            // the last chunk of a band tile is on the critical path of the chain: it issues first
            if (flags & DAG_FINAL) __builtin_amdgcn_s_setprio(3);
            // (the two-buffer loop: the three-buffer one is ~10 % slower at three workgroups per CU -- tools/hip/time_kstep.hip)
            if (!gemm_nt_mainloop<2, 2, 4, 0, 8>(acc, Ak, in->ld, Bk, in->ld, (kend - kbeg) * 16, smem_raw, tid, gate)) return false;
            if (tr) tr[1] = wall_clock64();  // K-loop done
            if (!wait_chunk_order()) return false;
            // C(I, J) -= acc
            if (kend > kbeg)
                gemm_nt_epilogue<2, 2, 4, 2, false, true>(acc, row0, col0, (int64_t)1 << 40, (int64_t)1 << 40, in->F, in->ld, nullptr, nullptr, 0, tid);
            if (tr) tr[3] = wall_clock64();
        } else {
            // A chunk of a DIAGONAL tile, accumulated in subtract order (gemm_nt_load_neg_lower; its chunks run one after the
            // other).  A branch of its own: the body chunks' K-loop below must keep the registers it was tuned with.
            v4f64 acc[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = v4f64{0.0, 0.0, 0.0, 0.0};
            __builtin_amdgcn_s_setprio(3);   // (every chunk: the next one of the tile -- in the end the chain -- waits for it)
            if (!wait_chunk_order()) return false;
            gemm_nt_load_neg_lower<2, 2, 4>(acc, row0, col0, in->F, in->ld, tid);
            if (!gemm_nt_mainloop<2, 2, 4, 0, 8>(acc, Ak, in->ld, Bk, in->ld, (kend - kbeg) * 16, smem_raw, tid, gate)) return false;
            if (tr) tr[1] = wall_clock64();  // K-loop done
            gemm_nt_store_neg_lower<2, 2, 4>(acc, row0, col0, in->F, in->ld, tid);
            if (tr) tr[3] = wall_clock64();
        }
        return true;
        };
        const bool done = run_task();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (done && tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!(flags & DAG_FINAL)) {
                __hip_atomic_store(in->tprog + (int64_t)I * a.ntile + J, q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (!(flags & DAG_BANDACC)) {
                __hip_atomic_store(in->front + 2 * I, J + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(in->front + 2 * I + 1, J + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                __hip_atomic_store(in->af + (int64_t)I * a.ntile + J, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (tr) tr[5] = wall_clock64();
        }
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();  // (s_val and the LDS tiles are reused by the next task)
        };
        do_task((DagInstP)(uintptr_t)(a.insts + inst));
    }
}

// The bulk kernel of a SINGLE factorization: the arguments of the one instance travel in the kernel's argument block (flat,
// as in round 3) and the task loop below is round 3's, statement for statement, plus the DAG_FILL branch.  Not folded into
// the batch kernel above on purpose: with the identical K-loop instructions (compared block by block in the assembly) the
// generic body compiled to a different VGPR assignment of the MFMA operands and ran the k-steps 7 % slower (44.8 -> 48.1 us,
// 9.20 -> 9.45 ms per factorize! at C3, same box, alternating runs) -- this loop is the one the headline number rests on.
struct DagArgs1 {
    double* F;
    int64_t ld;
    double* V;            // LDL^T: V = L D, same layout as F (the B operand of the updates); Cholesky: nullptr (V = L)
    const double* dinv;   // 1 / d_k
    const double* dblk;   // factored 64x64 diagonal blocks
    const double* inv16;  // inverses of their 16x16 diagonal sub-blocks
    const int4* tasks;    // (flags | chunk index << 8, I, J, kbeg | kend << 16)
    int ntasks;
    int* front;
    int* af;
    int* tprog;           // [I * ntile + J]: chunks of tile (I, J) that are in memory
    int ntile;
    int* qctr;
    int* info;
    long spin_limit;
    unsigned long long* trace;  // diagnostics (option dag_trace): 8 time stamps per task
    unsigned long long* wgstat; // diagnostics: per workgroup {first grab, exit, ticks waited, tasks, ticks in finalize}
    unsigned long long* vmax;   // growth monitor of the static-pivot LDL^T (factor.hip growth_fold): receives max|V|, or NULL
    int fake_share;             // DIAGNOSTIC (env MNK_DAG_FAKE_SHARE, wrong results): every chunk reads the operand rows of tile rows 0..n-1
    double* zfill;              // DAG_FILL tasks: the buffer of the next factorization (or NULL)
    int64_t N;
#if MNK_DIAG_BULK_DBG
    int* bdbg;                  // 16 words per workgroup: {task, stage, words, target, value, spins >> 18, tasks done, -, the 4 front words seen}
#endif
};

#if MNK_DIAG_BULK_DBG
#define MNK_BDBG(a) ((a).bdbg != nullptr ? (a).bdbg + 16 * blockIdx.x : nullptr)
int* g_diag_bdbg = nullptr;   // (set by the host side before a launch; diagnostic builds only)
#else
#define MNK_BDBG(a) nullptr
#endif

template <bool LDL>
__global__ __launch_bounds__(256, 3) void dag_bulk_kernel1(DagArgs1 a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ int s_val;
    // per-workgroup statistics of the trace option live in LDS (thread 0 only): no registers across the task loop
    __shared__ unsigned long long s_stat[4];  // first grab, ticks waited, tasks, ticks in the finalization
    if (threadIdx.x < 4) s_stat[threadIdx.x] = 0;
    __syncthreads();
    for (;;) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));  // (keeps the thread-id arithmetic out of the task loop's live ranges, as in gemm_nt_tile)
        if (tid == 0)
            s_val = __hip_atomic_load(a.info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ? INT_MAX : atomicAdd(a.qctr, 1);
        __syncthreads();
        const int t = s_val;
        __syncthreads();
        if (t >= a.ntasks) {
            if (a.wgstat != nullptr && tid == 0) {
                unsigned long long* w = a.wgstat + (int64_t)blockIdx.x * 8;
                w[0] = s_stat[0]; w[1] = wall_clock64(); w[2] = s_stat[1]; w[3] = s_stat[2]; w[4] = s_stat[3];
            }
            return;
        }
        if (a.wgstat != nullptr && tid == 0) { if (s_stat[0] == 0) s_stat[0] = wall_clock64(); ++s_stat[2]; }
#if MNK_DIAG_BULK_DBG
        if (a.bdbg != nullptr && tid == 0) {
            int* d8 = a.bdbg + 16 * blockIdx.x;
            d8[0] = t; d8[1] = 1; d8[6] += 1;
            // where the workgroup runs (HW_ID: cu [11:8], sh [12], se [15:13]; XCC_ID [3:0]) and when it took the task (100 MHz)
            d8[7] = (int)((__builtin_amdgcn_s_getreg(63492) & 0xffffu) | ((__builtin_amdgcn_s_getreg(63508) & 15u) << 16));
            const unsigned long long now = wall_clock64();
            d8[12] = (int)(unsigned)now; d8[13] = (int)(unsigned)(now >> 32);
        }
#endif
        const int4 tk = a.tasks[t];
        const int flags = __builtin_amdgcn_readfirstlane(tk.x) & 255, q = __builtin_amdgcn_readfirstlane(tk.x) >> 8;
        const int I = __builtin_amdgcn_readfirstlane(tk.y), J = __builtin_amdgcn_readfirstlane(tk.z);
        const int kbeg = __builtin_amdgcn_readfirstlane(tk.w) & 0xffff, kend = __builtin_amdgcn_readfirstlane(tk.w) >> 16;
        const int64_t row0 = (int64_t)128 * I, col0 = (int64_t)128 * J;
        unsigned long long* tr = a.trace != nullptr && tid == 0 ? a.trace + (int64_t)t * 8 : nullptr;
        if (tr) { tr[0] = wall_clock64(); tr[6] = 0; tr[7] = 0; }
        if (flags & DAG_FILL) {   // zero-fill of the next factorization's buffer (see dag_fill_tile)
            if (a.zfill != nullptr) dag_fill_tile(&a, I, J, tid);
            continue;
        }

        // k-tiles [0, limit) of this chunk have final operands; the gate blocks at the first k-tile of a tile column that is
        // not final yet (a rare event: the queue is sorted by readiness)
        int limit = 0;
        auto gate = [&](int kt) -> bool {
            if (kt < limit) return true;
            const unsigned long long w0 = tr ? wall_clock64() : 0;
            const int r = dag_wait_front(&a, a.spin_limit, 2 * I, 2 * I + 1, 2 * J, 2 * J + 1, kbeg + (kt >> 4), kend, &s_val, MNK_BDBG(a));
            if (tr) { const unsigned long long w1 = wall_clock64(); tr[2] = w1; tr[6] += 1; tr[7] += w1 - w0; s_stat[1] += w1 - w0; }
            if (r < 0) return false;
            limit = (__builtin_amdgcn_readfirstlane(r) - kbeg) * 16;
            return true;
        };
        const double* Ak = a.F + (a.fake_share > 0 ? (int64_t)128 * (kend + I % a.fake_share) : row0) + (int64_t)128 * kbeg * a.ld;
        const double* Bk = (LDL ? a.V : a.F) + (a.fake_share > 0 ? (int64_t)128 * (kend + J % a.fake_share) : col0) + (int64_t)128 * kbeg * a.ld;
        auto wait_chunk_order = [&]() -> bool {   // the chunks of one tile are applied in order
            if (flags & DAG_FIRST) return true;
            const int* word = a.tprog + (int64_t)I * a.ntile + J;
            const unsigned long long w0 = tr ? wall_clock64() : 0;
            if (!dag_wait_words(&a, a.spin_limit, word, q, word, q, &s_val, MNK_BDBG(a))) return false;
            if (tr) s_stat[1] += wall_clock64() - w0;
            return true;
        };
        if ((flags & DAG_FINAL) && !(flags & DAG_BANDACC)) {
            // Tile-closing task: on the critical path of its row of tiles (and through it of the band the row enters).  The
            // last tile column is multiplied with every wave owning 32 full rows -- the register layout of the
            // finalization -- so the tile goes from the accumulators through the two substitutions to memory ONCE:
            // X = C - acc without a store, no reload (the round trip through memory was 14 + ~5 us of the task's ~78).
            __builtin_amdgcn_s_setprio(3);
            // the tile as the earlier chunks left it goes into the accumulators BEFORE the wait for the row's previous tile
            // column: its load is off the critical path; the K-loop then subtracts (NEG)
            if (!wait_chunk_order()) return;
            v4f64 X[8][2];
            {
                const int lane = tid & 63, w = tid >> 6;
                const double* Cs = a.F + (row0 + 32 * w + (lane & 15)) + (col0 + (lane >> 4)) * a.ld;
#pragma unroll
                for (int cb = 0; cb < 8; ++cb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double* cp = Cs + (int64_t)(16 * cb + 4 * r) * a.ld;
                        X[cb][0][r] = cp[0];
                        X[cb][1][r] = cp[16];
                    }
            }
            if (kend > kbeg && !gate(0)) return;   // one tile column: its operands are final at once
            (void)gemm_nt_mainloop3<2, 8, true>(X, Ak, a.ld, Bk, a.ld, (kend - kbeg) * 16, smem_raw, tid);
            // (new live ranges: the register pressure of the finalization below must not push the accumulators of the
            // K-loop above into scratch)
#pragma unroll
            for (int i = 0; i < 8; ++i) { asm volatile("" : "+v"(X[i][0])); asm volatile("" : "+v"(X[i][1])); }
            if (tr) tr[1] = tr[3] = wall_clock64();  // K-loop done = tile applied
            // the diagonal blocks of tile column J and L(2J + 1, 2J)
            if (dag_wait_front(&a, a.spin_limit, 2 * J, 2 * J + 1, 2 * J, 2 * J + 1, J, J + 1, &s_val, MNK_BDBG(a)) < 0) return;
            if (tr) { tr[4] = wall_clock64(); s_stat[1] += tr[4] - tr[3]; }
            dag_finalize_tile<LDL>(&a, row0, col0, 2 * J, smem_raw, tid, X, tr);
            if (tr) s_stat[3] += wall_clock64() - tr[4];
        } else if (!(MNK_DAG_DIAG_SUB && I == J && kend > kbeg)) {
            v4f64 acc[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = v4f64{0.0, 0.0, 0.0, 0.0};
            // the last chunk of a band tile is on the critical path of the chain: it issues first
            if (flags & DAG_FINAL) __builtin_amdgcn_s_setprio(3);
            // (the two-buffer loop: the three-buffer one is ~10 % slower at three workgroups per CU -- tools/hip/time_kstep.hip)
            if (!gemm_nt_mainloop<2, 2, 4, 0, 8>(acc, Ak, a.ld, Bk, a.ld, (kend - kbeg) * 16, smem_raw, tid, gate)) return;
            if (tr) tr[1] = wall_clock64();  // K-loop done
            if (!wait_chunk_order()) return;
            // C(I, J) -= acc
            if (kend > kbeg)
                gemm_nt_epilogue<2, 2, 4, 2, false, true>(acc, row0, col0, (int64_t)1 << 40, (int64_t)1 << 40, a.F, a.ld, nullptr, nullptr, 0, tid);
            if (tr) tr[3] = wall_clock64();
        } else {
            // A chunk of a DIAGONAL tile, accumulated in subtract order (gemm_nt_load_neg_lower; its chunks run one after the
            // other).  A branch of its own: the body chunks' K-loop below must keep the registers it was tuned with.
            v4f64 acc[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = v4f64{0.0, 0.0, 0.0, 0.0};
            __builtin_amdgcn_s_setprio(3);   // (every chunk: the next one of the tile -- in the end the chain -- waits for it)
            if (!wait_chunk_order()) return;
            gemm_nt_load_neg_lower<2, 2, 4>(acc, row0, col0, a.F, a.ld, tid);
            if (!gemm_nt_mainloop<2, 2, 4, 0, 8>(acc, Ak, a.ld, Bk, a.ld, (kend - kbeg) * 16, smem_raw, tid, gate)) return;
            if (tr) tr[1] = wall_clock64();  // K-loop done
            gemm_nt_store_neg_lower<2, 2, 4>(acc, row0, col0, a.F, a.ld, tid);
            if (tr) tr[3] = wall_clock64();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!(flags & DAG_FINAL)) {
                __hip_atomic_store(a.tprog + (int64_t)I * a.ntile + J, q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (!(flags & DAG_BANDACC)) {
                __hip_atomic_store(a.front + 2 * I, J + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.front + 2 * I + 1, J + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                __hip_atomic_store(a.af + (int64_t)I * a.ntile + J, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (tr) tr[5] = wall_clock64();
#if MNK_DIAG_BULK_DBG
            if (a.bdbg != nullptr) {   // published, and when
                int* d8 = a.bdbg + 16 * blockIdx.x;
                const unsigned long long now = wall_clock64();
                d8[1] = 4; d8[14] = (int)(unsigned)now; d8[15] = (int)(unsigned)(now >> 32);
            }
#endif
        }
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();  // (s_val and the LDS tiles are reused by the next task)
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
// Task list for a matrix of `ntile` 128-row tiles, chunks of at most `chunk` tile columns and a band of `band_tiles` tile
// rows for the strip-columns Js < js2, every remaining row from js2 on (depends on these four only; cached by the caller).
// 4 ints per task: flags | chunk index << 8, I, J, kbeg | kend << 16.  Returns the number of first-phase tasks: the ones
// that are ready before the chain enters strip-column js2 (they come first in the list).
int dag_build_tasks(int ntile, int chunk, int band_tiles, int js2, std::vector<int>& out, int taper0, std::vector<int>* ready) {
    struct T { int ready, cls, J, I, flags, q, kbeg, kend; };
    std::vector<T> ts;
    // Tile (I, J), tile columns [0, K) to accumulate.  A bulk tile's last tile column is a task of its own (paced by the pivot
    // chain: K = 128 step, then the finalization); the columns in front of it -- all columns of a band tile -- go in chunks:
    //   * STAGGERED from tile to tile (first chunk 1 .. chunk tile columns long, by a hash of the tile's coordinates): at
    //     every chain position about 1 / chunk of the tiles have a chunk that just became ready, so ready work arrives
    //     evenly.  (With boundaries at multiples of `chunk` the queue alternated between a big batch of ready chunks and a
    //     stretch of tile-closing tasks that are serialized along every row of tiles: 23 % of the slot-time waiting.)
    //   * TAPERED towards K (..., chunk, 4, 2, 1): a chunk [b, e) cannot start before the chain has passed tile column
    //     e - 1 and the tile is due when the chain reaches K, so a full-length chunk that becomes ready one chain step
    //     before the tile is due was 100-300 us late for the band, every time (measured).
    auto add_tile = [&](int I, int J, int K, bool band) {
        const int body = band ? K : std::max(0, K - 1);
        std::vector<int> cuts{body};  // chunk boundaries, back to front
        int e = body;
        for (int len = taper0; len < chunk && e > 0; len *= 2) { e = std::max(0, e - len); cuts.push_back(e); }   // (factors 3 / 4 instead of 2: +2 / +5 %)
        const int first = 1 + (I * 5 + J * 3) % chunk;
        while (e > first) { e = std::max(first, e - chunk); cuts.push_back(e); }
        if (e > 0) cuts.push_back(0);
        int q = 0;
        for (int c = (int)cuts.size() - 1; c > 0; --c, ++q) {
            const int kb = cuts[c], ke = cuts[c - 1];
            const bool last = band && ke == K;
            ts.push_back({ke, band ? 0 : 2, J, I, (band ? DAG_BANDACC : 0) | (last ? DAG_FINAL : 0) | (q == 0 ? DAG_FIRST : 0), q, kb, ke});
        }
        if (!band) ts.push_back({K, 1, J, I, DAG_FINAL | (q == 0 ? DAG_FIRST : 0), q, body, K});
    };
    for (int Jt = 0; Jt < ntile; ++Jt) {
        const int Js = Jt / 2;
        const int bt = Js >= js2 ? ntile : band_tiles;
        for (int I = 2 * Js + bt; I < ntile; ++I) add_tile(I, Jt, Jt, false);
        // band tiles of strip-column Js (tile rows 2Js .. 2Js + bt - 1, lower part): accumulated over the tile columns
        // k < 2Js - 2 (the chain's prologue applies strip-column Js - 1 itself)
        if (2 * Js - 2 > 0)
            for (int I = std::max(2 * Js, Jt); I < 2 * Js + bt && I < ntile; ++I) add_tile(I, Jt, 2 * Js - 2, true);
    }
    // by the chain position that makes a task ready (the last tile column it reads), then band tiles, then the tile-closing
    // tasks, then by column and row (rows next to the band first)
    std::stable_sort(ts.begin(), ts.end(), [](const T& x, const T& y) {
        if (x.ready != y.ready) return x.ready < y.ready;
        if (x.cls != y.cls) return x.cls < y.cls;
        if (x.J != y.J) return x.J < y.J;
        return x.I < y.I;
    });
    out.clear();
    out.reserve(ts.size() * 4);
    if (ready != nullptr) { ready->clear(); ready->reserve(ts.size()); }
    int n1 = 0;
    for (const T& t : ts) {
        if (t.ready <= 2 * js2) ++n1;  // needs only tile columns the first phase's chain has passed (sorted by `ready`)
        if (ready != nullptr) ready->push_back(t.ready);
        out.push_back(t.flags | (t.q << 8));
        out.push_back(t.I);
        out.push_back(t.J);
        out.push_back(t.kbeg | (t.kend << 16));
    }
    return n1;
}

// Zero-fill tasks (DAG_FILL) for the lower tiles of the NEXT factorization's buffer, one after every few tasks of the first
// phase: 128 KB of stores each, absorbed by the queue while the matrix cores are the bound.  `ready` gets the value of the
// task in front (the merge of several instances keeps them where they are).  Returns the new number of first-phase tasks.
int dag_add_fill_tasks(int ntile, std::vector<int>& tasks, std::vector<int>& ready, int n1) {
    const int nt = (int)(tasks.size() / 4), nfill = ntile * (ntile + 1) / 2;
    const int span = n1 > 0 ? n1 : nt;   // (spread over the first phase; a list without one: over everything)
    if (span <= 0) return n1;
    std::vector<int> out, rdy;
    out.reserve(tasks.size() + 4 * (size_t)nfill);
    rdy.reserve(ready.size() + nfill);
    int fi = 0, fj = 0, done = 0;
    auto emit_fill = [&](int r) {
        out.push_back(DAG_FILL); out.push_back(fi); out.push_back(fj); out.push_back(0);
        rdy.push_back(r);
        if (++fi >= ntile) { ++fj; fi = fj; }
        ++done;
    };
    for (int t = 0; t < nt; ++t) {
        for (int k = 0; k < 4; ++k) out.push_back(tasks[4 * t + k]);
        rdy.push_back(ready[t]);
        // after task t of the span, (t + 1) * nfill / span fill tasks are out
        if (t < span)
            while (done < (int)((int64_t)(t + 1) * nfill / span)) emit_fill(ready[t]);
    }
    tasks.swap(out);
    ready.swap(rdy);
    return n1 > 0 ? n1 + nfill : 0;
}

// One queue for `ninst` independent factorizations of the same order (mnk_factorize_batch_*): instance i's tasks keep their
// order and are shifted by i * period chain positions (tile columns) against instance 0's, so that an instance's
// saturated middle runs under the chain-bound ends of its neighbours.  Two pivot chains run at a time (instances i and
// i + 1; chain i + 2 follows chain i on its stream), hence period >= ntile / 2: every task of instance i then precedes
// every task of instance i + 2, and a workgroup that holds a task waits only for tasks in front of it or for a chain that
// is running or will run once tasks in front of it are done.
void dag_merge_tasks(const std::vector<int>& tasks, const std::vector<int>& ready, int ninst, int period, std::vector<int>& out) {
    const int nt = (int)ready.size();
    out.clear();
    out.reserve((size_t)ninst * nt * 4);
    std::vector<int> pos(ninst, 0);
    for (;;) {   // k-way merge by (i * period + ready, i); every list is sorted by `ready`
        int best = -1;
        long bkey = 0;
        for (int i = 0; i < ninst; ++i) {
            if (pos[i] >= nt) continue;
            const long key = (long)i * period + ready[pos[i]];
            if (best < 0 || key < bkey) { best = i; bkey = key; }
        }
        if (best < 0) break;
        const int t = pos[best]++;
        out.push_back(tasks[4 * t]);
        out.push_back(tasks[4 * t + 1] | (best << 16));
        out.push_back(tasks[4 * t + 2]);
        out.push_back(tasks[4 * t + 3]);
    }
}

template <bool LDL>
static int launch_bulk_t(hipStream_t s, const DagArgs& a, int nwg) {
    const size_t smem = TILE3_LDS_BYTES;  // three k-tile buffers of the tile-closing tasks (gemm_nt_mainloop3); chunks use two, the finalization 32 KB
    auto kern = dag_bulk_kernel<LDL>;
    static std::atomic<uint64_t> attr_devs{0};
    int dev = 0;
    MNK_HIP(hipGetDevice(&dev));
    if (!(attr_devs.load(std::memory_order_relaxed) >> (dev & 63) & 1)) {
        MNK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_devs.fetch_or(1ull << (dev & 63), std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), smem, s, a);
    MNK_HIP(hipGetLastError());
    return 0;
}

template <bool LDL>
static int launch_bulk1_t(hipStream_t s, const DagArgs1& a, int nwg) {
    const size_t smem = TILE3_LDS_BYTES;
    auto kern = dag_bulk_kernel1<LDL>;
    static std::atomic<uint64_t> attr_devs{0};
    int dev = 0;
    MNK_HIP(hipGetDevice(&dev));
    if (!(attr_devs.load(std::memory_order_relaxed) >> (dev & 63) & 1)) {
        MNK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_devs.fetch_or(1ull << (dev & 63), std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), smem, s, a);
    MNK_HIP(hipGetLastError());
    return 0;
}

// `insts` == nullptr: one factorization (dag_bulk_kernel1, the record `one` flattened into its argument block); otherwise a batch
static int launch_dag_bulk(hipStream_t s, bool ldl, const DagInst& one, const DagInst* insts, const int* tasks, int ntasks, int ntile,
                           int* qctr, long spin_limit, int nwg, unsigned long long* trace, unsigned long long* wgstat) {
    if (ntasks <= 0) return 0;
    static const int fake = getenv("MNK_DAG_FAKE_SHARE") ? atoi(getenv("MNK_DAG_FAKE_SHARE")) : 0;
    if (insts == nullptr) {
        DagArgs1 a{one.F, one.ld, one.V, one.dinv, one.dblk, one.inv16, reinterpret_cast<const int4*>(tasks), ntasks, one.front, one.af,
                   one.tprog, ntile, qctr, one.info, spin_limit, trace, wgstat, one.vmax, fake, one.zfill, one.N};
#if MNK_DIAG_BULK_DBG
        a.bdbg = g_diag_bdbg;
#endif
        return ldl ? launch_bulk1_t<true>(s, a, nwg) : launch_bulk1_t<false>(s, a, nwg);
    }
    DagArgs a{insts, reinterpret_cast<const int4*>(tasks), ntasks, ntile, qctr, spin_limit, trace, wgstat, fake};
    return ldl ? launch_bulk_t<true>(s, a, nwg) : launch_bulk_t<false>(s, a, nwg);
}

}  // namespace mnk

using namespace mnk;

// First launch of the bulk kernels on a stream, with an EMPTY queue (every workgroup leaves at once): the code objects are
// loaded and the stream's hardware queue gets its scratch memory (these kernels spill a few registers) while nothing is
// resident that could wait for them.  Without it the first real factorization of a process launches a pivot chain that
// spins for a bulk kernel the runtime is still setting up -- normally microseconds, but seen to exceed the bound of the
// device-side waits (info = -7, schedule 1 for the next 16 factorizations: round 3 met it in 2 of ~150 bench processes, this
// round in 1 of 8 runs of tools/dag_time.py).  Called once per device when its task-DAG streams are created (ls.hip).
int mnk_dag_warmup(hipStream_t* streams, int n, int nwg) {
    static mnk::DevBuf<int> ctr;   // (process lifetime)
    if (!ctr.p) {
        if (ctr.alloc(4)) return -2;
        MNK_HIP(hipMemset(ctr.p, 0, 4 * sizeof(int)));
    }
    for (int i = 0; i < n; ++i) {
        if (streams[i] == nullptr) continue;
        mnk::DagArgs1 a1{};
        a1.qctr = ctr.p;
        a1.info = ctr.p + 1;
        mnk::DagArgs a{};
        a.qctr = ctr.p;
        // (a FULL-size grid: the runtime sizes a queue's scratch by the dispatch that asks for it)
        int rc = mnk::launch_bulk1_t<true>(streams[i], a1, nwg);
        if (!rc) rc = mnk::launch_bulk1_t<false>(streams[i], a1, nwg);
        if (!rc) rc = mnk::launch_bulk_t<true>(streams[i], a, nwg);
        if (!rc) rc = mnk::launch_bulk_t<false>(streams[i], a, nwg);
        if (rc) return rc;
    }
    for (int i = 0; i < n; ++i)
        if (streams[i] != nullptr) MNK_HIP(mnk::stream_wait(streams[i]));
    return 0;
}

// Bound of the schedule's device-side waits, in polls of ~0.17 us.  Every wait is on work that is queued or resident, so a
// bound only ever expires when something outside the schedule keeps a kernel of the group from running: another process'
// kernels on the CUs, a copy between pageable host memory and the device by OTHER code of this process, a device-synchronizing
// runtime call between the group's launches.  The recovery is cheap (the factorization is redone
// with one launch per piece, the schedule is tried again 16 factorizations later), a long bound is not: whatever stops the
// group costs the bound.  History: 2^24 polls (~3 s) in round 3; ~100 x the factorization's time, at least 1 s, in round 4 --
// a bound of 0.3 s then expired about once in 1000 factorizations on transient stalls, which turned out to be the workgroups of
// an oversized bulk grid that the hardware placed in mid-kernel (grid 720 -> 672).  With that
// grid, round 5 ran 3 x 10 240 factorizations of the C3 system at bounds of 0.05 / 0.1 / 0.3 s without a single expiry
// (profiles/r05_spin_bound_sweep.txt): the default is now ~10 x the time the factorization should take, at least 0.1 s -- the
// longest legitimate wait is a fraction of one factorization (a chain strip waiting for its band tiles: ~0.3 ms at N = 11 192).
long mnk_ls_dag_spin_limit(const mnk_ls* ls) {
    if (ls->dag_spin_limit > 0) return ls->dag_spin_limit;
    const double n = (double)ls->Np;
    const double t_est = n * n * n / 3.0 / 50e12;   // seconds at ~0.64 of the fp64 peak
    const double polls = 10.0 * t_est / 0.17e-6;
    //  several contexts alive on the device: the soak runs behind the 0.1 s floor were single-context; beside other
    // contexts' persistent groups (the case mnk_release_idle_streams exists for) short stalls are legitimate, and an expiry costs a
    // whole redo plus 16 / 64 / 256 factorizations on the slow schedule -- the floor is the old 1 s there
    const double floor_polls = mnk_live_contexts(ls->ctx->device) > 1 ? 5880000.0 : 588000.0;
    return (long)std::min(16777216.0, std::max(floor_polls, polls));
}

static mnk::DagInst dag_instance(mnk_ls* ls) {
    const int ntile = (int)(ls->Np / 128), nblk = (int)(ls->Np / NBI);
    int* front = ls->dag_flags.p + 2;
    int* af = front + nblk;
    int* tprog = af + (size_t)ntile * ntile;
    // the spare factor buffer is zeroed by the queue's DAG_FILL tasks when the NEXT transfer will want it (a sparse source was
    // transferred for this factorization: mnk_ls::spare_pending) and no background fill of it is in flight
    double* zfill = (ls->dag_fill && ls->dag_has_fill && ls->spare_pending && ls->fact_spare.p && !ls->spare_zeroed) ? ls->fact_spare.p : nullptr;
    return mnk::DagInst{ls->fact.p, ls->ld, ls->algo == MNK_LDL ? ls->vfull.p : nullptr, ls->dinv.p, ls->dblk.p, ls->inv16.p,
                        front, af, tprog, ls->info_dev.p, mnk_ls_growth_word(ls), zfill, ls->N};
}

// Buffers of the task-DAG schedule: the task list (built once per matrix order and option set), the progress words and,
// for LDL^T, V = L D of every column (a second N x N array).  Returns 0 when everything is there; non-zero when the device
// cannot hold it -- the caller keeps schedule 4 then (round 3 returned -2 from factorize! instead).  The spare factor buffer
// of the background zero-fill is given up first: with many solvers per GPU it can be the allocation that took V's memory.
int mnk_ls_dag_prepare(mnk_ls* ls) {
    mnk_ctx* ctx = ls->ctx;
    hipStream_t s = ctx->stream;
    const int64_t Np = ls->Np, ld = ls->ld;
    const bool ldl = ls->algo == MNK_LDL;
    const int ntile = (int)(Np / 128), nblk = (int)(Np / NBI), nsc = (int)((Np + 255) / 256);
    const size_t nflags = (size_t)2 + nblk + 2 * (size_t)ntile * ntile;  // two queue counters | front | af | tprog
    auto give_up = [&]() {
        (void)hipGetLastError();
        ls->dag_tasks.release();
        ls->dag_flags.release();
        ls->vfull.release();
        return 1;
    };
    if (!ls->dag_tasks.p) {
        // Two band shapes.  Large systems (more than dag_cus2 strips of 64 rows): a shallow band on a handful of CUs, the
        // rows below it closed by the bulk kernel's own tasks -- the factorization is bound by the bulk work for most of its
        // columns.  Small systems: EVERY row is a strip of the chain's band (one CU each, on a larger partition) and the
        // bulk kernel only accumulates: they are bound by the pivot chain from the first column on, and a row of tiles that
        // the bulk kernel closes advances one tile column per {K = 128 step + finalization} ~ 80 us against the chain's
        // ~50 us.  (Switching from the first shape to the second in mid-factorization, once dag_cus2 strips remain, is
        // supported by the task list -- dag_js2 -- and was measured: the deep band needs one CU per strip, the remaining bulk
        // work of a C3-size system then no longer fits the smaller bulk partition, 10.7 vs 10.3 ms.)
        // (round 6: the deep band pays up to ~5300 rows -- all rows in the band against band + bulk kernel throughout, LDL^T, one box:
        // N = 4096 1.61 vs 1.73 ms, 4608 1.86 vs 1.99, 5120 2.20 vs 2.27, 5632 2.70 vs 2.59, 6144 3.24 vs 2.93 (Cholesky 5632: 2.75 vs 2.46) --
        // above that every strip's left-looking prologue on its one CU costs more than the bulk kernel's closing tasks; the limit
        // was the partition's 96 strips = 6144 rows: dag_deep_rows)
        const int64_t deep_rows = std::min<int64_t>((int64_t)ctx->dag_cus2 * NBI, ls->dag_deep_rows);
        ls->dag_js2 = (ctx->dag_cus2 > 0 && Np <= deep_rows) ? 0 : nsc;
        if (ls->dag_js2_override >= 0) ls->dag_js2 = std::min(nsc, ls->dag_js2_override);
        std::vector<int>& h = ls->dag_host_tasks;   // (kept on the host: a batch merges the lists of its instances)
        ls->dag_ntasks1 = mnk::dag_build_tasks(ntile, ls->dag_chunk, ls->dag_band / 2, ls->dag_js2, h, ls->dag_taper0, &ls->dag_host_ready);
        // zero-fill of the next factorization's buffer as tasks of the queue (sparse sources; see dag_fill_tile)
        ls->dag_has_fill = ls->dag_fill && ls->prefill && Np <= ls->prefill_max_rows;
        if (ls->dag_has_fill && h.empty()) ls->dag_has_fill = false;   // (a system of a few hundred rows has no bulk task: nothing to ride on)
        if (ls->dag_has_fill) ls->dag_ntasks1 = mnk::dag_add_fill_tasks(ntile, h, ls->dag_host_ready, ls->dag_ntasks1);
        ls->dag_ntasks = (int)(h.size() / 4);
        if (ls->dag_tasks.alloc(h.size() + 4)) return give_up();
        {
            mnk::H2DGuard h2d;   // (a pageable upload beside another context's persistent group would stop that group: common.h)
            if (!h.empty() && hipMemcpyAsync(ls->dag_tasks.p, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice, s) != hipSuccess) return give_up();
            if (mnk::stream_wait(s) != hipSuccess) return give_up();
        }
        if (ls->dag_flags.alloc(nflags)) return give_up();
    }
    if (ldl && !ls->vfull.p && ls->vfull.alloc((size_t)ld * Np + SLACK)) {   // V = L D of every column (LDL^T)
        (void)hipGetLastError();
        bool ok = false;
        if (ls->fact_spare.p && !ls->spare_zeroed) {
            // (spare_zeroed: a background fill of the spare buffer has been queued and the next transfer will swap it in)
            ls->fact_spare.release();
            ls->prefill = 0;
            ls->spare_pending = false;
            ok = ls->vfull.alloc((size_t)ld * Np + SLACK) == 0;
        }
        if (!ok) return give_up();
    }
    if (ls->dag_trace_on) {
        const size_t ntr = (size_t)ls->dag_ntasks * 8 + 4096 * 8 + 1024 * 8;  // tasks | chain strips | per-workgroup statistics (2 x 512)
        if (ls->dag_trace.n < ntr && ls->dag_trace.alloc(ntr)) return give_up();   // (the task list may have been rebuilt with more tasks)
    }
    return 0;
}

// Host driver of the task-DAG schedule (panel_algo = 5): ONE persistent pivot-chain launch (factor.hip: pchain_kernel, one
// workgroup per 64-row strip of the band, on the chain's CU partition) beside ONE persistent bulk launch on all the other
// CUs; the two sides meet through progress words in device memory, no event is recorded or awaited between the fork and
// the join.  The buffers come from mnk_ls_dag_prepare (called by mnk_ls_run_factorization before it settles on this schedule).
int mnk_ls_run_factorization_dag(mnk_ls* ls) {
    mnk_ctx* ctx = ls->ctx;
    hipStream_t s = ctx->stream;
    const int64_t Np = ls->Np;
    const bool ldl = ls->algo == MNK_LDL;
    const int ntile = (int)(Np / 128), nblk = (int)(Np / NBI), nsc = (int)((Np + 255) / 256);
    const size_t nflags = (size_t)2 + nblk + 2 * (size_t)ntile * ntile;  // two queue counters | front | af | tprog
    MNK_REQUIRE(ls->dag_tasks.p && ls->dag_flags.p && (!ldl || ls->vfull.p), "task-DAG schedule: mnk_ls_dag_prepare was not called");
    // progress words and `info` in ONE launch (two memsets are two fill kernels, ~8 us each in front of the pivot chain)
    const mnk::DagInst inst = dag_instance(ls);
    ls->dag_filled = inst.zfill != nullptr;
    hipLaunchKernelGGL(mnk::dag_reset_kernel, dim3((unsigned)std::min<size_t>((nflags + 1023) / 1024, 64)), dim3(256), 0, s,
                       ls->dag_flags.p, (int64_t)nflags, ls->info_dev.p, inst, (mnk::DagInst*)nullptr);
    int* qctr = ls->dag_flags.p;
    int* front = inst.front;
    const long spin_limit = mnk_ls_dag_spin_limit(ls);
    unsigned long long* trace = nullptr;
    if (ls->dag_trace_on && ls->dag_trace.p) {
        const size_t ntr = (size_t)ls->dag_ntasks * 8 + 4096 * 8 + 1024 * 8;
        MNK_HIP(hipMemsetAsync(ls->dag_trace.p, 0, ntr * sizeof(unsigned long long), s));
        trace = ls->dag_trace.p;
    }
    const int js2 = ls->dag_js2;
    // one phase = one persistent bulk launch (update stream) beside one persistent chain launch (panel stream)
    auto phase = [&](hipStream_t sp, hipStream_t su, int chain_cus, int task0, int ntask, int* counter, int js_begin, int js_end,
                     unsigned strips) -> int {
        // A small system (every row a strip of the chain, one phase) runs its chain on the CALLER's stream: no fork in front
        // of the first diagonal block, no join behind the last one, and the inverses simply follow the chain.  Its workgroups
        // are dispatched before the bulk kernel's (which waits for the event) and need a CU each to themselves (registers);
        // the CUs outside the bulk stream's mask -- at least as many as there are strips -- cannot be taken from them.
        const bool small = js_begin == 0 && js_end == nsc && js2 == 0 && ls->dag_chain_inline;
        if (small) sp = s;
        MNK_HIP(hipEventRecord(ctx->ev_a, s));
        if (!small) MNK_HIP(hipStreamWaitEvent(sp, ctx->ev_a, 0));
        MNK_HIP(hipStreamWaitEvent(su, ctx->ev_a, 0));
        mnk::PpDag dag{front, inst.af, ntile, 0, 0, -1, spin_limit, trace ? trace + (size_t)ls->dag_ntasks * 8 + (size_t)js_begin * 8 * 16 : nullptr,
                  mnk_ls_growth_word(ls), ls->dag_debug ? ls->dag_dbg.p : nullptr};
        // (the chain first: its first diagonal block is the start of the critical path, the bulk kernel has nothing to do before it)
        int rc = mnk_launch_pchain(ls, sp, dag, js_begin, js_end, strips);
        if (rc) return rc;
        // (more than three workgroups per CU -- to fill slots a shader-engine-imbalanced mask might leave empty -- measured
        // no difference: 4 / 5 / 6 per CU 9.63-9.67 vs 9.64-9.68 ms)
#if MNK_DIAG_BULK_DBG
        mnk::g_diag_bdbg = ls->dag_debug && ls->dag_dbg.p ? ls->dag_dbg.p + 8 * 128 : nullptr;
#endif
        rc = mnk::launch_dag_bulk(su, ldl, inst, nullptr, ls->dag_tasks.p + 4 * (size_t)task0, ntask, ntile, counter, spin_limit,
                                  std::min(ntask, mnk_ctx_bulk_wgs(ctx, chain_cus, 3)), trace ? trace + 8 * (size_t)task0 : nullptr,
                                  trace ? trace + (size_t)ls->dag_ntasks * 8 + 4096 * 8 + (task0 > 0 ? 512 * 8 : 0) : nullptr);
        if (rc) return rc;
        // Once the bulk kernel has run out of tasks every tile-closing task is done, hence every strip-column that still
        // had rows below the band is final: its diagonal blocks are inverted for the solves here, behind the bulk kernel
        // on its stream, while the chain works on the last strip-columns (all rows in the band: no bulk task left).
        // (Measured and dropped in round 4: the inverses of the LAST strip-columns one by one on this stream behind device-side
        // gates on the chain's progress.  The bulk kernel leaves only ~130 us before the chain's end -- its last tile-closing
        // tasks -- and every gated launch pair costs ~170 us whatever its size: C3 factorize! 9.66 -> 10.78 ms.)
        if (js_begin == 0 && js_end > 0) {
            const int64_t safe = std::min<int64_t>(js_end, std::max<int64_t>(0, (ntile - ls->dag_band / 2 - 1) / 2)) & ~(int64_t)1;  // (even: 512-row triangles)
            if (strips == (unsigned)ls->dag_band && safe > 0) {
                rc = mnk_ls_invert_blocks(ls, su, 0, safe);
                if (rc) return rc;
                ls->inv_done = safe;
            }
        }
        if (small) {
            rc = mnk_ls_invert_blocks(ls, s, 0, nsc);
            if (rc) return rc;
            ls->inv_done = nsc;
        } else if (js_begin == 0 && js_end == nsc && js2 == 0) {
            hipLaunchKernelGGL(mnk::dag_gate_kernel, dim3(1), dim3(1), 0, su, front + (nblk - 1), ntile, ls->info_dev.p, spin_limit);
            rc = mnk_ls_invert_blocks(ls, su, 0, nsc);
            if (rc) return rc;
            ls->inv_done = nsc;
        }
        if (!small) MNK_HIP(hipEventRecord(ctx->ev_a, sp));
        MNK_HIP(hipEventRecord(ctx->ev_b, su));
        if (!small) MNK_HIP(hipStreamWaitEvent(s, ctx->ev_a, 0));
        MNK_HIP(hipStreamWaitEvent(s, ctx->ev_b, 0));
        return 0;
    };
    if (js2 > 0) {
        int rc = phase(ctx->sp_dag, ctx->su_dag, ctx->dag_cus, 0, ls->dag_ntasks1, qctr, 0, std::min(js2, nsc),
                       (unsigned)std::min<int64_t>(ls->dag_band, Np / NBI));
        if (rc) return rc;
    }
    if (js2 < nsc) {
        const int64_t rows2 = Np - 256 * (int64_t)js2;
        MNK_REQUIRE(ctx->sp_dag2 != nullptr, "task-DAG schedule: the deep-band streams are missing");   // (mnk_ls_run_factorization made them)
        int rc = phase(ctx->sp_dag2, ctx->su_dag2, ctx->dag_cus2, ls->dag_ntasks1, ls->dag_ntasks - ls->dag_ntasks1,
                       qctr + 1, js2, nsc, (unsigned)(rows2 / NBI));
        if (rc) return rc;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Batches of independent factorizations (mnk_factorize_batch_begin / _end; BASELINE config C5: scenario batches per GPU).
// A single factorization of N ~ 1e4 leaves the bulk CUs nearly idle at both ends -- 0.5 ms of start-up and ~2.6 ms of a
// tail that is bound by the pivot chain -- which no schedule of ONE matrix can fill.  Between begin and end the
// factorize! calls of the calling thread only transfer their matrices; `end` launches them together: ONE bulk kernel
// drains the merged task queue of all instances (dag_merge_tasks) beside TWO pivot chains on two CU partitions (even
// instances on one, odd ones on the other), so that instance i + 1's saturated middle runs under instance i's tail.  The
// arithmetic per instance is exactly that of a single factorization (same task list, same order per tile): bit-identical
// factors.  Reference behaviour: distinct solver instances are driven concurrently (src/KKT/Schur/schur.jl:953).
// ---------------------------------------------------------------------------------------------------------------------
namespace {
struct BatchState {
    int depth = 0;             // (begin / end pairs nest: the outermost end launches)
    bool active = false;
    std::vector<mnk_ls*> pend;
};
thread_local BatchState t_batch;

struct BatchBuffers {   // per device, reused from batch to batch (guarded by the arbiter's mutex while in use)
    mnk::DevBuf<int> tasks;
    mnk::DevBuf<int> qctr;
    mnk::DevBuf<char> insts, pcsys, aux;
    char* stage = nullptr;        // pinned host memory: the records of one batch (instances | chain table | small-system records)
    size_t stage_bytes = 0;
    hipEvent_t stage_ev = nullptr;   // recorded behind the last upload from `stage`
    int ntile = 0, ninst = 0, period = 0, chunk = 0, taper0 = 0, band = 0, ntasks = 0;
    bool fill = false;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
};
BatchBuffers g_batch[64];
}  // namespace

bool mnk_batch_active() { return t_batch.active; }

bool mnk_batch_defer(mnk_ls* ls) {
    if (!t_batch.active) return false;
    const int nsc = (int)((ls->Np + 255) / 256);
    // (two merged launches: the large-system mode of the schedule -- band + bulk + two alternating chains -- and the small
    // one, every row in the chain's band, many chains side by side; a second phase in mid-factorization is simply run now)
    if (ls->algo_now != 5 || (ls->dag_js2 != nsc && ls->dag_js2 != 0) || ls->dag_trace_on || ls->ctx->partitioned) return false;
    if (mnk_ctx_ensure_batch_streams(ls->ctx) != 0) (void)hipGetLastError();   // (made with the first batch of a process)
    if (!ls->ctx->sp_dagB) return false;
    if (!ls->ev_defer && hipEventCreateWithFlags(&ls->ev_defer, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (hipEventRecord(ls->ev_defer, ls->ctx->stream) != hipSuccess) { (void)hipGetLastError(); return false; }
    ls->deferred = true;
    t_batch.pend.push_back(ls);
    return true;
}

static void batch_mark_done(mnk_ls* ls) {
    ls->factorized = true;
    ls->info_valid = false;
    ls->bk_active = false;
    ls->deferred = false;
}

// the merged launch of g.size() >= 2 deferred factorizations of one order / algorithm / option set on one device
static int batch_run_group(std::vector<mnk_ls*>& g) {
    mnk_ls* l0 = g[0];
    mnk_ctx* c0 = l0->ctx;
    hipStream_t h = c0->stream;
    MNK_HIP(hipSetDevice(c0->device));
    const int64_t Np = l0->Np;
    const bool ldl = l0->algo == MNK_LDL;
    const int ntile = (int)(Np / 128), nsc = (int)((Np + 255) / 256), ninst = (int)g.size();
    const size_t nflags = (size_t)2 + (size_t)(Np / NBI) + 2 * (size_t)ntile * ntile;
    for (mnk_ls* ls : g)
        if (ls->ctx->stream != h) MNK_HIP(hipStreamWaitEvent(h, ls->ev_defer, 0));
    int rc = mnk_persist_begin(c0, h);
    if (rc) return rc;
    auto body = [&]() -> int {
        BatchBuffers& B = g_batch[c0->device & 63];
        const int period = std::max((ntile + 1) / 2, l0->batch_period > 0 ? l0->batch_period : 0);
        if (B.ntile != ntile || B.ninst != ninst || B.period != period || B.chunk != l0->dag_chunk || B.taper0 != l0->dag_taper0 ||
            B.band != l0->dag_band || B.fill != l0->dag_has_fill || !B.tasks.p) {
            std::vector<int> merged;
            mnk::dag_merge_tasks(l0->dag_host_tasks, l0->dag_host_ready, ninst, period, merged);
            if (B.tasks.alloc(merged.size() + 4)) return -2;
            { mnk::H2DGuard h2d; MNK_HIP(hipMemcpy(B.tasks.p, merged.data(), merged.size() * sizeof(int), hipMemcpyHostToDevice)); }
            if (!B.qctr.p && B.qctr.alloc(4)) return -2;
            if (B.insts.alloc(sizeof(mnk::DagInst) * (size_t)ninst)) return -2;
            for (hipEvent_t& e : B.ev)
                if (!e) MNK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            B.ntile = ntile; B.ninst = ninst; B.period = period; B.chunk = l0->dag_chunk; B.taper0 = l0->dag_taper0;
            B.band = l0->dag_band; B.fill = l0->dag_has_fill; B.ntasks = (int)(merged.size() / 4);
        }
        std::vector<mnk::DagInst> hin;
        mnk::DagInst* insts_dev = reinterpret_cast<mnk::DagInst*>(B.insts.p);
        for (mnk_ls* ls : g) {
            hin.push_back(dag_instance(ls));
            ls->dag_filled = hin.back().zfill != nullptr;
            // (the instance's record for the bulk kernel rides along with the reset of its progress words)
            hipLaunchKernelGGL(mnk::dag_reset_kernel, dim3((unsigned)std::min<size_t>((nflags + 1023) / 1024, 64)), dim3(256), 0, h,
                               ls->dag_flags.p, (int64_t)nflags, ls->info_dev.p, hin.back(), insts_dev + (hin.size() - 1));
        }
        MNK_HIP(hipMemsetAsync(B.qctr.p, 0, 4 * sizeof(int), h));
        hipStream_t sp[2] = {c0->sp_dag, c0->sp_dagB}, su = c0->su_dagB;
        MNK_HIP(hipEventRecord(B.ev[0], h));
        MNK_HIP(hipStreamWaitEvent(sp[0], B.ev[0], 0));
        MNK_HIP(hipStreamWaitEvent(sp[1], B.ev[0], 0));
        MNK_HIP(hipStreamWaitEvent(su, B.ev[0], 0));
        const unsigned strips = (unsigned)std::min<int64_t>(l0->dag_band, Np / NBI);
        const int bulk_wgs = mnk_ctx_bulk_wgs(c0, 2 * c0->dag_cus, 3);   // (the grid that is resident at launch: see ls.hip)
        // Launch order: the first two chains, THEN the bulk kernel, then everything else -- the host needs ~3 ms for the
        // 4 x ninst launches of the chains' streams (measured: the bulk kernel used to start 3 ms after the first chain,
        // 2 % of a 16-instance step).  Per chain stream: chain i, its inverses for the solves and its inertia / info words
        // (the next chain of the stream starts behind them; the bulk kernel has work of the other instance meanwhile).
        auto launch_chain = [&](int i) -> int {
            mnk_ls* ls = g[i];
            mnk::PpDag dag{hin[i].front, hin[i].af, ntile, 0, 0, -1, mnk_ls_dag_spin_limit(ls), nullptr, mnk_ls_growth_word(ls)};
            return mnk_launch_pchain(ls, sp[i & 1], dag, 0, nsc, strips);
        };
        auto launch_tail = [&](int i) -> int {
            mnk_ls* ls = g[i];
            int r = mnk_ls_invert_blocks(ls, sp[i & 1], 0, nsc);
            if (r) return r;
            ls->inv_done = nsc;
            return mnk_ls_launch_finish_info(ls, sp[i & 1]);
        };
        for (int i = 0; i < std::min(2, ninst); ++i) {
            int r = launch_chain(i);
            if (r) return r;
        }
        int r = mnk::launch_dag_bulk(su, ldl, hin[0], insts_dev, B.tasks.p, B.ntasks, ntile, B.qctr.p,
                                     mnk_ls_dag_spin_limit(l0), std::min(B.ntasks, bulk_wgs), nullptr, nullptr);
        if (r) return r;
        for (int i = 0; i < ninst; ++i) {
            r = launch_tail(i);
            if (r) return r;
            if (i + 2 < ninst) {
                r = launch_chain(i + 2);
                if (r) return r;
            }
        }
        MNK_HIP(hipEventRecord(B.ev[1], sp[0]));
        MNK_HIP(hipEventRecord(B.ev[2], sp[1]));
        MNK_HIP(hipEventRecord(B.ev[3], su));
        for (int e = 1; e < 4; ++e) MNK_HIP(hipStreamWaitEvent(h, B.ev[e], 0));
        return 0;
    };
    rc = body();
    rc = mnk_persist_end(c0, h, rc);
    if (rc) return rc;
    // everybody's stream continues behind the batch
    MNK_HIP(hipEventRecord(l0->ev_defer, h));
    for (mnk_ls* ls : g) {
        if (ls->ctx->stream != h) MNK_HIP(hipStreamWaitEvent(ls->ctx->stream, l0->ev_defer, 0));
        batch_mark_done(ls);
        int r = mnk_ls_prefill_spare(ls);
        if (r) return r;
    }
    return 0;
}

// Small systems (every row a strip of the chain, dag_js2 == 0): rounds of K systems whose chains run SIDE BY SIDE in one
// launch (factor.hip: pchain_multi_kernel) on K * strips CUs, the band tiles of all of them accumulated by one bulk
// launch on the other CUs (systems of up to 768 rows have no bulk task at all).  A system this small is bound by its own
// pivot chain from the first column on -- 0.9 ms for N = 2048 on the whole chip, of which it uses 32 CUs.
static int batch_run_group_small(std::vector<mnk_ls*>& g) {
    mnk_ls* l0 = g[0];
    mnk_ctx* c0 = l0->ctx;
    hipStream_t h = c0->stream;
    MNK_HIP(hipSetDevice(c0->device));
    const int64_t Np = l0->Np;
    const bool ldl = l0->algo == MNK_LDL;
    const int ntile = (int)(Np / 128), nsc = (int)((Np + 255) / 256), strips = (int)(Np / NBI);
    const size_t nflags = (size_t)2 + (size_t)(Np / NBI) + 2 * (size_t)ntile * ntile;
    const int ntasks1 = l0->dag_ntasks;   // (single phase: all tasks)
    const int bulk_min = ntasks1 > 0 ? std::max(32, c0->num_cu / 4) : 0;
    // The chains' CU partition is a multiple of 32 CUs: every workgroup of the chain launch must be RESIDENT (one per CU), and
    // the dispatcher places them all only when the mask gives every shader engine of every XCD the same number of CUs --
    // 8 XCDs x 4 SEs = 32 (tools/hip/mask_resident.hip: masks of 32 / 64 / 96 / 192 / 256 bits hold as many 96-KB workgroups
    // as they have bits; 40, 48, 56, 72, 104, 176 do not, e.g. 34 workgroups on a 40-bit mask: 32 resident -- five 34-strip
    // systems on 176 CUs stalled into the schedule's time-out).
    // (... and of 64, so that a process ends up with three such stream pairs at most: every masked stream is a hardware queue,
    // and the device runs only so many of them side by side -- ls.hip, ctx_create_common)
    auto chain_cus_for = [&](int k) { return std::min(c0->num_cu, (k * strips + 63) / 64 * 64); };
    int kmax = std::min(32, (c0->num_cu - bulk_min) / strips);
    while (kmax > 1 && chain_cus_for(kmax) > c0->num_cu - bulk_min) --kmax;
    if (kmax < 2) {   // (no room for two chains: one after the other, as without a batch)
        for (mnk_ls* ls : g) {
            ls->deferred = false;
            int rc = mnk_ls_run_factorization_now(ls);
            if (rc) return rc;
        }
        return 0;
    }
    for (mnk_ls* ls : g)
        if (ls->ctx->stream != h) MNK_HIP(hipStreamWaitEvent(h, ls->ev_defer, 0));
    BatchBuffers& B = g_batch[c0->device & 63];
    for (size_t r0 = 0; r0 < g.size(); r0 += (size_t)kmax) {
        const int k = (int)std::min<size_t>(kmax, g.size() - r0);
        hipStream_t sp = nullptr, su = nullptr;
        int rc = mnk_masked_stream_pair(c0, chain_cus_for(k), &sp, &su);
        if (rc) return rc;
        if (ntasks1 > 0 && su == nullptr) { set_error("factorize batch: no CUs left for the bulk kernel"); return -1; }
        rc = mnk_persist_begin(c0, h);
        if (rc) return rc;
        auto body = [&]() -> int {
            if (ntasks1 > 0 && (B.ntile != ntile || B.ninst != k || B.period != 0 || B.chunk != l0->dag_chunk || B.taper0 != l0->dag_taper0 ||
                                B.band != l0->dag_band || B.fill != l0->dag_has_fill || !B.tasks.p)) {
                std::vector<int> merged;
                mnk::dag_merge_tasks(l0->dag_host_tasks, l0->dag_host_ready, k, 0, merged);
                if (B.tasks.alloc(merged.size() + 4)) return -2;
                { mnk::H2DGuard h2d; MNK_HIP(hipMemcpy(B.tasks.p, merged.data(), merged.size() * sizeof(int), hipMemcpyHostToDevice)); }
                B.ntile = ntile; B.ninst = k; B.period = 0; B.chunk = l0->dag_chunk; B.taper0 = l0->dag_taper0;
                B.band = l0->dag_band; B.fill = l0->dag_has_fill; B.ntasks = (int)(merged.size() / 4);
            }
            if (!B.qctr.p && B.qctr.alloc(4)) return -2;
            if (B.insts.n < sizeof(mnk::DagInst) * (size_t)k && B.insts.alloc(sizeof(mnk::DagInst) * (size_t)std::max(k, 32))) return -2;
            if (B.pcsys.n < mnk_pchain_sys_bytes() * (size_t)k && B.pcsys.alloc(mnk_pchain_sys_bytes() * (size_t)std::max(k, 32))) return -2;
            for (hipEvent_t& e : B.ev)
                if (!e) MNK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            // The records of this round (instances for the bulk kernel, the chains' table, what the kernels around them
            // need) go through pinned host memory in ONE piece each: a kernel launch per record and per small kernel made the
            // host the bound of a batch of many small systems (~8 launches per system).
            std::vector<mnk::DagInst> hin;
            std::vector<int*> fronts;
            std::vector<const int*> afs;
            mnk::DagInst* insts_dev = reinterpret_cast<mnk::DagInst*>(B.insts.p);
            const size_t pc_bytes = mnk_pchain_sys_bytes();
            const size_t need = (size_t)k * (sizeof(mnk::DagInst) + pc_bytes + sizeof(mnk::SmallSysRec));
            if (B.aux.n < sizeof(mnk::SmallSysRec) * (size_t)k && B.aux.alloc(sizeof(mnk::SmallSysRec) * (size_t)std::max(k, 32))) return -2;
            if (B.stage_ev == nullptr) MNK_HIP(hipEventCreateWithFlags(&B.stage_ev, hipEventDisableTiming));
            else MNK_HIP(hipEventSynchronize(B.stage_ev));   // (the previous round's uploads have left the staging memory)
            if (B.stage_bytes < need) {
                mnk::LaunchLock lock;
                if (B.stage) { mnk::quiesce_persistent(); (void)hipHostFree(B.stage); B.stage = nullptr; }
                const size_t cap = (size_t)std::max(k, 32) * (sizeof(mnk::DagInst) + pc_bytes + sizeof(mnk::SmallSysRec));
                MNK_HIP(hipHostMalloc((void**)&B.stage, cap, hipHostMallocDefault));
                B.stage_bytes = cap;
            }
            mnk::DagInst* st_inst = reinterpret_cast<mnk::DagInst*>(B.stage);
            char* st_pc = B.stage + (size_t)k * sizeof(mnk::DagInst);
            mnk::SmallSysRec* st_aux = reinterpret_cast<mnk::SmallSysRec*>(st_pc + (size_t)k * pc_bytes);
            const bool batch_aux = l0->linv_mfma && !l0->solve512;   // (the other inverse kernels have no batched form)
            for (int i = 0; i < k; ++i) {
                mnk_ls* ls = g[r0 + i];
                hin.push_back(dag_instance(ls));
                ls->dag_filled = hin.back().zfill != nullptr;
                fronts.push_back(hin.back().front);
                afs.push_back(hin.back().af);
                st_inst[i] = hin.back();
                mnk_pchain_fill_sys(ls, st_pc + (size_t)i * pc_bytes, hin.back().front, hin.back().af);
                mnk_ls_fill_small_rec(ls, st_aux + i);
            }
            MNK_HIP(hipMemcpyAsync(B.insts.p, st_inst, (size_t)k * sizeof(mnk::DagInst), hipMemcpyHostToDevice, h));
            MNK_HIP(hipMemcpyAsync(B.pcsys.p, st_pc, (size_t)k * pc_bytes, hipMemcpyHostToDevice, h));
            MNK_HIP(hipMemcpyAsync(B.aux.p, st_aux, (size_t)k * sizeof(mnk::SmallSysRec), hipMemcpyHostToDevice, h));
            MNK_HIP(hipEventRecord(B.stage_ev, h));
            const mnk::SmallSysRec* aux_dev = reinterpret_cast<const mnk::SmallSysRec*>(B.aux.p);
            hipLaunchKernelGGL(mnk::dag_reset_kernel, dim3((unsigned)std::min<size_t>((nflags + 1023) / 1024, 64), (unsigned)k), dim3(256), 0, h,
                               (int*)nullptr, (int64_t)0, (int*)nullptr, hin[0], (mnk::DagInst*)nullptr, aux_dev);
            MNK_HIP(hipMemsetAsync(B.qctr.p, 0, 4 * sizeof(int), h));
            MNK_HIP(hipEventRecord(B.ev[0], h));
            MNK_HIP(hipStreamWaitEvent(sp, B.ev[0], 0));
            if (su != nullptr) MNK_HIP(hipStreamWaitEvent(su, B.ev[0], 0));
            // (the records of the chains' table are written on the home stream before the fork)
            int r = mnk_launch_pchain_multi(g.data() + r0, k, sp, nullptr, B.pcsys.p, fronts.data(), afs.data());
            if (r) return r;
            if (ntasks1 > 0) {
                const int bulk_wgs = mnk_ctx_bulk_wgs(c0, chain_cus_for(k), 3);
                r = mnk::launch_dag_bulk(su, ldl, hin[0], insts_dev, B.tasks.p, B.ntasks, ntile, B.qctr.p, mnk_ls_dag_spin_limit(l0),
                                         std::min(B.ntasks, bulk_wgs), nullptr, nullptr);
                if (r) return r;
                MNK_HIP(hipEventRecord(B.ev[2], su));
                MNK_HIP(hipStreamWaitEvent(h, B.ev[2], 0));
            }
            MNK_HIP(hipEventRecord(B.ev[1], sp));
            MNK_HIP(hipStreamWaitEvent(h, B.ev[1], 0));
            if (su != nullptr) MNK_HIP(hipStreamWaitEvent(su, B.ev[1], 0));   // (all chains done: every system's factor is final)
            if (su != nullptr) MNK_HIP(hipStreamWaitEvent(sp, B.ev[2], 0));
            // inverses for the solves, inertia / info words: small latency-bound kernels, spread over the three streams
            if (batch_aux) {
                // ONE launch each for all systems of the round: 64 x 64 inverses, 256 x 256 inverses, inertia / info words
                r = mnk_ls_invert_blocks_batch(h, ldl, aux_dev, k, Np);
                if (r) return r;
                r = mnk_ls_finish_info_batch(h, aux_dev, k, ldl ? (l0->N >= 4096 ? 1024 : 256) : 64);
                if (r) return r;
                for (int i = 0; i < k; ++i) {
                    mnk_ls* ls = g[r0 + i];
                    ls->inv_done = nsc;
                    if (!ls->ev_info) MNK_HIP(hipEventCreateWithFlags(&ls->ev_info, hipEventDisableTiming));
                    MNK_HIP(hipEventRecord(ls->ev_info, h));
                    ls->ev_info_recorded = true;
                }
                return 0;
            }
            hipStream_t inv_s[3] = {h, sp, su != nullptr ? su : sp};
            for (int i = 0; i < k; ++i) {
                mnk_ls* ls = g[r0 + i];
                hipStream_t si = inv_s[i % 3];
                r = mnk_ls_invert_blocks(ls, si, 0, nsc);
                if (r) return r;
                ls->inv_done = nsc;
                r = mnk_ls_launch_finish_info(ls, si);
                if (r) return r;
            }
            MNK_HIP(hipEventRecord(B.ev[1], sp));
            MNK_HIP(hipStreamWaitEvent(h, B.ev[1], 0));
            if (su != nullptr) {
                MNK_HIP(hipEventRecord(B.ev[2], su));
                MNK_HIP(hipStreamWaitEvent(h, B.ev[2], 0));
            }
            return 0;
        };
        rc = body();
        rc = mnk_persist_end(c0, h, rc);
        if (rc) return rc;
    }
    MNK_HIP(hipEventRecord(l0->ev_defer, h));
    for (mnk_ls* ls : g) {
        if (ls->ctx->stream != h) MNK_HIP(hipStreamWaitEvent(ls->ctx->stream, l0->ev_defer, 0));
        batch_mark_done(ls);
        int r = mnk_ls_prefill_spare(ls);
        if (r) return r;
    }
    return 0;
}

static int batch_flush() {
    std::vector<mnk_ls*> pend;
    pend.swap(t_batch.pend);
    int rc_all = 0;
    while (!pend.empty()) {
        // group: same device, order, algorithm and schedule options as the first one pending
        mnk_ls* l0 = pend.front();
        std::vector<mnk_ls*> g, rest;
        for (mnk_ls* ls : pend) {
            const bool same = ls->ctx->device == l0->ctx->device && ls->Np == l0->Np && ls->algo == l0->algo && ls->dag_js2 == l0->dag_js2 && ls->dag_chunk == l0->dag_chunk &&
                              ls->dag_taper0 == l0->dag_taper0 && ls->dag_band == l0->dag_band && ls->dag_has_fill == l0->dag_has_fill;
            (same && std::find(g.begin(), g.end(), ls) == g.end() ? g : rest).push_back(ls);
        }
        pend.swap(rest);
        int rc;
        if (g.size() >= 2) {
            rc = l0->dag_js2 == 0 ? batch_run_group_small(g) : batch_run_group(g);
            if (rc)   // (nothing of the group counts as factorized: the factor buffers already hold the NEW, unfactored matrices, so
                      // a solve or an inertia call that ignores this error must not find the previous factorization's flag)
                for (mnk_ls* ls : g) { ls->deferred = false; ls->factorized = false; ls->info_valid = false; }
        } else {
            g[0]->deferred = false;
            if (g[0]->Np < g[0]->dag_min_rows) g[0]->algo_now = 4;   // (alone, a system below the schedule's window keeps its usual one)
            rc = mnk_ls_run_factorization_now(g[0]);
            if (rc) { g[0]->factorized = false; g[0]->info_valid = false; }
        }
        if (rc && !rc_all) rc_all = rc;
    }
    return rc_all;
}

int mnk_ls_sync_deferred(mnk_ls* ls) {
    {   // (queued solves of this solver run before anything else touches it)
        int rc_s = mnk_solve_sync_deferred(ls);
        if (rc_s) return rc_s;
    }
    return mnk_ls_sync_deferred_fact(ls);
}

// true: a factorize! call of this solver is queued in a batch that ANOTHER thread opened (its thread-local list holds the pointer)
bool mnk_ls_pending_elsewhere(const mnk_ls* ls) {
    return ls->deferred && std::find(t_batch.pend.begin(), t_batch.pend.end(), ls) == t_batch.pend.end();
}

int mnk_ls_sync_deferred_fact(mnk_ls* ls) {
    if (!ls->deferred) return 0;
    if (std::find(t_batch.pend.begin(), t_batch.pend.end(), ls) == t_batch.pend.end()) {
        set_error("a factorize! call of this solver is pending in a batch that another thread opened (mnk_factorize_batch_end must come first)");
        return -1;
    }
    return batch_flush();   // (the batch stays open: later factorize! calls form the next group)
}

extern "C" int mnk_factorize_batch_begin(void) {
    ++t_batch.depth;
    t_batch.active = true;
    return 0;
}

extern "C" int mnk_factorize_batch_end(void) {
    if (t_batch.depth <= 0) { set_error("mnk_factorize_batch_end: no batch is open on this thread"); return -1; }
    if (--t_batch.depth > 0) return 0;
    t_batch.active = false;
    return batch_flush();
}

// Diagnostics / tests: the task list of the schedule for a matrix of `ntile` 128-row tiles (host only, no device needed).
// out: 4 ints per task as the bulk kernel reads them (flags | chunk index << 8, I, J, kbeg | kend << 16), at most `cap`
// tasks are written; returns the number of tasks, *first_phase receives the number of first-phase tasks.
extern "C" int mnk_debug_dag_tasks(int ntile, int chunk, int band_tiles, int js2, int taper0, int* out, int cap, int* first_phase) {
    if (ntile <= 0 || chunk <= 0 || band_tiles <= 0 || taper0 <= 0) return -1;
    std::vector<int> h;
    const int n1 = mnk::dag_build_tasks(ntile, chunk, band_tiles, js2, h, taper0, nullptr);
    const int n = (int)(h.size() / 4);
    if (first_phase != nullptr) *first_phase = n1;
    if (out != nullptr)
        for (int i = 0; i < std::min(n, cap) * 4; ++i) out[i] = h[i];
    return n;
}

// Diagnostics / tests: the merged queue of `ninst` instances (with the zero-fill tasks if `fill`), 4 ints per task with the
// instance index in the upper half of the second one; returns the number of tasks.
extern "C" int mnk_debug_dag_merged_tasks(int ntile, int chunk, int band_tiles, int js2, int taper0, int fill, int ninst, int period,
                                          int* out, int cap) {
    if (ntile <= 0 || chunk <= 0 || band_tiles <= 0 || taper0 <= 0 || ninst <= 0 || period < 0) return -1;
    std::vector<int> h, ready, merged;
    int n1 = mnk::dag_build_tasks(ntile, chunk, band_tiles, js2, h, taper0, &ready);
    if (fill) n1 = mnk::dag_add_fill_tasks(ntile, h, ready, n1);
    mnk::dag_merge_tasks(h, ready, ninst, period, merged);
    const int n = (int)(merged.size() / 4);
    if (out != nullptr)
        for (int i = 0; i < std::min(n, cap) * 4; ++i) out[i] = merged[i];
    return n;
}
