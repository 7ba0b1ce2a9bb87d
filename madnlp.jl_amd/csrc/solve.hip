// Triangular solves with the blocked factor: `solve_linear_system!` of the
// AbstractLinearSolver contract (reference src/LinearSolvers/lapack_common.jl:75-81),
// replacing LAPACK dpotrs / dsytrs (reference src/LinearSolvers/lapack.jl:150-153,169-172).
//
// HBM-bound: each sweep reads the lower triangle once (8*N^2/2 bytes).  Both sweeps are
// right-looking over 256-column steps; the 256x256 diagonal triangles are applied as GEMVs
// with their explicit inverses, which the factorization computes once (`linv_tri_kernel<256>`, from
// the 64x64 inv(L_jj) blocks), so a step has no substitution chain at all:
//   forward : x_j = inv(L_jj) b_j (1 workgroup) ; b[below] -= L[below, j] x_j  (64 rows per
//             workgroup, 4 column quarters per row, coalesced down the columns)
//   backward: x_j = inv(L_jj)^T z_j             ; z[before] -= L[j, before]^T x_j (16 lanes per
//             column, 4 rows each = 512-byte column segments, shuffle reduction)
// Launches per solve: 4 * N/256.
#include <mutex>

#include <atomic>

#include "gemm_tile.h"
#include "ls.h"

namespace mnk {

constexpr int SB = 256;
typedef double v2d __attribute__((ext_vector_type(2)));

// ---- explicit inverse of every 256x256 diagonal triangle (unit diagonal for LDL) ----------------
//   X_qq = inv(L_qq);  X_bq = -inv(L_bb) * sum_{q<=b'<b} L_{b,b'} X_{b'q}   for b = q+1..nb-1
// 64x64x16 products with both operands staged in LDS.  Writes Inv (column-major, ld 256) and its
// transpose (for the backward sweep).
// c[i] += sum_k A[4ty+i][k] * B[k][tx]  (64 x 16 output slice);  As[k*64 + r] = A[r][k] ; Bs[j*64 + k] = B[k][j]
__device__ __forceinline__ void mm64x16_acc(const double* As, const double* Bs, double (&c)[4], int ty, int tx) {
#pragma unroll 8
    for (int k = 0; k < 64; ++k) {
        const double b = Bs[tx * 64 + k];
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = fma(As[k * 64 + 4 * ty + i], b, c[i]);
    }
}

// One workgroup per (diagonal S-block, 64-column block q of its inverse, 16-column slice cs of that block): the
// columns of an inverse are independent, so 16 workgroups share one 256x256 triangle (102 -> ~30 us critical path).
// S = 256: the step of the stepwise solves and of the 64-row persistent solve; S = 512: the 32-row persistent solve.
template <int S>
__global__ __launch_bounds__(256) void linv_tri_kernel(const double* __restrict__ F, int64_t ld,
                                                      const double* __restrict__ Linv64, double* __restrict__ Inv,
                                                      double* __restrict__ InvT, int64_t Np,
                                                      const int* __restrict__ info, int blk0) {
    __shared__ double As[64 * 64];
    __shared__ double Bs[16 * 64];
    if (*info != 0) return;
    const int64_t blk = (int64_t)blockIdx.x + blk0;  // S-row diagonal block
    const int q = blockIdx.y;            // column block of the inverse
    const int cs = blockIdx.z;           // 16-column slice of that block
    const int64_t j0 = blk * S;
    const int nb = (int)((Np - j0 < S ? Np - j0 : S) / 64);
    double* out = Inv + blk * (int64_t)(S * S);
    double* outT = InvT + blk * (int64_t)(S * S);
    const int t = threadIdx.x, ty = t & 15, tx = t >> 4;
    const int col = 64 * q + 16 * cs + tx;  // this thread's column of the inverse
    // zero the part of the slice above the diagonal block (rows of earlier blocks)
    for (int e = t; e < 64 * q * 16; e += 256) {
        const int r = e % (64 * q), c = 64 * q + 16 * cs + e / (64 * q);
        out[r + (int64_t)c * S] = 0.0;
        outT[c + (int64_t)r * S] = 0.0;
    }
    if (q >= nb) {  // padding block of a short last step: zeros
        for (int e = t; e < 16 * S; e += 256) {
            const int r = e % S, c = 64 * q + 16 * cs + e / S;
            out[r + (int64_t)c * S] = 0.0;
            outT[c + (int64_t)r * S] = 0.0;
        }
        return;
    }
    for (int b = q; b < S / 64; ++b) {
        double c[4] = {0.0, 0.0, 0.0, 0.0};
        if (b < nb) {
            if (b == q) {
                // X_qq = inv(L_qq): straight copy
                const double* Li = Linv64 + ((j0 >> 6) + q) * 4096;
#pragma unroll
                for (int i = 0; i < 4; ++i) c[i] = Li[(4 * ty + i) + 64 * (16 * cs + tx)];
            } else {
                // S = sum_{b'} L_{b,b'} X_{b',q}   (this slice's 16 columns)
                for (int bp = q; bp < b; ++bp) {
                    __syncthreads();
                    for (int e = t; e < 4096; e += 256) {
                        const int r = e & 63, k = e >> 6;
                        As[k * 64 + r] = F[(j0 + 64 * b + r) + (j0 + 64 * bp + k) * ld];  // L_{b,bp}[r][k]
                    }
                    for (int e = t; e < 1024; e += 256) {
                        const int k = e & 63, j = e >> 6;
                        Bs[j * 64 + k] = out[(64 * bp + k) + (int64_t)(64 * q + 16 * cs + j) * S];  // X_{bp,q}[k][j]
                    }
                    __syncthreads();
                    mm64x16_acc(As, Bs, c, ty, tx);
                }
                // X_bq = -inv(L_bb) * S : stage S (as B operand) and inv(L_bb) (as A operand)
                __syncthreads();
#pragma unroll
                for (int i = 0; i < 4; ++i) Bs[tx * 64 + 4 * ty + i] = c[i];
                const double* Li = Linv64 + ((j0 >> 6) + b) * 4096;
                for (int e = t; e < 4096; e += 256) As[e] = Li[e];  // As[k*64+r] = inv(L_bb)[r][k]
                __syncthreads();
#pragma unroll
                for (int i = 0; i < 4; ++i) c[i] = 0.0;
                mm64x16_acc(As, Bs, c, ty, tx);
#pragma unroll
                for (int i = 0; i < 4; ++i) c[i] = -c[i];
            }
        }
        // store the slice of block (b, q) and its transpose (zeros for the padding rows of a short last step)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 64 * b + 4 * ty + i;
            out[r + (int64_t)col * S] = c[i];
            outT[col + (int64_t)r * S] = c[i];
        }
        __threadfence_block();
        __syncthreads();
    }
}

// The same inverses on the matrix cores (the default).  Rows 64 q .. 64 q + 63 of Y = inv(L)^T of one 256-row triangle are
// what block row operations make of 64 rows of the identity: for j = q..3  Y_j = T_j inv(L_jj)^T (one product with the 64x64
// inverse of linv64_kernel -- the very formula of the scalar kernel above, transposed), then T_c -= Y_j L_cj^T for the later
// column blocks c.  One workgroup per (triangle, q), wave w = 16 rows in registers (C^T layout: register r of 16-column
// block g at lane (l15, l4) is Y[16 w + l15][16 g + l4 + 4 r]); the 64x64 blocks L_cj go through LDS.  ~25 us per
// workgroup against ~70 of the scalar kernel: 167 + 69 us of every factorize! at C3.
// (Block substitution with the 16x16 inverses of the pivot kernel instead of the 64x64 ones -- what the factorization's own
// triangular solves use -- was built first and is NOT good enough here: an explicit inverse is applied to every right-hand
// side, and on the AC-OPF systems (condition numbers ~1e19) the interior-point loop's Richardson refinement then diverged
// in one solve out of four: case118 16 iterations / 98 back-solves instead of 13 / 48, tools/acopf_richardson_ab.py.)
__global__ __launch_bounds__(256) void linv256_mfma_kernel(const double* __restrict__ F, int64_t ld, const double* __restrict__ Linv64,
                                                           double* __restrict__ Inv,
                                                           double* __restrict__ InvT, int64_t Np, const int* __restrict__ info,
                                                           int blk0, const mnk::SmallSysRec* __restrict__ recs = nullptr) {
    __shared__ v4f64 tile[1024];
    if (recs != nullptr) {   // (a batch of small systems: blockIdx.z = system)
        const mnk::SmallSysRec r = recs[blockIdx.z];
        F = r.F; ld = r.ld; Linv64 = r.linv; Inv = r.linv256; InvT = r.linv256t; Np = r.Np; info = r.info;
    }
    if (*info != 0) return;
    const int64_t blk = (int64_t)blockIdx.x + blk0;
    const int q = blockIdx.y;
    const int64_t j0 = blk * SB;
    const int nb = (int)((Np - j0 < SB ? Np - j0 : SB) / 64);
    double* out = Inv + blk * (int64_t)(SB * SB);
    double* outT = InvT + blk * (int64_t)(SB * SB);
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    const int row = 64 * q + 16 * w + l15;   // this lane's row of Y = column of the inverse
    v4f64 X[16];
#pragma unroll
    for (int g = 0; g < 16; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) X[g][r] = (q < nb && 16 * g + l4 + 4 * r == row) ? 1.0 : 0.0;
    auto step = [&](auto Jc) __attribute__((always_inline)) {
        constexpr int j = decltype(Jc)::value;
        if (j >= q && j < nb && q < nb) {   // (uniform)
            // Y_j^T = inv(L_jj) T_j^T, 16x16 block by block of the lower-triangular 64x64 inverse (descending, in place)
            const double* Li = Linv64 + ((j0 >> 6) + j) * 4096;
#pragma unroll
            for (int cb = 3; cb >= 0; --cb) {
                v4f64 x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ib = 0; ib <= cb; ++ib)
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        x = __builtin_amdgcn_mfma_f64_16x16x4f64(Li[(16 * cb + l15) + 64 * (16 * ib + 4 * s + l4)], X[4 * j + ib][s], x, 0, 0, 0);
                X[4 * j + cb] = x;
            }
            // T_c -= Y_j L_cj^T for the later column blocks (L_cj staged as in the persistent panel kernel)
#pragma unroll
            for (int c = j + 1; c < 4; ++c) {
                if (c >= nb) break;
                __syncthreads();
                {
                    const double* src = F + (j0 + 64 * c + lane) + (j0 + 64 * j + w) * ld;
#pragma unroll
                    for (int ib = 0; ib < 4; ++ib) {
                        v4f64 pre;
#pragma unroll
                        for (int s = 0; s < 4; ++s) pre[s] = src[(16 * ib + 4 * s) * ld];
                        tile[((lane >> 4) * 4 + ib) * 64 + (lane & 15) + 16 * w] = pre;
                    }
                }
                __syncthreads();
#pragma unroll
                for (int cb2 = 0; cb2 < 4; ++cb2)
#pragma unroll
                    for (int ib = 0; ib < 4; ++ib) {
                        const v4f64 a = tile[(cb2 * 4 + ib) * 64 + lane];
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                            X[4 * c + cb2] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[s], X[4 * j + ib][s], X[4 * c + cb2], 0, 0, 0);
                    }
            }
        }
    };
    step(std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{});
    step(std::integral_constant<int, 3>{});
    // Y[row][c] = inverse[c][row]: the transposed copy gets the rows as they are, the inverse itself the transposition
    // (zeros included: the solves read the full 256 x 256 squares)
#pragma unroll
    for (int g = 0; g < 16; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 16 * g + l4 + 4 * r;
            const double v = (c >= row && c < 64 * nb) ? X[g][r] : 0.0;
            outT[row + (int64_t)c * SB] = v;
            out[c + (int64_t)row * SB] = v;
        }
}

// x = M b for the 256x256 (column-major, ld 256) triangular matrix M of one diagonal step: one
// workgroup of 1024 threads, 4 column quarters per row (quarters that are structurally zero are
// skipped: M is lower triangular in the forward sweep, upper in the backward one).
__global__ __launch_bounds__(1024) void diag256_kernel(const double* __restrict__ M, const double* __restrict__ b,
                                                       double* __restrict__ x, int upper, int nrow) {
    __shared__ double bs[SB];
    __shared__ double part[4][SB];
    const int t = threadIdx.x, r = t & 255, p = t >> 8;
    if (t < SB) bs[t] = t < nrow ? b[t] : 0.0;
    __syncthreads();
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    const int rb = r >> 6;
    if (upper ? p >= rb : p <= rb) {
        const double* Mr = M + r + (int64_t)(64 * p) * SB;
        const double* bp = bs + 64 * p;
#pragma unroll 8
        for (int c = 0; c < 64; c += 4) {
            a0 += Mr[(int64_t)(c + 0) * SB] * bp[c + 0];
            a1 += Mr[(int64_t)(c + 1) * SB] * bp[c + 1];
            a2 += Mr[(int64_t)(c + 2) * SB] * bp[c + 2];
            a3 += Mr[(int64_t)(c + 3) * SB] * bp[c + 3];
        }
    }
    part[p][r] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (p == 0 && r < nrow) x[r] = (part[0][r] + part[1][r]) + (part[2][r] + part[3][r]);
}

__global__ void scale_vec_kernel(double* __restrict__ y, const double* __restrict__ dinv, int64_t n) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) y[i] *= dinv[i];
}

// forward panel: b[r] -= sum_c L[r, j0+c] x[c], rows r >= j0+ncol; 64 rows per workgroup
__global__ __launch_bounds__(256) void fwd_panel_kernel(const double* __restrict__ F, int64_t ld,
                                                        const double* __restrict__ x, double* __restrict__ b,
                                                        int64_t j0, int ncol, int64_t Np) {
    __shared__ double xs[SB];
    __shared__ double part[4][64];
    const int t = threadIdx.x, row = t & 63, p = t >> 6;
    xs[t] = t < ncol ? x[j0 + t] : 0.0;
    __syncthreads();
    const int64_t r = j0 + ncol + (int64_t)blockIdx.x * 64 + row;
    double s0 = 0.0, s1 = 0.0;
    if (r < Np) {
        const int c0 = p * (ncol / 4), c1 = c0 + ncol / 4;
        const double* Fr = F + r + j0 * ld;
#pragma unroll 8
        for (int c = c0; c < c1; c += 2) {
            s0 += Fr[(int64_t)c * ld] * xs[c];
            s1 += Fr[(int64_t)(c + 1) * ld] * xs[c + 1];
        }
    }
    part[p][row] = s0 + s1;
    __syncthreads();
    if (p == 0 && r < Np) b[r] -= (part[0][row] + part[1][row]) + (part[2][row] + part[3][row]);
}

// backward panel: z[col] -= sum_r L[j0+r, col] x[r], cols < j0; 64 columns per workgroup
__global__ __launch_bounds__(256) void bwd_panel_kernel(const double* __restrict__ F, int64_t ld,
                                                        const double* __restrict__ x, double* __restrict__ z,
                                                        int64_t j0, int nrow) {
    __shared__ double xs[SB];
    const int t = threadIdx.x;
    xs[t] = t < nrow ? x[j0 + t] : 0.0;
    __syncthreads();
    const int64_t cb = (int64_t)blockIdx.x * 64;
    const int lane = t & 63, w = t >> 6;
    const int sub = lane & 15, colq = lane >> 4;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int rb = 0; rb < nrow; rb += 64) {
        const double x0 = xs[rb + 4 * sub], x1 = xs[rb + 4 * sub + 1], x2 = xs[rb + 4 * sub + 2],
                     x3 = xs[rb + 4 * sub + 3];
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int64_t col = cb + pass * 16 + w * 4 + colq;
            const double* Fp = F + j0 + rb + 4 * sub + col * ld;
            acc[pass] += (Fp[0] * x0 + Fp[1] * x1) + (Fp[2] * x2 + Fp[3] * x3);
        }
    }
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        double s = acc[pass];
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        s += __shfl_xor(s, 8);
        const int64_t col = cb + pass * 16 + w * 4 + colq;
        if (sub == 0) z[col] -= s;
    }
}


// ================================================================================================
// Persistent solve: both sweeps in ONE launch.
//
// The multi-kernel path above spends 4*N/256 dependent launches of a few microseconds each, i.e. it
// is bound by launch/drain latency, not by HBM (1.6 ms for a 1 GB factor).  Here every 64-row block
// i of the right-hand side is owned by workgroup i mod G (G <= #CUs, so all workgroups are resident
// together) and the blocks talk through global memory, value by value:
//   * a published value is an 8-byte agent-scope store; a consumer polls the very element it needs
//     until it differs from the sentinel (all-ones NaN) the host wrote before the launch.  One hop is
//     ~0.6 us on gfx950 (tools/hop_latency.hip); there are no separate flags or fences.
//   * forward, 256-column step k (blocks b0..b0+3):  the owner of block i in the step publishes its
//     final right-hand side bfin[i], gathers bfin of the blocks above it inside the step and applies
//     its 64 rows of inv(L_kk) (explicit inverse, no substitution chain) -> publishes y[i].  Owners of
//     the blocks below poll y[b0..b0+3] and subtract L[i, step k] * y_k from their running block,
//     which lives in LDS for the whole solve.
//   * the 64x256 slice a workgroup is going to multiply (of L or of inv(L_kk)) is loaded into
//     registers BEFORE it starts polling, so the memory latency overlaps the wait and the critical
//     path per step is two hops plus two in-register 64-term dot products.
//   * backward is the mirror image with column blocks (the same owners: z = D^-1 y stays in LDS),
//     stepping from the last block to the first with inv(L_kk)^T.
// Every workgroup walks its tasks in the global order (sweep, step, diagonal role before update role);
// each wait is on a task earlier in that order, so with all workgroups resident there is no deadlock.
// A wait that exceeds the spin budget raises the abort flag and every workgroup leaves (the host
// reports a SolveException instead of hanging the device).
// ================================================================================================
constexpr int PS_MAXOWN = 12;            // owned blocks per workgroup: Np <= 12 * 64 * G
constexpr int PS_NEAR = 3;               // blocks within this many steps of the front poll eagerly
#ifndef PS_DAHEAD
#define PS_DAHEAD 3                      // steps between the request of a block's inverse slice and its diagonal role (1: rounds 2-5)
#endif

__device__ __forceinline__ void ps_publish(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Workgroup barrier for LDS traffic only.  __syncthreads() also waits for every global load the wave has in flight (its fence names no
// address space): in the persistent solve that is the slice of L requested on purpose a step before it is needed.
__device__ __forceinline__ void ps_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// threads t < n poll src[t] into dst[t]; returns false (uniformly) if the solve was aborted.
// (`s_bad`: a word in LDS, zero while the solve is alive -- the barrier is then ps_barrier, not __syncthreads)
// `relaxed`: the caller is several steps away from the critical path -- nap between polls, so that the
// ~170 workgroups that merely follow the front do not hammer the 16 cache lines the front is
// publishing into (their polls queue in front of the critical stores and loads on the same channel).
__device__ __forceinline__ bool ps_gather(const double* src, int n, double* dst, int* abort_flag, long spin_limit,
                                          bool relaxed = false, int nap = 6, int* s_bad = nullptr) {
    const int t = threadIdx.x;
    int bad = 0;
    if (t < n) {
        const unsigned long long* p = reinterpret_cast<const unsigned long long*>(src + t);
        unsigned long long u = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        while (u == ~0ull) {
            // back off: short naps while the value is probably about to land, long ones for the
            // workgroups that wait for a distant step (keeps their polling off the memory fabric)
            if (relaxed) { for (int z = 0; z < nap; ++z) __builtin_amdgcn_s_sleep(8); }
            else if (spins < 256) __builtin_amdgcn_s_sleep(1);
            else __builtin_amdgcn_s_sleep(24);
            if ((++spins & 1023) == 0) {
                if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { bad = 1; break; }
                if (spins > spin_limit) {
                    __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    bad = 1;
                    break;
                }
            }
            u = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        dst[t] = __longlong_as_double((long long)u);
    }
    if (s_bad != nullptr) {
        if (bad) *s_bad = 1;
        ps_barrier();
        return *s_bad == 0;
    }
    return __syncthreads_or(bad) == 0;
}

__global__ void pad_copy_kernel(double* __restrict__ dst, const double* __restrict__ src, int64_t N, int64_t Np) {
    const int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (k < Np) dst[k] = k < N ? src[k] : 0.0;
}

// (the abort word is cleared by the host only, after it has dealt with an abort: a solve that follows an
// aborted one on the same stream must not hide it)
__global__ void ps_reset_kernel(unsigned long long* pub, int64_t n) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) pub[i] = ~0ull;
}

constexpr int PS_NT = 1024;          // threads per workgroup: a 64 x 256 slice is 16 values per thread
constexpr int PS_NQ = PS_NT / 64;    // 16-column chunks of a step (forward) / waves (backward)
constexpr int PS_CW = 256 / PS_NQ;   // columns per chunk

// (the body of one system's solve: workgroup g of the G that share the system; the kernels below wrap it)
template <bool LDL>
__device__ __forceinline__ void persistent_solve_body(
    const double* __restrict__ F, int64_t ld, const double* __restrict__ Inv, const double* __restrict__ InvT,
    const double* __restrict__ dinv, const double* xin /* N: right-hand side (rows N..Np-1 are zero) */,
    double* xout /* N: solution (may alias xin) */, int64_t N,
    double* __restrict__ pub /* 4*Np sentinel-filled: bfin | y | zfin | x */,
    double* __restrict__ pub_other /* the buffer of the NEXT solve: this launch fills it with the sentinel (or NULL) */,
    int64_t Np, int* abort_flag,
    const int* __restrict__ info, unsigned long long* __restrict__ trace /* optional: 8 stamps per block */,
    int near_steps, int nap, long spin_limit, int missing_wg, const int G, const int g) {
#define PS_STAMP(blk, slot) do { if (trace != nullptr && t == 0) trace[(int64_t)(blk) * 8 + (slot)] = wall_clock64(); } while (0)
    __shared__ double run[PS_MAXOWN][64];   // running rhs of the owned blocks (forward: b, backward: z)
    __shared__ double ysol[PS_MAXOWN][64];  // forward solution of the owned blocks
    __shared__ double xs[256];              // the step's published vector
    __shared__ double part[PS_NQ][64];      // partial sums
    __shared__ int s_bad;                   // set by a thread whose wait was given up (ps_gather)
    const int t = threadIdx.x, r = t & 63, q = t >> 6;
    const int nb = (int)(Np / 64);
    const int nsteps = (nb + 3) / 4;
    const int nown = g < nb ? (nb - g + G - 1) / G : 0;
    // The sentinel fill for the NEXT solve (it uses the other of two publication buffers): every workgroup clears the four
    // words of each element of its own blocks -- round 3 launched a reset kernel in front of every solve.  First thing,
    // whatever happens to this solve.
    if (pub_other != nullptr)
        for (int e = t; e < nown * 256; e += PS_NT) {
            const int m = e >> 8, arr = (e >> 6) & 3;
            reinterpret_cast<unsigned long long*>(pub_other)[(int64_t)arr * Np + (int64_t)(g + m * G) * 64 + (e & 63)] = ~0ull;
        }
    if (*info != 0) return;
    if (nown == 0) return;
    if (g == missing_wg) return;  // tests: a peer that never became resident (everyone else must give up, not hang)
    double* bfin = pub;
    double* ypub = pub + Np;
    double* zfin = pub + 2 * Np;
    double* xpub = pub + 3 * Np;
    for (int e = t; e < nown * 64; e += PS_NT) {
        const int64_t row = (int64_t)(g + (e >> 6) * G) * 64 + (e & 63);
        run[e >> 6][e & 63] = row < N ? xin[row] : 0.0;   // (the caller's vector itself: no padded staging copy)
    }
    if (t == 0) s_bad = 0;
    __syncthreads();

    double a[PS_CW];  // slice of L the next update multiplies
    double d[PS_CW];  // slice of inv(L_kk) / inv(L_kk)^T the next diagonal role multiplies
    // dot product of a register slice with chunk q of xs, and the 16-way reduction over the chunks
    auto dot_chunk = [&](const double (&v)[PS_CW], bool active) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        if (active) {
            const double* xq = xs + q * PS_CW;
#pragma unroll
            for (int j = 0; j < PS_CW; j += 4) {
                s0 = fma(v[j], xq[j], s0);
                s1 = fma(v[j + 1], xq[j + 1], s1);
                s2 = fma(v[j + 2], xq[j + 2], s2);
                s3 = fma(v[j + 3], xq[j + 3], s3);
            }
        }
        part[q][r] = (s0 + s1) + (s2 + s3);
    };
    auto reduce_parts = [&](int c) {
        double sm = 0.0;
#pragma unroll
        for (int u = 0; u < PS_NQ; ++u) sm += part[u][c];
        return sm;
    };
    const int qb = (q * PS_CW) >> 6;  // 64-block (within the step) that column chunk q belongs to

    // ------------------------------------------------------------------ forward: L y = b
    int m0 = 0;           // first owned block whose forward solution is not yet known
    bool have_d = false;  // d holds the inverse's slice for block own[m0]
    bool pub_b = false;   // ... and its final right-hand side is already published
    // The slice of the inverse a block multiplies in its diagonal role is requested PS_DAHEAD steps before that step, not in the step
    // right before it: a workgroup's requests go through ONE texture addresser at ~14 cycles per wave instruction, so the 128 KB of L for
    // the last update (256 wave instructions) plus up to 128 KB of the inverse took 2.5 (first block of a step) to 4.4 us (fourth) to
    // arrive -- longer than the step's y took to get there: the gather's barrier waited for the slices, not for the hop (round 6,
    // tools/solve_trace.py on a build with -DMNK_DIAG_SOLVE_PREFETCH).  Same arithmetic, same bits.
    for (int k = 0; k < nsteps && m0 < nown; ++k) {
        const int b0 = 4 * k, nbk = nb - b0 < 4 ? nb - b0 : 4;
        int i = g + m0 * G;
        if (i < b0 + nbk) {
            // diagonal role: y_i = sum_{c <= li} inv(L_kk)[li, c] * bfin[b0 + c]
            const int li = i - b0;
            if (!have_d) {  // first steps, or a block that was never near (tiny systems)
                const double* Mk = Inv + (int64_t)k * (SB * SB) + (li * 64 + r) + (int64_t)(q * PS_CW) * SB;
                if (qb <= li) {
#pragma unroll
                    for (int j = 0; j < PS_CW; ++j) d[j] = Mk[(int64_t)j * SB];
                }
            }
            if (!pub_b && t < 64) ps_publish(bfin + (int64_t)i * 64 + t, run[m0][t]);
            if (t < 64) xs[li * 64 + t] = run[m0][t];
            if (!ps_gather(bfin + (int64_t)b0 * 64, li * 64, xs, abort_flag, spin_limit, false, 6, &s_bad)) return;
            PS_STAMP(i, 3);
            dot_chunk(d, qb <= li);
            ps_barrier();
            if (t < 64) {
                const double v = reduce_parts(t);
                ps_publish(ypub + (int64_t)i * 64 + t, v);
                ysol[m0][t] = v;
            }
            PS_STAMP(i, 4);
            // (no trailing barrier: the next touch of xs / part comes behind the barrier of the next gather)
            have_d = false;
            pub_b = false;
            ++m0;
            if (m0 >= nown) break;
            i = g + m0 * G;
        }
        // update role: run_i -= L[i, step k] * y_k for the owned blocks below the step.  Everything the
        // critical block needs next is requested before the wait: its slice of L for this step and, PS_DAHEAD steps ahead
        // of its diagonal step, its slice of that step's inverse.
        const bool diag_next = i < b0 + nbk + 4;
        {
            const double* Fs = F + ((int64_t)i * 64 + r) + ((int64_t)b0 * 64 + q * PS_CW) * ld;
            if (qb < nbk) {
#pragma unroll
                for (int j = 0; j < PS_CW; ++j) a[j] = Fs[(int64_t)j * ld];
            }
            if (!have_d && i < b0 + nbk + 4 * PS_DAHEAD) {   // (block i sits in a later, hence full, step i >> 2)
                const int li = i & 3;
                const double* Mk = Inv + (int64_t)(i >> 2) * (SB * SB) + (li * 64 + r) + (int64_t)(q * PS_CW) * SB;
                if (qb <= li) {
#pragma unroll
                    for (int j = 0; j < PS_CW; ++j) d[j] = Mk[(int64_t)j * SB];
                }
                have_d = true;
            }
        }
        if (diag_next) PS_STAMP(i, 0);
        if (!ps_gather(ypub + (int64_t)b0 * 64, nbk * 64, xs, abort_flag, spin_limit, i >= b0 + nbk + 4 * near_steps, nap, &s_bad)) return;
        if (diag_next) PS_STAMP(i, 1);
        for (int m = m0; m < nown; ++m) {
            const int im = g + m * G;
            if (m > m0 && qb < nbk) {
                const double* Fs = F + ((int64_t)im * 64 + r) + ((int64_t)b0 * 64 + q * PS_CW) * ld;
#pragma unroll
                for (int j = 0; j < PS_CW; ++j) a[j] = Fs[(int64_t)j * ld];
            }
            dot_chunk(a, qb < nbk);
            ps_barrier();
            if (t < 64) {
                const double v = run[m][t] - reduce_parts(t);
                run[m][t] = v;
                if (m == m0 && diag_next) ps_publish(bfin + (int64_t)im * 64 + t, v);  // final: hand it on at once
            }
            if (m == m0 && diag_next) PS_STAMP(im, 2);
            if (m + 1 < nown) ps_barrier();  // part is rewritten by the next owned block; after the last one the
                                                // next gather's barrier does it
        }
        pub_b = diag_next;
    }

    // ------------------------------------------------------------------ backward: L^T x = D^-1 y
    __syncthreads();  // ysol of the last diagonal role is read by other threads below
    for (int e = t; e < nown * 64; e += PS_NT) {
        const int m = e >> 6, c = e & 63;
        const double y = ysol[m][c];
        run[m][c] = LDL ? y * dinv[(int64_t)(g + m * G) * 64 + c] : y;
    }
    __syncthreads();
    // update-role mapping: wave q -> row chunk rc = q & 3 (64 rows of the step), column group cg = q >> 2
    // (16 columns of the block); lane (sub, colq): rows 4*sub..4*sub+3, columns cg*16 + 4p + colq, p < 4.
    const int lane = t & 63, sub = lane & 15, colq = lane >> 4, rc = q & 3, cg = q >> 2;
    int m1 = nown - 1;  // last owned block whose solution is not yet known
    have_d = false;
    pub_b = false;
    for (int k = nsteps - 1; k >= 0 && m1 >= 0; --k) {
        const int b0 = 4 * k, nbk = nb - b0 < 4 ? nb - b0 : 4;
        int i = g + m1 * G;
        if (i >= b0) {
            // diagonal role: x_i = sum_{rc >= li} inv(L_kk)^T[li, rc] * zfin[b0 + rc]
            const int li = i - b0;
            const bool act = qb >= li && qb < nbk;
            if (!have_d) {
                const double* Mk = InvT + (int64_t)k * (SB * SB) + (li * 64 + r) + (int64_t)(q * PS_CW) * SB;
                if (act) {
#pragma unroll
                    for (int j = 0; j < PS_CW; ++j) d[j] = Mk[(int64_t)j * SB];
                }
            }
            if (!pub_b && t < 64) ps_publish(zfin + (int64_t)i * 64 + t, run[m1][t]);
            if (t < 64) xs[li * 64 + t] = run[m1][t];
            // blocks li+1 .. nbk-1 of the step come from their owners
            if (!ps_gather(zfin + (int64_t)(i + 1) * 64, (nbk - 1 - li) * 64, xs + (li + 1) * 64, abort_flag, spin_limit, false, 6, &s_bad)) return;
            dot_chunk(d, act);
            ps_barrier();
            if (t < 64) {
                const double v = reduce_parts(t);
                ps_publish(xpub + (int64_t)i * 64 + t, v);
                if ((int64_t)i * 64 + t < N) xout[(int64_t)i * 64 + t] = v;
            }
            have_d = false;
            pub_b = false;
            --m1;
            if (m1 < 0) break;
            i = g + m1 * G;
        }
        // update role: run_i -= L[step k rows, block i columns]^T * x_k for the owned blocks before the step
        auto load_slice = [&](int im, double (&dst)[PS_CW], int bs) {
            const double* Fs = F + ((int64_t)bs * 64 + 64 * rc + 4 * sub) + ((int64_t)im * 64 + cg * 16 + colq) * ld;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const v2d lo = *reinterpret_cast<const v2d*>(Fs + (int64_t)(4 * p) * ld);
                const v2d hi = *reinterpret_cast<const v2d*>(Fs + (int64_t)(4 * p) * ld + 2);
                dst[4 * p] = lo[0];
                dst[4 * p + 1] = lo[1];
                dst[4 * p + 2] = hi[0];
                dst[4 * p + 3] = hi[1];
            }
        };
        const bool diag_next = i >= b0 - 4;  // the critical block sits in the step right before this one
        if (rc < nbk) load_slice(i, a, b0);
        if (!have_d && i >= b0 - 4 * PS_DAHEAD) {   // (block i sits in an earlier step i >> 2: a full one)
            const int li = i & 3;
            const double* Mk = InvT + (int64_t)(i >> 2) * (SB * SB) + (li * 64 + r) + (int64_t)(q * PS_CW) * SB;
            if (qb >= li) {
#pragma unroll
                for (int j = 0; j < PS_CW; ++j) d[j] = Mk[(int64_t)j * SB];
            }
            have_d = true;
        }
        if (!ps_gather(xpub + (int64_t)b0 * 64, nbk * 64, xs, abort_flag, spin_limit, i < b0 - 4 * near_steps, nap, &s_bad)) return;
        for (int m = m1; m >= 0; --m) {
            const int im = g + m * G;
            if (m < m1 && rc < nbk) load_slice(im, a, b0);
            // part[rc * 4 + (sub >> 2)][col]: 16 partial sums per column
            if (rc < nbk) {
                const double x0 = xs[64 * rc + 4 * sub], x1 = xs[64 * rc + 4 * sub + 1],
                             x2 = xs[64 * rc + 4 * sub + 2], x3 = xs[64 * rc + 4 * sub + 3];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    double sp = (a[4 * p] * x0 + a[4 * p + 1] * x1) + (a[4 * p + 2] * x2 + a[4 * p + 3] * x3);
                    sp += __shfl_xor(sp, 1);
                    sp += __shfl_xor(sp, 2);
                    if ((sub & 3) == 0) part[rc * 4 + (sub >> 2)][cg * 16 + 4 * p + colq] = sp;
                }
            } else if (lane < 16) {
                // idle row chunks contribute zeros to their 4 rows of `part` (16 columns per wave)
#pragma unroll
                for (int u = 0; u < 4; ++u) part[rc * 4 + u][cg * 16 + lane] = 0.0;
            }
            ps_barrier();
            if (t < 64) {
                const double v = run[m][t] - reduce_parts(t);
                run[m][t] = v;
                if (m == m1 && diag_next) ps_publish(zfin + (int64_t)im * 64 + t, v);
            }
            if (m > 0) ps_barrier();
        }
        pub_b = diag_next;
    }
#undef PS_STAMP
}

template <bool LDL>
__global__ __launch_bounds__(PS_NT) void persistent_solve_kernel(
    const double* __restrict__ F, int64_t ld, const double* __restrict__ Inv, const double* __restrict__ InvT,
    const double* __restrict__ dinv, const double* xin, double* xout, int64_t N, double* __restrict__ pub,
    double* __restrict__ pub_other, int64_t Np, int* abort_flag, const int* __restrict__ info,
    unsigned long long* __restrict__ trace, int near_steps, int nap, long spin_limit, int missing_wg) {
    persistent_solve_body<LDL>(F, ld, Inv, InvT, dinv, xin, xout, N, pub, pub_other, Np, abort_flag, info, trace, near_steps, nap,
                               spin_limit, missing_wg, (int)gridDim.x, (int)blockIdx.x);
}

// Several INDEPENDENT systems of the same order in one launch (mnk_solve_batch_begin / _end; scenario batches): workgroups
// [G i, G (i + 1)) work on system i exactly as the G workgroups of a single launch would (every workgroup of the launch must
// be resident: Q G <= number of CUs).  One solve is bound by its chain of hops, not by HBM (0.45 ms for 1 GB at N = 11 192,
// 2.2 TB/s): four of them side by side share the chip's bandwidth and the launch takes hardly longer than one.  Same
// arithmetic per system as a lone solve with G workgroups (the block -> workgroup map only decides who computes what).
struct PsSys {
    const double* F;
    int64_t ld;
    const double* Inv;
    const double* InvT;
    const double* dinv;
    const double* xin;
    double* xout;
    int64_t N;
    double* pub;
    double* pub_other;
    int* abort_flag;
    const int* info;
};
typedef const PsSys __attribute__((address_space(4))) * PsSysP;

template <bool LDL>
__global__ __launch_bounds__(PS_NT) void persistent_solve_multi_kernel(const PsSys* __restrict__ sys, int G, int64_t Np, int near_steps,
                                                                      int nap, long spin_limit) {
    const int isys = (int)blockIdx.x / G, g = (int)blockIdx.x % G;
    const PsSysP y = (PsSysP)(uintptr_t)(sys + isys);
    persistent_solve_body<LDL>(y->F, y->ld, y->Inv, y->InvT, y->dinv, y->xin, y->xout, y->N, y->pub, y->pub_other, Np, y->abort_flag,
                               y->info, nullptr, near_steps, nap, spin_limit, -1, G, g);
}

__global__ void ps_set_sys_kernel(PsSys rec, PsSys* __restrict__ dst) { *dst = rec; }


// ================================================================================================
// The same solve with steps of 512 columns: half as many steps, each still two hops.  A 64 x 512 slice per thread would be
// 32 values each for L and for the inverse -- more than the 128 registers a 1024-thread workgroup leaves per lane -- so the
// blocks are 32 ROWS here (a 32 x 512 slice is 16 values per thread, as before): 16 blocks per step, Np / 32 blocks owned
// round-robin by G <= #CUs workgroups, the explicit inverses are those of the 512 x 512 diagonal triangles (linv_tri_kernel
// <512>).  Thread t: row r = t & 31 of a block, chunk q = t >> 5 (32 chunks of 16 step columns) in the forward sweep and in
// both diagonal roles; in the backward update wave w = t >> 6 takes the 32 rows w of the step, lane (sub = lane & 7: rows
// 4 sub .. 4 sub + 3, colq = lane >> 3: columns colq + 8 p, p < 4).
// ================================================================================================
constexpr int P5_RB = 32, P5_NBS = 16, P5_STEP = 512, P5_NQ = 32, P5_CW = 16;

template <bool LDL>
__global__ __launch_bounds__(PS_NT) void persistent_solve512_kernel(
    const double* __restrict__ F, int64_t ld, const double* __restrict__ Inv, const double* __restrict__ InvT,
    const double* __restrict__ dinv, double* __restrict__ xio, double* __restrict__ pub, int64_t Np, int* abort_flag,
    const int* __restrict__ info, int near_steps, int nap, long spin_limit, int missing_wg) {
    __shared__ double run[PS_MAXOWN][P5_RB];
    __shared__ double ysol[PS_MAXOWN][P5_RB];
    __shared__ double xs[P5_STEP];
    __shared__ double part[P5_NQ][P5_RB];
    if (*info != 0) return;
    const int t = threadIdx.x, r = t & 31, q = t >> 5;
    const int G = gridDim.x, g = blockIdx.x;
    const int nb = (int)(Np / P5_RB);
    const int nsteps = (nb + P5_NBS - 1) / P5_NBS;
    const int nown = g < nb ? (nb - g + G - 1) / G : 0;
    if (nown == 0) return;
    if (g == missing_wg) return;
    double* bfin = pub;
    double* ypub = pub + Np;
    double* zfin = pub + 2 * Np;
    double* xpub = pub + 3 * Np;
    for (int e = t; e < nown * P5_RB; e += PS_NT) run[e >> 5][e & 31] = xio[(int64_t)(g + (e >> 5) * G) * P5_RB + (e & 31)];
    __syncthreads();

    double a[P5_CW], d[P5_CW];
    auto dot_chunk = [&](const double (&v)[P5_CW], bool active) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        if (active) {
            const double* xq = xs + q * P5_CW;
#pragma unroll
            for (int j = 0; j < P5_CW; j += 4) {
                s0 = fma(v[j], xq[j], s0);
                s1 = fma(v[j + 1], xq[j + 1], s1);
                s2 = fma(v[j + 2], xq[j + 2], s2);
                s3 = fma(v[j + 3], xq[j + 3], s3);
            }
        }
        part[q][r] = (s0 + s1) + (s2 + s3);
    };
    auto reduce_parts = [&](int c, int n) {
        double sm = 0.0;
        for (int u = 0; u < n; ++u) sm += part[u][c];
        return sm;
    };
    const int qb = q >> 1;  // 32-row block (within the step) that column chunk q belongs to

    // ------------------------------------------------------------------ forward: L y = b
    int m0 = 0;
    bool have_d = false;
    for (int k = 0; k < nsteps && m0 < nown; ++k) {
        const int b0 = P5_NBS * k, nbk = nb - b0 < P5_NBS ? nb - b0 : P5_NBS;
        int i = g + m0 * G;
        if (i < b0 + nbk) {
            const int li = i - b0;
            if (!have_d) {
                const double* Mk = Inv + (int64_t)k * (P5_STEP * P5_STEP) + (li * P5_RB + r) + (int64_t)(q * P5_CW) * P5_STEP;
                if (qb <= li) {
#pragma unroll
                    for (int j = 0; j < P5_CW; ++j) d[j] = Mk[(int64_t)j * P5_STEP];
                }
                if (t < P5_RB) ps_publish(bfin + (int64_t)i * P5_RB + t, run[m0][t]);
            }
            if (t < P5_RB) xs[li * P5_RB + t] = run[m0][t];
            if (!ps_gather(bfin + (int64_t)b0 * P5_RB, li * P5_RB, xs, abort_flag, spin_limit)) return;
            dot_chunk(d, qb <= li);
            __syncthreads();
            if (t < P5_RB) {
                const double v = reduce_parts(t, P5_NQ);
                ps_publish(ypub + (int64_t)i * P5_RB + t, v);
                ysol[m0][t] = v;
            }
            have_d = false;
            ++m0;
            if (m0 >= nown) break;
            i = g + m0 * G;
        }
        const bool diag_next = i < b0 + nbk + P5_NBS;
        {
            const double* Fs = F + ((int64_t)i * P5_RB + r) + ((int64_t)b0 * P5_RB + q * P5_CW) * ld;
            if (qb < nbk) {
#pragma unroll
                for (int j = 0; j < P5_CW; ++j) a[j] = Fs[(int64_t)j * ld];
            }
            if (diag_next) {
                const int li = i - (b0 + nbk);
                const double* Mk = Inv + (int64_t)(k + 1) * (P5_STEP * P5_STEP) + (li * P5_RB + r) + (int64_t)(q * P5_CW) * P5_STEP;
                if (qb <= li) {
#pragma unroll
                    for (int j = 0; j < P5_CW; ++j) d[j] = Mk[(int64_t)j * P5_STEP];
                }
            }
        }
        if (!ps_gather(ypub + (int64_t)b0 * P5_RB, nbk * P5_RB, xs, abort_flag, spin_limit, i >= b0 + nbk + P5_NBS * near_steps, nap)) return;
        for (int m = m0; m < nown; ++m) {
            const int im = g + m * G;
            if (m > m0 && qb < nbk) {
                const double* Fs = F + ((int64_t)im * P5_RB + r) + ((int64_t)b0 * P5_RB + q * P5_CW) * ld;
#pragma unroll
                for (int j = 0; j < P5_CW; ++j) a[j] = Fs[(int64_t)j * ld];
            }
            dot_chunk(a, qb < nbk);
            __syncthreads();
            if (t < P5_RB) {
                const double v = run[m][t] - reduce_parts(t, P5_NQ);
                run[m][t] = v;
                if (m == m0 && diag_next) ps_publish(bfin + (int64_t)im * P5_RB + t, v);
            }
            if (m + 1 < nown) __syncthreads();
        }
        have_d = diag_next;
    }

    // ------------------------------------------------------------------ backward: L^T x = D^-1 y
    __syncthreads();
    for (int e = t; e < nown * P5_RB; e += PS_NT) {
        const int m = e >> 5, c = e & 31;
        const double y = ysol[m][c];
        run[m][c] = LDL ? y * dinv[(int64_t)(g + m * G) * P5_RB + c] : y;
    }
    __syncthreads();
    const int lane = t & 63, sub = lane & 7, colq = lane >> 3, rc = t >> 6;   // update role: wave rc -> rows 32 rc .. of the step
    int m1 = nown - 1;
    have_d = false;
    for (int k = nsteps - 1; k >= 0 && m1 >= 0; --k) {
        const int b0 = P5_NBS * k, nbk = nb - b0 < P5_NBS ? nb - b0 : P5_NBS;
        int i = g + m1 * G;
        if (i >= b0) {
            const int li = i - b0;
            const bool act = qb >= li && qb < nbk;
            if (!have_d) {
                const double* Mk = InvT + (int64_t)k * (P5_STEP * P5_STEP) + (li * P5_RB + r) + (int64_t)(q * P5_CW) * P5_STEP;
                if (act) {
#pragma unroll
                    for (int j = 0; j < P5_CW; ++j) d[j] = Mk[(int64_t)j * P5_STEP];
                }
                if (t < P5_RB) ps_publish(zfin + (int64_t)i * P5_RB + t, run[m1][t]);
            }
            if (t < P5_RB) xs[li * P5_RB + t] = run[m1][t];
            if (!ps_gather(zfin + (int64_t)(i + 1) * P5_RB, (nbk - 1 - li) * P5_RB, xs + (li + 1) * P5_RB, abort_flag, spin_limit)) return;
            dot_chunk(d, act);
            __syncthreads();
            if (t < P5_RB) {
                const double v = reduce_parts(t, P5_NQ);
                ps_publish(xpub + (int64_t)i * P5_RB + t, v);
                xio[(int64_t)i * P5_RB + t] = v;
            }
            have_d = false;
            --m1;
            if (m1 < 0) break;
            i = g + m1 * G;
        }
        // update role: run_i -= L[step k rows, block i columns]^T * x_k for the owned blocks before the step
        auto load_slice = [&](int im) {
            const double* Fs = F + ((int64_t)b0 * P5_RB + 32 * rc + 4 * sub) + ((int64_t)im * P5_RB + colq) * ld;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const v2d lo = *reinterpret_cast<const v2d*>(Fs + (int64_t)(8 * p) * ld);
                const v2d hi = *reinterpret_cast<const v2d*>(Fs + (int64_t)(8 * p) * ld + 2);
                a[4 * p] = lo[0];
                a[4 * p + 1] = lo[1];
                a[4 * p + 2] = hi[0];
                a[4 * p + 3] = hi[1];
            }
        };
        const bool diag_next = i >= b0 - P5_NBS;
        if (rc < nbk) load_slice(i);
        if (diag_next) {
            const int li = i - (b0 - P5_NBS);
            const double* Mk = InvT + (int64_t)(k - 1) * (P5_STEP * P5_STEP) + (li * P5_RB + r) + (int64_t)(q * P5_CW) * P5_STEP;
            if (qb >= li) {  // step k-1 is a full step
#pragma unroll
                for (int j = 0; j < P5_CW; ++j) d[j] = Mk[(int64_t)j * P5_STEP];
            }
        }
        if (!ps_gather(xpub + (int64_t)b0 * P5_RB, nbk * P5_RB, xs, abort_flag, spin_limit, i < b0 - P5_NBS * near_steps, nap)) return;
        for (int m = m1; m >= 0; --m) {
            const int im = g + m * G;
            if (m < m1 && rc < nbk) load_slice(im);
            // part[rc][col]: one partial sum per 32-row chunk of the step and column of the block
            if (rc < nbk) {
                const double x0 = xs[32 * rc + 4 * sub], x1 = xs[32 * rc + 4 * sub + 1], x2 = xs[32 * rc + 4 * sub + 2],
                             x3 = xs[32 * rc + 4 * sub + 3];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    double sp = (a[4 * p] * x0 + a[4 * p + 1] * x1) + (a[4 * p + 2] * x2 + a[4 * p + 3] * x3);
                    sp += __shfl_xor(sp, 1);
                    sp += __shfl_xor(sp, 2);
                    sp += __shfl_xor(sp, 4);
                    if (sub == 0) part[rc][colq + 8 * p] = sp;
                }
            } else if (lane < P5_RB) {
                part[rc][lane] = 0.0;
            }
            __syncthreads();
            if (t < P5_RB) {
                const double v = run[m][t] - reduce_parts(t, P5_NBS);
                run[m][t] = v;
                if (m == m1 && diag_next) ps_publish(zfin + (int64_t)im * P5_RB + t, v);
            }
            if (m > 0) __syncthreads();
        }
        have_d = diag_next;
    }
}


// ---- 512 x 512 inverses from the 256 x 256 ones (see mnk_ls_build_inverses) --------------------------------------------
// element (row, col) of triangle t0 + blockIdx.x: the diagonal quadrants are copies, the others start as zeros
__global__ __launch_bounds__(512) void inv512_assemble_kernel(const double* __restrict__ Inv256, const double* __restrict__ InvT256,
                                                              double* __restrict__ Inv512, double* __restrict__ InvT512, int64_t Np,
                                                              int t0, const int* __restrict__ info) {
    if (*info != 0) return;
    const int64_t t = (int64_t)blockIdx.x + t0;
    const int col = blockIdx.y, row = threadIdx.x;
    const int64_t nblk256 = (Np + 255) / 256;
    double v = 0.0, vt = 0.0;
    const int qr = row >> 8, qc = col >> 8;
    if (qr == qc && 2 * t + qr < nblk256) {
        const int64_t off = (2 * t + qr) * 65536 + (row & 255) + (int64_t)(col & 255) * 256;
        v = Inv256[off];
        vt = InvT256[off];
    }
    Inv512[t * 262144 + row + (int64_t)col * 512] = v;
    InvT512[t * 262144 + row + (int64_t)col * 512] = vt;
}
// STAGE 1: Tt = A^-T C'; STAGE 2 (blockIdx.z = 0): Inv512[lower left] -= D^-1 Tt' ; (1): InvT512[upper right] -= Tt D^-T
template <int STAGE>
__global__ __launch_bounds__(256, 3) void inv512_gemm_kernel(const double* __restrict__ F, int64_t ld, const double* __restrict__ Inv256,
                                                             const double* __restrict__ InvT256, double* __restrict__ Tt,
                                                             double* __restrict__ Inv512, double* __restrict__ InvT512, int64_t Np,
                                                             int t0, const int* __restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    if (*info != 0) return;
    const int64_t t = (int64_t)blockIdx.x + t0;
    if ((2 * t + 1) * 256 >= Np) return;   // no second 256-block in this (last) triangle
    const int tm = blockIdx.y & 1, tn = blockIdx.y >> 1;
    const double* IA_t = InvT256 + (2 * t) * 65536;
    const double* ID = Inv256 + (2 * t + 1) * 65536;
    double* T = Tt + t * 65536;
    if (STAGE == 1) {
        const double* C = F + (512 * t + 256) + (512 * t) * ld;
        gemm_nt_tile<2, 2, 4, 1, false, 0, 8>(tm, tn, 256, 256, 256, IA_t, 256, C, ld, T, 256, nullptr, nullptr, 0, smem_raw);
    } else if (blockIdx.z == 0) {
        gemm_nt_tile<2, 2, 4, 0, false, 0, 8>(tm, tn, 256, 256, 256, ID, 256, T, 256, Inv512 + t * 262144 + 256, 512, nullptr, nullptr, 0, smem_raw);
    } else {
        gemm_nt_tile<2, 2, 4, 0, false, 0, 8>(tm, tn, 256, 256, 256, T, 256, ID, 256, InvT512 + t * 262144 + (int64_t)256 * 512, 512, nullptr, nullptr, 0, smem_raw);
    }
}

}  // namespace mnk

using namespace mnk;

// called at the end of the factorization (after linv64_kernel)
int mnk_ls_build_inverses(mnk_ls* ls, hipStream_t s, int64_t sc0, int64_t sc1) {
    const int64_t nblk = std::min<int64_t>(sc1, (ls->Np + SB - 1) / SB) - sc0;
    if (nblk <= 0) return 0;
    if (!ls->linv_mfma)
        hipLaunchKernelGGL(linv_tri_kernel<256>, dim3((unsigned)nblk, 4, 4), dim3(256), 0, s, ls->fact.p, ls->ld, ls->linv.p,
                           ls->linv256.p, ls->linv256t.p, ls->Np, ls->info_dev.p, (int)sc0);
    else
        hipLaunchKernelGGL(linv256_mfma_kernel, dim3((unsigned)nblk, 4), dim3(256), 0, s, ls->fact.p, ls->ld, ls->linv.p,
                           ls->linv256.p, ls->linv256t.p, ls->Np, ls->info_dev.p, (int)sc0);
    // 512-row triangles (the 32-row persistent solve): those that lie entirely inside [sc0, sc1) -- a caller that inverts in
    // two ranges cuts at an even strip-column.  Built from the 256-row inverses on the MFMA tile kernel:
    //   inv [A 0; C D] = [A^-1 0; -D^-1 C A^-1  D^-1],   T' = A^-T C' (stage 1),  X = -D^-1 T,  X' = -T' D^-T (stage 2)
    // (a 512 x 512 triangle inverted block by block with the scalar kernel above costs 35 dependent 64x64x16 products:
    // +0.63 ms on factorize! at C3, measured).
    if (ls->solve512 && ls->Np >= ls->solve512_min_rows && ls->Np % 512 != 384 && !ls->linv512.p) {
        const size_t ntri = (size_t)((ls->Np + 511) / 512);
        if (ls->linv512.alloc(ntri * 512 * 512) || ls->linv512t.alloc(ntri * 512 * 512) || ls->linv512tmp.alloc(ntri * 256 * 256)) {
            (void)hipGetLastError();
            ls->linv512.release(); ls->linv512t.release(); ls->linv512tmp.release();
            ls->solve512 = 0;
        }
    }
    if (ls->solve512 && ls->linv512.p) {
        const int64_t nsc = (ls->Np + SB - 1) / SB;
        const int64_t t0 = (sc0 + 1) / 2, t1 = std::min<int64_t>(sc1, nsc) == nsc ? (ls->Np + 511) / 512 : sc1 / 2;
        if (t1 > t0) {
            const unsigned nt = (unsigned)(t1 - t0);
            static std::atomic<uint64_t> attr_devs{0};
            int dev = 0;
            MNK_HIP(hipGetDevice(&dev));
            const int smem = 2 * 8 * ((128 + 16) + (128 + 16)) * (int)sizeof(double);
            if (!(attr_devs.load(std::memory_order_relaxed) >> (dev & 63) & 1)) {
                MNK_HIP(hipFuncSetAttribute((const void*)inv512_gemm_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
                MNK_HIP(hipFuncSetAttribute((const void*)inv512_gemm_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
                attr_devs.fetch_or(1ull << (dev & 63), std::memory_order_relaxed);
            }
            hipLaunchKernelGGL(inv512_assemble_kernel, dim3(nt, 512), dim3(512), 0, s, ls->linv256.p, ls->linv256t.p, ls->linv512.p,
                               ls->linv512t.p, ls->Np, (int)t0, ls->info_dev.p);
            hipLaunchKernelGGL(inv512_gemm_kernel<1>, dim3(nt, 4, 1), dim3(256), smem, s, ls->fact.p, ls->ld, ls->linv256.p, ls->linv256t.p,
                               ls->linv512tmp.p, ls->linv512.p, ls->linv512t.p, ls->Np, (int)t0, ls->info_dev.p);
            hipLaunchKernelGGL(inv512_gemm_kernel<2>, dim3(nt, 4, 2), dim3(256), smem, s, ls->fact.p, ls->ld, ls->linv256.p, ls->linv256t.p,
                               ls->linv512tmp.p, ls->linv512.p, ls->linv512t.p, ls->Np, (int)t0, ls->info_dev.p);
        }
    }
    MNK_HIP(hipGetLastError());
    return 0;
}

int mnk_ls_build_inverses_batch(hipStream_t s, const mnk::SmallSysRec* recs_dev, int n, int64_t Np) {
    const unsigned nblk = (unsigned)((Np + SB - 1) / SB);
    hipLaunchKernelGGL(linv256_mfma_kernel, dim3(nblk, 4, (unsigned)n), dim3(256), 0, s, (const double*)nullptr, (int64_t)0,
                       (const double*)nullptr, (double*)nullptr, (double*)nullptr, (int64_t)0, (const int*)nullptr, 0, recs_dev);
    MNK_HIP(hipGetLastError());
    return 0;
}

// First launch of linv256_mfma_kernel on a stream (it returns at once: `info` != 0).  The kernel spills a few registers,
// so the stream's hardware queue needs scratch memory for it, and the runtime's first allocation of that takes ~1.6 ms
// (seen in the kernel trace of the first factorize! behind a batch: two such gaps, +0.64 ms on a five-call average).
int mnk_solve_warmup(hipStream_t s) {
    static mnk::DevBuf<int> one;   // (process lifetime; holds 1)
    if (!one.p) {
        if (one.alloc(1)) return -2;
        const int v = 1;
        { mnk::H2DGuard h2d; MNK_HIP(hipMemcpy(one.p, &v, sizeof(int), hipMemcpyHostToDevice)); }
    }
    hipLaunchKernelGGL(linv256_mfma_kernel, dim3(1, 4), dim3(256), 0, s, (const double*)nullptr, (int64_t)0, (const double*)nullptr,
                       (double*)nullptr, (double*)nullptr, (int64_t)256, one.p, 0);
    MNK_HIP(hipGetLastError());
    return 0;
}

// xdev: the solver's work vector (10*Np doubles: x | y | two publication buffers of 4*Np).
// xuser == NULL: on entry xdev[0:Np] = rhs (zero padded); on exit xdev[0:Np] = solution.
// xuser != NULL: a device vector of N entries, rhs on entry, solution on exit; the one-launch solve reads and writes it
// directly (ONE launch per solve: no padded staging copy, no reset kernel, no copy back), every other path stages through xdev.
int mnk_ls_run_solve(mnk_ls* ls, double* xdev, double* xuser) {
    hipStream_t s = ls->ctx->stream;
    const int64_t Np = ls->Np, ld = ls->ld, N = ls->N;
    const int ldl = ls->algo == MNK_LDL;
    const int64_t nb64 = Np / 64;
    const int G = (int)std::min<int64_t>(nb64, ls->ctx->num_cu);
    // Bunch-Kaufman factor (bk.hip): P A P^T = L D L^T with 2x2 blocks in D -- gather, stepwise unit-lower sweeps,
    // block-diagonal D^-1, scatter (a 2x2 block may straddle two 64-row blocks, which the one-launch solve's
    // block ownership does not allow)
    const bool bk = ls->bk_active;
    const bool persistent = !bk && ls->persistent_solve && G >= 1 && (nb64 + G - 1) / G <= PS_MAXOWN && (G >= 4 || nb64 <= G);
    const int64_t nb32 = Np / P5_RB;
    const int G5 = (int)std::min<int64_t>(nb32, ls->ctx->num_cu);
    const bool use512 = persistent && ls->solve512 && ls->linv512.p && (nb32 + G5 - 1) / G5 <= PS_MAXOWN && (G5 >= P5_NBS || nb32 <= G5);
    const bool direct = persistent && !use512;
    auto stage_in = [&]() -> int {
        if (xuser != nullptr) hipLaunchKernelGGL(pad_copy_kernel, dim3((unsigned)((Np + 255) / 256)), dim3(256), 0, s, xdev, xuser, N, Np);
        MNK_HIP(hipGetLastError());
        return 0;
    };
    auto stage_out = [&]() -> int {
        if (xuser != nullptr) MNK_HIP(hipMemcpyAsync(xuser, xdev, N * sizeof(double), hipMemcpyDeviceToDevice, s));
        return 0;
    };
    if (!direct) {
        int rc = stage_in();
        if (rc) return rc;
    }
    if (bk) {
        // (the permutation's scratch is the first publication buffer of the one-launch solve: it no longer holds the sentinel --
        // a later solve on the static-pivot tier must reset it before it polls it)
        ls->pub_clean[0] = false;
        int rc = mnk_ls_bk_permute(ls, xdev, xdev + 2 * Np, true);
        if (rc) return rc;
    }
    if (persistent) {
        // The kernel needs all its workgroups resident at once (one per CU).  Two of them launched from different contexts
        // could each grab part of the chip and wait for the rest forever, and the same holds next to a persistent
        // factorization: the persistent operations of one process take turns on the device (common.h: mnk_persist_begin).
        int rc = mnk_persist_begin(ls->ctx, s);
        if (rc) return rc;
        const int ps_near = PS_NEAR, ps_nap = 6;
        double* pubs[2] = {xdev + 2 * Np, xdev + 6 * Np};
        auto ensure_clean = [&](int b) {
            if (!ls->pub_clean[b])
                hipLaunchKernelGGL(ps_reset_kernel, dim3((unsigned)((4 * Np + 255) / 256)), dim3(256), 0, s,
                                   reinterpret_cast<unsigned long long*>(pubs[b]), 4 * Np);
            ls->pub_clean[b] = true;
        };
        if (use512) {
            // steps of 512 columns, blocks of 32 rows (half the steps, the same two hops per step)
            ls->pub_clean[0] = false;
            ensure_clean(0);
            double* pub = pubs[0];
            if (ldl)
                hipLaunchKernelGGL(persistent_solve512_kernel<true>, dim3(G5), dim3(PS_NT), 0, s, ls->fact.p, ld, ls->linv512.p,
                                   ls->linv512t.p, ls->dinv.p, xdev, pub, Np, ls->solve_abort, ls->info_dev.p, ps_near, ps_nap,
                                   ls->ps_spin_limit, ls->debug_ps_missing);
            else
                hipLaunchKernelGGL(persistent_solve512_kernel<false>, dim3(G5), dim3(PS_NT), 0, s, ls->fact.p, ld, ls->linv512.p,
                                   ls->linv512t.p, ls->dinv.p, xdev, pub, Np, ls->solve_abort, ls->info_dev.p, ps_near, ps_nap,
                                   ls->ps_spin_limit, ls->debug_ps_missing);
            ls->pub_clean[0] = false;
            rc = hipGetLastError() == hipSuccess ? 0 : -2;
            rc = mnk_persist_end(ls->ctx, s, rc);
            return rc ? rc : stage_out();
        }
        // two publication buffers: this solve polls `cur` (all sentinel) and fills `oth` with the sentinel for the next one
        const int cur = ls->pub_next, oth = cur ^ 1;
        ensure_clean(cur);
        const double* xin = xuser != nullptr ? xuser : xdev;
        double* xout = xuser != nullptr ? xuser : xdev;
        const int64_t nio = xuser != nullptr ? N : Np;
        if (ldl)
            hipLaunchKernelGGL(persistent_solve_kernel<true>, dim3(G), dim3(PS_NT), 0, s, ls->fact.p, ld, ls->linv256.p,
                               ls->linv256t.p, ls->dinv.p, xin, xout, nio, pubs[cur], pubs[oth], Np, ls->solve_abort, ls->info_dev.p,
                               ls->solve_trace.p, ps_near, ps_nap, ls->ps_spin_limit, ls->debug_ps_missing);
        else
            hipLaunchKernelGGL(persistent_solve_kernel<false>, dim3(G), dim3(PS_NT), 0, s, ls->fact.p, ld,
                               ls->linv256.p, ls->linv256t.p, ls->dinv.p, xin, xout, nio, pubs[cur], pubs[oth], Np, ls->solve_abort,
                               ls->info_dev.p, ls->solve_trace.p, ps_near, ps_nap, ls->ps_spin_limit,
                               ls->debug_ps_missing);
        ls->pub_clean[cur] = false;
        // (a test's "missing" workgroup does not clear its blocks either; a solve that gives up marks both buffers dirty: mnk_ls_take_solve_abort)
        ls->pub_clean[oth] = ls->debug_ps_missing < 0;
        ls->pub_next = oth;
        rc = hipGetLastError() == hipSuccess ? 0 : -2;
        return mnk_persist_end(ls->ctx, s, rc);
    }
    double* b = xdev;       // forward: running right-hand side; backward: solution
    double* y = xdev + Np;  // forward: solution of L y = b (scaled by D^-1 for LDL); backward: running rhs
    const int64_t nsteps = (Np + SB - 1) / SB;
    for (int64_t k = 0; k < nsteps; ++k) {
        const int64_t j0 = k * SB;
        const int ncol = (int)std::min<int64_t>(SB, Np - j0);
        hipLaunchKernelGGL(diag256_kernel, dim3(1), dim3(1024), 0, s, ls->linv256.p + k * (int64_t)(SB * SB), b + j0,
                           y + j0, 0, ncol);
        const int64_t below = Np - j0 - ncol;
        if (below > 0)
            hipLaunchKernelGGL(fwd_panel_kernel, dim3((unsigned)((below + 63) / 64)), dim3(256), 0, s, ls->fact.p, ld,
                               y, b, j0, ncol, Np);
    }
    // L D L^T: z = D^-1 y before the backward sweep
    if (bk) {
        int rc = mnk_ls_bk_dsolve(ls, y);
        if (rc) return rc;
    } else if (ldl)
        hipLaunchKernelGGL(scale_vec_kernel, dim3((unsigned)((Np + 255) / 256)), dim3(256), 0, s, y, ls->dinv.p, Np);
    for (int64_t k = nsteps - 1; k >= 0; --k) {
        const int64_t j0 = k * SB;
        const int nrow = (int)std::min<int64_t>(SB, Np - j0);
        hipLaunchKernelGGL(diag256_kernel, dim3(1), dim3(1024), 0, s, ls->linv256t.p + k * (int64_t)(SB * SB), y + j0,
                           b + j0, 1, nrow);
        if (j0 > 0)
            hipLaunchKernelGGL(bwd_panel_kernel, dim3((unsigned)(j0 / 64)), dim3(256), 0, s, ls->fact.p, ld, b, y, j0,
                               nrow);
    }
    MNK_HIP(hipGetLastError());
    if (bk) {
        ls->pub_clean[0] = false;
        int rc = mnk_ls_bk_permute(ls, xdev, xdev + 2 * Np, false);
        if (rc) return rc;
    }
    return stage_out();
}

// ---------------------------------------------------------------------------------------------------------------------
// Batches of independent solves (mnk_solve_batch_begin / _end): between the two calls the single-right-hand-side solves of
// the calling thread on DEVICE vectors are queued; `end` runs them up to four systems per launch
// (persistent_solve_multi_kernel).  Systems of one launch belong to different solvers (a solver's second right-hand side goes
// into the next launch: its publication buffers are in use); solves that do not qualify (host vectors, several right-hand
// sides, the pivoted tier, the stepwise solve) run at once as usual.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
struct SolveReq { mnk_ls* ls; double* x; };
struct SolveBatchState { int depth = 0; bool active = false; std::vector<SolveReq> pend; };
thread_local SolveBatchState t_sbatch;
struct SolveBatchBuffers { mnk::DevBuf<char> sys; hipEvent_t ev = nullptr; };
SolveBatchBuffers g_sbatch[64];
constexpr int PS_BATCH_MAXQ = 32;
}  // namespace

static bool solve_eligible(const mnk_ls* ls) {
    const int64_t nb64 = ls->Np / 64;
    const int G = (int)std::min<int64_t>(nb64, ls->ctx->num_cu);
    return !ls->bk_active && ls->persistent_solve && !(ls->solve512 && ls->linv512.p) && !ls->ctx->partitioned &&
           (nb64 + G - 1) / G <= PS_MAXOWN && (G >= 4 || nb64 <= G) && !(ls->solve_abort && *ls->solve_abort != 0) &&
           ls->debug_ps_missing < 0 && !ls->solve_trace.p;
}

bool mnk_solve_defer(mnk_ls* ls, double* xuser) {
    if (!t_sbatch.active || !solve_eligible(ls)) return false;
    if (!ls->ev_defer && hipEventCreateWithFlags(&ls->ev_defer, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (hipEventRecord(ls->ev_defer, ls->ctx->stream) != hipSuccess) { (void)hipGetLastError(); return false; }
    t_sbatch.pend.push_back({ls, xuser});
    return true;
}

// one launch: systems v[0..q) (distinct solvers of one device, order and algorithm)
static int solve_batch_launch(const std::vector<SolveReq>& v) {
    const int q = (int)v.size();
    mnk_ls* l0 = v[0].ls;
    mnk_ctx* c0 = l0->ctx;
    hipStream_t h = c0->stream;
    MNK_HIP(hipSetDevice(c0->device));
    const int64_t Np = l0->Np, nb64 = Np / 64;
    const int G = (int)std::min<int64_t>(nb64, c0->num_cu / q);
    SolveBatchBuffers& B = g_sbatch[c0->device & 63];
    for (const SolveReq& r : v)
        if (r.ls->ctx->stream != h) MNK_HIP(hipStreamWaitEvent(h, r.ls->ev_defer, 0));
    int rc = mnk_persist_begin(c0, h);
    if (rc) return rc;
    auto body = [&]() -> int {
        if (B.sys.n < sizeof(PsSys) * PS_BATCH_MAXQ && B.sys.alloc(sizeof(PsSys) * PS_BATCH_MAXQ)) return -2;
        if (!B.ev) MNK_HIP(hipEventCreateWithFlags(&B.ev, hipEventDisableTiming));
        PsSys* dsys = reinterpret_cast<PsSys*>(B.sys.p);
        for (int i = 0; i < q; ++i) {
            mnk_ls* ls = v[i].ls;
            double* pubs[2] = {ls->xwork.p + 2 * Np, ls->xwork.p + 6 * Np};
            const int cur = ls->pub_next, oth = cur ^ 1;
            if (!ls->pub_clean[cur])
                hipLaunchKernelGGL(ps_reset_kernel, dim3((unsigned)((4 * Np + 255) / 256)), dim3(256), 0, h,
                                   reinterpret_cast<unsigned long long*>(pubs[cur]), 4 * Np);
            PsSys rec{ls->fact.p, ls->ld, ls->linv256.p, ls->linv256t.p, ls->dinv.p, v[i].x, v[i].x, ls->N, pubs[cur], pubs[oth],
                      ls->solve_abort, ls->info_dev.p};
            hipLaunchKernelGGL(ps_set_sys_kernel, dim3(1), dim3(1), 0, h, rec, dsys + i);
            ls->pub_clean[cur] = false;
            ls->pub_clean[oth] = true;
            ls->pub_next = oth;
        }
        const bool ldl = l0->algo == MNK_LDL;
        if (ldl)
            hipLaunchKernelGGL(persistent_solve_multi_kernel<true>, dim3((unsigned)(q * G)), dim3(PS_NT), 0, h, dsys, G, Np, PS_NEAR, 6, l0->ps_spin_limit);
        else
            hipLaunchKernelGGL(persistent_solve_multi_kernel<false>, dim3((unsigned)(q * G)), dim3(PS_NT), 0, h, dsys, G, Np, PS_NEAR, 6, l0->ps_spin_limit);
        MNK_HIP(hipGetLastError());
        return 0;
    };
    rc = body();
    rc = mnk_persist_end(c0, h, rc);
    if (rc) return rc;
    bool other = false;
    for (const SolveReq& r : v) other = other || r.ls->ctx->stream != h;
    if (other) {
        MNK_HIP(hipEventRecord(B.ev, h));
        for (const SolveReq& r : v)
            if (r.ls->ctx->stream != h) MNK_HIP(hipStreamWaitEvent(r.ls->ctx->stream, B.ev, 0));
    }
    return 0;
}

static int solve_batch_flush() {
    std::vector<SolveReq> pend;
    pend.swap(t_sbatch.pend);
    // systems per launch: four for large systems (measured at N = 11 192: 2 -> 104.0, 4 -> 107.4, 8 -> 106.2 it/s of the C5
    // step), as many as the CUs hold for small ones (a 512-row block has 8 blocks of 64 rows: 32 of them per launch)
    static const int envq = []() { const char* e = getenv("MNK_SOLVE_BATCH_Q"); return e ? std::max(1, std::min(PS_BATCH_MAXQ, atoi(e))) : 0; }();
    int rc_all = 0;
    while (!pend.empty()) {
        mnk_ls* l0 = pend.front().ls;
        const int64_t nb64 = l0->Np / 64;
        const int g4 = (int)std::max<int64_t>(1, std::min<int64_t>(nb64, l0->ctx->num_cu / 4));   // workgroups of a system when four share the chip
        const int maxq = envq > 0 ? envq : std::max(1, std::min(PS_BATCH_MAXQ, l0->ctx->num_cu / g4));
        std::vector<SolveReq> g, rest;
        for (const SolveReq& r : pend) {
            bool take = (int)g.size() < maxq && r.ls->ctx->device == l0->ctx->device && r.ls->Np == l0->Np && r.ls->algo == l0->algo;
            for (const SolveReq& t : g) take = take && t.ls != r.ls;      // (one right-hand side per solver and launch)
            for (const SolveReq& t : rest) take = take && t.ls != r.ls;   // (and a solver's right-hand sides keep their order)
            if (take) {   // all workgroups of the launch resident, at most PS_MAXOWN blocks each
                const int Gq = (int)std::min<int64_t>(nb64, l0->ctx->num_cu / ((int)g.size() + 1));
                take = (Gq >= 4 || nb64 <= Gq) && (nb64 + Gq - 1) / Gq <= PS_MAXOWN;
                take = take || g.empty();   // (the first one always goes: alone it is an ordinary solve)
            }
            (take ? g : rest).push_back(r);
        }
        pend.swap(rest);
        int rc;
        if (g.size() >= 2) rc = solve_batch_launch(g);
        else rc = mnk_ls_run_solve(g[0].ls, g[0].ls->xwork.p, g[0].x);
        if (rc && !rc_all) rc_all = rc;
    }
    return rc_all;
}

extern "C" int mnk_solve_batch_begin(void) {
    ++t_sbatch.depth;
    t_sbatch.active = true;
    return 0;
}

extern "C" int mnk_solve_batch_end(void) {
    if (t_sbatch.depth <= 0) { mnk::set_error("mnk_solve_batch_end: no batch is open on this thread"); return -1; }
    if (--t_sbatch.depth > 0) return 0;
    t_sbatch.active = false;
    return solve_batch_flush();
}

// Library-internal consumers of a solve batch (schur.hip) read the solutions right behind their own begin / end pair.  When
// the CALLER already has a batch open on this thread the inner end is a no-op (pairs nest, the outermost end launches) and
// the solves would still be queued: this runs whatever the thread has queued, now, and leaves the caller's batch open.
int mnk_solve_batch_flush_pending(void) { return t_sbatch.pend.empty() ? 0 : solve_batch_flush(); }

// (a solver that is destroyed, factorized again or asked for a non-batchable solve while one of its solves is queued)
int mnk_solve_sync_deferred(mnk_ls* ls) {
    for (const SolveReq& r : t_sbatch.pend)
        if (r.ls == ls) return solve_batch_flush();
    return 0;
}
