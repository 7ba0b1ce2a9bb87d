// Bunch-Kaufman LDL^T with 1x1 / 2x2 pivots and GLOBAL partner search on the device: the robust tier
// behind `factorize_bunchkaufman!` = dsytrf('L') (reference src/LinearSolvers/lapack.jl:164-167) and its
// inertia rule `num_neg_ev` (reference src/LinearSolvers/lapack.jl:240-268).
//
// Two tiers.  BUNCHKAUFMAN first runs the static-pivot blocked LDL^T of factor.hip (fp64 MFMA, the fast path:
// every KKT system that is quasi-definite in the given order -- all the condensed systems, the regularized
// augmented ones -- factors there, and by Sylvester's law its sign(D) inertia is the one dsytrf reports).
// When that factorization BREAKS DOWN (an exact zero / non-finite pivot in the given order, e.g. a zero
// diagonal block that needs a 2x2 pivot), the matrix is transferred again and factored here with the pivoting
// strategy of LAPACK's dsytf2 (alpha = (1 + sqrt(17)) / 8, partner = row of the largest off-diagonal entry of
// the pivot column, 1x1 or 2x2 pivot by the usual four tests), so the inertia equals the reference's on
// systems that are NOT quasi-definite in the given order instead of detouring through delta_c.
//
// This tier is BLOCKED (LAPACK dlasyf's scheme, lower variant): panels of NB = 64 columns are factored left-looking by ONE
// workgroup -- every column is brought up to date with the panel's previous columns just before its pivot search, the
// partner column likewise when the 1x1 test fails, so the search sees exactly the entries dsytf2 would -- and the trailing
// matrix receives the whole panel in one fp64-MFMA product A22 -= L21 (L21 D)^T (the tile k-loop of gemm_tile.h).  Two
// launches per panel, the panel's position and width (63 or 64 columns: a 2x2 pivot never straddles a panel end) live in
// device memory, so the host enqueues the whole factorization without a synchronization.  Unlike dsytf2 / dlasyf the
// interchanges are applied to the previous columns as well, so the result is a plain P A P^T = L D L^T with ONE
// permutation vector; the solve is gather, unit-lower sweeps (the same stepwise kernels as the static factor),
// block-diagonal D^-1, scatter.  (Round 2's tier was unblocked: four launches and an HBM-bound rank-1/2 update of the
// whole trailing triangle per pivot -- seconds at N ~ 1e4.)
#include <cfloat>
#include <cmath>

#include <atomic>

#include "gemm_tile.h"
#include "ls.h"

namespace mnk {

struct BkState {
    int k;      // first column that is not factored yet
    int p0;     // first column of the panel just factored
    int kb;     // its width (the trailing update applies columns p0 .. p0 + kb - 1)
    int info;   // LAPACK-style: 1-based index of the first exactly-zero pivot (0: none)
};

constexpr double BK_ALPHA = 0.6403882032022076;  // (1 + sqrt(17)) / 8
constexpr int BK_NB = 64;    // panel width
constexpr int BK_NBW = 72;   // columns of the panel work space (NB + 1 working column, padded to the k-tile depth 8)
constexpr int BK_T = 1024;   // threads of the panel workgroup

__device__ __forceinline__ void block_argmax(double v, int idx, double* sval, int* sidx, double& outv, int& outi) {
    // largest |value|, smallest index among ties (idamax)
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_down(v, off);
        const int oi = __shfl_down(idx, off);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { sval[w] = v; sidx[w] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double bv = sval[0];
        int bi = sidx[0];
        for (int q = 1; q < (int)(blockDim.x >> 6); ++q)
            if (sval[q] > bv || (sval[q] == bv && sidx[q] < bi)) { bv = sval[q]; bi = sidx[q]; }
        sval[0] = bv;
        sidx[0] = bi;
    }
    __syncthreads();
    outv = sval[0];
    outi = sidx[0];
    __syncthreads();
}

// One panel, one workgroup.  W[:, c] = the panel's c-th column brought up to date = (L D)[:, c] once it is eliminated (what
// the trailing update multiplies with), LW[:, c] = L[:, p0 + c] (a zero-padded copy: the trailing update reads K = 72
// columns); both Np x BK_NBW, leading dimension ldw.  Pivoting: dsytf2 / dlasyf, column by column:
//   absakk = |a_kk|, colmax = max_{i>k} |a_ik| (row imax) on the UPDATED column;  zero column -> info, no elimination;
//   absakk >= alpha colmax -> 1x1, no interchange;  else with rowmax = largest off-diagonal of the UPDATED row/column imax:
//   absakk >= alpha colmax (colmax / rowmax) -> 1x1, no interchange;  |a_imax,imax| >= alpha rowmax -> 1x1, interchange k <->
//   imax;  else 2x2 pivot {k, imax}.
__global__ __launch_bounds__(BK_T) void bkp_panel_kernel(double* __restrict__ F, int64_t ld, int Np, BkState* st,
                                                         double* __restrict__ W, double* __restrict__ LW, int64_t ldw,
                                                         double* __restrict__ dvec, double* __restrict__ doff,
                                                         int* __restrict__ ptype, int* __restrict__ perm) {
    __shared__ double sval[16];
    __shared__ int sidx[16];
    __shared__ double s_wk[BK_NBW], s_wi[BK_NBW];
    const int t = threadIdx.x;
    const int p0 = st->k;
    __syncthreads();  // (everyone has read st->k before thread 0 rewrites the state at the end)
    if (p0 >= Np) {
        if (t == 0) { st->p0 = p0; st->kb = 0; }
        return;
    }
    for (int c = 0; c < BK_NBW; ++c)
        for (int i = p0 + t; i < Np; i += BK_T) { W[i + c * ldw] = 0.0; LW[i + c * ldw] = 0.0; }
    __syncthreads();
    int kb = 0;
    while (kb < BK_NB - 1 && p0 + kb < Np) {
        const int k = p0 + kb;
        double* Wk = W + (int64_t)kb * ldw;        // working column: the pivot column
        double* Wn = W + (int64_t)(kb + 1) * ldw;  // second working column: the partner
        // ---- 1. column k, brought up to date with the panel's columns
        if (t < kb) s_wk[t] = W[k + (int64_t)t * ldw];
        __syncthreads();
        for (int i = k + t; i < Np; i += BK_T) {
            double acc = F[i + (int64_t)k * ld];
            for (int c = 0; c < kb; ++c) acc -= LW[i + (int64_t)c * ldw] * s_wk[c];
            Wk[i] = acc;
        }
        __syncthreads();
        // ---- 2. pivot search
        const double absakk = fabs(Wk[k]);
        double v = -1.0;
        int vi = 0x7fffffff;
        for (int i = k + 1 + t; i < Np; i += BK_T) {
            const double a = fabs(Wk[i]);
            if (a > v || !(a <= DBL_MAX)) { v = !(a <= DBL_MAX) ? DBL_MAX : a; vi = i; }
        }
        double colmax;
        int imax;
        block_argmax(v, vi, sval, sidx, colmax, imax);
        if (colmax < 0.0) colmax = 0.0;  // k is the last column
        int kstep = 1, kp = k;
        bool zero = false;
        if (!(fmax(absakk, colmax) > 0.0) || !(absakk <= DBL_MAX) || colmax >= DBL_MAX) {
            zero = true;  // the column is exactly zero (or not finite): no elimination, info reports it
        } else if (absakk < BK_ALPHA * colmax) {
            // row / column imax of the trailing matrix, brought up to date
            if (t < kb) s_wi[t] = W[imax + (int64_t)t * ldw];
            __syncthreads();
            for (int i = k + t; i < Np; i += BK_T) {
                double acc = i < imax ? F[imax + (int64_t)i * ld] : F[i + (int64_t)imax * ld];
                for (int c = 0; c < kb; ++c) acc -= LW[i + (int64_t)c * ldw] * s_wi[c];
                Wn[i] = acc;
            }
            __syncthreads();
            double rv = -1.0;
            int ri = 0x7fffffff;
            for (int i = k + t; i < Np; i += BK_T) {
                if (i == imax) continue;
                const double a = fabs(Wn[i]);
                if (a > rv) { rv = a; ri = i; }
            }
            double rowmax;
            int jmax;
            block_argmax(rv, ri, sval, sidx, rowmax, jmax);
            if (absakk >= BK_ALPHA * colmax * (colmax / rowmax)) {
                kp = k;
            } else if (fabs(Wn[imax]) >= BK_ALPHA * rowmax) {
                kp = imax;  // 1x1 pivot on a_imax,imax: its column becomes the pivot column
                for (int i = k + t; i < Np; i += BK_T) Wk[i] = Wn[i];
                __syncthreads();
            } else {
                kp = imax;
                kstep = 2;
            }
        }
        const int kk = k + kstep - 1;
        // ---- 3. symmetric interchange kk <-> kp (kp > kk): the part of A that is not factored yet, the same rows of every
        // previous column (one permutation for the whole factorization), of the panel copies and of the working columns
        if (kp != kk) {
            auto swp = [&](double* a, double* b) { const double x = *a; *a = *b; *b = x; };
            for (int j = t; j < Np; j += BK_T) {
                if (j < k) {
                    swp(F + kk + (int64_t)j * ld, F + kp + (int64_t)j * ld);
                } else if (j > kp) {
                    swp(F + j + (int64_t)kk * ld, F + j + (int64_t)kp * ld);
                } else if (j > kk && j < kp) {
                    swp(F + j + (int64_t)kk * ld, F + kp + (int64_t)j * ld);
                } else if (j == kk) {
                    swp(F + kk + (int64_t)kk * ld, F + kp + (int64_t)kp * ld);
                    if (kstep == 2) swp(F + (k + 1) + (int64_t)k * ld, F + kp + (int64_t)k * ld);
                    const int p = perm[kk]; perm[kk] = perm[kp]; perm[kp] = p;
                }
            }
            if (t < kb + kstep) swp(W + kk + (int64_t)t * ldw, W + kp + (int64_t)t * ldw);
            if (t >= 64 && t - 64 < kb) swp(LW + kk + (int64_t)(t - 64) * ldw, LW + kp + (int64_t)(t - 64) * ldw);
            __syncthreads();
        }
        // ---- 4. eliminate
        if (zero) {
            if (t == 0) {
                if (st->info == 0) st->info = k + 1;
                dvec[k] = 0.0; doff[k] = 0.0; ptype[k] = 1;
            }
            for (int i = k + 1 + t; i < Np; i += BK_T) { F[i + (int64_t)k * ld] = 0.0; Wk[i] = 0.0; }
        } else if (kstep == 1) {
            const double d = Wk[k];
            const double r = 1.0 / d;
            __syncthreads();  // (everyone has read the pivot before the column is rewritten)
            for (int i = k + 1 + t; i < Np; i += BK_T) {
                const double l = Wk[i] * r;
                F[i + (int64_t)k * ld] = l;
                LW[i + (int64_t)kb * ldw] = l;
            }
            if (t == 0) { F[k + (int64_t)k * ld] = d; dvec[k] = d; doff[k] = 0.0; ptype[k] = 1; }
        } else {
            // [l1 l2] = [w1 w2] inv([[p11 p21], [p21 p22]]), scaled as dsytf2 does (no overflow from a tiny p21)
            const double p11 = Wk[k], p21 = Wk[k + 1], p22 = Wn[k + 1];
            const double d11 = p22 / p21, d22 = p11 / p21;
            const double tt = 1.0 / (d11 * d22 - 1.0) / p21;
            __syncthreads();
            for (int i = k + 2 + t; i < Np; i += BK_T) {
                const double a1 = Wk[i], a2 = Wn[i];
                const double l1 = tt * (d11 * a1 - a2), l2 = tt * (d22 * a2 - a1);
                F[i + (int64_t)k * ld] = l1;
                F[i + (int64_t)(k + 1) * ld] = l2;
                LW[i + (int64_t)kb * ldw] = l1;
                LW[i + (int64_t)(kb + 1) * ldw] = l2;
            }
            if (t == 0) {
                dvec[k] = p11; dvec[k + 1] = p22; doff[k] = p21; doff[k + 1] = 0.0;
                ptype[k] = 2; ptype[k + 1] = 3;
                F[k + (int64_t)k * ld] = p11; F[(k + 1) + (int64_t)(k + 1) * ld] = p22;
                F[(k + 1) + (int64_t)k * ld] = 0.0;  // L is unit lower: the 2x2 block's off-diagonal lives in doff
            }
        }
        kb += kstep;
        __syncthreads();
    }
    // the working column behind the panel is not part of it: the trailing update must not see it
    for (int c = kb; c < BK_NBW; ++c)
        for (int i = p0 + t; i < Np; i += BK_T) { W[i + c * ldw] = 0.0; LW[i + c * ldw] = 0.0; }
    if (t == 0) { st->p0 = p0; st->kb = kb; st->k = p0 + kb; }
}

// Trailing update of one panel on the matrix cores: A[i, j] -= sum_c LW[i, c] W[j, c] for i >= j >= p0 + kb, 128 x 128
// tiles in absolute tile coordinates (tile (tm, tn) = rows 128 tm.., columns 128 tn..), K = BK_NBW zero-padded columns.
// The panel's position comes from device memory; the host launches the tiles of the largest region it can be.
__global__ __launch_bounds__(256, 3) void bkp_update_kernel(double* __restrict__ F, int64_t ld, int Np, const BkState* st,
                                                            const double* __restrict__ W, const double* __restrict__ LW,
                                                            int64_t ldw, int tile0) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int kb = st->kb;
    if (kb <= 0) return;
    const int r0 = st->p0 + kb;
    // lower tiles of the square of `nt` tile rows that starts at tile0, enumerated row by row
    const int b = blockIdx.x;
    int tm = (int)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
    while (tm * (tm + 1) / 2 > b) --tm;
    while ((tm + 1) * (tm + 2) / 2 <= b) ++tm;
    const int tn = tile0 + (b - tm * (tm + 1) / 2);
    tm += tile0;
    if (128 * (tm + 1) <= r0 || 128 * (tn + 1) <= r0 || 128 * tm >= Np) return;  // nothing of this tile is in the trailing matrix
    int tid = threadIdx.x;
    v4f64 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = v4f64{0.0, 0.0, 0.0, 0.0};
    (void)gemm_nt_mainloop<2, 2, 4, 0, 8>(acc, LW + (int64_t)128 * tm, ldw, W + (int64_t)128 * tn, ldw, BK_NBW / 8, smem_raw, tid);
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1, l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int col = 128 * tn + 64 * wn + 16 * ni + l4 + 4 * r;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int row = 128 * tm + 64 * wm + 16 * mi + l15;
                if (col >= r0 && row >= col && row < Np) F[row + (int64_t)col * ld] -= acc[ni][mi][r];
            }
        }
}

__global__ void bkp_init_kernel(BkState* st, int* __restrict__ perm, int Np) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < Np) perm[t] = t;
    if (t == 0) { st->k = 0; st->p0 = 0; st->kb = 0; st->info = 0; }
}

// dinv / dcoup of the block-diagonal D^-1, and the factored diagonal 64x64 blocks for linv64_kernel
__global__ void bk_finish_kernel(const double* __restrict__ F, int64_t ld, int Np, const double* __restrict__ dvec,
                                 const double* __restrict__ doff, const int* __restrict__ ptype,
                                 double* __restrict__ dinv, double* __restrict__ dcoup, double* __restrict__ dblk) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < Np) {
        const int pt = ptype[t];
        if (pt == 1) {
            dinv[t] = dvec[t] != 0.0 ? 1.0 / dvec[t] : 0.0;
            dcoup[t] = 0.0;
        } else {
            const int f = pt == 2 ? t : t - 1;  // first index of the pair
            const double p11 = dvec[f], p22 = dvec[f + 1], p21 = doff[f];
            const double d11 = p22 / p21, d22 = p11 / p21;
            const double tt = 1.0 / (d11 * d22 - 1.0) / p21;
            // inv = tt * [[d11, -1], [-1, d22]]
            dinv[t] = pt == 2 ? tt * d11 : tt * d22;
            dcoup[t] = -tt;
        }
    }
    // diagonal blocks (column-major 64x64, lower part; the diagonal entry is irrelevant for the unit-lower inverse)
    for (int64_t e = t; e < (int64_t)(Np / 64) * 4096; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = e >> 12;
        const int r = (int)(e & 63), c = (int)((e >> 6) & 63);
        dblk[e] = r > c ? F[(b * 64 + r) + (b * 64 + c) * ld] : (r == c ? 1.0 : 0.0);
    }
}

// reference `num_neg_ev` (src/LinearSolvers/lapack.jl:247-268) over the first N pivots: out[0] = #negative,
// out[1] = #(d == 0) hits
__global__ void bk_inertia_kernel(const double* __restrict__ dvec, const double* __restrict__ doff,
                                  const int* __restrict__ ptype, int64_t N, unsigned long long* out) {
    unsigned long long neg = 0, zer = 0;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < N; k += (int64_t)gridDim.x * blockDim.x) {
        const int pt = ptype[k];
        double d;
        if (pt == 1) d = dvec[k];
        else if (pt == 2) { const double tt = fabs(doff[k]); d = (dvec[k] / tt) * dvec[k + 1] - tt; }
        else d = 1.0;  // second index of a pair: the reference counts it as positive (d = t > 0)
        if (d < 0.0) ++neg;
        if (d == 0.0) ++zer;
    }
    for (int off = 32; off > 0; off >>= 1) { neg += __shfl_down(neg, off); zer += __shfl_down(zer, off); }
    if ((threadIdx.x & 63) == 0) {
        if (neg) atomicAdd(&out[0], neg);
        if (zer) atomicAdd(&out[1], zer);
    }
}

__global__ void bk_gather_kernel(double* __restrict__ dst, const double* __restrict__ src, const int* __restrict__ perm, int Np) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < Np) dst[t] = src[perm[t]];
}
__global__ void bk_scatter_kernel(double* __restrict__ dst, const double* __restrict__ src, const int* __restrict__ perm, int Np) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < Np) dst[perm[t]] = src[t];
}
// y <- D^-1 y with 1x1 / 2x2 blocks, in place (the first index of a pair writes both entries)
__global__ void bk_dsolve_kernel(double* __restrict__ y, const double* __restrict__ dinv, const double* __restrict__ dcoup,
                                 const int* __restrict__ ptype, int Np) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Np) return;
    const int pt = ptype[t];
    if (pt == 1) y[t] *= dinv[t];
    else if (pt == 2) {
        const double y1 = y[t], y2 = y[t + 1];
        y[t] = dinv[t] * y1 + dcoup[t] * y2;
        y[t + 1] = dcoup[t + 1] * y1 + dinv[t + 1] * y2;
    }
}

}  // namespace mnk

using namespace mnk;

// Factor the matrix currently in ls->fact (lower triangle, padded with a unit diagonal) by Bunch-Kaufman.
int mnk_ls_run_bunchkaufman(mnk_ls* ls) {
    hipStream_t s = ls->ctx->stream;
    const int Np = (int)ls->Np;
    const int64_t ld = ls->ld;
    double* F = ls->fact.p;
    int rc = 0;
    if (!ls->bk_perm.p) {
        rc |= ls->bk_perm.alloc(Np);
        rc |= ls->bk_ptype.alloc(Np);
        rc |= ls->bk_doff.alloc(Np);
        rc |= ls->bk_dcoup.alloc(Np);
        rc |= ls->bk_state.alloc(sizeof(BkState));
        rc |= ls->bk_work.alloc((size_t)2 * Np * BK_NBW + SLACK);
        if (rc) return -2;
    }
    BkState* st = reinterpret_cast<BkState*>(ls->bk_state.p);
    double* W = ls->bk_work.p;
    double* LW = W + (size_t)Np * BK_NBW;
    const size_t smem = 2 * 8 * ((128 + 16) + (128 + 16)) * sizeof(double);
    {
        static std::atomic<uint64_t> attr_devs{0};
        int dev = 0;
        MNK_HIP(hipGetDevice(&dev));
        if (!(attr_devs.load(std::memory_order_relaxed) >> (dev & 63) & 1)) {
            MNK_HIP(hipFuncSetAttribute((const void*)bkp_update_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            attr_devs.fetch_or(1ull << (dev & 63), std::memory_order_relaxed);
        }
    }
    MNK_HIP(hipMemsetAsync(ls->info_dev.p, 0, sizeof(int), s));
    hipLaunchKernelGGL(bkp_init_kernel, dim3((Np + 255) / 256), dim3(256), 0, s, st, ls->bk_perm.p, Np);
    // a panel takes 63 or 64 columns: the p-th panel starts at column >= 63 p, so its update touches no tile row above
    // floor(63 (p + 1) / 128); one more (empty) round covers the case that every panel was 63 wide
    const int npanel = (Np + BK_NB - 2) / (BK_NB - 1) + 1;
    for (int p = 0; p < npanel; ++p) {
        if ((int64_t)(BK_NB - 1) * p >= Np) break;
        hipLaunchKernelGGL(bkp_panel_kernel, dim3(1), dim3(BK_T), 0, s, F, ld, Np, st, W, LW, (int64_t)Np, ls->dvec.p, ls->bk_doff.p,
                           ls->bk_ptype.p, ls->bk_perm.p);
        const int tile0 = (int)(((int64_t)(BK_NB - 1) * (p + 1)) / 128);
        const int nt = Np / 128 - tile0;
        if (nt > 0)
            hipLaunchKernelGGL(bkp_update_kernel, dim3((unsigned)(nt * (nt + 1) / 2)), dim3(256), smem, s, F, ld, Np, st, W, LW,
                               (int64_t)Np, tile0);
    }
    hipLaunchKernelGGL(bk_finish_kernel, dim3((Np + 255) / 256), dim3(256), 0, s, F, ld, Np, ls->dvec.p, ls->bk_doff.p,
                       ls->bk_ptype.p, ls->dinv.p, ls->bk_dcoup.p, ls->dblk.p);
    MNK_HIP(hipGetLastError());
    // LAPACK-style info -> the solver's info word
    MNK_HIP(hipMemcpyAsync(ls->info_dev.p, &st->info, sizeof(int), hipMemcpyDeviceToDevice, s));
    ls->bk_active = true;
    return 0;
}

int mnk_ls_bk_permute(mnk_ls* ls, double* x, double* tmp, bool forward) {
    hipStream_t s = ls->ctx->stream;
    const int Np = (int)ls->Np;
    if (forward) hipLaunchKernelGGL(bk_gather_kernel, dim3((Np + 255) / 256), dim3(256), 0, s, tmp, x, ls->bk_perm.p, Np);
    else hipLaunchKernelGGL(bk_scatter_kernel, dim3((Np + 255) / 256), dim3(256), 0, s, tmp, x, ls->bk_perm.p, Np);
    MNK_HIP(hipMemcpyAsync(x, tmp, (size_t)Np * sizeof(double), hipMemcpyDeviceToDevice, s));
    return 0;
}

int mnk_ls_bk_dsolve(mnk_ls* ls, double* y) {
    const int Np = (int)ls->Np;
    hipLaunchKernelGGL(bk_dsolve_kernel, dim3((Np + 255) / 256), dim3(256), 0, ls->ctx->stream, y, ls->dinv.p,
                       ls->bk_dcoup.p, ls->bk_ptype.p, Np);
    MNK_HIP(hipGetLastError());
    return 0;
}

int mnk_ls_bk_inertia(mnk_ls* ls, unsigned long long* out_dev) {
    const int blocks = (int)std::min<int64_t>(256, (ls->N + 255) / 256);
    hipLaunchKernelGGL(bk_inertia_kernel, dim3(blocks), dim3(256), 0, ls->ctx->stream, ls->dvec.p, ls->bk_doff.p,
                       ls->bk_ptype.p, ls->N, out_dev);
    MNK_HIP(hipGetLastError());
    return 0;
}
