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
// This tier is unblocked and right-looking: three small launches per pivot step (decide / interchange /
// rank-1-or-2 update of the trailing triangle, HBM-bound, 8 N^3 / 3 bytes in total) -- seconds at N ~ 1e4,
// milliseconds at N ~ 1e3.  It is a fallback that a well-posed IPM iteration never takes, not a fast path.
// Unlike dsytf2 the interchanges are applied to the previous columns as well, so the result is a plain
// P A P^T = L D L^T with ONE permutation vector; the solve is gather, unit-lower sweeps (the same stepwise
// kernels as the static factor), block-diagonal D^-1, scatter.
#include <cfloat>
#include <cmath>

#include "ls.h"

namespace mnk {

struct BkState {
    int k;        // first column of the current pivot
    int kstep;    // 0: finished, 1 / 2: size of the current pivot block
    int kp;       // row/column interchanged with k + kstep - 1
    int info;     // LAPACK-style: 1-based index of the first exactly-zero pivot (0: none)
    int pk;       // pivot whose column(s) still have to be scaled (-1: none)
    int pkstep;
    double p11, p21, p22;  // that pivot block (unscaled)
};

constexpr double BK_ALPHA = 0.6403882032022076;  // (1 + sqrt(17)) / 8

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

// One workgroup: finish the previous pivot (L = A[:, pivot columns] * inv(D block), record D), then choose the
// next pivot exactly as dsytf2 does.
__global__ __launch_bounds__(1024) void bk_decide_kernel(double* __restrict__ F, int64_t ld, int Np, BkState* st,
                                                         double* __restrict__ dvec, double* __restrict__ doff,
                                                         int* __restrict__ ptype) {
    __shared__ double sval[16];
    __shared__ int sidx[16];
    const int t = threadIdx.x;
    // ---- finish the previous step
    const int pk = st->pk;
    if (pk >= 0) {
        if (st->pkstep == 1) {
            const double d = st->p11;
            const double r = d != 0.0 ? 1.0 / d : 0.0;
            for (int i = pk + 1 + t; i < Np; i += blockDim.x) F[i + (int64_t)pk * ld] *= r;
            if (t == 0) { dvec[pk] = d; doff[pk] = 0.0; ptype[pk] = 1; }
        } else {
            // [l1 l2] = [a1 a2] inv([[p11 p21],[p21 p22]]), scaled as dsytf2 does (no overflow from tiny p21)
            const double d21 = st->p21;
            const double d11 = st->p22 / d21, d22 = st->p11 / d21;
            const double tt = 1.0 / (d11 * d22 - 1.0) / d21;
            for (int i = pk + 2 + t; i < Np; i += blockDim.x) {
                const double a1 = F[i + (int64_t)pk * ld], a2 = F[i + (int64_t)(pk + 1) * ld];
                F[i + (int64_t)pk * ld] = tt * (d11 * a1 - a2);
                F[i + (int64_t)(pk + 1) * ld] = tt * (d22 * a2 - a1);
            }
            if (t == 0) {
                dvec[pk] = st->p11; dvec[pk + 1] = st->p22; doff[pk] = d21; doff[pk + 1] = 0.0;
                ptype[pk] = 2; ptype[pk + 1] = 3;
                F[(pk + 1) + (int64_t)pk * ld] = 0.0;  // L is unit lower: the 2x2 block's off-diagonal lives in doff
            }
        }
    }
    __syncthreads();
    const int k = st->k;
    if (k >= Np) {
        if (t == 0) { st->kstep = 0; st->pk = -1; }
        return;
    }
    // ---- pivot search in column k
    const double akk = F[k + (int64_t)k * ld];
    const double absakk = fabs(akk);
    double v = -1.0;
    int vi = 0x7fffffff;
    for (int i = k + 1 + t; i < Np; i += blockDim.x) {
        const double a = fabs(F[i + (int64_t)k * ld]);
        if (a > v || !(a <= DBL_MAX)) { v = !(a <= DBL_MAX) ? DBL_MAX : a; vi = i; }
    }
    double colmax;
    int imax;
    block_argmax(v, vi, sval, sidx, colmax, imax);
    if (colmax < 0.0) colmax = 0.0;  // k is the last column
    int kstep = 1, kp = k;
    bool zero = false;
    if (!(fmax(absakk, colmax) > 0.0) || !(absakk <= DBL_MAX) || colmax >= DBL_MAX) {
        zero = true;  // column is exactly zero (or not finite): no elimination, info reports it
    } else if (absakk < BK_ALPHA * colmax) {
        // largest off-diagonal entry in row/column imax of the trailing matrix
        double rv = -1.0;
        int ri = 0x7fffffff;
        for (int j = k + t; j < imax; j += blockDim.x) {
            const double a = fabs(F[imax + (int64_t)j * ld]);
            if (a > rv) { rv = a; ri = j; }
        }
        for (int i = imax + 1 + t; i < Np; i += blockDim.x) {
            const double a = fabs(F[i + (int64_t)imax * ld]);
            if (a > rv) { rv = a; ri = i; }
        }
        double rowmax;
        int jmax;
        block_argmax(rv, ri, sval, sidx, rowmax, jmax);
        if (absakk >= BK_ALPHA * colmax * (colmax / rowmax)) {
            kp = k;
        } else if (fabs(F[imax + (int64_t)imax * ld]) >= BK_ALPHA * rowmax) {
            kp = imax;
        } else {
            kp = imax;
            kstep = 2;
        }
    }
    if (t == 0) {
        st->kstep = zero ? -1 : kstep;  // -1: zero pivot (treated as a 1x1 step without elimination)
        st->kp = kp;
        if (zero && st->info == 0) st->info = k + 1;
    }
}

// Symmetric interchange of rows/columns kk = k + kstep - 1 and kp (> kk) in the lower triangle, the same
// rows of the previous columns, and the permutation vector.
__global__ void bk_swap_kernel(double* __restrict__ F, int64_t ld, int Np, const BkState* st, int* __restrict__ perm) {
    const int ks = st->kstep;
    if (ks == 0) return;
    const int k = st->k, kstep = ks < 0 ? 1 : ks, kk = k + kstep - 1, kp = st->kp;
    if (kp == kk) return;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Np) return;
    auto swp = [&](int64_t a, int64_t b) { const double x = F[a]; F[a] = F[b]; F[b] = x; };
    if (t < k) {
        swp(kk + (int64_t)t * ld, kp + (int64_t)t * ld);  // previous columns (P A P^T = L D L^T with ONE permutation)
    } else if (t > kp) {
        swp(t + (int64_t)kk * ld, t + (int64_t)kp * ld);
    } else if (t > kk && t < kp) {
        swp(t + (int64_t)kk * ld, kp + (int64_t)t * ld);
    } else if (t == kk) {
        swp(kk + (int64_t)kk * ld, kp + (int64_t)kp * ld);
        if (kstep == 2) swp((k + 1) + (int64_t)k * ld, kp + (int64_t)k * ld);
        const int p = perm[kk]; perm[kk] = perm[kp]; perm[kp] = p;
    }
}

// Trailing update with the UNSCALED pivot columns (the scaling is the next decide kernel's first job):
//   1x1:  A[i,j] -= a_i a_j / d              2x2:  A[i,j] -= [a_i1 a_i2] inv(D2) [a_j1 a_j2]^T
// One thread per entry of the lower triangle below the pivot block; the grid is sized by the host for the
// smallest k this step can have.  Thread (0,0) advances the state.
__global__ __launch_bounds__(256) void bk_update_kernel(double* __restrict__ F, int64_t ld, int Np, BkState* st, int kmin) {
    const int ks = st->kstep;
    if (ks == 0) return;
    const int k = st->k, kstep = ks < 0 ? 1 : ks, j0 = k + kstep;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i = kmin + 1 + blockIdx.x * 16 + tx, j = kmin + 1 + blockIdx.y * 16 + ty;
    const bool first = blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0;
    if (ks > 0 && i >= j0 && j >= j0 && i >= j && i < Np) {
        if (kstep == 1) {
            const double d = F[k + (int64_t)k * ld];
            F[i + (int64_t)j * ld] -= F[i + (int64_t)k * ld] * (F[j + (int64_t)k * ld] / d);
        } else {
            const double p11 = F[k + (int64_t)k * ld], p21 = F[(k + 1) + (int64_t)k * ld],
                         p22 = F[(k + 1) + (int64_t)(k + 1) * ld];
            const double d11 = p22 / p21, d22 = p11 / p21;
            const double tt = 1.0 / (d11 * d22 - 1.0) / p21;
            const double aj1 = F[j + (int64_t)k * ld], aj2 = F[j + (int64_t)(k + 1) * ld];
            const double w1 = tt * (d11 * aj1 - aj2), w2 = tt * (d22 * aj2 - aj1);
            F[i + (int64_t)j * ld] -= F[i + (int64_t)k * ld] * w1 + F[i + (int64_t)(k + 1) * ld] * w2;
        }
    }
    if (first) {
        // Every thread of this launch reads st->k / kstep before it can see the new values?  No ordering is
        // guaranteed inside a launch, so the state for the NEXT step goes to a shadow (pk fields) and `k` itself
        // is advanced by bk_advance_kernel, a separate launch.
        st->pk = ks > 0 ? k : -1;
        st->pkstep = kstep;
        st->p11 = F[k + (int64_t)k * ld];
        st->p21 = kstep == 2 ? F[(k + 1) + (int64_t)k * ld] : 0.0;
        st->p22 = kstep == 2 ? F[(k + 1) + (int64_t)(k + 1) * ld] : 0.0;
    }
}

__global__ void bk_advance_kernel(BkState* st, double* __restrict__ dvec, double* __restrict__ doff,
                                  int* __restrict__ ptype) {
    const int ks = st->kstep;
    if (ks == 0) return;
    if (ks < 0) {  // zero pivot: recorded as a 1x1 block with d = 0
        dvec[st->k] = 0.0; doff[st->k] = 0.0; ptype[st->k] = 1;
    }
    st->k += ks < 0 ? 1 : ks;
}

__global__ void bk_init_kernel(BkState* st, int* __restrict__ perm, int Np) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < Np) perm[t] = t;
    if (t == 0) { st->k = 0; st->kstep = 1; st->kp = 0; st->info = 0; st->pk = -1; st->pkstep = 1; st->p11 = st->p21 = st->p22 = 0.0; }
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
        if (rc) return -2;
    }
    BkState* st = reinterpret_cast<BkState*>(ls->bk_state.p);
    MNK_HIP(hipMemsetAsync(ls->info_dev.p, 0, sizeof(int), s));
    hipLaunchKernelGGL(bk_init_kernel, dim3((Np + 255) / 256), dim3(256), 0, s, st, ls->bk_perm.p, Np);
    for (int step = 0; step < Np; ++step) {
        hipLaunchKernelGGL(bk_decide_kernel, dim3(1), dim3(1024), 0, s, F, ld, Np, st, ls->dvec.p, ls->bk_doff.p, ls->bk_ptype.p);
        hipLaunchKernelGGL(bk_swap_kernel, dim3((Np + 255) / 256), dim3(256), 0, s, F, ld, Np, st, ls->bk_perm.p);
        const int rem = Np - step - 1;  // rows below the smallest possible k of this step
        if (rem > 0) {
            const unsigned g = (unsigned)((rem + 15) / 16);
            hipLaunchKernelGGL(bk_update_kernel, dim3(g, g), dim3(256), 0, s, F, ld, Np, st, step);
        } else {
            hipLaunchKernelGGL(bk_update_kernel, dim3(1, 1), dim3(256), 0, s, F, ld, Np, st, step);
        }
        hipLaunchKernelGGL(bk_advance_kernel, dim3(1), dim3(1), 0, s, st, ls->dvec.p, ls->bk_doff.p, ls->bk_ptype.p);
    }
    // finish the last pivot's column scaling (k >= Np now: no new pivot is chosen)
    hipLaunchKernelGGL(bk_decide_kernel, dim3(1), dim3(1024), 0, s, F, ld, Np, st, ls->dvec.p, ls->bk_doff.p, ls->bk_ptype.p);
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
