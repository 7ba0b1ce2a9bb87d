// Elementwise pieces of solve_kkt! / mul! shared by the sparse and the dense KKT units (reference
// src/IPM/kernels.jl:161-204): reduce_rhs!, finish_aug_solve!, the bound part of _kktmul!.
#pragma once
#include "common.h"

namespace mnk {

// ---- device-side solve_kkt! / mul! pieces (reference src/IPM/kernels.jl:161-204, factorization.jl:143-167,289-308)
// reduce_rhs!: xp_lr -= wl ./ l_diag (one launch per bound side: a variable may carry both bounds)
static __global__ void reduce_rhs_kernel(double* __restrict__ w, const int64_t* __restrict__ ind, const double* __restrict__ wb,
                                  const double* __restrict__ diag, int64_t nb) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < nb) w[ind[i]] -= wb[i] / diag[i];
}
// finish_aug_solve!: dlb = (-dlb + l_lower .* xp_lr) ./ l_diag ; dub = (dub - u_lower .* xp_ur) ./ u_diag
static __global__ void finish_aug_kernel(double* __restrict__ db, const double* __restrict__ w, const int64_t* __restrict__ ind,
                                  const double* __restrict__ lower, const double* __restrict__ diag, int64_t nb, int upper) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < nb) db[i] = upper ? (db[i] - lower[i] * w[ind[i]]) / diag[i] : (-db[i] + lower[i] * w[ind[i]]) / diag[i];
}
// _kktmul!, bound part.  side 0: xp_lr -= alpha dlb(x) ; dlb(w) = beta dlb(w) + alpha (x_lr l_lower - dlb(x) l_diag)
//                        side 1: xp_ur += alpha dub(x) ; dub(w) = beta dub(w) + alpha (x_ur u_lower + dub(x) u_diag)
static __global__ void kktmul_bound_kernel(double* __restrict__ w, double* __restrict__ wb, const double* __restrict__ x,
                                    const double* __restrict__ xb, const int64_t* __restrict__ ind,
                                    const double* __restrict__ lower, const double* __restrict__ diag, double alpha,
                                    double beta, int64_t nb, int upper) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= nb) return;
    const int64_t p = ind[i];
    if (upper) {
        w[p] += alpha * xb[i];
        wb[i] = beta * wb[i] + alpha * (x[p] * lower[i] + xb[i] * diag[i]);
    } else {
        w[p] -= alpha * xb[i];
        wb[i] = beta * wb[i] + alpha * (x[p] * lower[i] - xb[i] * diag[i]);
    }
}

// ---- device-side feeders of build_kkt!: set_aug_diagonal! (reference src/IPM/kernels.jl:4-27) and
// regularize_diagonal! (src/KKT/KKTsystem.jl:222-226), so that an iteration needs no host vector at all:
//   l_diag = xl_r - x_lr, l_lower = zl_r      (upper = 0)      u_diag = x_ur - xu_r, u_lower = zu_r     (upper = 1)
static __global__ void aug_terms_kernel(double* __restrict__ diag, double* __restrict__ lower, const double* __restrict__ x,
                                        const double* __restrict__ xb, const double* __restrict__ z,
                                        const int64_t* __restrict__ ind, int64_t nb, int upper) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= nb) return;
    const int64_t p = ind[i];
    diag[i] = upper ? x[p] - xb[p] : xb[p] - x[p];
    lower[i] = z[p];
}
//   pr_diag[ind] -= lower ./ diag   (one launch per bound side: a variable may carry both bounds)
static __global__ void aug_diag_sub_kernel(double* __restrict__ pr_diag, const double* __restrict__ lower,
                                           const double* __restrict__ diag, const int64_t* __restrict__ ind, int64_t nb) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < nb) pr_diag[ind[i]] -= lower[i] / diag[i];
}
static __global__ void vec_fill2_kernel(double* __restrict__ a, double* __restrict__ b, double v, int64_t n) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) { a[i] = v; if (b) b[i] = v; }
}
static __global__ void vec_shift2_kernel(double* __restrict__ a, double* __restrict__ b, double v, int64_t n) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) { a[i] += v; if (b) b[i] += v; }
}

// One view over the diagonal state of a KKT handle (sparse condensed, dense condensed, dense augmented).
struct AugDiagView {
    mnk_ctx* ctx;
    int64_t npr, ndu, nlb, nub;  // lengths of pr_diag / du_diag / the bound sides
    double *reg, *pr_diag, *du_diag, *l_diag, *u_diag, *l_lower, *u_lower;
    const int64_t *ind_lb, *ind_ub;
    DevBuf<double>* feed;        // staging for host-resident iterates (5 * npr)
};

static inline int kkt_set_aug_diagonal(const AugDiagView& v, const double* x, const double* xl, const double* xu,
                                       const double* zl, const double* zu, double primal_reg, double dual_reg, int loc) {
    hipStream_t s = v.ctx->stream;
    const double* in[5] = {x, xl, xu, zl, zu};
    if (loc != MNK_DEVICE) {
        if (v.feed->n < (size_t)(5 * v.npr)) {
            int rc = v.feed->alloc((size_t)(5 * v.npr));
            if (rc) return rc;
        }
        for (int k = 0; k < 5; ++k) {
            MNK_HIP(mnk::h2d_copy(v.feed->p + k * v.npr, in[k], v.npr * sizeof(double), s));
            in[k] = v.feed->p + k * v.npr;
        }
        MNK_HIP(mnk::stream_wait(s));  // the caller's arrays are only valid for the duration of the call
    }
#define MNK_G1(cnt) dim3((unsigned)(((cnt) + 255) / 256)), dim3(256), 0, s
    hipLaunchKernelGGL(vec_fill2_kernel, MNK_G1(v.npr), v.reg, v.pr_diag, primal_reg, v.npr);
    if (v.ndu > 0) hipLaunchKernelGGL(vec_fill2_kernel, MNK_G1(v.ndu), v.du_diag, (double*)nullptr, -dual_reg, v.ndu);
    if (v.nlb > 0) {
        hipLaunchKernelGGL(aug_terms_kernel, MNK_G1(v.nlb), v.l_diag, v.l_lower, in[0], in[1], in[3], v.ind_lb, v.nlb, 0);
        hipLaunchKernelGGL(aug_diag_sub_kernel, MNK_G1(v.nlb), v.pr_diag, v.l_lower, v.l_diag, v.ind_lb, v.nlb);
    }
    if (v.nub > 0) {
        hipLaunchKernelGGL(aug_terms_kernel, MNK_G1(v.nub), v.u_diag, v.u_lower, in[0], in[2], in[4], v.ind_ub, v.nub, 1);
        hipLaunchKernelGGL(aug_diag_sub_kernel, MNK_G1(v.nub), v.pr_diag, v.u_lower, v.u_diag, v.ind_ub, v.nub);
    }
    MNK_HIP(hipGetLastError());
    return 0;
}

// set_aug_RR! (reference src/IPM/kernels.jl:72-87) + _set_aug_diagonal! (:22-27): the robust restorer's diagonals from
// DEVICE-resident vectors: reg = primal_reg + zeta D_R^2, du_diag = -dual_reg - pp/zp - nn/zn, bound terms as above.
static __global__ void rr_reg_kernel(double* __restrict__ reg, double* __restrict__ pr_diag, const double* __restrict__ D,
                                     double primal_reg, double zeta, int64_t n) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = primal_reg + zeta * (D[i] * D[i]);
    reg[i] = v;
    pr_diag[i] = v;
}
static __global__ void rr_du_kernel(double* __restrict__ du_diag, const double* __restrict__ pp, const double* __restrict__ zp,
                                    const double* __restrict__ nn, const double* __restrict__ zn, double dual_reg, int64_t m) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < m) du_diag[i] = -dual_reg - pp[i] / zp[i] - nn[i] / zn[i];
}
static inline int kkt_set_aug_RR(const AugDiagView& v, const double* x, const double* xl, const double* xu, const double* zl,
                                 const double* zu, const double* D_R, const double* pp, const double* zp, const double* nn,
                                 const double* zn, double zeta, double primal_reg, double dual_reg) {
    hipStream_t s = v.ctx->stream;
    hipLaunchKernelGGL(rr_reg_kernel, MNK_G1(v.npr), v.reg, v.pr_diag, D_R, primal_reg, zeta, v.npr);
    if (v.ndu > 0) hipLaunchKernelGGL(rr_du_kernel, MNK_G1(v.ndu), v.du_diag, pp, zp, nn, zn, dual_reg, v.ndu);
    if (v.nlb > 0) {
        hipLaunchKernelGGL(aug_terms_kernel, MNK_G1(v.nlb), v.l_diag, v.l_lower, x, xl, zl, v.ind_lb, v.nlb, 0);
        hipLaunchKernelGGL(aug_diag_sub_kernel, MNK_G1(v.nlb), v.pr_diag, v.l_lower, v.l_diag, v.ind_lb, v.nlb);
    }
    if (v.nub > 0) {
        hipLaunchKernelGGL(aug_terms_kernel, MNK_G1(v.nub), v.u_diag, v.u_lower, x, xu, zu, v.ind_ub, v.nub, 1);
        hipLaunchKernelGGL(aug_diag_sub_kernel, MNK_G1(v.nub), v.pr_diag, v.u_lower, v.u_diag, v.ind_ub, v.nub);
    }
    MNK_HIP(hipGetLastError());
    return 0;
}

static inline int kkt_regularize_diagonal(const AugDiagView& v, double primal, double dual) {
    hipStream_t s = v.ctx->stream;
    hipLaunchKernelGGL(vec_shift2_kernel, MNK_G1(v.npr), v.reg, v.pr_diag, primal, v.npr);
    if (v.ndu > 0) hipLaunchKernelGGL(vec_shift2_kernel, MNK_G1(v.ndu), v.du_diag, (double*)nullptr, -dual, v.ndu);
    MNK_HIP(hipGetLastError());
    return 0;
}

static inline int kkt_get_diagonals(const AugDiagView& v, double* pr_diag, double* du_diag, double* reg, double* l_diag,
                                    double* u_diag, double* l_lower, double* u_lower) {
    hipStream_t s = v.ctx->stream;
    auto get = [&](double* dst, const double* src, int64_t n) -> int {
        if (dst && n > 0) MNK_HIP(mnk::d2h_copy(dst, src, n * sizeof(double), s));
        return 0;
    };
    int rc = get(pr_diag, v.pr_diag, v.npr) | get(du_diag, v.du_diag, v.ndu) | get(reg, v.reg, v.npr) |
             get(l_diag, v.l_diag, v.nlb) | get(u_diag, v.u_diag, v.nub) | get(l_lower, v.l_lower, v.nlb) |
             get(u_lower, v.u_lower, v.nub);
    if (rc) return rc;
    MNK_HIP(mnk::stream_wait(s));
    return 0;
}
#undef MNK_G1

}  // namespace mnk
