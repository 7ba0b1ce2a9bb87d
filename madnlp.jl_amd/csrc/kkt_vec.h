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

}  // namespace mnk
