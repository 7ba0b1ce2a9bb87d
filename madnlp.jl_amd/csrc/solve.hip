// Triangular solves with the blocked factor: `solve_linear_system!` of the
// AbstractLinearSolver contract (reference src/LinearSolvers/lapack_common.jl:75-81),
// replacing LAPACK dpotrs / dsytrs (reference src/LinearSolvers/lapack.jl:150-153,169-172).
//
// HBM-bound: each sweep reads the lower triangle once (8*N^2/2 bytes).  Both sweeps
// are right-looking over 64-column blocks; the 64x64 diagonal solves are GEMVs with
// the inv(L_jj) blocks produced by the factorization, so a step has no sequential
// substitution chain.
//   forward : y_j = inv(L_jj) b_j ;  b[below] -= L[below, j] y_j      (row per thread, coalesced)
//   backward: x_j = inv(L_jj)^T z_j ; z[before] -= L[j, before]^T x_j  (16 lanes per column,
//             4 rows each = one 512-byte column segment, DPP/shuffle reduction)
#include "ls.h"

namespace mnk {

__global__ __launch_bounds__(256) void fwd_step_kernel(const double* __restrict__ F, int64_t ld,
                                                       const double* __restrict__ Linv,
                                                       double* __restrict__ b, double* __restrict__ y,
                                                       const double* __restrict__ dinv, int ldl,
                                                       int64_t j0, int64_t Np) {
    __shared__ double part[4][64];
    __shared__ double xj[64];
    const int t = threadIdx.x, row = t & 63, p = t >> 6;
    double acc = 0.0;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int cc = p * 16 + c;
        acc += Linv[row + 64 * cc] * b[j0 + cc];
    }
    part[p][row] = acc;
    __syncthreads();
    if (p == 0) {
        const double v = (part[0][row] + part[1][row]) + (part[2][row] + part[3][row]);
        xj[row] = v;
        if (blockIdx.x == 0) y[j0 + row] = ldl ? v * dinv[j0 + row] : v;
    }
    __syncthreads();
    const int64_t r = j0 + 64 + (int64_t)blockIdx.x * 256 + t;
    if (r < Np) {
        const double* Fr = F + r + j0 * ld;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int c = 0; c < 64; c += 4) {
            s0 += Fr[(int64_t)(c + 0) * ld] * xj[c + 0];
            s1 += Fr[(int64_t)(c + 1) * ld] * xj[c + 1];
            s2 += Fr[(int64_t)(c + 2) * ld] * xj[c + 2];
            s3 += Fr[(int64_t)(c + 3) * ld] * xj[c + 3];
        }
        b[r] -= (s0 + s1) + (s2 + s3);
    }
}

__global__ __launch_bounds__(256) void bwd_step_kernel(const double* __restrict__ F, int64_t ld,
                                                       const double* __restrict__ Linv,
                                                       double* __restrict__ z, double* __restrict__ x,
                                                       int64_t j0) {
    __shared__ double part[4][64];
    __shared__ double xj[64];
    const int t = threadIdx.x, c = t & 63, p = t >> 6;
    // xj[c] = sum_r inv(L)[r][c] * z[j0 + r]
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rr = p * 16 + r;
        acc += Linv[rr + 64 * c] * z[j0 + rr];
    }
    part[p][c] = acc;
    __syncthreads();
    if (p == 0) {
        const double v = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
        xj[c] = v;
        if (blockIdx.x == 0) x[j0 + c] = v;
    }
    __syncthreads();
    const int64_t cb = (int64_t)blockIdx.x * 64;
    if (cb >= j0) return;
    const int lane = t & 63, w = t >> 6;
    const int sub = lane & 15, colq = lane >> 4;
    const double x0 = xj[4 * sub], x1 = xj[4 * sub + 1], x2 = xj[4 * sub + 2], x3 = xj[4 * sub + 3];
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int64_t col = cb + pass * 16 + w * 4 + colq;
        const double* Fp = F + j0 + 4 * sub + col * ld;
        double s = (Fp[0] * x0 + Fp[1] * x1) + (Fp[2] * x2 + Fp[3] * x3);
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        s += __shfl_xor(s, 8);
        if (sub == 0) z[col] -= s;
    }
}

}  // namespace mnk

using namespace mnk;

// xdev: 2*Np doubles; on entry xdev[0:Np] = rhs (zero padded); on exit xdev[0:Np] = solution.
int mnk_ls_run_solve(mnk_ls* ls, double* xdev) {
    hipStream_t s = ls->ctx->stream;
    const int64_t Np = ls->Np, ld = ls->ld;
    const int ldl = ls->algo == MNK_LDL;
    double* b = xdev;
    double* y = xdev + Np;
    const int64_t nb = Np / NBI;
    for (int64_t jb = 0; jb < nb; ++jb) {
        const int64_t j0 = jb * NBI;
        const int64_t below = Np - j0 - NBI;
        const int grid = (int)std::max<int64_t>(1, (below + 255) / 256);
        hipLaunchKernelGGL(fwd_step_kernel, dim3(grid), dim3(256), 0, s, ls->fact.p, ld,
                           ls->linv.p + jb * NBI * NBI, b, y, ls->dinv.p, ldl, j0, Np);
    }
    for (int64_t jb = nb - 1; jb >= 0; --jb) {
        const int64_t j0 = jb * NBI;
        const int grid = (int)std::max<int64_t>(1, j0 / 64);
        hipLaunchKernelGGL(bwd_step_kernel, dim3(grid), dim3(256), 0, s, ls->fact.p, ld,
                           ls->linv.p + jb * NBI * NBI, y, b, j0);
    }
    MNK_HIP(hipGetLastError());
    return 0;
}
