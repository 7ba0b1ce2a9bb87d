// Triangular solves with the blocked factor: `solve_linear_system!` of the
// AbstractLinearSolver contract (reference src/LinearSolvers/lapack_common.jl:75-81),
// replacing LAPACK dpotrs / dsytrs (reference src/LinearSolvers/lapack.jl:150-153,169-172).
//
// HBM-bound: each sweep reads the lower triangle once (8*N^2/2 bytes).  Both sweeps are
// right-looking over 256-column steps, ONE launch per step: every workgroup first solves the
// 256x256 diagonal triangle redundantly in LDS (four 64-blocks: GEMV with the inv(L_jj) block
// produced by the factorization, then 64x64 GEMV updates -- no substitution chain longer than
// 4), then applies the 256-wide panel to its own rows (forward: row per thread, coalesced down
// the columns) or columns (backward: 16 lanes per column, 4 rows each = 512-byte column
// segments, shuffle reduction).  Launch count per solve: 2 * N/256 instead of 2 * N/64.
#include "ls.h"

namespace mnk {

constexpr int SB = 256;  // step width (columns per launch)

// y = M (64x64, column-major, ld) * x   or   y = M^T * x, by 256 threads; result in out[64] (LDS)
template <bool TRANS>
__device__ __forceinline__ void gemv64(const double* __restrict__ M, int64_t ld, const double* xs, double* part,
                                       double* out, int t) {
    const int r = t & 63, p = t >> 6;
    double acc = 0.0;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int cc = p * 16 + c;
        // TRANS: out[r] = sum_cc M[cc][r] * x[cc] ; else out[r] = sum_cc M[r][cc] * x[cc]
        acc += (TRANS ? M[cc + r * ld] : M[r + cc * ld]) * xs[cc];
    }
    part[p * 64 + r] = acc;
    __syncthreads();
    if (p == 0) out[r] = (part[r] + part[64 + r]) + (part[128 + r] + part[192 + r]);
    __syncthreads();
}

// Forward step over columns [j0, j0+nb*64): x = L_diag^{-1} b (in LDS), then b[rows below] -= L x.
__global__ __launch_bounds__(256) void fwd_step256_kernel(const double* __restrict__ F, int64_t ld,
                                                          const double* __restrict__ Linv,
                                                          double* __restrict__ b, double* __restrict__ y,
                                                          const double* __restrict__ dinv, int ldl, int64_t j0,
                                                          int nb, int64_t Np) {
    __shared__ double xs[SB];
    __shared__ double part[256];
    __shared__ double tmp[64];
    const int t = threadIdx.x;
    if (t < nb * 64) xs[t] = b[j0 + t];
    __syncthreads();
    for (int s = 0; s < nb; ++s) {
        // x_s = inv(L_ss) * b_s
        gemv64<false>(Linv + ((j0 >> 6) + s) * 4096, 64, xs + s * 64, part, tmp, t);
        if (t < 64) xs[s * 64 + t] = tmp[t];
        __syncthreads();
        // b_s' -= L[s', s] x_s for the later blocks of this step
        for (int s2 = s + 1; s2 < nb; ++s2) {
            gemv64<false>(F + (j0 + s2 * 64) + (j0 + s * 64) * ld, ld, xs + s * 64, part, tmp, t);
            if (t < 64) xs[s2 * 64 + t] -= tmp[t];
            __syncthreads();
        }
    }
    if (blockIdx.x == 0 && t < nb * 64) y[j0 + t] = ldl ? xs[t] * dinv[j0 + t] : xs[t];
    const int64_t r = j0 + nb * 64 + (int64_t)blockIdx.x * 256 + t;
    if (r < Np) {
        const double* Fr = F + r + j0 * ld;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        const int ncol = nb * 64;
#pragma unroll 4
        for (int c = 0; c < ncol; c += 4) {
            s0 += Fr[(int64_t)(c + 0) * ld] * xs[c + 0];
            s1 += Fr[(int64_t)(c + 1) * ld] * xs[c + 1];
            s2 += Fr[(int64_t)(c + 2) * ld] * xs[c + 2];
            s3 += Fr[(int64_t)(c + 3) * ld] * xs[c + 3];
        }
        b[r] -= (s0 + s1) + (s2 + s3);
    }
}

// Backward step over rows/cols [j0, j0+nb*64): x = L_diag^{-T} z, then z[cols before] -= L[j-rows, cols]^T x.
__global__ __launch_bounds__(256) void bwd_step256_kernel(const double* __restrict__ F, int64_t ld,
                                                          const double* __restrict__ Linv,
                                                          double* __restrict__ z, double* __restrict__ x,
                                                          int64_t j0, int nb) {
    __shared__ double xs[SB];
    __shared__ double part[256];
    __shared__ double tmp[64];
    const int t = threadIdx.x;
    if (t < nb * 64) xs[t] = z[j0 + t];
    __syncthreads();
    for (int s = nb - 1; s >= 0; --s) {
        // x_s = inv(L_ss)^T z_s
        gemv64<true>(Linv + ((j0 >> 6) + s) * 4096, 64, xs + s * 64, part, tmp, t);
        if (t < 64) xs[s * 64 + t] = tmp[t];
        __syncthreads();
        // z_s' -= L[s, s']^T x_s for the earlier blocks of this step
        for (int s2 = s - 1; s2 >= 0; --s2) {
            gemv64<true>(F + (j0 + s * 64) + (j0 + s2 * 64) * ld, ld, xs + s * 64, part, tmp, t);
            if (t < 64) xs[s2 * 64 + t] -= tmp[t];
            __syncthreads();
        }
    }
    if (blockIdx.x == 0 && t < nb * 64) x[j0 + t] = xs[t];
    // columns before j0: each workgroup takes 64 columns, 16 lanes per column read 4 rows each
    const int64_t cb = (int64_t)blockIdx.x * 64;
    if (cb >= j0) return;
    const int lane = t & 63, w = t >> 6;
    const int sub = lane & 15, colq = lane >> 4;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int rb = 0; rb < nb * 64; rb += 64) {
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

}  // namespace mnk

using namespace mnk;

// xdev: 2*Np doubles; on entry xdev[0:Np] = rhs (zero padded); on exit xdev[0:Np] = solution.
int mnk_ls_run_solve(mnk_ls* ls, double* xdev) {
    hipStream_t s = ls->ctx->stream;
    const int64_t Np = ls->Np, ld = ls->ld;
    const int ldl = ls->algo == MNK_LDL;
    double* b = xdev;
    double* y = xdev + Np;
    for (int64_t j0 = 0; j0 < Np; j0 += SB) {
        const int nb = (int)(std::min<int64_t>(SB, Np - j0) / NBI);
        const int64_t below = Np - j0 - nb * NBI;
        const int grid = (int)std::max<int64_t>(1, (below + 255) / 256);
        hipLaunchKernelGGL(fwd_step256_kernel, dim3(grid), dim3(256), 0, s, ls->fact.p, ld, ls->linv.p, b, y,
                           ls->dinv.p, ldl, j0, nb, Np);
    }
    const int64_t nsteps = (Np + SB - 1) / SB;
    for (int64_t k = nsteps - 1; k >= 0; --k) {
        const int64_t j0 = k * SB;
        const int nb = (int)(std::min<int64_t>(SB, Np - j0) / NBI);
        const int grid = (int)std::max<int64_t>(1, j0 / 64);
        hipLaunchKernelGGL(bwd_step256_kernel, dim3(grid), dim3(256), 0, s, ls->fact.p, ld, ls->linv.p, y, b, j0,
                           nb);
    }
    MNK_HIP(hipGetLastError());
    return 0;
}
