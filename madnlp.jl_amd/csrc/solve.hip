// Triangular solves with the blocked factor: `solve_linear_system!` of the
// AbstractLinearSolver contract (reference src/LinearSolvers/lapack_common.jl:75-81),
// replacing LAPACK dpotrs / dsytrs (reference src/LinearSolvers/lapack.jl:150-153,169-172).
//
// HBM-bound: each sweep reads the lower triangle once (8*N^2/2 bytes).  Both sweeps are
// right-looking over 256-column steps; the 256x256 diagonal triangles are applied as GEMVs
// with their explicit inverses, which the factorization computes once (`linv256_kernel`, from
// the 64x64 inv(L_jj) blocks), so a step has no substitution chain at all:
//   forward : x_j = inv(L_jj) b_j (1 workgroup) ; b[below] -= L[below, j] x_j  (64 rows per
//             workgroup, 4 column quarters per row, coalesced down the columns)
//   backward: x_j = inv(L_jj)^T z_j             ; z[before] -= L[j, before]^T x_j (16 lanes per
//             column, 4 rows each = 512-byte column segments, shuffle reduction)
// Launches per solve: 4 * N/256.
#include "ls.h"

namespace mnk {

constexpr int SB = 256;

// ---- explicit inverse of every 256x256 diagonal triangle (unit diagonal for LDL) ----------------
// One workgroup per (diagonal block, 64-column block q of the inverse):
//   X_qq = inv(L_qq);  X_bq = -inv(L_bb) * sum_{q<=b'<b} L_{b,b'} X_{b'q}   for b = q+1..nb-1
// 64x64x64 products with both operands staged in LDS, 4x4 outputs per thread.
// Writes Inv (column-major, ld 256) and its transpose (for the backward sweep).
__device__ __forceinline__ void mm64_acc(const double* As, const double* Bs, double (&c)[4][4], int ty, int tx) {
    // c[i][j] += sum_k A[4ty+i][k] * B[k][4tx+j];  As[k*64 + r] = A[r][k] ; Bs[j*64 + k] = B[k][j]
#pragma unroll 4
    for (int k = 0; k < 64; ++k) {
        double a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = As[k * 64 + 4 * ty + i];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = Bs[(4 * tx + j) * 64 + k];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) c[i][j] = fma(a[i], b[j], c[i][j]);
    }
}

__global__ __launch_bounds__(256) void linv256_kernel(const double* __restrict__ F, int64_t ld,
                                                      const double* __restrict__ Linv64, double* __restrict__ Inv,
                                                      double* __restrict__ InvT, int64_t Np,
                                                      const int* __restrict__ info) {
    __shared__ double As[64 * 64];
    __shared__ double Bs[64 * 64];
    if (*info != 0) return;
    const int64_t blk = blockIdx.x;      // 256-block
    const int q = blockIdx.y;            // column block of the inverse
    const int64_t j0 = blk * SB;
    const int nb = (int)((Np - j0 < SB ? Np - j0 : SB) / 64);
    double* out = Inv + blk * (int64_t)(SB * SB);
    double* outT = InvT + blk * (int64_t)(SB * SB);
    const int t = threadIdx.x, ty = t & 15, tx = t >> 4;
    // zero the part of column block q above the diagonal block (rows of earlier blocks)
    for (int e = t; e < 64 * 64 * q; e += 256) {
        const int r = e % (64 * q), c = e / (64 * q);
        out[r + (int64_t)(64 * q + c) * SB] = 0.0;
        outT[(64 * q + c) + (int64_t)r * SB] = 0.0;
    }
    if (q >= nb) {  // padding block of a short last step: identity-free zeros
        for (int e = t; e < 64 * SB; e += 256) {
            const int r = e % SB, c = e / SB;
            out[r + (int64_t)(64 * q + c) * SB] = 0.0;
            outT[(64 * q + c) + (int64_t)r * SB] = 0.0;
        }
        return;
    }
    for (int b = q; b < 4; ++b) {
        double c[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) c[i][j] = 0.0;
        if (b < nb) {
            if (b == q) {
                // X_qq = inv(L_qq): straight copy
                const double* Li = Linv64 + ((j0 >> 6) + q) * 4096;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) c[i][j] = Li[(4 * ty + i) + 64 * (4 * tx + j)];
            } else {
                // S = sum_{b'} L_{b,b'} X_{b',q}
                for (int bp = q; bp < b; ++bp) {
                    __syncthreads();
                    for (int e = t; e < 4096; e += 256) {
                        const int r = e & 63, k = e >> 6;
                        As[k * 64 + r] = F[(j0 + 64 * b + r) + (j0 + 64 * bp + k) * ld];   // L_{b,bp}[r][k]
                        Bs[k * 64 + r] = out[(64 * bp + r) + (int64_t)(64 * q + k) * SB];     // X_{bp,q}[r][k] -> Bs[j*64+k]
                    }
                    __syncthreads();
                    mm64_acc(As, Bs, c, ty, tx);
                }
                // X_bq = -inv(L_bb) * S : stage S (as B operand) and inv(L_bb) (as A operand)
                __syncthreads();
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) Bs[(4 * tx + j) * 64 + 4 * ty + i] = c[i][j];
                const double* Li = Linv64 + ((j0 >> 6) + b) * 4096;
                for (int e = t; e < 4096; e += 256) As[e] = Li[e];  // As[k*64+r] = inv(L_bb)[r][k]
                __syncthreads();
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) c[i][j] = 0.0;
                mm64_acc(As, Bs, c, ty, tx);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) c[i][j] = -c[i][j];
            }
        }
        // store block (b, q) and its transpose (zeros for the padding rows of a short last step)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 64 * b + 4 * ty + i, cc = 64 * q + 4 * tx + j;
                out[r + (int64_t)cc * SB] = c[i][j];
                outT[cc + (int64_t)r * SB] = c[i][j];
            }
        __threadfence_block();
        __syncthreads();
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

}  // namespace mnk

using namespace mnk;

// called at the end of the factorization (after linv64_kernel)
int mnk_ls_build_inverses(mnk_ls* ls, hipStream_t s) {
    const int64_t nblk = (ls->Np + SB - 1) / SB;
    hipLaunchKernelGGL(linv256_kernel, dim3((unsigned)nblk, 4), dim3(256), 0, s, ls->fact.p, ls->ld, ls->linv.p,
                       ls->linv256.p, ls->linv256t.p, ls->Np, ls->info_dev.p);
    MNK_HIP(hipGetLastError());
    return 0;
}

// xdev: 2*Np doubles; on entry xdev[0:Np] = rhs (zero padded); on exit xdev[0:Np] = solution.
int mnk_ls_run_solve(mnk_ls* ls, double* xdev) {
    hipStream_t s = ls->ctx->stream;
    const int64_t Np = ls->Np, ld = ls->ld;
    const int ldl = ls->algo == MNK_LDL;
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
    if (ldl)
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
    return 0;
}
