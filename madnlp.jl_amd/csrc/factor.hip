// Blocked right-looking Cholesky / static-pivot LDL^T on the device: the
// `factorize!` of the AbstractLinearSolver contract (reference
// src/LinearSolvers/lapack_common.jl:54-66), replacing `transfer_matrix!` +
// LAPACK dpotrf('L') / dsytrf('L') (reference src/LinearSolvers/lapack.jl:145-148,
// 164-167) and their rocSOLVER twins (reference
// lib/MadNLPGPU/ext/MadNLPGPUAMDGPUExt/rocsolver.jl:47-229).
//
// Structure (two-level blocking, everything in HBM, column-major, lower):
//   for each outer panel of NBO columns
//     for each inner block of NBI=64 columns inside it
//        diag64_kernel    : factor the 64x64 diagonal block in one workgroup
//                           (registers + LDS), also emits inv(L_jj)
//        gemm_nt mode 1   : panel solve  X = A_panel * inv(L_jj)^T  (* D^-1 for LDL)
//        gemm_nt mode 0/2 : update the remaining columns of the outer panel (K = 64)
//     gemm_nt mode 2      : trailing update with K = NBO (fp64 MFMA, lower tiles only)
// A device-side `info` word makes every later kernel a no-op once a pivot fails
// (LAPACK stops at the failing column; we cannot stop the host without a sync).
#include <cfloat>
#include <cmath>

#include "ls.h"

namespace mnk {

// ---------------------------------------------------------------------------------------
// 64x64 diagonal block: thread (row i = tid & 63, wave w = tid >> 6) owns row i of the
// columns c = 4*cl + w.  One barrier per pivot: the pivot column is published through a
// double-buffered LDS vector, every thread rescales it by the pivot itself.
// ---------------------------------------------------------------------------------------
template <bool LDL>
__global__ __launch_bounds__(256) void diag64_kernel(double* __restrict__ Ablk, int64_t ld,
                                                      double* __restrict__ Linv, double* __restrict__ dvec,
                                                      double* __restrict__ dinv, int* __restrict__ info,
                                                      int gj, double pivot_tol) {
    __shared__ double colbuf[2][64];
    __shared__ double Lt[64 * 64];  // Lt[k*64 + i] = L[i][k]; later reused to transpose inv(L)
    __shared__ double rd[64];
    if (*info != 0) return;

    const int tid = threadIdx.x;
    const int i = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);

    double a[16];
#pragma unroll
    for (int cl = 0; cl < 16; ++cl) a[cl] = Ablk[i + (int64_t)(4 * cl + w) * ld];

    bool failed = false;
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        const int wo = j & 3, clj = j >> 2;
        if (w == wo) colbuf[j & 1][i] = a[clj];
        __syncthreads();
        const double piv = colbuf[j & 1][j];
        double scale, ldiag;
        if (LDL) {
            double p = piv;
            const bool zero = !(fabs(p) > pivot_tol) || !(fabs(p) <= DBL_MAX);
            if (zero) p = 1.0;  // keep going with a harmless pivot; dvec records the zero
            scale = 1.0 / p;
            ldiag = zero ? 0.0 : p;  // value recorded in dvec
        } else {
            // not positive definite (also catches NaN/Inf): record the first failing pivot and
            // let the remaining (fully unrolled) steps run on harmless values.
            const bool bad = !(piv > 0.0) || !(piv <= DBL_MAX);
            if (bad && !failed) {
                failed = true;
                if (tid == 0) atomicCAS(info, 0, gj + j + 1);
            }
            ldiag = bad ? 1.0 : sqrt(piv);
            scale = 1.0 / ldiag;
        }
        // my_l = L[i][j] (for LDL: w_ij / d_j)
        const double my_w = colbuf[j & 1][i];
        const double my_l = my_w * scale;
#pragma unroll
        for (int cl = 0; cl < 16; ++cl) {
            const int c = 4 * cl + w;
            if (c > j) {
                const double wc = colbuf[j & 1][c];
                // Cholesky: a_ic -= l_ij * l_cj ; LDL: a_ic -= l_ij * w_cj
                a[cl] -= my_l * (LDL ? wc : wc * scale);
            }
        }
        if (w == wo) {
            a[clj] = (i == j) ? (LDL ? 1.0 : ldiag) : my_l;
            if (i == j) {
                dvec[j] = ldiag;
                dinv[j] = LDL ? scale : 1.0;
                rd[j] = LDL ? 1.0 : scale;
            }
        }
    }
    if (failed) return;

    // write L back (lower part only; LDL keeps d on the diagonal like LAPACK) and stage L^T in LDS
#pragma unroll
    for (int cl = 0; cl < 16; ++cl) {
        const int c = 4 * cl + w;
        if (i > c) Ablk[i + (int64_t)c * ld] = a[cl];
        if (i == c) Ablk[i + (int64_t)c * ld] = LDL ? dvec[c] : a[cl];
        Lt[c * 64 + i] = (i >= c) ? a[cl] : 0.0;
    }
    __syncthreads();

    // inv(L): wave 0, lane c solves L x = e_c by column-oriented forward substitution.
    if (w == 0) {
        double s[64];
#pragma unroll
        for (int r = 0; r < 64; ++r) s[r] = (r == i) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            const double x = s[k] * rd[k];
            s[k] = x;
#pragma unroll
            for (int r = k + 1; r < 64; ++r) s[r] -= Lt[k * 64 + r] * x;
        }
        // same wave: all reads of Lt above are issued before these writes (LDS is in-order per wave)
#pragma unroll
        for (int r = 0; r < 64; ++r) Lt[i * 64 + r] = s[r];  // Lt[c*64 + r] = inv(L)[r][c]
    }
    __syncthreads();
    // Linv is stored column-major 64x64: Linv[r + 64*c] = inv(L)[r][c]  (= Lt layout) -> coalesced copy
#pragma unroll
    for (int q = 0; q < 16; ++q) Linv[tid + 256 * q] = Lt[tid + 256 * q];
}

// Count signs of D over the first N pivots: out[0]=pos, out[1]=zero, out[2]=neg.
__global__ void inertia_kernel(const double* __restrict__ dvec, int64_t N, unsigned long long* out) {
    unsigned long long pos = 0, zer = 0, neg = 0;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < N; k += (int64_t)gridDim.x * blockDim.x) {
        const double d = dvec[k];
        if (d > 0.0) ++pos;
        else if (d < 0.0) ++neg;
        else ++zer;
    }
    for (int off = 32; off > 0; off >>= 1) {
        pos += __shfl_down(pos, off);
        zer += __shfl_down(zer, off);
        neg += __shfl_down(neg, off);
    }
    if ((threadIdx.x & 63) == 0) {
        if (pos) atomicAdd(&out[0], pos);
        if (zer) atomicAdd(&out[1], zer);
        if (neg) atomicAdd(&out[2], neg);
    }
}

}  // namespace mnk

using namespace mnk;

// ---------------------------------------------------------------------------------------
// factorization driver (host orchestration; every launch is asynchronous on ls->ctx->stream)
// ---------------------------------------------------------------------------------------
int mnk_ls_run_factorization(mnk_ls* ls) {
    hipStream_t s = ls->ctx->stream;
    const int64_t Np = ls->Np, ld = ls->ld;
    const bool ldl = ls->algo == MNK_LDL;
    double* F = ls->fact.p;
    const int64_t NBO = ls->nbo;
    MNK_HIP(hipMemsetAsync(ls->info_dev.p, 0, sizeof(int), s));

    for (int64_t ko = 0; ko < Np; ko += NBO) {
        const int64_t nbo = std::min<int64_t>(NBO, Np - ko);
        const int64_t kend = ko + nbo;
        for (int64_t j = ko; j < kend; j += NBI) {
            double* Ajj = F + j + j * ld;
            double* Linv = ls->linv.p + (j / NBI) * (NBI * NBI);
            if (ldl)
                hipLaunchKernelGGL(diag64_kernel<true>, dim3(1), dim3(256), 0, s, Ajj, ld, Linv,
                                   ls->dvec.p + j, ls->dinv.p + j, ls->info_dev.p, (int)j, ls->pivot_tol);
            else
                hipLaunchKernelGGL(diag64_kernel<false>, dim3(1), dim3(256), 0, s, Ajj, ld, Linv,
                                   ls->dvec.p + j, ls->dinv.p + j, ls->info_dev.p, (int)j, ls->pivot_tol);
            const int64_t r0 = j + NBI;
            const int64_t Mr = Np - r0;
            if (Mr <= 0) continue;
            double* Apanel = F + r0 + j * ld;  // Mr x 64
            // W (= L*D for LDL) lives in the workspace, at the same row and at column (j - ko)
            double* Wpanel = ldl ? ls->wbuf.p + r0 + (j - ko) * ls->ldw : Apanel;
            int rc = launch_gemm_nt(s, 1, Mr, NBI, NBI, Apanel, ld, Linv, NBI, Apanel, ld,
                                    ldl ? ls->dinv.p + j : nullptr, ldl ? Wpanel : nullptr, ls->ldw,
                                    ls->info_dev.p);
            if (rc) return rc;
            // LDL epilogue indexes colscale by the tile-local column: pass dinv + j (done above).
            const int64_t Nc = kend - r0;  // remaining columns of the outer panel
            if (Nc > 0) {
                rc = launch_gemm_nt(s, 2, Mr, Nc, NBI, Wpanel, ldl ? ls->ldw : ld, Apanel, ld,
                                    F + r0 + r0 * ld, ld, nullptr, nullptr, 0, ls->info_dev.p);
                if (rc) return rc;
            }
        }
        const int64_t Mt = Np - kend;
        if (Mt > 0) {
            const double* Wsrc = ldl ? ls->wbuf.p + kend : F + kend + ko * ld;
            int rc = launch_gemm_nt(s, 2, Mt, Mt, nbo, Wsrc, ldl ? ls->ldw : ld, F + kend + ko * ld, ld,
                                    F + kend + kend * ld, ld, nullptr, nullptr, 0, ls->info_dev.p);
            if (rc) return rc;
        }
    }
    ls->factorized = true;
    ls->info_valid = false;
    return 0;
}

int mnk_ls_fetch_info(mnk_ls* ls) {
    if (ls->info_valid) return 0;
    hipStream_t s = ls->ctx->stream;
    MNK_HIP(hipMemsetAsync(ls->inertia_dev.p, 0, 3 * sizeof(unsigned long long), s));
    if (ls->algo == MNK_LDL) {
        const int blocks = (int)std::min<int64_t>(256, (ls->N + 255) / 256);
        hipLaunchKernelGGL(inertia_kernel, dim3(blocks), dim3(256), 0, s, ls->dvec.p, ls->N, ls->inertia_dev.p);
    }
    unsigned long long h[3];
    int hinfo = 0;
    MNK_HIP(hipMemcpyAsync(h, ls->inertia_dev.p, sizeof(h), hipMemcpyDeviceToHost, s));
    MNK_HIP(hipMemcpyAsync(&hinfo, ls->info_dev.p, sizeof(int), hipMemcpyDeviceToHost, s));
    MNK_HIP(hipStreamSynchronize(s));
    ls->info = hinfo;
    if (ls->algo == MNK_LDL) {
        ls->npos = (int64_t)h[0];
        ls->nzero = (int64_t)h[1];
        ls->nneg = (int64_t)h[2];
        if (ls->nzero > 0 && ls->info == 0) ls->info = 1;  // LAPACK-style "singular D" signal
    } else {
        // inertia_cholesky, reference src/LinearSolvers/lapack_common.jl:96-98
        if (hinfo == 0) { ls->npos = ls->N; ls->nzero = 0; ls->nneg = 0; }
        else { ls->npos = 0; ls->nzero = ls->N; ls->nneg = 0; }
    }
    ls->info_valid = true;
    return 0;
}
