// DenseCondensedKKTSystem / DenseKKTSystem assembly on the device (mnk_dc_*).
//
// build_kkt!(::DenseCondensedKKTSystem), reference src/KKT/Dense/condensed.jl:157-186:
//     D   = Sigma_s ./ (1 - Sigma_d[ineq] .* Sigma_s)
//     K   = [ H + diag(pr_diag[1:n]) + J_i' D J_i   J_eq' ;  J_eq   diag(du_diag[eq]) ]
// The reference scales J_i by sqrt(D) (`_build_ineq_jac!` :146-155), calls dgemm (:178) and
// then runs a scalar scatter loop (`_build_condensed_kkt_system!` :120-144; device twins
// lib/MadNLPGPU/src/KKT/kernels_dense.jl:81-119).  Here:
//   1. scale_transpose_kernel  : A = (sqrt(D) J_i)^T, n x ns, zero padded      (HBM-bound)
//   2. init_condensed_kernel   : lower(K) = H + diag + equality rows, rest 0    (HBM-bound)
//   3. gemm_nt mode 4          : lower tiles of K += A A^T  (fp64 MFMA, SYRK flop count)
//   4. mirror_kernel           : upper(K) = lower(K)^T (the reference writes both triangles)
// build_kkt!(::DenseKKTSystem), reference src/KKT/Dense/augmented.jl:116-161, is a pure
// scatter (one thread per entry).
#pragma clang fp contract(off)

#include "ls.h"
#include "kkt_vec.h"

namespace mnk {

// A[i + k*lda] = jac[ind_ineq[k] + i*m] * sqrt(D[k]); zero outside (i < n, k < ns).  D[k] = pr_s[k] / (1 - du[ind_ineq[k]] pr_s[k])
// (the slack block's condensation weight) is computed here and stored by the first row of blocks: one launch less per build_kkt!.
__global__ __launch_bounds__(256) void scale_transpose_kernel(double* __restrict__ A, int64_t lda, int64_t npad,
                                                              int64_t kpad, const double* __restrict__ jac,
                                                              int64_t m, int64_t n, int64_t ns,
                                                              const int64_t* __restrict__ ind_ineq,
                                                              double* __restrict__ D, const double* __restrict__ pr_s,
                                                              const double* __restrict__ du) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int64_t k0 = (int64_t)blockIdx.x * 32, i0 = (int64_t)blockIdx.y * 32;
    double Dk = 0.0;
    {
        const int64_t k = k0 + tx;
        if (k < ns) {
            Dk = pr_s[k] / (1.0 - du[ind_ineq[k]] * pr_s[k]);
            if (blockIdx.y == 0 && ty == 0) D[k] = Dk;
        }
    }
    const double sD = sqrt(Dk);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t k = k0 + tx, i = i0 + ty + 8 * q;
        double v = 0.0;
        if (k < ns && i < n) v = jac[ind_ineq[k] + i * m] * sD;
        tile[ty + 8 * q][tx] = v;  // tile[i][k]
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t i = i0 + tx, k = k0 + ty + 8 * q;
        if (i < npad && k < kpad) A[i + k * lda] = tile[tx][ty + 8 * q];
    }
}

// Lower part (row >= first row of the column's 128-tile ... simply all rows >= col) of K:
// Hessian + primal diagonal, equality Jacobian rows, dual regularization; zero elsewhere.
__global__ __launch_bounds__(256) void init_condensed_kernel(double* __restrict__ K, int64_t ldk, int64_t ordpad,
                                                             const double* __restrict__ hess,
                                                             const double* __restrict__ jac, int64_t m, int64_t n,
                                                             int64_t n_eq, const int64_t* __restrict__ ind_eq,
                                                             const double* __restrict__ pr_diag,
                                                             const double* __restrict__ du_diag) {
    const int64_t j = blockIdx.x;
    const int64_t i = (int64_t)blockIdx.y * 256 + threadIdx.x;
    if (i >= ordpad) return;
    double v = 0.0;
    const int64_t order = n + n_eq;
    if (i < order && j < order && i >= j) {
        if (i < n) {  // j <= i < n
            v = hess[i + j * n];
            if (i == j) v = pr_diag[i] + v;
        } else if (j < n) {
            v = jac[ind_eq[i - n] + j * m];
        } else if (i == j) {
            v = du_diag[ind_eq[i - n]];
        }
    }
    K[i + j * ldk] = v;
}

// upper(K) = lower(K)^T on 32x32 tiles (tile row > tile col are read and written transposed;
// diagonal tiles mirror themselves).
__global__ __launch_bounds__(256) void mirror_kernel(double* __restrict__ K, int64_t ldk, int64_t order) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t bi = blockIdx.x, bj = blockIdx.y;
    if (bi < bj) return;
    const int64_t i0 = bi * 32, j0 = bj * 32;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t i = i0 + tx, j = j0 + ty + 8 * q;
        tile[ty + 8 * q][tx] = (i < order && j < order) ? K[i + j * ldk] : 0.0;  // tile[j][i]
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        // write element (row = j0 + tx, col = i0 + ty + 8q) = K[i0 + ty + 8q, j0 + tx]
        const int64_t r = j0 + tx, c = i0 + ty + 8 * q;
        if (r < order && c < order && r < c) K[r + c * ldk] = tile[tx][ty + 8 * q];
    }
}

// DenseKKTSystem: one thread per entry of the (n+ns+m)^2 augmented matrix.
__global__ __launch_bounds__(256) void dense_aug_kernel(double* __restrict__ K, int64_t ldk, int64_t ordpad,
                                                        const double* __restrict__ hess,
                                                        const double* __restrict__ jac, int64_t m, int64_t n,
                                                        int64_t ns, const int64_t* __restrict__ ind_ineq,
                                                        const int32_t* __restrict__ ineq_slot,
                                                        const double* __restrict__ pr_diag,
                                                        const double* __restrict__ du_diag) {
    const int64_t j = blockIdx.x;
    const int64_t i = (int64_t)blockIdx.y * 256 + threadIdx.x;
    if (i >= ordpad) return;
    const int64_t order = n + ns + m;
    double v = 0.0;
    if (i < order && j < order) {
        const int64_t a = i > j ? i : j, b = i > j ? j : i;  // a >= b
        if (a < n) {
            v = (a == b) ? pr_diag[a] + hess[a + a * n] : hess[i + j * n];
        } else if (a < n + ns) {
            if (a == b) v = pr_diag[a];
        } else {
            const int64_t r = a - n - ns;  // constraint row
            if (b < n) v = jac[r + b * m];
            else if (b < n + ns) v = (ineq_slot[r] == (int32_t)(b - n)) ? -1.0 : 0.0;
            else if (a == b) v = du_diag[r];
        }
    }
    K[i + j * ldk] = v;
}


// ---- device-side solve_kkt! / mul! of the dense systems (reference src/IPM/factorization.jl:41-46,190-229,310-330)
// y[j] = alpha * dot(A[:, j], x) + beta * y[j]   (A is rows x cols, column-major): one wave per column
__global__ __launch_bounds__(256) void gemv_t_kernel(double* __restrict__ y, const double* __restrict__ A, int64_t lda,
                                                     const double* __restrict__ x, int64_t rows, int64_t cols,
                                                     double alpha, double beta) {
    const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (j >= cols) return;
    const double* a = A + j * lda;
    double s = 0.0;
    for (int64_t r = lane; r < rows; r += 64) s += a[r] * x[r];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if (lane == 0) y[j] = beta == 0.0 ? alpha * s : alpha * s + beta * y[j];
}
// y[r] = alpha * dot(A[r, :], x) + beta * y[r]: 64 rows per workgroup, 4 column quarters, LDS reduction
__global__ __launch_bounds__(256) void gemv_n_kernel(double* __restrict__ y, const double* __restrict__ A, int64_t lda,
                                                     const double* __restrict__ x, int64_t rows, int64_t cols,
                                                     double alpha, double beta) {
    __shared__ double part[4][64];
    const int row = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * 64 + row;
    double s = 0.0;
    if (r < rows) {
        const int64_t c0 = cols * q / 4, c1 = cols * (q + 1) / 4;
        for (int64_t c = c0; c < c1; ++c) s += A[r + c * lda] * x[c];
    }
    part[q][row] = s;
    __syncthreads();
    if (q == 0 && r < rows) {
        const double t = (part[0][row] + part[1][row]) + (part[2][row] + part[3][row]);
        y[r] = beta == 0.0 ? alpha * t : alpha * t + beta * y[r];
    }
}
// y[i] = alpha * (Symmetric(H, :L) x)[i] + beta * y[i]: one wave per row
__global__ __launch_bounds__(256) void symv_l_kernel(double* __restrict__ y, const double* __restrict__ H, int64_t n,
                                                     const double* __restrict__ x, double alpha, double beta) {
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= n) return;
    double s = 0.0;
    for (int64_t j = lane; j < n; j += 64) s += (j <= i ? H[i + j * n] : H[j + i * n]) * x[j];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if (lane == 0) y[i] = beta == 0.0 ? alpha * s : alpha * s + beta * y[i];
}
// buffer = 0 ; buffer[ind_ineq] = D .* (wz + ws ./ Sigma_s)      (one thread per constraint)
__global__ void dc_condense_rhs_kernel(double* __restrict__ buffer, const double* __restrict__ dual,
                                       const double* __restrict__ ws, const double* __restrict__ D,
                                       const double* __restrict__ Ss, const int32_t* __restrict__ slot, int64_t m) {
    const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (r >= m) return;
    const int32_t i = slot[r];
    buffer[r] = i >= 0 ? D[i] * (dual[r] + ws[i] / Ss[i]) : 0.0;
}
// pd = [xx + wx ; wy]
__global__ void dc_pack_kernel(double* __restrict__ pd, const double* __restrict__ wx, const double* __restrict__ dual,
                               const int64_t* __restrict__ ind_eq, int64_t n, int64_t n_eq) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) pd[i] += wx[i];
    else if (i < n + n_eq) pd[i] = dual[ind_eq[i - n]];
}
// after dual = jac * wx:  wy = xy ; wz = wz .* D - buffer ; ws = (ws + wz) ./ Sigma_s
__global__ void dc_expand_kernel(double* __restrict__ dual, double* __restrict__ ws, const double* __restrict__ buffer,
                                 const double* __restrict__ D, const double* __restrict__ Ss,
                                 const int32_t* __restrict__ slot, const int32_t* __restrict__ eq_slot,
                                 const double* __restrict__ xy, int64_t m) {
    const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (r >= m) return;
    const int32_t i = slot[r];
    if (i >= 0) {
        const double z = dual[r] * D[i] - buffer[r];
        dual[r] = z;
        ws[i] = (ws[i] + z) / Ss[i];
    } else {
        dual[r] = xy[eq_slot[r]];
    }
}
// mul!: ws = beta ws - alpha xz ; wz -= alpha xs ; primal(w) += alpha reg .* primal(x) ; dual(w) += alpha du .* dual(x)
__global__ void dc_kktmul_diag_kernel(double* __restrict__ w, const double* __restrict__ x, const double* __restrict__ reg,
                                      const double* __restrict__ du, const int64_t* __restrict__ ind_ineq,
                                      const int32_t* __restrict__ slot, double alpha, double beta, int64_t n, int64_t ns,
                                      int64_t m) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) {
        w[i] += alpha * reg[i] * x[i];
    } else if (i < n + ns) {
        const int64_t c = i - n;
        w[i] = (beta * w[i] - alpha * x[n + ns + ind_ineq[c]]) + alpha * reg[i] * x[i];
    } else if (i < n + ns + m) {
        const int64_t r = i - n - ns;
        const int32_t c = slot[r];
        double v = w[i];
        if (c >= 0) v -= alpha * x[n + c];
        w[i] = v + alpha * du[r] * x[i];
    }
}
}  // namespace mnk

using namespace mnk;

struct mnk_dc_extra {
    DevBuf<int32_t> ineq_slot;   // constraint -> inequality slot or -1
    DevBuf<int32_t> eq_slot;     // constraint -> equality slot or -1
    // device-side solve_kkt! / mul!
    int64_t nlb = 0, nub = 0;
    DevBuf<int64_t> ind_lb, ind_ub;
    DevBuf<double> reg, l_diag, u_diag, l_lower, u_lower, buffer, pd, wdev, xdev, feed;
    bool have_bounds = false, have_terms = false, have_diag = false;
};
static mnk_dc_extra* extra_of(mnk_dc* dc) { return static_cast<mnk_dc_extra*>(dc->extra); }

extern "C" {

int mnk_dc_create(mnk_ctx* ctx, int condensed, int64_t n, int64_t m, int64_t ns, const int64_t* ind_ineq,
                  const int64_t* ind_eq, int index_base, mnk_dc** out) {
    MNK_REQUIRE(ctx && out, "mnk_dc_create: NULL argument");
    MNK_REQUIRE(n > 0 && m >= 0 && ns >= 0 && ns <= m, "mnk_dc_create: bad sizes");
    MNK_REQUIRE(ns == 0 || ind_ineq, "mnk_dc_create: ind_ineq is NULL");
    MNK_REQUIRE(m - ns == 0 || ind_eq || !condensed, "mnk_dc_create: ind_eq is NULL");
    MNK_HIP(hipSetDevice(ctx->device));
    mnk_dc* dc = new mnk_dc();
    dc->ctx = ctx;
    dc->condensed = condensed ? 1 : 0;
    dc->n = n; dc->m = m; dc->ns = ns; dc->n_eq = m - ns;
    dc->order = condensed ? n + dc->n_eq : n + ns + m;
    dc->ind_ineq.resize(ns);
    std::vector<int32_t> slot(std::max<int64_t>(m, 1), -1);
    for (int64_t k = 0; k < ns; ++k) {
        dc->ind_ineq[k] = ind_ineq[k] - index_base;
        MNK_REQUIRE(dc->ind_ineq[k] >= 0 && dc->ind_ineq[k] < m, "mnk_dc_create: ind_ineq out of range");
        slot[dc->ind_ineq[k]] = (int32_t)k;
    }
    dc->ind_eq.resize(dc->n_eq);
    if (condensed)
        for (int64_t k = 0; k < dc->n_eq; ++k) {
            dc->ind_eq[k] = ind_eq[k] - index_base;
            MNK_REQUIRE(dc->ind_eq[k] >= 0 && dc->ind_eq[k] < m, "mnk_dc_create: ind_eq out of range");
        }
    const int64_t ordpad = round_up(dc->order, PAD);
    dc->npad = round_up(n, 64);
    dc->kpad = round_up(std::max<int64_t>(ns, 1), 16);
    dc->ld_jis = dc->npad;
    hipStream_t s = ctx->stream;
    mnk_dc_extra* ex = new mnk_dc_extra();
    int rc = 0;
    rc |= dc->d_ind_ineq.upload(dc->ind_ineq, s);
    rc |= dc->d_ind_eq.upload(dc->ind_eq, s);
    rc |= ex->ineq_slot.upload(slot, s);
    {
        std::vector<int32_t> eslot(std::max<int64_t>(m, 1), -1);
        for (int64_t k = 0; k < (int64_t)dc->ind_eq.size(); ++k) eslot[dc->ind_eq[k]] = (int32_t)k;
        rc |= ex->eq_slot.upload(eslot, s);
    }
    rc |= dc->hess.alloc((size_t)n * n);
    rc |= dc->jac.alloc((size_t)std::max<int64_t>(m, 1) * n);
    rc |= dc->aug.alloc((size_t)ordpad * ordpad + SLACK);
    rc |= dc->pr_diag.alloc(n + ns);
    rc |= dc->du_diag.alloc(std::max<int64_t>(m, 1));
    rc |= dc->diag_buffer.alloc(std::max<int64_t>(ns, 1));
    if (condensed) rc |= dc->jis.alloc((size_t)dc->ld_jis * dc->kpad + SLACK);
    if (rc) { delete ex; delete dc; return -2; }
    MNK_HIP(hipMemsetAsync(dc->hess.p, 0, dc->hess.n * sizeof(double), s));
    MNK_HIP(hipMemsetAsync(dc->jac.p, 0, dc->jac.n * sizeof(double), s));
    MNK_HIP(hipMemsetAsync(dc->aug.p, 0, dc->aug.n * sizeof(double), s));
    dc->extra = ex;
    mnk_ctx_child_added(ctx);
    *out = dc;
    return 0;
}

int mnk_dc_destroy(mnk_dc* dc) {
    if (!dc) return 0;
    (void)hipSetDevice(dc->ctx->device);
    (void)mnk::stream_wait(dc->ctx->stream);
    delete extra_of(dc);
    dc->extra = nullptr;
    mnk_ctx* ctx = dc->ctx;
    delete dc;
    mnk_ctx_child_gone(ctx);
    return 0;
}

int64_t mnk_dc_order(mnk_dc* dc) { return dc ? dc->order : -1; }

static int copy_in_2d(mnk_ctx* ctx, double* dst, int64_t rows, int64_t cols, const double* src, int64_t ld, int loc) {
    if (rows == 0 || cols == 0) return 0;
    MNK_REQUIRE(ld >= rows, "leading dimension smaller than the number of rows");
    if (loc == MNK_DEVICE)
        MNK_HIP(hipMemcpy2DAsync(dst, rows * sizeof(double), src, ld * sizeof(double), rows * sizeof(double), cols,
                                 hipMemcpyDeviceToDevice, ctx->stream));
    else
        MNK_HIP(mnk::h2d_copy_2d(dst, rows * sizeof(double), src, ld * sizeof(double), rows * sizeof(double), cols, ctx->stream));
    return 0;
}

int mnk_dc_set_hess(mnk_dc* dc, const double* hess, int64_t ld, int loc) {
    MNK_REQUIRE(dc && hess, "mnk_dc_set_hess: NULL argument");
    MNK_HIP(hipSetDevice(dc->ctx->device));
    return copy_in_2d(dc->ctx, dc->hess.p, dc->n, dc->n, hess, ld, loc);
}

int mnk_dc_set_jac(mnk_dc* dc, const double* jac, int64_t ld, int loc) {
    MNK_REQUIRE(dc && (jac || dc->m == 0), "mnk_dc_set_jac: NULL argument");
    MNK_HIP(hipSetDevice(dc->ctx->device));
    return copy_in_2d(dc->ctx, dc->jac.p, dc->m, dc->n, jac, ld, loc);
}

int mnk_dc_build(mnk_dc* dc, const double* pr_diag, const double* du_diag, int loc) {
    MNK_REQUIRE(dc, "mnk_dc_build: NULL argument");
    MNK_HIP(hipSetDevice(dc->ctx->device));
    hipStream_t s = dc->ctx->stream;
    if (pr_diag == nullptr && du_diag == nullptr) {
        // the diagonals the handle keeps itself (mnk_dc_set_aug_diagonal / mnk_dc_regularize_diagonal)
        mnk_dc_extra* ex0 = extra_of(dc);
        MNK_REQUIRE(ex0 != nullptr && ex0->have_diag, "mnk_dc_build: no diagonals given and mnk_dc_set_aug_diagonal was not called");
    } else {
        MNK_REQUIRE(pr_diag && (du_diag || dc->m == 0), "mnk_dc_build: NULL argument");
        if (loc == MNK_DEVICE) {
            MNK_HIP(hipMemcpyAsync(dc->pr_diag.p, pr_diag, (dc->n + dc->ns) * sizeof(double), hipMemcpyDeviceToDevice, s));
            if (dc->m > 0) MNK_HIP(hipMemcpyAsync(dc->du_diag.p, du_diag, dc->m * sizeof(double), hipMemcpyDeviceToDevice, s));
        } else {
            MNK_HIP(mnk::h2d_copy(dc->pr_diag.p, pr_diag, (dc->n + dc->ns) * sizeof(double), s));
            if (dc->m > 0) MNK_HIP(mnk::h2d_copy(dc->du_diag.p, du_diag, dc->m * sizeof(double), s));
        }
    }
    const int64_t ordpad = round_up(dc->order, PAD);
    const int64_t ldk = ordpad;
    dim3 egrid((unsigned)ordpad, (unsigned)((ordpad + 255) / 256));
    if (!dc->condensed) {
        mnk_dc_extra* ex = extra_of(dc);
        MNK_REQUIRE(ex != nullptr, "mnk_dc_build: unknown handle");
        hipLaunchKernelGGL(dense_aug_kernel, egrid, dim3(256), 0, s, dc->aug.p, ldk, ordpad, dc->hess.p, dc->jac.p,
                           dc->m, dc->n, dc->ns, dc->d_ind_ineq.p, ex->ineq_slot.p, dc->pr_diag.p, dc->du_diag.p);
        MNK_HIP(hipGetLastError());
        dc->mirror_pending = false;
        return 0;
    }
    dim3 tgrid((unsigned)(dc->kpad + 31) / 32, (unsigned)(dc->npad + 31) / 32);
    hipLaunchKernelGGL(scale_transpose_kernel, tgrid, dim3(256), 0, s, dc->jis.p, dc->ld_jis, dc->npad, dc->kpad,
                       dc->jac.p, dc->m, dc->n, dc->ns, dc->d_ind_ineq.p, dc->diag_buffer.p, dc->pr_diag.p + dc->n, dc->du_diag.p);
    hipLaunchKernelGGL(init_condensed_kernel, egrid, dim3(256), 0, s, dc->aug.p, ldk, ordpad, dc->hess.p, dc->jac.p,
                       dc->m, dc->n, dc->n_eq, dc->d_ind_eq.p, dc->pr_diag.p, dc->du_diag.p);
    MNK_HIP(hipGetLastError());
    if (dc->ns > 0) {
        int rc = launch_gemm_nt(s, 4, dc->npad, dc->npad, dc->kpad, dc->jis.p, dc->ld_jis, dc->jis.p, dc->ld_jis,
                                dc->aug.p, ldk, nullptr, nullptr, 0, nullptr);
        if (rc) return rc;
    }
    // (the upper triangle: nothing on the path reads it -- the factorization copies the lower one -- so it is written when
    // somebody asks for the matrix, mnk_dc_get_aug)
    dc->mirror_pending = true;
    MNK_HIP(hipGetLastError());
    return 0;
}

int mnk_dc_get_aug(mnk_dc* dc, double* out, int loc) {
    MNK_REQUIRE(dc && out, "mnk_dc_get_aug: NULL argument");
    MNK_HIP(hipSetDevice(dc->ctx->device));
    const int64_t ldk = round_up(dc->order, PAD);
    if (dc->mirror_pending) {
        dim3 mgrid((unsigned)((dc->order + 31) / 32), (unsigned)((dc->order + 31) / 32));
        hipLaunchKernelGGL(mirror_kernel, mgrid, dim3(256), 0, dc->ctx->stream, dc->aug.p, ldk, dc->order);
        MNK_HIP(hipGetLastError());
        dc->mirror_pending = false;
    }
    if (loc == MNK_DEVICE)
        MNK_HIP(hipMemcpy2DAsync(out, dc->order * sizeof(double), dc->aug.p, ldk * sizeof(double), dc->order * sizeof(double),
                                 dc->order, hipMemcpyDeviceToDevice, dc->ctx->stream));
    else
        MNK_HIP(mnk::d2h_copy_2d(out, dc->order * sizeof(double), dc->aug.p, ldk * sizeof(double), dc->order * sizeof(double),
                                 dc->order, dc->ctx->stream));
    return 0;
}


// ---- device-side solve_kkt! / mul! -------------------------------------------------------------------
int mnk_dc_set_bounds(mnk_dc* dc, int64_t nlb, const int64_t* ind_lb, int64_t nub, const int64_t* ind_ub, int index_base) {
    MNK_REQUIRE(dc && nlb >= 0 && nub >= 0 && (nlb == 0 || ind_lb) && (nub == 0 || ind_ub), "mnk_dc_set_bounds: bad argument");
    MNK_HIP(hipSetDevice(dc->ctx->device));
    mnk_dc_extra* ex = extra_of(dc);
    MNK_REQUIRE(ex != nullptr, "mnk_dc_set_bounds: unknown handle");
    const int64_t np = dc->n + dc->ns;
    std::vector<int64_t> lb(nlb), ub(nub);
    for (int64_t i = 0; i < nlb; ++i) {
        lb[i] = ind_lb[i] - index_base;
        MNK_REQUIRE(lb[i] >= 0 && lb[i] < np, "mnk_dc_set_bounds: lower-bound index out of range");
    }
    for (int64_t i = 0; i < nub; ++i) {
        ub[i] = ind_ub[i] - index_base;
        MNK_REQUIRE(ub[i] >= 0 && ub[i] < np, "mnk_dc_set_bounds: upper-bound index out of range");
    }
    hipStream_t s = dc->ctx->stream;
    const size_t lw = (size_t)(np + dc->m + nlb + nub);
    int rc = ex->ind_lb.upload(lb, s);
    rc |= ex->ind_ub.upload(ub, s);
    rc |= ex->reg.alloc(np);
    rc |= ex->l_diag.alloc(nlb);
    rc |= ex->l_lower.alloc(nlb);
    rc |= ex->u_diag.alloc(nub);
    rc |= ex->u_lower.alloc(nub);
    rc |= ex->buffer.alloc(dc->m);
    rc |= ex->pd.alloc(dc->order);
    rc |= ex->wdev.alloc(lw);
    rc |= ex->xdev.alloc(lw);
    if (rc) return rc;
    ex->nlb = nlb;
    ex->nub = nub;
    ex->have_bounds = true;
    return 0;
}

static int dc_diag_view(mnk_dc* dc, AugDiagView& v, const char* who) {
    if (!dc) { set_error("%s: NULL argument", who); return -1; }
    MNK_HIP(hipSetDevice(dc->ctx->device));
    mnk_dc_extra* ex = extra_of(dc);
    if (!(ex != nullptr && ex->have_bounds)) { set_error("%s: call mnk_dc_set_bounds first", who); return -1; }
    v = AugDiagView{dc->ctx, dc->n + dc->ns, dc->m, ex->nlb, ex->nub, ex->reg.p, dc->pr_diag.p, dc->du_diag.p, ex->l_diag.p,
                    ex->u_diag.p, ex->l_lower.p, ex->u_lower.p, ex->ind_lb.p, ex->ind_ub.p, &ex->feed};
    return 0;
}

int mnk_dc_set_aug_diagonal(mnk_dc* dc, const double* x, const double* xl, const double* xu, const double* zl,
                            const double* zu, double primal_reg, double dual_reg, int loc) {
    AugDiagView v;
    int rc = dc_diag_view(dc, v, "mnk_dc_set_aug_diagonal");
    if (rc) return rc;
    MNK_REQUIRE(x && xl && xu && zl && zu, "mnk_dc_set_aug_diagonal: NULL vector");
    rc = kkt_set_aug_diagonal(v, x, xl, xu, zl, zu, primal_reg, dual_reg, loc);
    if (rc) return rc;
    extra_of(dc)->have_terms = extra_of(dc)->have_diag = true;
    return 0;
}

// set_aug_RR!(kkt, solver, RR) (reference src/IPM/kernels.jl:72-87): device-resident vectors only
int mnk_dc_set_aug_RR(mnk_dc* dc, const double* x, const double* xl, const double* xu, const double* zl, const double* zu,
                      const double* D_R, const double* pp, const double* zp, const double* nn, const double* zn, double zeta,
                      double primal_reg, double dual_reg) {
    AugDiagView v;
    int rc = dc_diag_view(dc, v, "mnk_dc_set_aug_RR");
    if (rc) return rc;
    MNK_REQUIRE(x && xl && xu && zl && zu && D_R && (v.ndu == 0 || (pp && zp && nn && zn)), "mnk_dc_set_aug_RR: NULL vector");
    rc = kkt_set_aug_RR(v, x, xl, xu, zl, zu, D_R, pp, zp, nn, zn, zeta, primal_reg, dual_reg);
    if (rc) return rc;
    extra_of(dc)->have_terms = extra_of(dc)->have_diag = true;
    return 0;
}

int mnk_dc_regularize_diagonal(mnk_dc* dc, double primal, double dual) {
    AugDiagView v;
    int rc = dc_diag_view(dc, v, "mnk_dc_regularize_diagonal");
    if (rc) return rc;
    MNK_REQUIRE(extra_of(dc)->have_diag, "mnk_dc_regularize_diagonal: call mnk_dc_set_aug_diagonal first");
    return kkt_regularize_diagonal(v, primal, dual);
}

int mnk_dc_get_diagonals(mnk_dc* dc, double* pr_diag, double* du_diag, double* reg, double* l_diag, double* u_diag,
                         double* l_lower, double* u_lower) {
    AugDiagView v;
    int rc = dc_diag_view(dc, v, "mnk_dc_get_diagonals");
    if (rc) return rc;
    return kkt_get_diagonals(v, pr_diag, du_diag, reg, l_diag, u_diag, l_lower, u_lower);
}

int mnk_dc_set_barrier_terms(mnk_dc* dc, const double* reg, const double* l_diag, const double* u_diag,
                             const double* l_lower, const double* u_lower, int loc) {
    MNK_REQUIRE(dc, "mnk_dc_set_barrier_terms: NULL argument");
    MNK_HIP(hipSetDevice(dc->ctx->device));
    mnk_dc_extra* ex = extra_of(dc);
    MNK_REQUIRE(ex != nullptr && ex->have_bounds, "mnk_dc_set_barrier_terms: call mnk_dc_set_bounds first");
    hipStream_t s = dc->ctx->stream;
    auto put = [&](double* dst, const double* src, int64_t cnt) -> int {
        if (cnt <= 0) return 0;
        MNK_REQUIRE(src != nullptr, "mnk_dc_set_barrier_terms: NULL vector");
        if (loc == MNK_DEVICE) MNK_HIP(hipMemcpyAsync(dst, src, cnt * sizeof(double), hipMemcpyDeviceToDevice, s));
        else MNK_HIP(mnk::h2d_copy(dst, src, cnt * sizeof(double), s));
        return 0;
    };
    int rc = put(ex->reg.p, reg, dc->n + dc->ns);
    rc |= put(ex->l_diag.p, l_diag, ex->nlb);
    rc |= put(ex->u_diag.p, u_diag, ex->nub);
    rc |= put(ex->l_lower.p, l_lower, ex->nlb);
    rc |= put(ex->u_lower.p, u_lower, ex->nub);
    if (rc) return rc;
    if (loc != MNK_DEVICE) MNK_HIP(mnk::stream_wait(s));
    ex->have_terms = true;
    return 0;
}

#define MNK_GRID(cnt) dim3((unsigned)(((cnt) + 255) / 256)), dim3(256), 0, s

int mnk_dc_solve_kkt(mnk_dc* dc, mnk_ls* ls, double* w, int loc) {
    MNK_REQUIRE(dc && ls && w, "mnk_dc_solve_kkt: NULL argument");
    MNK_REQUIRE(ls->ctx == dc->ctx && ls->N == dc->order, "mnk_dc_solve_kkt: the solver does not belong to this system");
    MNK_HIP(hipSetDevice(dc->ctx->device));
    mnk_dc_extra* ex = extra_of(dc);
    MNK_REQUIRE(ex != nullptr && ex->have_bounds && ex->have_terms,
                "mnk_dc_solve_kkt: call mnk_dc_set_bounds / mnk_dc_set_barrier_terms / mnk_dc_build first");
    hipStream_t s = dc->ctx->stream;
    const int64_t n = dc->n, ns = dc->ns, m = dc->m, n_eq = dc->n_eq, nlb = ex->nlb, nub = ex->nub;
    const int64_t lw = n + ns + m + nlb + nub;
    // see mnk_sc_solve_kkt: an aborted persistent solve is detected before the copy-back and redone stepwise
    for (int attempt = 0; attempt < 2; ++attempt) {
        double* d = w;
        if (loc != MNK_DEVICE) {
            d = ex->wdev.p;
            MNK_HIP(mnk::h2d_copy(d, w, lw * sizeof(double), s));
        }
        double *ws = d + n, *dual = d + n + ns, *wl = dual + m, *wu = wl + nlb;
        if (nlb > 0) hipLaunchKernelGGL(reduce_rhs_kernel, MNK_GRID(nlb), d, ex->ind_lb.p, wl, ex->l_diag.p, nlb);
        if (nub > 0) hipLaunchKernelGGL(reduce_rhs_kernel, MNK_GRID(nub), d, ex->ind_ub.p, wu, ex->u_diag.p, nub);
        int rc = 0;
        if (!dc->condensed) {
            // reduced solve (reference src/IPM/factorization.jl:41-46): the solver acts on primal_dual(w) in place
            rc = mnk_ls_solve(ls, d, 1, dc->order, MNK_DEVICE);
            if (rc) return rc;
        } else {
            const double* Ss = dc->pr_diag.p + n;
            if (m > 0) {
                hipLaunchKernelGGL(dc_condense_rhs_kernel, MNK_GRID(m), ex->buffer.p, dual, ws, dc->diag_buffer.p, Ss,
                                   ex->ineq_slot.p, m);
                hipLaunchKernelGGL(gemv_t_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, ex->pd.p, dc->jac.p, m,
                                   ex->buffer.p, m, n, 1.0, 0.0);  // xx = jac' * buffer
            } else {
                MNK_HIP(hipMemsetAsync(ex->pd.p, 0, n * sizeof(double), s));
            }
            hipLaunchKernelGGL(dc_pack_kernel, MNK_GRID(n + n_eq), ex->pd.p, d, dual, dc->d_ind_eq.p, n, n_eq);
            rc = mnk_ls_solve(ls, ex->pd.p, 1, dc->order, MNK_DEVICE);
            if (rc) return rc;
            MNK_HIP(hipMemcpyAsync(d, ex->pd.p, n * sizeof(double), hipMemcpyDeviceToDevice, s));  // wx = xx
            if (m > 0) {
                hipLaunchKernelGGL(gemv_n_kernel, dim3((unsigned)((m + 63) / 64)), dim3(256), 0, s, dual, dc->jac.p, m, d, m, n,
                                   1.0, 0.0);  // dual(w) = jac * wx
                hipLaunchKernelGGL(dc_expand_kernel, MNK_GRID(m), dual, ws, ex->buffer.p, dc->diag_buffer.p, Ss,
                                   ex->ineq_slot.p, ex->eq_slot.p, ex->pd.p + n, m);
            }
        }
        if (nlb > 0) hipLaunchKernelGGL(finish_aug_kernel, MNK_GRID(nlb), wl, d, ex->ind_lb.p, ex->l_lower.p, ex->l_diag.p, nlb, 0);
        if (nub > 0) hipLaunchKernelGGL(finish_aug_kernel, MNK_GRID(nub), wu, d, ex->ind_ub.p, ex->u_lower.p, ex->u_diag.p, nub, 1);
        MNK_HIP(hipGetLastError());
        if (loc == MNK_DEVICE) break;
        MNK_HIP(mnk::stream_wait(s));
        if (attempt == 0 && mnk_ls_take_solve_abort(ls)) continue;
        MNK_HIP(mnk::d2h_copy(w, d, lw * sizeof(double), s));
        break;
    }
    return 0;
}

int mnk_dc_mul(mnk_dc* dc, double* w, const double* x, double alpha, double beta, int loc) {
    MNK_REQUIRE(dc && w && x, "mnk_dc_mul: NULL argument");
    MNK_HIP(hipSetDevice(dc->ctx->device));
    mnk_dc_extra* ex = extra_of(dc);
    MNK_REQUIRE(ex != nullptr && ex->have_bounds && ex->have_terms,
                "mnk_dc_mul: call mnk_dc_set_bounds / mnk_dc_set_barrier_terms first");
    hipStream_t s = dc->ctx->stream;
    const int64_t n = dc->n, ns = dc->ns, m = dc->m, nlb = ex->nlb, nub = ex->nub;
    const int64_t lw = n + ns + m + nlb + nub;
    double* dw = w;
    const double* dx = x;
    if (loc != MNK_DEVICE) {
        dw = ex->wdev.p;
        MNK_HIP(mnk::h2d_copy(ex->wdev.p, w, lw * sizeof(double), s));
        MNK_HIP(mnk::h2d_copy(ex->xdev.p, x, lw * sizeof(double), s));
        dx = ex->xdev.p;
    }
    // wx = alpha Sym(H) xx + beta wx ; wx += alpha jac' dual(x) ; dual(w) = alpha jac xx + beta dual(w)
    hipLaunchKernelGGL(symv_l_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, dw, dc->hess.p, n, dx, alpha, beta);
    if (m > 0) {
        hipLaunchKernelGGL(gemv_t_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, dw, dc->jac.p, m, dx + n + ns, m, n,
                           alpha, 1.0);
        hipLaunchKernelGGL(gemv_n_kernel, dim3((unsigned)((m + 63) / 64)), dim3(256), 0, s, dw + n + ns, dc->jac.p, m, dx, m,
                           n, alpha, beta);
    }
    hipLaunchKernelGGL(dc_kktmul_diag_kernel, MNK_GRID(n + ns + m), dw, dx, ex->reg.p, dc->du_diag.p, dc->d_ind_ineq.p,
                       ex->ineq_slot.p, alpha, beta, n, ns, m);
    if (nlb > 0)
        hipLaunchKernelGGL(kktmul_bound_kernel, MNK_GRID(nlb), dw, dw + n + ns + m, dx, dx + n + ns + m, ex->ind_lb.p,
                           ex->l_lower.p, ex->l_diag.p, alpha, beta, nlb, 0);
    if (nub > 0)
        hipLaunchKernelGGL(kktmul_bound_kernel, MNK_GRID(nub), dw, dw + n + ns + m + nlb, dx, dx + n + ns + m + nlb,
                           ex->ind_ub.p, ex->u_lower.p, ex->u_diag.p, alpha, beta, nub, 1);
    MNK_HIP(hipGetLastError());
    if (loc != MNK_DEVICE) {
        MNK_HIP(mnk::d2h_copy(w, dw, lw * sizeof(double), s));
    }
    return 0;
}
#undef MNK_GRID

}  // extern "C"
