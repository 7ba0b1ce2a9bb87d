// SparseCondensedKKTSystem on the device (mnk_sc_*).
//
// Host side (C++, integer work, once per problem): force_lower_triangular!,
// coo_to_csc + get_mapping (reference src/matrixtools.jl:55-95,129-137) and
// build_condensed_aug_symbolic (reference src/KKT/Sparse/condensed.jl:158-301).
// Device side (per IPM iteration, HBM-bound gather kernels):
//   compress_*      = transfer! (reference src/matrixtools.jl:79-88; device twin
//                     lib/MadNLPGPU/src/KKT/kernels_sparse.jl:161-167): segmented COO->CSC sum
//   build_kkt!      = diag_buffer + _build_condensed_aug_coord! (reference
//                     src/KKT/Sparse/condensed.jl:328-366; device twins kernels_sparse.jl:127-139)
// Every sum is accumulated in the reference's order (see DESIGN.md "bit-exact assembly"),
// and FMA contraction is disabled so results equal the CPU restatement bit for bit.
#pragma clang fp contract(off)

#include <algorithm>
#include <numeric>

#include "ls.h"
#include "kkt_vec.h"

namespace mnk {

// dst[s] = sum_{k in [ptr[s], ptr[s+1])} src[idx[k]]   (idx ascending inside a segment)
__global__ void segsum_kernel(double* __restrict__ dst, const double* __restrict__ src,
                              const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx, int64_t nseg) {
    const int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (s >= nseg) return;
    double acc = 0.0;
    for (int32_t k = ptr[s]; k < ptr[s + 1]; ++k) acc += src[idx[k]];
    dst[s] = acc;
}

// build_kkt! in ONE launch (round 3: two copies + diag_buffer_kernel + condense_kernel, host-enqueue bound at ~0.1 ms):
//   thread i < m:      diag_buffer[i] = Sigma_s ./ (1 - Sigma_d .* Sigma_s)   (reference condensed.jl:364), kept for solve_kkt!
//   thread i < n + m:  the handle's own copies of pr_diag / du_diag (the device-side solve_kkt! / mul! read them later)
//   thread s < nslot:  K.nz[s] = ((sum_h H.nz) + pr_diag) + sum_j D[c]*Jt[k]*Jt[l]  with D[c] evaluated in place by the
//                      same expression (one rounding each, so the bits equal the two-kernel version's)
__global__ void condense_kernel(double* __restrict__ K, const double* __restrict__ Hnz,
                                const double* __restrict__ pr_diag, const double* __restrict__ du,
                                double* __restrict__ Dout, double* __restrict__ pr_keep, double* __restrict__ du_keep,
                                const double* __restrict__ Jt, const int32_t* __restrict__ hptr,
                                const int32_t* __restrict__ hsrc, const int32_t* __restrict__ dsrc,
                                const int32_t* __restrict__ jptr, const int32_t* __restrict__ jc,
                                const int32_t* __restrict__ jk, const int32_t* __restrict__ jl, int64_t nslot, int64_t n,
                                int64_t m) {
    const int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const double* pr_s = pr_diag + n;
    if (s < m) {
        Dout[s] = pr_s[s] / (1.0 - du[s] * pr_s[s]);
        if (du_keep != nullptr) du_keep[s] = du[s];
    }
    if (pr_keep != nullptr && s < n + m) pr_keep[s] = pr_diag[s];
    if (s >= nslot) return;
    double acc = 0.0;
    for (int32_t k = hptr[s]; k < hptr[s + 1]; ++k) acc += Hnz[hsrc[k]];
    const int32_t d = dsrc[s];
    if (d >= 0) acc += pr_diag[d];
    for (int32_t k = jptr[s]; k < jptr[s + 1]; ++k) {
        const int32_t c = jc[k];
        const double Dc = pr_s[c] / (1.0 - du[c] * pr_s[c]);
        acc += (Dc * Jt[jk[k]]) * Jt[jl[k]];
    }
    K[s] = acc;
}

// y[i] = alpha * sum_k val[perm[k]] * x[idx[k]] + beta * y[i]
__global__ void gather_spmv_kernel(double* __restrict__ y, const double* __restrict__ x,
                                   const double* __restrict__ val, const int32_t* __restrict__ ptr,
                                   const int32_t* __restrict__ idx, const int32_t* __restrict__ perm,
                                   double alpha, double beta, int64_t nrow) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= nrow) return;
    double acc = 0.0;
    for (int32_t k = ptr[i]; k < ptr[i + 1]; ++k) acc += val[perm ? perm[k] : k] * x[idx[k]];
    y[i] = beta == 0.0 ? alpha * acc : alpha * acc + beta * y[i];
}

// buffer = diag_buffer .* (wz .+ ws ./ Sigma_s)
__global__ void condense_rhs_kernel(double* __restrict__ buffer, const double* __restrict__ D, const double* __restrict__ ws,
                                    const double* __restrict__ wz, const double* __restrict__ Ss, int64_t m) {
    const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (c < m) buffer[c] = D[c] * (wz[c] + ws[c] / Ss[c]);
}
// wz = -buffer + diag_buffer .* buffer2 ; ws = (ws + wz) ./ Sigma_s
__global__ void expand_sol_kernel(double* __restrict__ ws, double* __restrict__ wz, const double* __restrict__ buffer,
                                  const double* __restrict__ buffer2, const double* __restrict__ D,
                                  const double* __restrict__ Ss, int64_t m) {
    const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (c < m) {
        const double z = -buffer[c] + D[c] * buffer2[c];
        wz[c] = z;
        ws[c] = (ws[c] + z) / Ss[c];
    }
}
// mul!, slack/dual coupling: wz -= alpha xs ; ws = beta ws - alpha xz ; then _kktmul!'s diagonal part
//   primal(w) += alpha reg .* primal(x) ; dual(w) += alpha du_diag .* dual(x)
__global__ void kktmul_diag_kernel(double* __restrict__ w, const double* __restrict__ x, const double* __restrict__ reg,
                                   const double* __restrict__ du, double alpha, double beta, int64_t n, int64_t m) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) {
        w[i] += alpha * reg[i] * x[i];
    } else if (i < n + m) {  // slack block
        const int64_t c = i - n;
        w[i] = (beta * w[i] - alpha * x[n + m + c]) + alpha * reg[i] * x[i];
    } else if (i < n + 2 * m) {  // dual block (already holds alpha Jt' xx + beta wz)
        const int64_t c = i - n - m;
        w[i] = (w[i] - alpha * x[n + c]) + alpha * du[c] * x[i];
    }
}

// ---- host symbolic ---------------------------------------------------------------------
struct CscPattern {
    std::vector<int32_t> colptr, rowval;
    std::vector<int64_t> map;  // COO entry -> slot
};

// reference coo_to_csc + get_mapping (src/matrixtools.jl:55-95): structure of sparse(I,J,1).
static CscPattern coo_to_csc(int64_t nrow, int64_t ncol, const std::vector<int32_t>& I,
                             const std::vector<int32_t>& J) {
    const int64_t nnz = (int64_t)I.size();
    std::vector<int64_t> order(nnz);
    std::iota(order.begin(), order.end(), 0);
    auto key = [&](int64_t k) { return (int64_t)J[k] * nrow + I[k]; };
    std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {
        const int64_t ka = key(a), kb = key(b);
        return ka != kb ? ka < kb : a < b;
    });
    CscPattern out;
    out.colptr.assign(ncol + 1, 0);
    out.map.assign(nnz, 0);
    int64_t slot = -1, prev = -1;
    for (int64_t t = 0; t < nnz; ++t) {
        const int64_t k = order[t], kk = key(k);
        if (kk != prev) {
            ++slot;
            prev = kk;
            out.rowval.push_back(I[k]);
            out.colptr[J[k] + 1]++;
        }
        out.map[k] = slot;
    }
    for (int64_t c = 0; c < ncol; ++c) out.colptr[c + 1] += out.colptr[c];
    return out;
}

// group COO sources by destination slot: ptr[nslot+1], src (ascending inside each slot)
static void group_by_slot(const std::vector<int64_t>& map, int64_t nslot, std::vector<int32_t>& ptr,
                          std::vector<int32_t>& src) {
    ptr.assign(nslot + 1, 0);
    for (int64_t v : map) ptr[v + 1]++;
    for (int64_t s = 0; s < nslot; ++s) ptr[s + 1] += ptr[s];
    src.assign(map.size(), 0);
    std::vector<int32_t> cur(ptr.begin(), ptr.end() - 1);
    for (int64_t k = 0; k < (int64_t)map.size(); ++k) src[cur[map[k]]++] = (int32_t)k;
}

static std::vector<int32_t> col_index(const std::vector<int32_t>& colptr) {
    std::vector<int32_t> ci(colptr.back());
    for (int64_t c = 0; c + 1 < (int64_t)colptr.size(); ++c)
        for (int32_t k = colptr[c]; k < colptr[c + 1]; ++k) ci[k] = (int32_t)c;
    return ci;
}

}  // namespace mnk

using namespace mnk;

// extra device structures for spmv (kept out of ls.h: only this unit uses them)
struct mnk_sc_spmv_data {
    DevBuf<int32_t> jtr_ptr, jtr_idx, jtr_perm;   // CSR of Jt (rows = variables)
    DevBuf<int32_t> hs_ptr, hs_idx, hs_perm;      // rows of Symmetric(hess_com, :L)
    // device-side solve_kkt! / mul!: bound structure, barrier terms, work vectors
    int64_t nlb = 0, nub = 0;
    DevBuf<int64_t> ind_lb, ind_ub;               // positions in the primal block [0, n+m)
    DevBuf<double> reg, l_diag, u_diag, l_lower, u_lower;
    DevBuf<double> buffer, buffer2, wdev, xdev;   // m, m, len(w), len(w)
    DevBuf<double> feed;                          // staging of host iterates for mnk_sc_set_aug_diagonal
    bool have_bounds = false, have_terms = false, have_diag = false;
    DevBuf<double> saved_diag;                    // reg | pr_diag | du_diag of mnk_sc_save_diagonals
    bool have_saved_diag = false;
};
static mnk_sc_spmv_data* spmv_of(mnk_sc* sc) { return static_cast<mnk_sc_spmv_data*>(sc->extra); }

extern "C" {

int mnk_sc_create(mnk_ctx* ctx, int64_t n, int64_t m, int64_t nnzj, const int32_t* jac_I, const int32_t* jac_J,
                  int64_t nnzh, const int32_t* hess_I, const int32_t* hess_J, int index_base, mnk_sc** out) {
    MNK_REQUIRE(out, "mnk_sc_create: NULL argument");
    MNK_REQUIRE(n > 0 && m >= 0 && nnzj >= 0 && nnzh >= 0, "mnk_sc_create: bad sizes");
    MNK_REQUIRE(index_base == 0 || index_base == 1, "mnk_sc_create: index_base must be 0 or 1");
    MNK_REQUIRE((nnzj == 0 || (jac_I && jac_J)) && (nnzh == 0 || (hess_I && hess_J)), "mnk_sc_create: NULL pattern");
    // ctx == NULL: host-only handle (symbolic analysis only; structure getters work, no device state)
    if (ctx) MNK_HIP(hipSetDevice(ctx->device));
    // jt_coo = J': rows = variable (jac_J), cols = constraint (jac_I)   (condensed.jl:104-109)
    std::vector<int32_t> jtI(nnzj), jtJ(nnzj), hI(nnzh), hJ(nnzh);
    for (int64_t k = 0; k < nnzj; ++k) {
        jtI[k] = jac_J[k] - index_base;
        jtJ[k] = jac_I[k] - index_base;
        MNK_REQUIRE(jtI[k] >= 0 && jtI[k] < n && jtJ[k] >= 0 && jtJ[k] < m, "mnk_sc_create: Jacobian index out of range");
    }
    for (int64_t k = 0; k < nnzh; ++k) {
        int32_t i = hess_I[k] - index_base, j = hess_J[k] - index_base;
        MNK_REQUIRE(i >= 0 && i < n && j >= 0 && j < n, "mnk_sc_create: Hessian index out of range");
        if (j > i) std::swap(i, j);  // force_lower_triangular! (matrixtools.jl:129-137)
        hI[k] = i;
        hJ[k] = j;
    }
    mnk_sc* sc = new mnk_sc();
    sc->ctx = ctx;
    sc->n = n; sc->m = m; sc->nnzj = nnzj; sc->nnzh = nnzh;
    CscPattern jt = coo_to_csc(n, m, jtI, jtJ);
    CscPattern hh = coo_to_csc(n, n, hI, hJ);
    sc->jt_colptr = jt.colptr; sc->jt_rowval = jt.rowval; sc->jt_map = jt.map;
    sc->h_colptr = hh.colptr; sc->h_rowval = hh.rowval; sc->h_map = hh.map;
    sc->nnz_jt = (int64_t)jt.rowval.size();
    sc->nnz_hess = (int64_t)hh.rowval.size();

    // ---- build_condensed_aug_symbolic (condensed.jl:201-301) ----
    int64_t L = 0;  // _sym_length (condensed.jl:158-165)
    for (int64_t c = 0; c < m; ++c) {
        const int64_t k = jt.colptr[c + 1] - jt.colptr[c];
        L += k * (k + 1) / 2;
    }
    const int64_t T = n + sc->nnz_hess + L;
    MNK_REQUIRE(T < (int64_t)2000000000, "mnk_sc_create: symbolic structure exceeds int32 range");
    std::vector<int64_t> key(T);
    std::vector<int8_t> kind(T);          // -1 diag, 0 hess, 1 jt
    std::vector<int32_t> sa(T), sb(T), scc(T);
    int64_t t = 0;
    for (int64_t i = 0; i < n; ++i, ++t) { key[t] = i * n + i; kind[t] = -1; sa[t] = (int32_t)i; sb[t] = 0; scc[t] = 0; }
    {
        std::vector<int32_t> hcol = col_index(hh.colptr);
        for (int64_t k = 0; k < sc->nnz_hess; ++k, ++t) {
            key[t] = (int64_t)hcol[k] * n + hh.rowval[k]; kind[t] = 0; sa[t] = (int32_t)k; sb[t] = 0; scc[t] = 0;
        }
    }
    for (int64_t c = 0; c < m; ++c)
        for (int32_t j = jt.colptr[c]; j < jt.colptr[c + 1]; ++j)
            for (int32_t k = j; k < jt.colptr[c + 1]; ++k, ++t) {
                const int64_t c1 = jt.rowval[j], c2 = jt.rowval[k];  // c2 >= c1: rows sorted in a column
                key[t] = c1 * n + c2;  // (row, col) = (c2, c1), sorted by (col, row)
                kind[t] = 1; sa[t] = (int32_t)c; sb[t] = j; scc[t] = k;
            }
    std::vector<int64_t> ord(T);
    std::iota(ord.begin(), ord.end(), 0);
    std::stable_sort(ord.begin(), ord.end(), [&](int64_t a, int64_t b) { return key[a] < key[b]; });

    std::vector<int32_t> a_hptr, a_hsrc, a_dsrc, a_jptr, a_jc, a_jk, a_jl, a_row, a_col;
    sc->aug_colptr.assign(n + 1, 0);
    int64_t slot = -1, prev = -1;
    for (int64_t q = 0; q < T; ++q) {
        const int64_t e = ord[q];
        if (key[e] != prev) {
            prev = key[e];
            ++slot;
            const int32_t col = (int32_t)(key[e] / n), row = (int32_t)(key[e] % n);
            a_row.push_back(row); a_col.push_back(col);
            sc->aug_rowval.push_back(row);
            sc->aug_colptr[col + 1]++;
            a_hptr.push_back((int32_t)a_hsrc.size());
            a_jptr.push_back((int32_t)a_jc.size());
            a_dsrc.push_back(-1);
        }
        if (kind[e] == -1) {
            a_dsrc[slot] = sa[e];
            sc->d_dst.push_back((int32_t)slot); sc->d_src.push_back(sa[e]);
        } else if (kind[e] == 0) {
            a_hsrc.push_back(sa[e]);
            sc->hp_dst.push_back((int32_t)slot); sc->hp_src.push_back(sa[e]);
        } else {
            a_jc.push_back(sa[e]); a_jk.push_back(sb[e]); a_jl.push_back(scc[e]);
            sc->j_dst.push_back((int32_t)slot); sc->j_c.push_back(sa[e]); sc->j_k.push_back(sb[e]); sc->j_l.push_back(scc[e]);
        }
    }
    a_hptr.push_back((int32_t)a_hsrc.size());
    a_jptr.push_back((int32_t)a_jc.size());
    for (int64_t c = 0; c < n; ++c) sc->aug_colptr[c + 1] += sc->aug_colptr[c];
    sc->nnz_aug = slot + 1;
    sc->len_jptr = L;

    if (!ctx) { *out = sc; return 0; }
    // ---- upload ----
    hipStream_t s = ctx->stream;
    std::vector<int32_t> ptr, src;
    int rc = 0;
    group_by_slot(jt.map, sc->nnz_jt, ptr, src);
    rc |= sc->jt_seg_ptr.upload(ptr, s); rc |= sc->jt_seg_src.upload(src, s);
    group_by_slot(hh.map, sc->nnz_hess, ptr, src);
    rc |= sc->h_seg_ptr.upload(ptr, s); rc |= sc->h_seg_src.upload(src, s);
    rc |= sc->aug_hptr.upload(a_hptr, s); rc |= sc->aug_hsrc.upload(a_hsrc, s);
    rc |= sc->aug_dsrc.upload(a_dsrc, s);
    rc |= sc->aug_jptr.upload(a_jptr, s); rc |= sc->aug_jc.upload(a_jc, s);
    rc |= sc->aug_jk.upload(a_jk, s); rc |= sc->aug_jl.upload(a_jl, s);
    rc |= sc->aug_row.upload(a_row, s); rc |= sc->aug_col.upload(a_col, s);
    rc |= sc->d_jt_colptr.upload(sc->jt_colptr, s); rc |= sc->d_jt_rowval.upload(sc->jt_rowval, s);
    rc |= sc->d_h_colptr.upload(sc->h_colptr, s); rc |= sc->d_h_rowval.upload(sc->h_rowval, s);
    rc |= sc->jac_coo.alloc(nnzj); rc |= sc->hess_coo.alloc(nnzh);
    rc |= sc->jt_nz.alloc(sc->nnz_jt); rc |= sc->h_nz.alloc(sc->nnz_hess); rc |= sc->aug_nz.alloc(sc->nnz_aug);
    rc |= sc->diag_buffer.alloc(m); rc |= sc->pr_diag.alloc(n + m); rc |= sc->du_diag.alloc(m);

    // spmv structures: CSR of Jt and the row lists of Symmetric(hess_com, :L)
    mnk_sc_spmv_data* sp = new mnk_sc_spmv_data();
    {
        std::vector<int32_t> rp(n + 1, 0), ri(sc->nnz_jt), rperm(sc->nnz_jt);
        for (int32_t r : jt.rowval) rp[r + 1]++;
        for (int64_t i = 0; i < n; ++i) rp[i + 1] += rp[i];
        std::vector<int32_t> cur(rp.begin(), rp.end() - 1);
        for (int64_t c = 0; c < m; ++c)
            for (int32_t k = jt.colptr[c]; k < jt.colptr[c + 1]; ++k) {
                const int32_t p = cur[jt.rowval[k]]++;
                ri[p] = (int32_t)c; rperm[p] = k;
            }
        rc |= sp->jtr_ptr.upload(rp, s); rc |= sp->jtr_idx.upload(ri, s); rc |= sp->jtr_perm.upload(rperm, s);
    }
    {
        // row i of Symmetric(H,:L): entries (r, i) of column i (r >= i) with x_r, and entries (i, c), c < i, with x_c
        std::vector<int32_t> cnt(n + 1, 0);
        std::vector<int32_t> hcol = col_index(hh.colptr);
        for (int64_t k = 0; k < sc->nnz_hess; ++k) {
            cnt[hcol[k] + 1]++;
            if (hh.rowval[k] != hcol[k]) cnt[hh.rowval[k] + 1]++;
        }
        for (int64_t i = 0; i < n; ++i) cnt[i + 1] += cnt[i];
        std::vector<int32_t> idx(cnt[n]), perm(cnt[n]), cur(cnt.begin(), cnt.end() - 1);
        // strictly-lower row entries first (ascending column), then the column part: order of
        // Symmetric(H,:L)*x in ascending index = row part (cols < i), then column part (rows >= i)
        for (int64_t k = 0; k < sc->nnz_hess; ++k)
            if (hh.rowval[k] != hcol[k]) { const int32_t p = cur[hh.rowval[k]]++; idx[p] = hcol[k]; perm[p] = (int32_t)k; }
        for (int64_t k = 0; k < sc->nnz_hess; ++k) { const int32_t p = cur[hcol[k]]++; idx[p] = hh.rowval[k]; perm[p] = (int32_t)k; }
        rc |= sp->hs_ptr.upload(cnt, s); rc |= sp->hs_idx.upload(idx, s); rc |= sp->hs_perm.upload(perm, s);
    }
    if (rc) { delete sp; delete sc; return -2; }
    MNK_HIP(hipMemsetAsync(sc->jt_nz.p, 0, sc->jt_nz.n * sizeof(double), s));
    MNK_HIP(hipMemsetAsync(sc->h_nz.p, 0, sc->h_nz.n * sizeof(double), s));
    MNK_HIP(hipMemsetAsync(sc->aug_nz.p, 0, sc->aug_nz.n * sizeof(double), s));
    sc->extra = sp;
    mnk_ctx_child_added(ctx);
    *out = sc;
    return 0;
}

int mnk_sc_destroy(mnk_sc* sc) {
    if (!sc) return 0;
    if (sc->ctx) {
        (void)hipSetDevice(sc->ctx->device);
        (void)mnk::stream_wait(sc->ctx->stream);
    }
    delete spmv_of(sc);
    sc->extra = nullptr;
    mnk_ctx* ctx = sc->ctx;
    delete sc;
    mnk_ctx_child_gone(ctx);
    return 0;
}

int mnk_sc_sizes(mnk_sc* sc, int64_t* nnz_jt, int64_t* nnz_hess, int64_t* nnz_aug, int64_t* len_jptr) {
    MNK_REQUIRE(sc, "mnk_sc_sizes: NULL argument");
    if (nnz_jt) *nnz_jt = sc->nnz_jt;
    if (nnz_hess) *nnz_hess = sc->nnz_hess;
    if (nnz_aug) *nnz_aug = sc->nnz_aug;
    if (len_jptr) *len_jptr = sc->len_jptr;
    return 0;
}

int mnk_sc_get_structure(mnk_sc* sc, int which, int32_t* colptr, int32_t* rowval) {
    MNK_REQUIRE(sc && colptr && rowval, "mnk_sc_get_structure: NULL argument");
    const std::vector<int32_t>*cp, *rv;
    if (which == MNK_SC_JT) { cp = &sc->jt_colptr; rv = &sc->jt_rowval; }
    else if (which == MNK_SC_HESS) { cp = &sc->h_colptr; rv = &sc->h_rowval; }
    else if (which == MNK_SC_AUG) { cp = &sc->aug_colptr; rv = &sc->aug_rowval; }
    else { set_error("mnk_sc_get_structure: bad selector %d", which); return -1; }
    std::copy(cp->begin(), cp->end(), colptr);
    std::copy(rv->begin(), rv->end(), rowval);
    return 0;
}

int mnk_sc_get_map(mnk_sc* sc, int which, int64_t* map) {
    MNK_REQUIRE(sc && map, "mnk_sc_get_map: NULL argument");
    const std::vector<int64_t>* mp = which == MNK_SC_JT ? &sc->jt_map : which == MNK_SC_HESS ? &sc->h_map : nullptr;
    MNK_REQUIRE(mp != nullptr, "mnk_sc_get_map: bad selector");
    std::copy(mp->begin(), mp->end(), map);
    return 0;
}

int mnk_sc_get_ptrs(mnk_sc* sc, int32_t* d_dst, int32_t* d_src, int32_t* h_dst, int32_t* h_src, int32_t* j_dst,
                    int32_t* j_c, int32_t* j_k, int32_t* j_l) {
    MNK_REQUIRE(sc, "mnk_sc_get_ptrs: NULL argument");
    auto cp = [](const std::vector<int32_t>& v, int32_t* o) { if (o) std::copy(v.begin(), v.end(), o); };
    cp(sc->d_dst, d_dst); cp(sc->d_src, d_src); cp(sc->hp_dst, h_dst); cp(sc->hp_src, h_src);
    cp(sc->j_dst, j_dst); cp(sc->j_c, j_c); cp(sc->j_k, j_k); cp(sc->j_l, j_l);
    return 0;
}

static int stage_in(mnk_ctx* ctx, double* dev, const double* src, int64_t n, int loc, const double** use) {
    if (loc == MNK_DEVICE) { *use = src; return 0; }
    if (n > 0) {
        MNK_HIP(mnk::h2d_copy(dev, src, n * sizeof(double), ctx->stream));  // (complete on return: the caller's buffer is only valid for the duration of the call)
    }
    *use = dev;
    return 0;
}

int mnk_sc_compress_jacobian(mnk_sc* sc, const double* jac_coo, int loc) {
    MNK_REQUIRE(sc && sc->ctx && (jac_coo || sc->nnzj == 0), "mnk_sc_compress_jacobian: NULL argument or host-only handle");
    MNK_HIP(hipSetDevice(sc->ctx->device));
    const double* src;
    int rc = stage_in(sc->ctx, sc->jac_coo.p, jac_coo, sc->nnzj, loc, &src);
    if (rc) return rc;
    if (sc->nnz_jt > 0)
        hipLaunchKernelGGL(segsum_kernel, dim3((unsigned)((sc->nnz_jt + 255) / 256)), dim3(256), 0, sc->ctx->stream,
                           sc->jt_nz.p, src, sc->jt_seg_ptr.p, sc->jt_seg_src.p, sc->nnz_jt);
    MNK_HIP(hipGetLastError());
    return 0;
}

int mnk_sc_compress_hessian(mnk_sc* sc, const double* hess_coo, int loc) {
    MNK_REQUIRE(sc && sc->ctx && (hess_coo || sc->nnzh == 0), "mnk_sc_compress_hessian: NULL argument or host-only handle");
    MNK_HIP(hipSetDevice(sc->ctx->device));
    const double* src;
    int rc = stage_in(sc->ctx, sc->hess_coo.p, hess_coo, sc->nnzh, loc, &src);
    if (rc) return rc;
    if (sc->nnz_hess > 0)
        hipLaunchKernelGGL(segsum_kernel, dim3((unsigned)((sc->nnz_hess + 255) / 256)), dim3(256), 0, sc->ctx->stream,
                           sc->h_nz.p, src, sc->h_seg_ptr.p, sc->h_seg_src.p, sc->nnz_hess);
    MNK_HIP(hipGetLastError());
    return 0;
}

int mnk_sc_build(mnk_sc* sc, const double* pr_diag, const double* du_diag, int loc) {
    MNK_REQUIRE(sc && sc->ctx, "mnk_sc_build: NULL argument or host-only handle");
    MNK_HIP(hipSetDevice(sc->ctx->device));
    hipStream_t s = sc->ctx->stream;
    if (pr_diag == nullptr && du_diag == nullptr) {
        // the diagonals the handle keeps itself (mnk_sc_set_aug_diagonal / mnk_sc_regularize_diagonal)
        mnk_sc_spmv_data* sp0 = spmv_of(sc);
        MNK_REQUIRE(sp0 != nullptr && sp0->have_diag, "mnk_sc_build: no diagonals given and mnk_sc_set_aug_diagonal was not called");
        pr_diag = sc->pr_diag.p;
        du_diag = sc->du_diag.p;
        loc = MNK_DEVICE;
    }
    MNK_REQUIRE(pr_diag && (du_diag || sc->m == 0), "mnk_sc_build: NULL argument");
    const double *pr, *du;
    int rc = stage_in(sc->ctx, sc->pr_diag.p, pr_diag, sc->n + sc->m, loc, &pr);
    rc |= stage_in(sc->ctx, sc->du_diag.p, du_diag, sc->m, loc, &du);
    if (rc) return rc;
    // (caller's device vectors: the kernel also keeps our own copies -- the device-side solve_kkt!/mul! read Sigma_s and
    // du_diag later)
    const bool keep = loc == MNK_DEVICE && pr != sc->pr_diag.p;
    const int64_t work = std::max<int64_t>(sc->nnz_aug, sc->n + sc->m);
    hipLaunchKernelGGL(condense_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, sc->aug_nz.p, sc->h_nz.p, pr, du,
                       sc->diag_buffer.p, keep ? sc->pr_diag.p : nullptr, keep && sc->m > 0 ? sc->du_diag.p : nullptr,
                       sc->jt_nz.p, sc->aug_hptr.p, sc->aug_hsrc.p, sc->aug_dsrc.p, sc->aug_jptr.p, sc->aug_jc.p, sc->aug_jk.p,
                       sc->aug_jl.p, sc->nnz_aug, sc->n, sc->m);
    MNK_HIP(hipGetLastError());
    return 0;
}

int mnk_sc_get_values(mnk_sc* sc, int which, double* out, int loc) {
    MNK_REQUIRE(sc && sc->ctx && out, "mnk_sc_get_values: NULL argument or host-only handle");
    MNK_HIP(hipSetDevice(sc->ctx->device));
    const double* src; int64_t cnt;
    if (which == MNK_SC_JT) { src = sc->jt_nz.p; cnt = sc->nnz_jt; }
    else if (which == MNK_SC_HESS) { src = sc->h_nz.p; cnt = sc->nnz_hess; }
    else if (which == MNK_SC_AUG) { src = sc->aug_nz.p; cnt = sc->nnz_aug; }
    else if (which == MNK_SC_DIAGBUF) { src = sc->diag_buffer.p; cnt = sc->m; }
    else { set_error("mnk_sc_get_values: bad selector %d", which); return -1; }
    if (cnt > 0) {
        if (loc == MNK_DEVICE) MNK_HIP(hipMemcpyAsync(out, src, cnt * sizeof(double), hipMemcpyDeviceToDevice, sc->ctx->stream));
        else MNK_HIP(mnk::d2h_copy(out, src, cnt * sizeof(double), sc->ctx->stream));
    }
    return 0;
}

int mnk_sc_spmv(mnk_sc* sc, int which, int trans, double alpha, const double* x, double beta, double* y) {
    MNK_REQUIRE(sc && sc->ctx && x && y, "mnk_sc_spmv: NULL argument or host-only handle");
    MNK_HIP(hipSetDevice(sc->ctx->device));
    mnk_sc_spmv_data* sp = spmv_of(sc);
    MNK_REQUIRE(sp != nullptr, "mnk_sc_spmv: unknown handle");
    hipStream_t s = sc->ctx->stream;
    if (which == MNK_SC_JT && trans == 0) {
        hipLaunchKernelGGL(gather_spmv_kernel, dim3((unsigned)((sc->n + 255) / 256)), dim3(256), 0, s, y, x,
                           sc->jt_nz.p, sp->jtr_ptr.p, sp->jtr_idx.p, sp->jtr_perm.p, alpha, beta, sc->n);
    } else if (which == MNK_SC_JT) {
        if (sc->m > 0)
            hipLaunchKernelGGL(gather_spmv_kernel, dim3((unsigned)((sc->m + 255) / 256)), dim3(256), 0, s, y, x,
                               sc->jt_nz.p, sc->d_jt_colptr.p, sc->d_jt_rowval.p, (const int32_t*)nullptr, alpha,
                               beta, sc->m);
    } else if (which == MNK_SC_HESS) {
        hipLaunchKernelGGL(gather_spmv_kernel, dim3((unsigned)((sc->n + 255) / 256)), dim3(256), 0, s, y, x,
                           sc->h_nz.p, sp->hs_ptr.p, sp->hs_idx.p, sp->hs_perm.p, alpha, beta, sc->n);
    } else {
        set_error("mnk_sc_spmv: bad selector %d", which);
        return -1;
    }
    MNK_HIP(hipGetLastError());
    return 0;
}

// ---- device-side solve_kkt! / mul! ------------------------------------------------------------------
int mnk_sc_set_bounds(mnk_sc* sc, int64_t nlb, const int64_t* ind_lb, int64_t nub, const int64_t* ind_ub, int index_base) {
    MNK_REQUIRE(sc && sc->ctx && nlb >= 0 && nub >= 0 && (nlb == 0 || ind_lb) && (nub == 0 || ind_ub),
                "mnk_sc_set_bounds: bad argument or host-only handle");
    MNK_HIP(hipSetDevice(sc->ctx->device));
    mnk_sc_spmv_data* sp = spmv_of(sc);
    MNK_REQUIRE(sp != nullptr, "mnk_sc_set_bounds: unknown handle");
    const int64_t np = sc->n + sc->m;
    std::vector<int64_t> lb(nlb), ub(nub);
    for (int64_t i = 0; i < nlb; ++i) {
        lb[i] = ind_lb[i] - index_base;
        MNK_REQUIRE(lb[i] >= 0 && lb[i] < np, "mnk_sc_set_bounds: lower-bound index out of range");
    }
    for (int64_t i = 0; i < nub; ++i) {
        ub[i] = ind_ub[i] - index_base;
        MNK_REQUIRE(ub[i] >= 0 && ub[i] < np, "mnk_sc_set_bounds: upper-bound index out of range");
    }
    hipStream_t s = sc->ctx->stream;
    int rc = sp->ind_lb.upload(lb, s);
    rc |= sp->ind_ub.upload(ub, s);
    rc |= sp->reg.alloc(np);
    rc |= sp->l_diag.alloc(nlb);
    rc |= sp->l_lower.alloc(nlb);
    rc |= sp->u_diag.alloc(nub);
    rc |= sp->u_lower.alloc(nub);
    rc |= sp->buffer.alloc(sc->m);
    rc |= sp->buffer2.alloc(sc->m);
    const size_t lw = (size_t)(sc->n + 2 * sc->m + nlb + nub);
    rc |= sp->wdev.alloc(lw);
    rc |= sp->xdev.alloc(lw);
    if (rc) return rc;
    sp->nlb = nlb;
    sp->nub = nub;
    sp->have_bounds = true;
    return 0;
}

static int copy_in(mnk_ctx* ctx, double* dst, const double* src, int64_t n, int loc) {
    if (n <= 0) return 0;
    MNK_REQUIRE(src != nullptr, "NULL vector");
    if (loc == MNK_DEVICE) MNK_HIP(hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    else MNK_HIP(mnk::h2d_copy(dst, src, n * sizeof(double), ctx->stream));
    return 0;
}

int mnk_sc_set_barrier_terms(mnk_sc* sc, const double* reg, const double* l_diag, const double* u_diag,
                             const double* l_lower, const double* u_lower, int loc) {
    MNK_REQUIRE(sc && sc->ctx, "mnk_sc_set_barrier_terms: NULL argument or host-only handle");
    MNK_HIP(hipSetDevice(sc->ctx->device));
    mnk_sc_spmv_data* sp = spmv_of(sc);
    MNK_REQUIRE(sp != nullptr && sp->have_bounds, "mnk_sc_set_barrier_terms: call mnk_sc_set_bounds first");
    int rc = copy_in(sc->ctx, sp->reg.p, reg, sc->n + sc->m, loc);
    rc |= copy_in(sc->ctx, sp->l_diag.p, l_diag, sp->nlb, loc);
    rc |= copy_in(sc->ctx, sp->u_diag.p, u_diag, sp->nub, loc);
    rc |= copy_in(sc->ctx, sp->l_lower.p, l_lower, sp->nlb, loc);
    rc |= copy_in(sc->ctx, sp->u_lower.p, u_lower, sp->nub, loc);
    if (rc) return rc;
    if (loc != MNK_DEVICE) MNK_HIP(mnk::stream_wait(sc->ctx->stream));  // the host arrays may change after return
    sp->have_terms = true;
    return 0;
}

static int sc_diag_view(mnk_sc* sc, AugDiagView& v, const char* who) {
    if (!(sc && sc->ctx)) { set_error("%s: NULL argument or host-only handle", who); return -1; }
    MNK_HIP(hipSetDevice(sc->ctx->device));
    mnk_sc_spmv_data* sp = spmv_of(sc);
    if (!(sp != nullptr && sp->have_bounds)) { set_error("%s: call mnk_sc_set_bounds first", who); return -1; }
    v = AugDiagView{sc->ctx, sc->n + sc->m, sc->m, sp->nlb, sp->nub, sp->reg.p, sc->pr_diag.p, sc->du_diag.p, sp->l_diag.p,
                    sp->u_diag.p, sp->l_lower.p, sp->u_lower.p, sp->ind_lb.p, sp->ind_ub.p, &sp->feed};
    return 0;
}

int mnk_sc_set_aug_diagonal(mnk_sc* sc, const double* x, const double* xl, const double* xu, const double* zl,
                            const double* zu, double primal_reg, double dual_reg, int loc) {
    AugDiagView v;
    int rc = sc_diag_view(sc, v, "mnk_sc_set_aug_diagonal");
    if (rc) return rc;
    MNK_REQUIRE(x && xl && xu && zl && zu, "mnk_sc_set_aug_diagonal: NULL vector");
    rc = kkt_set_aug_diagonal(v, x, xl, xu, zl, zu, primal_reg, dual_reg, loc);
    if (rc) return rc;
    spmv_of(sc)->have_terms = spmv_of(sc)->have_diag = true;
    return 0;
}

// set_aug_RR!(kkt, solver, RR) (reference src/IPM/kernels.jl:72-87): device-resident vectors only
int mnk_sc_set_aug_RR(mnk_sc* sc, const double* x, const double* xl, const double* xu, const double* zl, const double* zu,
                      const double* D_R, const double* pp, const double* zp, const double* nn, const double* zn, double zeta,
                      double primal_reg, double dual_reg) {
    AugDiagView v;
    int rc = sc_diag_view(sc, v, "mnk_sc_set_aug_RR");
    if (rc) return rc;
    MNK_REQUIRE(x && xl && xu && zl && zu && D_R && (v.ndu == 0 || (pp && zp && nn && zn)), "mnk_sc_set_aug_RR: NULL vector");
    rc = kkt_set_aug_RR(v, x, xl, xu, zl, zu, D_R, pp, zp, nn, zn, zeta, primal_reg, dual_reg);
    if (rc) return rc;
    spmv_of(sc)->have_terms = spmv_of(sc)->have_diag = true;
    return 0;
}

int mnk_sc_regularize_diagonal(mnk_sc* sc, double primal, double dual) {
    AugDiagView v;
    int rc = sc_diag_view(sc, v, "mnk_sc_regularize_diagonal");
    if (rc) return rc;
    MNK_REQUIRE(spmv_of(sc)->have_diag, "mnk_sc_regularize_diagonal: call mnk_sc_set_aug_diagonal first");
    return kkt_regularize_diagonal(v, primal, dual);
}

// reg, pr_diag and du_diag as they are, into a buffer of the handle / back from it (device copies on the handle's stream): the
// bracket of a SPECULATIVE trial of inertia_correction! (madnlp_jl_amd.ipm_dev: the trial with the next perturbation is factorized
// together with the unperturbed one; when the unperturbed matrix is accepted after all, the diagonals return to the bits they had
// -- pr_diag + dw - dw would not)
int mnk_sc_save_diagonals(mnk_sc* sc) {
    AugDiagView v;
    int rc = sc_diag_view(sc, v, "mnk_sc_save_diagonals");
    if (rc) return rc;
    mnk_sc_spmv_data* sp = spmv_of(sc);
    MNK_REQUIRE(sp->have_diag, "mnk_sc_save_diagonals: call mnk_sc_set_aug_diagonal first");
    const size_t npr = (size_t)v.npr, ndu = (size_t)v.ndu;
    if (sp->saved_diag.n < 2 * npr + ndu && sp->saved_diag.alloc(2 * npr + ndu)) return -1;
    hipStream_t s = sc->ctx->stream;
    MNK_HIP(hipMemcpyAsync(sp->saved_diag.p, v.reg, npr * sizeof(double), hipMemcpyDeviceToDevice, s));
    MNK_HIP(hipMemcpyAsync(sp->saved_diag.p + npr, v.pr_diag, npr * sizeof(double), hipMemcpyDeviceToDevice, s));
    if (ndu > 0) MNK_HIP(hipMemcpyAsync(sp->saved_diag.p + 2 * npr, v.du_diag, ndu * sizeof(double), hipMemcpyDeviceToDevice, s));
    sp->have_saved_diag = true;
    return 0;
}

int mnk_sc_restore_diagonals(mnk_sc* sc) {
    AugDiagView v;
    int rc = sc_diag_view(sc, v, "mnk_sc_restore_diagonals");
    if (rc) return rc;
    mnk_sc_spmv_data* sp = spmv_of(sc);
    MNK_REQUIRE(sp->have_saved_diag, "mnk_sc_restore_diagonals: nothing saved (mnk_sc_save_diagonals)");
    const size_t npr = (size_t)v.npr, ndu = (size_t)v.ndu;
    hipStream_t s = sc->ctx->stream;
    MNK_HIP(hipMemcpyAsync(v.reg, sp->saved_diag.p, npr * sizeof(double), hipMemcpyDeviceToDevice, s));
    MNK_HIP(hipMemcpyAsync(v.pr_diag, sp->saved_diag.p + npr, npr * sizeof(double), hipMemcpyDeviceToDevice, s));
    if (ndu > 0) MNK_HIP(hipMemcpyAsync(v.du_diag, sp->saved_diag.p + 2 * npr, ndu * sizeof(double), hipMemcpyDeviceToDevice, s));
    return 0;
}

int mnk_sc_get_diagonals(mnk_sc* sc, double* pr_diag, double* du_diag, double* reg, double* l_diag, double* u_diag,
                         double* l_lower, double* u_lower) {
    AugDiagView v;
    int rc = sc_diag_view(sc, v, "mnk_sc_get_diagonals");
    if (rc) return rc;
    return kkt_get_diagonals(v, pr_diag, du_diag, reg, l_diag, u_diag, l_lower, u_lower);
}

#define MNK_GRID(cnt) dim3((unsigned)(((cnt) + 255) / 256)), dim3(256), 0, s

int mnk_sc_solve_kkt(mnk_sc* sc, mnk_ls* ls, double* w, int loc) {
    MNK_REQUIRE(sc && sc->ctx && ls && w, "mnk_sc_solve_kkt: NULL argument or host-only handle");
    MNK_REQUIRE(ls->ctx == sc->ctx && ls->N == sc->n, "mnk_sc_solve_kkt: the solver does not belong to this system");
    MNK_HIP(hipSetDevice(sc->ctx->device));
    mnk_sc_spmv_data* sp = spmv_of(sc);
    MNK_REQUIRE(sp != nullptr && sp->have_bounds && sp->have_terms,
                "mnk_sc_solve_kkt: call mnk_sc_set_bounds / mnk_sc_set_barrier_terms / mnk_sc_build first");
    hipStream_t s = sc->ctx->stream;
    const int64_t n = sc->n, m = sc->m, nlb = sp->nlb, nub = sp->nub, lw = n + 2 * m + nlb + nub;
    // Host-resident caller: the host still owns `w` until the final copy-back, so a persistent solve that gave
    // up (abort word raised, solve.hip) is detected HERE, after the stream synchronization and before anything
    // is copied back, and the whole solve_kkt! is redone once with the stepwise solve.
    for (int attempt = 0; attempt < 2; ++attempt) {
        double* d = w;
        if (loc != MNK_DEVICE) {
            d = sp->wdev.p;
            MNK_HIP(mnk::h2d_copy(d, w, lw * sizeof(double), s));
        }
        double *ws = d + n, *wz = d + n + m, *wl = d + n + 2 * m, *wu = wl + nlb;
        const double* Ss = sc->pr_diag.p + n;
        if (nlb > 0) hipLaunchKernelGGL(reduce_rhs_kernel, MNK_GRID(nlb), d, sp->ind_lb.p, wl, sp->l_diag.p, nlb);
        if (nub > 0) hipLaunchKernelGGL(reduce_rhs_kernel, MNK_GRID(nub), d, sp->ind_ub.p, wu, sp->u_diag.p, nub);
        if (m > 0) {
            hipLaunchKernelGGL(condense_rhs_kernel, MNK_GRID(m), sp->buffer.p, sc->diag_buffer.p, ws, wz, Ss, m);
            int rc = mnk_sc_spmv(sc, MNK_SC_JT, 0, 1.0, sp->buffer.p, 1.0, d);  // wx += Jt * buffer
            if (rc) return rc;
        }
        int rc = mnk_ls_solve(ls, d, 1, n, MNK_DEVICE);
        if (rc) return rc;
        if (m > 0) {
            rc = mnk_sc_spmv(sc, MNK_SC_JT, 1, 1.0, d, 0.0, sp->buffer2.p);  // buffer2 = Jt' * wx
            if (rc) return rc;
            hipLaunchKernelGGL(expand_sol_kernel, MNK_GRID(m), ws, wz, sp->buffer.p, sp->buffer2.p, sc->diag_buffer.p, Ss, m);
        }
        if (nlb > 0) hipLaunchKernelGGL(finish_aug_kernel, MNK_GRID(nlb), wl, d, sp->ind_lb.p, sp->l_lower.p, sp->l_diag.p, nlb, 0);
        if (nub > 0) hipLaunchKernelGGL(finish_aug_kernel, MNK_GRID(nub), wu, d, sp->ind_ub.p, sp->u_lower.p, sp->u_diag.p, nub, 1);
        MNK_HIP(hipGetLastError());
        if (loc == MNK_DEVICE) break;  // device-resident caller: mnk_ls_check_solve() reports an abort
        MNK_HIP(mnk::stream_wait(s));
        if (attempt == 0 && mnk_ls_take_solve_abort(ls)) continue;  // redo with the stepwise solve
        MNK_HIP(mnk::d2h_copy(w, d, lw * sizeof(double), s));
        break;
    }
    return 0;
}

int mnk_sc_mul(mnk_sc* sc, double* w, const double* x, double alpha, double beta, int loc) {
    MNK_REQUIRE(sc && sc->ctx && w && x, "mnk_sc_mul: NULL argument or host-only handle");
    MNK_HIP(hipSetDevice(sc->ctx->device));
    mnk_sc_spmv_data* sp = spmv_of(sc);
    MNK_REQUIRE(sp != nullptr && sp->have_bounds && sp->have_terms,
                "mnk_sc_mul: call mnk_sc_set_bounds / mnk_sc_set_barrier_terms first");
    hipStream_t s = sc->ctx->stream;
    const int64_t n = sc->n, m = sc->m, nlb = sp->nlb, nub = sp->nub, lw = n + 2 * m + nlb + nub;
    double* dw = w;
    const double* dx = x;
    if (loc != MNK_DEVICE) {
        dw = sp->wdev.p;
        MNK_HIP(mnk::h2d_copy(sp->wdev.p, w, lw * sizeof(double), s));
        MNK_HIP(mnk::h2d_copy(sp->xdev.p, x, lw * sizeof(double), s));
        dx = sp->xdev.p;
    }
    // wx = alpha Sym(H) xx + beta wx ; wx += alpha Jt xz ; wz = alpha Jt' xx + beta wz
    int rc = mnk_sc_spmv(sc, MNK_SC_HESS, 0, alpha, dx, beta, dw);
    if (rc) return rc;
    if (m > 0) {
        rc = mnk_sc_spmv(sc, MNK_SC_JT, 0, alpha, dx + n + m, 1.0, dw);
        if (rc) return rc;
        rc = mnk_sc_spmv(sc, MNK_SC_JT, 1, alpha, dx, beta, dw + n + m);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(kktmul_diag_kernel, MNK_GRID(n + 2 * m), dw, dx, sp->reg.p, sc->du_diag.p, alpha, beta, n, m);
    if (nlb > 0)
        hipLaunchKernelGGL(kktmul_bound_kernel, MNK_GRID(nlb), dw, dw + n + 2 * m, dx, dx + n + 2 * m, sp->ind_lb.p,
                           sp->l_lower.p, sp->l_diag.p, alpha, beta, nlb, 0);
    if (nub > 0)
        hipLaunchKernelGGL(kktmul_bound_kernel, MNK_GRID(nub), dw, dw + n + 2 * m + nlb, dx, dx + n + 2 * m + nlb,
                           sp->ind_ub.p, sp->u_lower.p, sp->u_diag.p, alpha, beta, nub, 1);
    MNK_HIP(hipGetLastError());
    if (loc != MNK_DEVICE) {
        MNK_HIP(mnk::d2h_copy(w, dw, lw * sizeof(double), s));
    }
    return 0;
}
#undef MNK_GRID

}  // extern "C"
