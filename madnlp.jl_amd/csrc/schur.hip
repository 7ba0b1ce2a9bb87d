// Dense `S` stage of the Schur-complement KKT system (reference src/KKT/Schur/schur.jl:927-1001 `build_kkt!`,
// :1003-1005 `factorize_kkt!`, :1040-1065 steps 3-5 of `solve_kkt!`): SURVEY 8(f).3.
//
// Two-stage structure: ns scenario blocks A_k (blk x blk, symmetric indefinite) coupled to nd design variables by
// dense C_dk (nd x blk); the Schur complement the dense linear solver sees is
//       S = S0 - sum_k C_dk A_k^-1 C_dk',        S0 = H_dd + Sigma_dd + inequality terms,
// and a solve is   r_k <- A_k^-1 r_k ;  r_d <- r_d - sum_k C_dk r_k ;  S x_d = r_d ;  x_k = r_k - (A_k^-1 C_dk') x_d.
// The reference factors the A_k with a sparse solver per scenario (MUMPS; sparse direct solvers are out of this
// path's scope); here the scenario blocks arrive dense and are factored by the same blocked fp64-MFMA factorization
// as S itself (BUNCHKAUFMAN tiers: the blocks are indefinite).  With a static-pivot factor A_k = L D L' the term is
// C A^-1 C' = (C L^-T) D^-1 (C L^-T)': ONE forward sweep over the nd rows of C_dk (right-side block triangular solve
// on MFMA) and one MFMA rank-blk update of S; a block that needed the pivoted tier falls back to the reference's
// column-by-column T_k = A_k^-1 C_dk' and S -= C_dk T_k.
//
// Multi-GPU: scenarios are sharded over the ranks.  mnk_schur_build_local produces THIS rank's contribution
// (S0 on the rank that owns it, minus its scenarios' terms) into a caller-owned device buffer; the caller sums the
// contributions with ONE all-reduce of nd^2 doubles (RCCL over xGMI) and every rank factors S; a solve needs one more
// all-reduce of nd doubles (mnk_schur_forward's contribution).  The library itself does no communication.
#include <vector>

#include "ls.h"

struct mnk_schur {
    mnk_ctx* ctx = nullptr;
    int64_t ns = 0, blk = 0, nd = 0, ndp = 0, blkp = 0;
    int algo = MNK_BUNCHKAUFMAN;
    mnk::DevBuf<double> A;    // ns x (blk x blk)   scenario blocks (lower triangle read)
    mnk::DevBuf<double> C;    // ns x (nd x blk)    coupling blocks
    mnk::DevBuf<double> T;    // blk x nd    A_k^-1 C_dk' of ONE scenario (pivoted-tier fallback only)
    mnk::DevBuf<double> Cp;   // ndp x Npb   X = C_dk L^-T D^-1 (fast path) / zero-padded C_dk (fallback)   (MFMA operand)
    mnk::DevBuf<double> Tt;   // ndp x Npb   V = C_dk L^-T (fast path) / zero-padded T_k' (fallback)         (MFMA operand)
    mnk::DevBuf<double> tmpk; // Npb         C_dk' x_d of one scenario (back-substitution)
    int64_t Npb = 0;          // padded order of a scenario factor (multiple of 128)
    mnk::DevBuf<double> Sp;   // ndp x ndp   accumulator
    std::vector<mnk_ls*> ls_k;
    mnk_ls* ls_s = nullptr;
    std::vector<int> info_k;
    bool built = false;
    // build_kkt!: the forward sweeps of ALL scenarios with a static-pivot factor run as grouped launches (one per 64-column
    // step: blockIdx.y = scenario) on per-scenario X / V buffers, their products X_k V_k' as one batched product into
    // per-scenario partial sums that are added to S in scenario order (the same bits whatever the timing)
    mnk::DevBuf<double> Xall, Vall, Pall;   // ns x (ndp x Npb) | the same (LDL^T only) | ns x (ndp x ndp)
    mnk::DevBuf<char> recs;                 // device copies of the launch records
    mnk::DevBuf<int> fast_k;                // the scenarios on the grouped path
    int64_t chunk = 32;                     // scenarios per pass of the grouped build (their X / V / P buffers are reused)
    char* stage = nullptr;                  // pinned host memory: the launch records + scenario list of one pass (no synchronizing upload)
    size_t stage_bytes = 0;
    hipEvent_t stage_ev = nullptr;          // recorded behind the last upload from `stage`
    mnk::DevBuf<double> Sown;               // nd x nd: the handle's own copy of S (mnk_schur_s_buffer: callers without device memory of their own)
    mnk::DevBuf<double> hostrk, hostrd;     // mnk_schur_solve with host vectors: ns x blk | 2 nd (right-hand side, contribution)
    // ---- device-side assembly of A_k / C_dk / S0 from the callbacks' COO values (mnk_schur_set_structure / mnk_schur_assemble):
    // every touched entry of the three buffers is one SEGMENT of sources, summed by one thread in a fixed order (no atomics)
    struct {
        bool have = false;
        int64_t n = 0, m = 0, nv = 0, nc = 0, nnzh = 0, nnzj = 0, n_ineq = 0, nseg = 0, nsrc = 0;
        mnk::DevBuf<int64_t> seg_dst;           // nseg: virtual offset into [A | C | S0]
        mnk::DevBuf<int32_t> seg_ptr;           // nseg + 1
        mnk::DevBuf<int8_t> src_kind;           // 0 hess[a], 1 jac[a], 2 pr_diag[a], 3 du_diag[a], 4 jac[a] * D[w] * jac[b]
        mnk::DevBuf<int32_t> src_a, src_b, src_w;
        mnk::DevBuf<int64_t> ind_ineq;          // constraint row of slack p
        mnk::DevBuf<double> D, S0, vals;        // n_ineq | nd x nd | staging of host inputs: hess | jac | pr_diag | du_diag
    } as;
};

namespace mnk {

// dst (rows_d x cols_d, ld ldd; zero padding beyond the source) = src' or src
__global__ void schur_copy_kernel(double* __restrict__ dst, int64_t ldd, int64_t rows_d, int64_t cols_d,
                                  const double* __restrict__ src, int64_t lds, int64_t rows_s, int64_t cols_s, int trans) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= rows_d * cols_d) return;
    const int64_t r = e % rows_d, c = e / rows_d;
    double v = 0.0;
    if (trans) { if (c < rows_s && r < cols_s) v = src[c + r * lds]; }   // dst[r][c] = src[c][r]
    else { if (r < rows_s && c < cols_s) v = src[r + c * lds]; }
    dst[r + c * ldd] = v;
}

// S += P_0 + P_1 + ... (the scenarios' partial sums, in scenario order)
__global__ void schur_partial_sum_kernel(double* __restrict__ S, const double* __restrict__ P, int64_t np, int64_t n) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = S[i];
    for (int64_t k = 0; k < np; ++k) v += P[k * n + i];
    S[i] = v;
}
// X_i (ndp x Npb, zero padded) = C_dk of scenario k = fast_k[i]   (blockIdx.y = i)
__global__ void schur_copy_batch_kernel(double* __restrict__ Xall, int64_t ndp, int64_t Npb, const double* __restrict__ Call,
                                        int64_t nd, int64_t blk, const int* __restrict__ fast_k) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= ndp * Npb) return;
    const int64_t r = e % ndp, c = e / ndp;
    const double* src = Call + (int64_t)fast_k[blockIdx.y] * nd * blk;
    Xall[(int64_t)blockIdx.y * ndp * Npb + e] = (r < nd && c < blk) ? src[r + c * nd] : 0.0;
}
// The per-scenario vector kernels of a solve, scenario k = blockIdx.y, ONE launch each (ns launches of a few microseconds were
// most of a solve).  Fixed summation orders: deterministic.
// P[k][r] = sum_c C_k[r + c*nd] x_k[c], then y[r] = sum_k alpha P[k][r] in scenario order
__global__ __launch_bounds__(256) void schur_gemv_batch_kernel(double* __restrict__ P, const double* __restrict__ Call, int64_t nd,
                                                               const double* __restrict__ xall, int64_t blk) {
    const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, k = blockIdx.y;
    if (r >= nd) return;
    const double* A = Call + k * nd * blk;
    const double* x = xall + k * blk;
    double s0 = 0.0, s1 = 0.0;
    int64_t c = 0;
    for (; c + 1 < blk; c += 2) {
        s0 = fma(A[r + c * nd], x[c], s0);
        s1 = fma(A[r + (c + 1) * nd], x[c + 1], s1);
    }
    if (c < blk) s0 = fma(A[r + c * nd], x[c], s0);
    P[k * nd + r] = s0 + s1;
}
__global__ void schur_contrib_sum_kernel(double* __restrict__ y, const double* __restrict__ P, int64_t ns, int64_t nd, double alpha) {
    const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (r >= nd) return;
    double v = 0.0;
    for (int64_t k = 0; k < ns; ++k) v += alpha * P[k * nd + r];
    y[r] = v;
}
// tmp_k[c] = sum_r C_k[r + c*nd] x_d[r]
__global__ __launch_bounds__(256) void schur_gemv_t_batch_kernel(double* __restrict__ tmp, int64_t stride, const double* __restrict__ Call,
                                                                 int64_t nd, const double* __restrict__ x, int64_t blk) {
    const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, k = blockIdx.y;
    if (c >= blk) return;
    const double* a = Call + k * nd * blk + c * nd;
    double s0 = 0.0, s1 = 0.0;
    int64_t r = 0;
    for (; r + 1 < nd; r += 2) {
        s0 = fma(a[r], x[r], s0);
        s1 = fma(a[r + 1], x[r + 1], s1);
    }
    if (r < nd) s0 = fma(a[r], x[r], s0);
    tmp[k * stride + c] = s0 + s1;
}
// rhs_k[i] -= tmp_k[i]
__global__ void schur_sub_batch_kernel(double* __restrict__ y, int64_t blk, const double* __restrict__ tmp, int64_t stride) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, k = blockIdx.y;
    if (i < blk) y[k * blk + i] -= tmp[k * stride + i];
}

// y += x (the design right-hand side receives the scenarios' contribution)
__global__ void schur_axpy_kernel(double* __restrict__ y, const double* __restrict__ x, int64_t n) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) y[i] += x[i];
}

}  // namespace mnk

namespace mnk {
// D = Sigma_s / (1 - Sigma_d Sigma_s) of every inequality row (reference schur.jl:929-933), rounded operation by operation
__global__ void schur_condense_weights_kernel(double* __restrict__ D, const double* __restrict__ pr_diag, const double* __restrict__ du_diag,
                                              const int64_t* __restrict__ ind_ineq, int64_t n, int64_t n_ineq) {
    const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (p >= n_ineq) return;
    const double Ss = pr_diag[n + p], Sd = du_diag[ind_ineq[p]];
    D[p] = __ddiv_rn(Ss, __dsub_rn(1.0, __dmul_rn(Sd, Ss)));
}
// One thread per touched entry of [A | C | S0]: its sources in the order the reference scatters them (Hessian entries, diagonal
// terms, equality-row Jacobian entries, the condensation pairs J' D J row by row: schur.jl:935-972), every operation rounded on
// its own (no contraction into fma): the entry carries the bits a sequential scatter-add produces.  The reference's GPU twin
// (lib/MadNLPGPU/ext/MadNLPGPUCUDAExt/kernels_schur.jl:14-174) adds with @atomic -- an order that changes from run to run.
__global__ void schur_assemble_kernel(int64_t nseg, const int64_t* __restrict__ seg_dst, const int32_t* __restrict__ seg_ptr,
                                      const int8_t* __restrict__ kind, const int32_t* __restrict__ sa, const int32_t* __restrict__ sb,
                                      const int32_t* __restrict__ sw, const double* __restrict__ hess, const double* __restrict__ jac,
                                      const double* __restrict__ pr_diag, const double* __restrict__ du_diag, const double* __restrict__ D,
                                      double* __restrict__ A, int64_t sizeA, double* __restrict__ Cb, int64_t sizeC, double* __restrict__ S0) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= nseg) return;
    double acc = 0.0;
    for (int32_t q = seg_ptr[t]; q < seg_ptr[t + 1]; ++q) {
        const int a = sa[q];
        double v;
        switch (kind[q]) {
            case 0: v = hess[a]; break;
            case 1: v = jac[a]; break;
            case 2: v = pr_diag[a]; break;
            case 3: v = du_diag[a]; break;
            default: v = __dmul_rn(__dmul_rn(jac[a], D[sw[q]]), jac[sb[q]]); break;
        }
        acc = __dadd_rn(acc, v);
    }
    const int64_t d = seg_dst[t];
    if (d < sizeA) A[d] = acc;
    else if (d < sizeA + sizeC) Cb[d - sizeA] = acc;
    else S0[d - sizeA - sizeC] = acc;
}
}  // namespace mnk

using namespace mnk;

extern "C" {

int mnk_schur_create(mnk_ctx* ctx, int64_t ns_local, int64_t blk, int64_t nd, int algo, mnk_schur** out) {
    MNK_REQUIRE(ctx && out, "mnk_schur_create: NULL argument");
    MNK_REQUIRE(ns_local >= 0 && blk > 0 && nd > 0, "mnk_schur_create: bad dimensions");
    MNK_HIP(hipSetDevice(ctx->device));
    mnk_schur* h = new mnk_schur();
    h->ctx = ctx;
    h->ns = ns_local; h->blk = blk; h->nd = nd; h->algo = algo;
    h->ndp = round_up(nd, 64);
    h->blkp = round_up(blk, 16);
    h->Npb = round_up(blk, PAD);
    int rc = 0;
    rc |= h->A.alloc((size_t)std::max<int64_t>(ns_local, 1) * blk * blk);
    rc |= h->C.alloc((size_t)std::max<int64_t>(ns_local, 1) * nd * blk);
    rc |= h->T.alloc((size_t)blk * nd);
    rc |= h->Cp.alloc((size_t)h->ndp * h->Npb + SLACK);
    rc |= h->Tt.alloc((size_t)h->ndp * h->Npb + SLACK);
    rc |= h->tmpk.alloc((size_t)h->Npb * (size_t)std::max<int64_t>(ns_local, 1));   // (one vector per scenario: their solves run as a batch)
    rc |= h->Sp.alloc((size_t)h->ndp * h->ndp + SLACK);
    {
        const size_t nsl = (size_t)std::max<int64_t>(ns_local, 1);
        const size_t nstep = (size_t)(h->Npb / NBI);
        //  the grouped build works on CHUNKS of at most `chunk` scenarios -- X / V / P buffers of that many, reused
        // from chunk to chunk, the partial sums added to S chunk by chunk in scenario order (the same additions in the same
        // order as one pass over all of them: the same bits) -- instead of ns x nd^2 doubles for P alone (ns = 128, nd = 4096:
        // 17 GB); the solves' per-scenario vectors need ns x nd doubles of P
        // chunk size: as many scenarios as ~4 GB of X / V / P hold, at least 32 (every chunk costs three record uploads with a
        // stream synchronization each: ns = 128, blk = 512, nd = 256 in chunks of 32 built in 10.2 instead of 4.6-7.9 ms)
        const size_t per_scen = (size_t)h->ndp * h->ndp * 8 + 2 * (size_t)h->ndp * h->Npb * 8;
        const size_t by_mem = std::max<size_t>(32, ((size_t)4 << 30) / std::max<size_t>(per_scen, 1));
        h->chunk = (int64_t)std::min<size_t>(nsl, getenv("MNK_SCHUR_CHUNK") ? (size_t)std::max(1, atoi(getenv("MNK_SCHUR_CHUNK"))) : by_mem);
        const size_t nch = (size_t)h->chunk;
        rc |= h->Xall.alloc(nch * h->ndp * h->Npb + SLACK);
        if (algo != MNK_CHOLESKY) rc |= h->Vall.alloc(nch * h->ndp * h->Npb + SLACK);
        rc |= h->Pall.alloc(std::max(nch * h->ndp * h->ndp, nsl * (size_t)h->ndp) + SLACK);
        rc |= h->recs.alloc(nsl * (sizeof(mnk::TrsmBatchRec) + nstep * sizeof(mnk::GemmBatchRec)) + 64);
        rc |= h->fast_k.alloc(nsl);
    }
    if (rc) { (void)hipGetLastError(); delete h; return -2; }
    for (int64_t k = 0; k < ns_local && !rc; ++k) {
        mnk_ls* ls = nullptr;
        rc = mnk_ls_create(ctx, blk, algo, &ls);
        if (!rc) h->ls_k.push_back(ls);
    }
    if (!rc) rc = mnk_ls_create(ctx, nd, algo, &h->ls_s);
    if (rc) {
        for (mnk_ls* l : h->ls_k) mnk_ls_destroy(l);
        if (h->ls_s) mnk_ls_destroy(h->ls_s);
        delete h;
        return rc;
    }
    h->info_k.assign(ns_local, 0);
    mnk_ctx_child_added(ctx);
    *out = h;
    return 0;
}

int mnk_schur_destroy(mnk_schur* h) {
    if (!h) return 0;
    (void)hipSetDevice(h->ctx->device);
    (void)mnk::stream_wait(h->ctx->stream);
    for (mnk_ls* l : h->ls_k) mnk_ls_destroy(l);
    if (h->ls_s) mnk_ls_destroy(h->ls_s);
    mnk_ctx* ctx = h->ctx;
    if (h->stage) { mnk::LaunchLock lock; mnk::quiesce_persistent(); (void)hipHostFree(h->stage); }
    if (h->stage_ev) (void)hipEventDestroy(h->stage_ev);
    delete h;
    mnk_ctx_child_gone(ctx);
    return 0;
}

int mnk_schur_set_block(mnk_schur* h, int64_t k, const double* A_kk, int64_t lda, const double* C_dk, int64_t ldc, int loc) {
    MNK_REQUIRE(h && A_kk && C_dk, "mnk_schur_set_block: NULL argument");
    MNK_REQUIRE(k >= 0 && k < h->ns && lda >= h->blk && ldc >= h->nd, "mnk_schur_set_block: bad scenario index / leading dimension");
    MNK_HIP(hipSetDevice(h->ctx->device));
    hipStream_t s = h->ctx->stream;
    if (loc == MNK_DEVICE) {
        MNK_HIP(hipMemcpy2DAsync(h->A.p + k * h->blk * h->blk, h->blk * sizeof(double), A_kk, lda * sizeof(double),
                                 h->blk * sizeof(double), h->blk, hipMemcpyDeviceToDevice, s));
        MNK_HIP(hipMemcpy2DAsync(h->C.p + k * h->nd * h->blk, h->nd * sizeof(double), C_dk, ldc * sizeof(double),
                                 h->nd * sizeof(double), h->blk, hipMemcpyDeviceToDevice, s));
    } else {
        MNK_HIP(mnk::h2d_copy_2d(h->A.p + k * h->blk * h->blk, h->blk * sizeof(double), A_kk, lda * sizeof(double),
                                 h->blk * sizeof(double), h->blk, s));
        MNK_HIP(mnk::h2d_copy_2d(h->C.p + k * h->nd * h->blk, h->nd * sizeof(double), C_dk, ldc * sizeof(double),
                                 h->nd * sizeof(double), h->blk, s));
    }
    h->built = false;
    return 0;
}

#define MNK_GRID1(cnt) dim3((unsigned)(((cnt) + 255) / 256)), dim3(256), 0, s

int mnk_schur_build_local(mnk_schur* h, const double* S0, int64_t lds0, int loc_s0, double* S_out, int64_t lds_out) {
    MNK_REQUIRE(h && S_out, "mnk_schur_build_local: NULL argument");
    MNK_REQUIRE(lds_out >= h->nd && (S0 == nullptr || lds0 >= h->nd), "mnk_schur_build_local: bad leading dimension");
    MNK_HIP(hipSetDevice(h->ctx->device));
    hipStream_t s = h->ctx->stream;
    const int64_t nd = h->nd, blk = h->blk, ndp = h->ndp, blkp = h->blkp;
    // S <- S0 (or 0), zero padded
    MNK_HIP(hipMemsetAsync(h->Sp.p, 0, (size_t)ndp * ndp * sizeof(double), s));
    if (S0 != nullptr) {
        if (loc_s0 == MNK_DEVICE)
            MNK_HIP(hipMemcpy2DAsync(h->Sp.p, ndp * sizeof(double), S0, lds0 * sizeof(double), nd * sizeof(double), nd,
                                     hipMemcpyDeviceToDevice, s));
        else
            MNK_HIP(mnk::h2d_copy_2d(h->Sp.p, ndp * sizeof(double), S0, lds0 * sizeof(double), nd * sizeof(double), nd, s));
    }
    // Phase 1a (reference :955-990, `@blas_safe_threads for k in 1:ns`): the scenario blocks are factored TOGETHER -- one
    // batch (dag.hip): the pivot chains of up to 32 blocks side by side in one launch instead of one whole-chip factorization
    // and one host synchronization per scenario (a 512-row block is chain-bound at 0.03 of the peak on its own)
    {
        int rc = mnk_factorize_batch_begin();
        if (rc) return rc;
        for (int64_t k = 0; k < h->ns && !rc; ++k) rc = mnk_ls_factorize_dense_dev_async(h->ls_k[k], h->A.p + k * blk * blk, blk);
        const int rc_end = mnk_factorize_batch_end();
        if (rc || rc_end) return rc ? rc : rc_end;
    }
    // (the tier that produced each factor is known first: the info fetch of a batch member waits for its own factorization)
    std::vector<int> fast, slow;
    for (int64_t k = 0; k < h->ns; ++k) {
        int rc = mnk_ls_fetch_info(h->ls_k[k]);
        if (rc) return rc;
        h->info_k[k] = h->ls_k[k]->info;
        const bool grouped = !h->ls_k[k]->bk_active && h->info_k[k] == 0 && h->ls_k[k]->algo == h->ls_k[0]->algo &&
                             h->ls_k[k]->ld == h->ls_k[0]->ld;
        (grouped ? fast : slow).push_back((int)k);
    }
    const int64_t Npb = h->Npb;
    // Phase 1b, fast path (static-pivot factor A_k = L D L' or L L'):  C A^-1 C' = (C L^-T) D^-1 (C L^-T)', so only the
    // FORWARD sweep is needed, as a right-side triangular solve of the nd rows of C_dk -- left-looking over the 64-column
    // blocks of L: an MFMA update with the finished blocks, then block substitution on MFMA against the diagonal block.
    // One launch of each per step for ALL scenarios of a chunk (round 3: ~20 launches per scenario, 13.3 ms for 16 of them;
    // four streams side by side: 5.2 ms; the host's launch rate was the bound).
    for (size_t c0 = 0; c0 < fast.size(); c0 += (size_t)h->chunk) {
        const int nf = (int)std::min<size_t>((size_t)h->chunk, fast.size() - c0);
        const int* fk = fast.data() + c0;
        const bool ldl = h->ls_k[fk[0]]->algo == MNK_LDL;
        const int nstep = (int)(Npb / NBI);
        // the records of this pass go through pinned host memory, laid out like the device buffer: one asynchronous copy (a
        // pageable source costs the launch mutex and a stream synchronization per upload -- three per pass)
        const size_t gr_off = (size_t)h->ns * sizeof(mnk::TrsmBatchRec);
        const size_t fk_off = gr_off + (size_t)nstep * h->chunk * sizeof(mnk::GemmBatchRec);
        const size_t need = fk_off + (size_t)h->chunk * sizeof(int);
        if (h->stage_ev == nullptr) MNK_HIP(hipEventCreateWithFlags(&h->stage_ev, hipEventDisableTiming));
        else MNK_HIP(hipEventSynchronize(h->stage_ev));   // (the previous pass' upload has left the staging memory)
        if (h->stage_bytes < need) {
            mnk::LaunchLock lock;
            if (h->stage) { mnk::quiesce_persistent(); (void)hipHostFree(h->stage); h->stage = nullptr; h->stage_bytes = 0; }
            MNK_HIP(hipHostMalloc((void**)&h->stage, need, hipHostMallocDefault));
            h->stage_bytes = need;
        }
        mnk::TrsmBatchRec* tr = reinterpret_cast<mnk::TrsmBatchRec*>(h->stage);
        mnk::GemmBatchRec* gr = reinterpret_cast<mnk::GemmBatchRec*>(h->stage + gr_off);
        int* fk_st = reinterpret_cast<int*>(h->stage + fk_off);
        for (int i = 0; i < nf; ++i) fk_st[i] = fk[i];
        for (int i = 0; i < nf; ++i) {
            mnk_ls* ls = h->ls_k[fk[i]];
            double* X = h->Xall.p + (size_t)i * ndp * Npb;
            double* V = ldl ? h->Vall.p + (size_t)i * ndp * Npb : X;
            tr[i] = {X, ldl ? V : nullptr, ls->dblk.p, ls->inv16.p, ls->dinv.p, ls->info_dev.p};
            for (int j = 1; j < nstep; ++j)   // X[:, block j] -= V[:, 0 : 64 j] L[block j, 0 : 64 j]'
                gr[(size_t)(j - 1) * nf + i] = {V, ls->fact.p + (int64_t)j * NBI, X + (int64_t)j * NBI * ndp, ls->info_dev.p};
            gr[(size_t)(nstep - 1) * nf + i] = {X, V, h->Pall.p + (size_t)i * ndp * ndp, ls->info_dev.p};   // P_i -= X V'
        }
        char* rdev = h->recs.p;
        mnk::TrsmBatchRec* tr_dev = reinterpret_cast<mnk::TrsmBatchRec*>(rdev);
        mnk::GemmBatchRec* gr_dev = reinterpret_cast<mnk::GemmBatchRec*>(rdev + (size_t)h->ns * sizeof(mnk::TrsmBatchRec));
        MNK_HIP(hipMemcpyAsync(rdev, h->stage, gr_off + (size_t)nstep * nf * sizeof(mnk::GemmBatchRec), hipMemcpyHostToDevice, s));
        MNK_HIP(hipMemcpyAsync(h->fast_k.p, fk_st, (size_t)nf * sizeof(int), hipMemcpyHostToDevice, s));
        MNK_HIP(hipEventRecord(h->stage_ev, s));
        hipLaunchKernelGGL(schur_copy_batch_kernel, dim3((unsigned)((ndp * Npb + 255) / 256), (unsigned)nf), dim3(256), 0, s,
                           h->Xall.p, ndp, Npb, h->C.p, nd, blk, h->fast_k.p);
        const int64_t ldf = h->ls_k[fk[0]]->ld;
        int rc = 0;
        for (int j = 0; j < nstep && !rc; ++j) {
            if (j > 0) rc = launch_gemm_nt_batch(s, ndp, NBI, (int64_t)j * NBI, gr_dev + (size_t)(j - 1) * nf, nf, ndp, ldf, ndp);
            if (!rc) rc = mnk_launch_trsm64_batch(s, ldl, tr_dev, nf, (int64_t)j * NBI, ndp, ndp);
        }
        if (rc) return rc;
        // Phase 2 (reference :993-999): S -= sum_k X_k V_k'
        MNK_HIP(hipMemsetAsync(h->Pall.p, 0, (size_t)nf * ndp * ndp * sizeof(double), s));
        rc = launch_gemm_nt_batch(s, ndp, ndp, Npb, gr_dev + (size_t)(nstep - 1) * nf, nf, ndp, ndp, ndp);
        if (rc) return rc;
        hipLaunchKernelGGL(schur_partial_sum_kernel, MNK_GRID1(ndp * ndp), h->Sp.p, h->Pall.p, (int64_t)nf, ndp * ndp);
        MNK_HIP(hipGetLastError());
    }
    for (int k : slow) {
        // Pivoted (Bunch-Kaufman tier) or failed factor: T_k = A_k^-1 C_dk' column by column, as the reference does
        mnk_ls* ls = h->ls_k[k];
        const double* Ck = h->C.p + (int64_t)k * nd * blk;
        double* Tk = h->T.p;
        hipLaunchKernelGGL(schur_copy_kernel, MNK_GRID1(blk * nd), Tk, blk, blk, nd, Ck, nd, nd, blk, 1);
        int rc = mnk_ls_solve(ls, Tk, nd, blk, MNK_DEVICE);
        if (!rc) rc = mnk_ls_check_solve(ls);   // (T_k feeds the Schur complement: an aborted solve must not get that far)
        if (rc) return rc;
        hipLaunchKernelGGL(schur_copy_kernel, MNK_GRID1(ndp * blkp), h->Cp.p, ndp, ndp, blkp, Ck, nd, nd, blk, 0);
        hipLaunchKernelGGL(schur_copy_kernel, MNK_GRID1(ndp * blkp), h->Tt.p, ndp, ndp, blkp, Tk, blk, blk, nd, 1);
        MNK_HIP(hipGetLastError());
        rc = launch_gemm_nt(s, 0, ndp, ndp, blkp, h->Cp.p, ndp, h->Tt.p, ndp, h->Sp.p, ndp, nullptr, nullptr, 0, nullptr);
        if (rc) return rc;
    }
    MNK_HIP(hipMemcpy2DAsync(S_out, lds_out * sizeof(double), h->Sp.p, ndp * sizeof(double), nd * sizeof(double), nd,
                             hipMemcpyDeviceToDevice, s));
    h->built = true;
    return 0;
}

int mnk_schur_factorize_s(mnk_schur* h, const double* S, int64_t lds, int loc, int* info) {
    MNK_REQUIRE(h && S, "mnk_schur_factorize_s: NULL argument");
    return mnk_ls_factorize_dense(h->ls_s, S, lds, loc, info);
}

int mnk_schur_inertia_s(mnk_schur* h, int64_t* num_pos, int64_t* num_zero, int64_t* num_neg) {
    MNK_REQUIRE(h, "mnk_schur_inertia_s: NULL argument");
    return mnk_ls_inertia(h->ls_s, num_pos, num_zero, num_neg);
}

int mnk_schur_scenario_inertia(mnk_schur* h, int64_t k, int64_t* num_pos, int64_t* num_zero, int64_t* num_neg) {
    MNK_REQUIRE(h && k >= 0 && k < h->ns, "mnk_schur_scenario_inertia: bad argument");
    return mnk_ls_inertia(h->ls_k[k], num_pos, num_zero, num_neg);
}

// The solves above run on device-resident vectors and return before the device is done.  A one-launch (persistent) solve
// that gives up raises a pinned abort word: every stage checks its solvers once at its end (one stream synchronization) and
// reports the failure now -- the solver has then switched to the stepwise solve, the caller repeats the stage.
static int schur_check_solves(mnk_schur* h, bool scenarios, bool design) {
    if (scenarios)
        for (mnk_ls* l : h->ls_k) {
            int rc = mnk_ls_check_solve(l);
            if (rc) return rc;
        }
    if (design && h->ls_s) return mnk_ls_check_solve(h->ls_s);
    return 0;
}

// Step 3 of solve_kkt! (reference :1040-1049): r_k <- A_k^-1 r_k for the local scenarios, and this rank's
// contribution  -sum_k C_dk r_k  to the design right-hand side (the caller adds r_d and all-reduces).
int mnk_schur_forward(mnk_schur* h, double* rhs_k, double* contrib_d) {
    MNK_REQUIRE(h && contrib_d && (rhs_k || h->ns == 0), "mnk_schur_forward: NULL argument");
    MNK_REQUIRE(h->built, "mnk_schur_forward: call mnk_schur_build_local first");
    MNK_HIP(hipSetDevice(h->ctx->device));
    hipStream_t s = h->ctx->stream;
    if (h->ns == 0) MNK_HIP(hipMemsetAsync(contrib_d, 0, h->nd * sizeof(double), s));
    {   // the scenarios' solves as ONE batch (solve.hip: up to 32 small systems per launch), then the contributions in order
        int rc = mnk_solve_batch_begin();
        for (int64_t k = 0; k < h->ns && !rc; ++k) rc = mnk_ls_solve(h->ls_k[k], rhs_k + k * h->blk, 1, h->blk, MNK_DEVICE);
        int rc_end = mnk_solve_batch_end();
        // (inside a batch the CALLER opened the end above is a no-op: the contributions below need the solutions now)
        if (!rc && !rc_end) rc_end = mnk_solve_batch_flush_pending();
        if (rc || rc_end) return rc ? rc : rc_end;
    }
    if (h->ns > 0) {
        hipLaunchKernelGGL(schur_gemv_batch_kernel, dim3((unsigned)((h->nd + 255) / 256), (unsigned)h->ns), dim3(256), 0, s, h->Pall.p,
                           h->C.p, h->nd, rhs_k, h->blk);
        hipLaunchKernelGGL(schur_contrib_sum_kernel, MNK_GRID1(h->nd), contrib_d, h->Pall.p, h->ns, h->nd, -1.0);
    }
    MNK_HIP(hipGetLastError());
    return schur_check_solves(h, true, false);
}

// Step 4: S x_d = r_d (in place, device vector)
int mnk_schur_solve_s(mnk_schur* h, double* rhs_d) {
    MNK_REQUIRE(h && rhs_d, "mnk_schur_solve_s: NULL argument");
    int rc = mnk_ls_solve(h->ls_s, rhs_d, 1, h->nd, MNK_DEVICE);
    return rc ? rc : schur_check_solves(h, false, true);
}

// Step 5 (reference :1055-1058): x_k = r_k - (A_k^-1 C_dk') x_d, applied as one more solve with A_k on C_dk' x_d
// (the reference multiplies by the stored blk x nd matrix A_k^-1 C_dk'; the fast build path never forms it).
int mnk_schur_backward(mnk_schur* h, double* rhs_k, const double* x_d) {
    MNK_REQUIRE(h && x_d && (rhs_k || h->ns == 0), "mnk_schur_backward: NULL argument");
    MNK_HIP(hipSetDevice(h->ctx->device));
    hipStream_t s = h->ctx->stream;
    if (h->ns > 0)
        hipLaunchKernelGGL(schur_gemv_t_batch_kernel, dim3((unsigned)((h->blk + 255) / 256), (unsigned)h->ns), dim3(256), 0, s, h->tmpk.p,
                           h->Npb, h->C.p, h->nd, x_d, h->blk);
    {
        int rc = mnk_solve_batch_begin();
        for (int64_t k = 0; k < h->ns && !rc; ++k) rc = mnk_ls_solve(h->ls_k[k], h->tmpk.p + k * h->Npb, 1, h->blk, MNK_DEVICE);
        int rc_end = mnk_solve_batch_end();
        if (!rc && !rc_end) rc_end = mnk_solve_batch_flush_pending();   // (as in the forward stage)
        if (rc || rc_end) return rc ? rc : rc_end;
    }
    if (h->ns > 0)
        hipLaunchKernelGGL(schur_sub_batch_kernel, dim3((unsigned)((h->blk + 255) / 256), (unsigned)h->ns), dim3(256), 0, s, rhs_k, h->blk,
                           h->tmpk.p, h->Npb);
    MNK_HIP(hipGetLastError());
    return schur_check_solves(h, true, false);
}

// A device buffer of nd x nd doubles owned by the handle: `S_out` of mnk_schur_build_local and `S` of mnk_schur_factorize_s for
// callers that keep no device memory themselves (the Julia glue's host-driven KKT system, julia/MadNLPHIP.jl).
int mnk_schur_set_structure(mnk_schur* h, int64_t n, int64_t m, int64_t nv, int64_t nc, int64_t nnzh, const int32_t* hess_I,
                            const int32_t* hess_J, int64_t nnzj, const int32_t* jac_I, const int32_t* jac_J, int64_t n_ineq,
                            const int64_t* ind_ineq, int64_t n_eq, const int64_t* ind_eq, int index_base, int64_t ns_global,
                            const int64_t* local_scen, int own_design) {
    MNK_REQUIRE(h, "mnk_schur_set_structure: NULL handle");
    MNK_REQUIRE(index_base == 0 || index_base == 1, "mnk_schur_set_structure: index_base must be 0 or 1");
    MNK_REQUIRE((nnzh == 0 || (hess_I && hess_J)) && (nnzj == 0 || (jac_I && jac_J)) && (n_ineq == 0 || ind_ineq) && (n_eq == 0 || ind_eq),
                "mnk_schur_set_structure: NULL pattern");
    const int64_t ns = local_scen ? ns_global : h->ns, nd = h->nd, blk = h->blk;
    MNK_REQUIRE(ns >= h->ns && nv > 0 && nc >= 0, "mnk_schur_set_structure: bad dimensions");
    MNK_REQUIRE(n == ns * nv + nd && m == ns * nc, "mnk_schur_set_structure: n, m do not match ns * nv + nd, ns * nc");
    MNK_REQUIRE(n_eq + n_ineq == m && n_eq % std::max<int64_t>(ns, 1) == 0 && n_ineq % std::max<int64_t>(ns, 1) == 0,
                "mnk_schur_set_structure: the scenarios have different numbers of equality / inequality constraints");
    const int64_t nc_eq = ns > 0 ? n_eq / ns : 0;
    MNK_REQUIRE(blk == nv + nc_eq, "mnk_schur_set_structure: the stage's block order is not nv + (equality rows per scenario)");
    MNK_REQUIRE(nnzh < 2000000000 && nnzj < 2000000000 && n + n_ineq < 2000000000, "mnk_schur_set_structure: structure exceeds int32 range");
    MNK_HIP(hipSetDevice(h->ctx->device));
    const int64_t off = ns * nv;
    // global scenario -> local block (or -1)
    std::vector<int64_t> loc_of(ns, -1);
    for (int64_t i = 0; i < h->ns; ++i) {
        const int64_t g = local_scen ? local_scen[i] : i;
        MNK_REQUIRE(g >= 0 && g < ns && loc_of[g] < 0, "mnk_schur_set_structure: bad local scenario list");
        loc_of[g] = i;
    }
    // equality rows: local index = rank among the scenario's equality rows (ascending); inequality rows: slack index
    std::vector<int64_t> eq_local(m, -1), slack_of(m, -1);
    {
        std::vector<int64_t> eq_sorted(ind_eq, ind_eq + n_eq);
        for (auto& r : eq_sorted) r -= index_base;
        std::sort(eq_sorted.begin(), eq_sorted.end());
        std::vector<int64_t> cnt(ns, 0);
        for (int64_t r : eq_sorted) {
            MNK_REQUIRE(r >= 0 && r < m, "mnk_schur_set_structure: equality index out of range");
            const int64_t k = nc > 0 ? r / nc : 0;
            eq_local[r] = cnt[k]++;
        }
        for (int64_t k = 0; k < ns; ++k)
            MNK_REQUIRE(cnt[k] == nc_eq, "mnk_schur_set_structure: the scenarios have different numbers of equality / inequality constraints");
        for (int64_t p = 0; p < n_ineq; ++p) {
            const int64_t r = ind_ineq[p] - index_base;
            MNK_REQUIRE(r >= 0 && r < m && eq_local[r] < 0 && slack_of[r] < 0, "mnk_schur_set_structure: bad inequality index");
            slack_of[r] = p;
        }
    }
    auto scen = [&](int64_t var) -> int64_t { return var < off ? var / nv : -1; };
    struct Src { int64_t dst; int8_t kind; int32_t a, b, w; };
    std::vector<Src> src;
    const int64_t sizeA = h->ns * blk * blk, sizeC = h->ns * nd * blk;
    auto putA = [&](int64_t kl, int64_t r, int64_t c, int8_t kind, int64_t a, int64_t b = 0, int64_t w = 0) {
        src.push_back(Src{kl * blk * blk + r + c * blk, kind, (int32_t)a, (int32_t)b, (int32_t)w});
    };
    auto putC = [&](int64_t kl, int64_t d, int64_t c, int8_t kind, int64_t a, int64_t b = 0, int64_t w = 0) {
        src.push_back(Src{sizeA + kl * nd * blk + d + c * nd, kind, (int32_t)a, (int32_t)b, (int32_t)w});
    };
    auto putS = [&](int64_t r, int64_t c, int8_t kind, int64_t a, int64_t b = 0, int64_t w = 0) {
        src.push_back(Src{sizeA + sizeC + r + c * nd, kind, (int32_t)a, (int32_t)b, (int32_t)w});
    };
    // Hessian entries (forced to the lower triangle, matrixtools.jl:129-137): scenario block, coupling block, design block
    for (int64_t e = 0; e < nnzh; ++e) {
        int64_t i = hess_I[e] - index_base, j = hess_J[e] - index_base;
        MNK_REQUIRE(i >= 0 && i < n && j >= 0 && j < n, "mnk_schur_set_structure: Hessian index out of range");
        if (j > i) std::swap(i, j);
        const int64_t si = scen(i), sj = scen(j);
        if (si >= 0 && sj >= 0) {
            MNK_REQUIRE(si == sj, "mnk_schur_set_structure: a Hessian entry couples two scenarios");
            const int64_t kl = loc_of[si];
            if (kl < 0) continue;
            const int64_t li = i - si * nv, lj = j - sj * nv;
            putA(kl, li, lj, 0, e);
            if (li != lj) putA(kl, lj, li, 0, e);
        } else if (si < 0 && sj < 0) {
            if (!own_design) continue;
            putS(i - off, j - off, 0, e);
            if (i != j) putS(j - off, i - off, 0, e);
        } else {   // (lower triangle: the design variable is the row)
            const int64_t kl = loc_of[sj];
            if (kl >= 0) putC(kl, i - off, j - sj * nv, 0, e);
        }
    }
    // diagonal terms
    for (int64_t k = 0; k < ns; ++k) {
        const int64_t kl = loc_of[k];
        if (kl < 0) continue;
        for (int64_t j = 0; j < nv; ++j) putA(kl, j, j, 2, k * nv + j);
    }
    for (int64_t r = 0; r < m; ++r)
        if (eq_local[r] >= 0 && loc_of[r / nc] >= 0) putA(loc_of[r / nc], nv + eq_local[r], nv + eq_local[r], 3, r);
    if (own_design)
        for (int64_t j = 0; j < nd; ++j) putS(j, j, 2, off + j);
    // Jacobian entries of equality rows; the entries of every inequality row in COO order
    std::vector<std::vector<std::pair<int32_t, int64_t>>> row_entries(m);
    for (int64_t e = 0; e < nnzj; ++e) {
        const int64_t r = jac_I[e] - index_base, c = jac_J[e] - index_base;
        MNK_REQUIRE(r >= 0 && r < m && c >= 0 && c < n, "mnk_schur_set_structure: Jacobian index out of range");
        const int64_t k = r / nc;
        MNK_REQUIRE(c >= off || c / nv == k, "mnk_schur_set_structure: a constraint reaches the variables of another scenario");
        const int64_t kl = loc_of[k];
        if (kl < 0) continue;
        if (eq_local[r] >= 0) {
            const int64_t li = eq_local[r];
            if (c < off) { putA(kl, nv + li, c - k * nv, 1, e); putA(kl, c - k * nv, nv + li, 1, e); }
            else putC(kl, c - off, nv + li, 1, e);
        } else {
            row_entries[r].push_back({(int32_t)e, c});
        }
    }
    // inequality rows: J' D J over A_k, C_dk and S0, one pair of entries at a time (_scatter_quad_add!, schur.jl:966-972)
    for (int64_t p = 0; p < n_ineq; ++p) {
        const int64_t r = ind_ineq[p] - index_base, k = r / nc, kl = loc_of[k];
        if (kl < 0) continue;
        const auto& ents = row_entries[r];
        for (const auto& e1 : ents)
            for (const auto& e2 : ents) {
                const int64_t c1 = e1.second, c2 = e2.second;
                if (c1 < off && c2 < off) putA(kl, c1 - k * nv, c2 - k * nv, 4, e1.first, e2.first, p);
                else if (c1 >= off && c2 < off) putC(kl, c1 - off, c2 - k * nv, 4, e1.first, e2.first, p);
                else if (c1 >= off && c2 >= off) putS(c1 - off, c2 - off, 4, e1.first, e2.first, p);
            }
    }
    MNK_REQUIRE(src.size() < (size_t)2000000000, "mnk_schur_set_structure: too many assembly terms");
    std::stable_sort(src.begin(), src.end(), [](const Src& x, const Src& y) { return x.dst < y.dst; });
    std::vector<int64_t> seg_dst;
    std::vector<int32_t> seg_ptr, sa(src.size()), sb(src.size()), sw(src.size());
    std::vector<int8_t> kind(src.size());
    for (size_t q = 0; q < src.size(); ++q) {
        if (q == 0 || src[q].dst != src[q - 1].dst) { seg_dst.push_back(src[q].dst); seg_ptr.push_back((int32_t)q); }
        kind[q] = src[q].kind; sa[q] = src[q].a; sb[q] = src[q].b; sw[q] = src[q].w;
    }
    seg_ptr.push_back((int32_t)src.size());
    auto& as = h->as;
    as.have = false;
    as.n = n; as.m = m; as.nv = nv; as.nc = nc; as.nnzh = nnzh; as.nnzj = nnzj; as.n_ineq = n_ineq;
    as.nseg = (int64_t)seg_dst.size(); as.nsrc = (int64_t)src.size();
    std::vector<int64_t> ineq0(n_ineq);
    for (int64_t p = 0; p < n_ineq; ++p) ineq0[p] = ind_ineq[p] - index_base;
    hipStream_t st = h->ctx->stream;
    int rc = as.seg_dst.upload(seg_dst, st) | as.seg_ptr.upload(seg_ptr, st) | as.src_kind.upload(kind, st) | as.src_a.upload(sa, st) |
             as.src_b.upload(sb, st) | as.src_w.upload(sw, st) | as.ind_ineq.upload(ineq0, st) | as.D.alloc((size_t)std::max<int64_t>(n_ineq, 1)) |
             as.S0.alloc((size_t)nd * nd + 8) | as.vals.alloc((size_t)(nnzh + nnzj + n + n_ineq + m) + 8);
    if (rc) { (void)hipGetLastError(); set_error("mnk_schur_set_structure: out of device memory"); return -2; }
    as.have = true;
    return 0;
}

int mnk_schur_assemble(mnk_schur* h, const double* hess, const double* jac, const double* pr_diag, const double* du_diag, int loc) {
    MNK_REQUIRE(h && h->as.have, "mnk_schur_assemble: call mnk_schur_set_structure first");
    auto& as = h->as;
    MNK_REQUIRE((hess || as.nnzh == 0) && (jac || as.nnzj == 0) && pr_diag && (du_diag || as.m == 0), "mnk_schur_assemble: NULL argument");
    MNK_HIP(hipSetDevice(h->ctx->device));
    hipStream_t s = h->ctx->stream;
    if (loc != MNK_DEVICE) {   // host callbacks: one staging buffer, four copies
        double* v = as.vals.p;
        if (as.nnzh > 0) MNK_HIP(mnk::h2d_copy(v, hess, (size_t)as.nnzh * sizeof(double), s));
        if (as.nnzj > 0) MNK_HIP(mnk::h2d_copy(v + as.nnzh, jac, (size_t)as.nnzj * sizeof(double), s));
        MNK_HIP(mnk::h2d_copy(v + as.nnzh + as.nnzj, pr_diag, (size_t)(as.n + as.n_ineq) * sizeof(double), s));
        if (as.m > 0) MNK_HIP(mnk::h2d_copy(v + as.nnzh + as.nnzj + as.n + as.n_ineq, du_diag, (size_t)as.m * sizeof(double), s));
        hess = v; jac = v + as.nnzh; pr_diag = v + as.nnzh + as.nnzj; du_diag = v + as.nnzh + as.nnzj + as.n + as.n_ineq;
    }
    const int64_t sizeA = h->ns * h->blk * h->blk, sizeC = h->ns * h->nd * h->blk;
    if (sizeA > 0) MNK_HIP(hipMemsetAsync(h->A.p, 0, (size_t)sizeA * sizeof(double), s));
    if (sizeC > 0) MNK_HIP(hipMemsetAsync(h->C.p, 0, (size_t)sizeC * sizeof(double), s));
    MNK_HIP(hipMemsetAsync(as.S0.p, 0, (size_t)h->nd * h->nd * sizeof(double), s));
    if (as.n_ineq > 0)
        hipLaunchKernelGGL(mnk::schur_condense_weights_kernel, dim3((unsigned)((as.n_ineq + 255) / 256)), dim3(256), 0, s, as.D.p, pr_diag, du_diag,
                           as.ind_ineq.p, as.n, as.n_ineq);
    if (as.nseg > 0)
        hipLaunchKernelGGL(mnk::schur_assemble_kernel, dim3((unsigned)((as.nseg + 255) / 256)), dim3(256), 0, s, as.nseg, as.seg_dst.p, as.seg_ptr.p,
                           as.src_kind.p, as.src_a.p, as.src_b.p, as.src_w.p, hess, jac, pr_diag, du_diag, as.D.p, h->A.p, sizeA, h->C.p, sizeC,
                           as.S0.p);
    MNK_HIP(hipGetLastError());
    h->built = false;
    return 0;
}

void* mnk_schur_s0_buffer(mnk_schur* h) { return (h && h->as.have) ? (void*)h->as.S0.p : nullptr; }

int mnk_schur_get_block(mnk_schur* h, int64_t k, double* A_kk, double* C_dk, double* S0) {
    MNK_REQUIRE(h && ((k >= 0 && k < h->ns) || (!A_kk && !C_dk)) && (!S0 || h->as.have), "mnk_schur_get_block: bad argument");
    MNK_HIP(hipSetDevice(h->ctx->device));
    hipStream_t s = h->ctx->stream;
    if (A_kk) MNK_HIP(mnk::d2h_copy(A_kk, h->A.p + k * h->blk * h->blk, (size_t)h->blk * h->blk * sizeof(double), s));
    if (C_dk) MNK_HIP(mnk::d2h_copy(C_dk, h->C.p + k * h->nd * h->blk, (size_t)h->nd * h->blk * sizeof(double), s));
    if (S0) MNK_HIP(mnk::d2h_copy(S0, h->as.S0.p, (size_t)h->nd * h->nd * sizeof(double), s));
    MNK_HIP(mnk::stream_wait(s));
    return 0;
}

void* mnk_schur_s_buffer(mnk_schur* h) {
    if (!h) return nullptr;
    (void)hipSetDevice(h->ctx->device);
    if (!h->Sown.p && h->Sown.alloc((size_t)h->nd * h->nd + 8)) { (void)hipGetLastError(); return nullptr; }
    return h->Sown.p;
}

// Steps 3-5 of solve_kkt! in one call on ONE rank (reference :1078-1092): rhs_k (ns x blk, scenario k's vector at k * blk) and
// rhs_d (nd) are overwritten with the solution; host or device vectors.  (Several ranks: forward / all-reduce / solve_s /
// backward separately, as above.)
int mnk_schur_solve(mnk_schur* h, double* rhs_k, double* rhs_d, int loc) {
    MNK_REQUIRE(h && rhs_d && (rhs_k || h->ns == 0), "mnk_schur_solve: NULL argument");
    MNK_HIP(hipSetDevice(h->ctx->device));
    hipStream_t s = h->ctx->stream;
    const size_t nk = (size_t)h->ns * h->blk;
    if (!h->hostrd.p && h->hostrd.alloc(2 * (size_t)h->nd + 8)) return -2;
    double* rk = rhs_k;
    double* rd = rhs_d;
    double* contrib = h->hostrd.p + h->nd;
    if (loc != MNK_DEVICE) {
        if (nk > 0 && !h->hostrk.p && h->hostrk.alloc(nk + 8)) return -2;
        rk = h->hostrk.p;
        rd = h->hostrd.p;
        if (nk > 0) MNK_HIP(mnk::h2d_copy(rk, rhs_k, nk * sizeof(double), s));
        MNK_HIP(mnk::h2d_copy(rd, rhs_d, (size_t)h->nd * sizeof(double), s));
    }
    int rc = mnk_schur_forward(h, rk, contrib);
    if (rc) return rc;
    hipLaunchKernelGGL(schur_axpy_kernel, dim3((unsigned)((h->nd + 255) / 256)), dim3(256), 0, s, rd, contrib, h->nd);
    MNK_HIP(hipGetLastError());
    rc = mnk_schur_solve_s(h, rd);
    if (!rc) rc = mnk_schur_backward(h, rk, rd);
    if (rc) return rc;
    if (loc != MNK_DEVICE) {
        if (nk > 0) MNK_HIP(mnk::d2h_copy(rhs_k, rk, nk * sizeof(double), s));
        MNK_HIP(mnk::d2h_copy(rhs_d, rd, (size_t)h->nd * sizeof(double), s));
    }
    return 0;
}

#undef MNK_GRID1

}  // extern "C"
