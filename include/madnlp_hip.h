/*
 * madnlp_hip.h -- C ABI of libmadnlp_hip.so: the MI355X (gfx950) implementation of
 * MadNLP's per-iteration KKT hot path.
 *
 * This is the drop-in boundary: every entry point below replaces a function (or a
 * group of functions) of MadNLP.jl v0.10.1; the reference file:line each one stands
 * in for is cited next to it.  A Julia maintainer binds these with `ccall`
 * exactly as `src/LinearSolvers/lapack.jl:50-139` binds LAPACK and
 * `src/LinearSolvers/mumps.jl:148-165` binds MUMPS.
 *
 * Conventions
 *   - plain C types only: int32/int64/double pointers and sizes, opaque handles.
 *   - return value: 0 = OK; < 0 = bad argument / HIP runtime error (text via
 *     mnk_last_error_string(); the Julia glue throws SymbolicException /
 *     FactorizationException / SolveException, reference
 *     `src/LinearSolvers/linearsolvers.jl:133-137`).  A *numerically* failed
 *     factorization is NOT an error: it is reported through `info` / the
 *     inertia, as `src/LinearSolvers/lapack_common.jl:96-102` does, so that the
 *     IPM regularizes and refactorizes.
 *   - `loc` arguments say where a caller buffer lives: MNK_HOST (pageable or
 *     pinned host memory, copied over PCIe on the context's stream) or
 *     MNK_DEVICE (HBM of the context's device; no copy).
 *   - index arrays passed in use `index_base` 0 or 1 (Julia passes 1); arrays
 *     handed back are 0-based unless stated.
 *   - all matrices are column-major Float64, symmetric matrices use the lower
 *     triangle ('L'), as everywhere in the reference.
 *   - all work is enqueued on the context's stream; entry points that return
 *     scalars to the host (info, inertia, host-destination copies) synchronize
 *     that stream, everything else is asynchronous.
 *   - one context / KKT system / solver per MadNLPSolver, driven from one
 *     thread; distinct handles may be used concurrently from distinct threads.
 */
#ifndef MADNLP_HIP_H
#define MADNLP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MNK_VERSION 100 /* 0.1.0 */

enum { MNK_HOST = 0, MNK_DEVICE = 1 };

/* Same order as the reference's `@enum LinearFactorization`
 * (`src/LinearSolvers/linearsolvers.jl:139-147`): BUNCHKAUFMAN LU QR CHOLESKY LDL EVD.
 * Implemented on device: CHOLESKY (dpotrf semantics) and LDL (static-pivot
 * LDL^T, inertia from sign(D); stands in for BUNCHKAUFMAN = dsytrf). */
enum { MNK_BUNCHKAUFMAN = 1, MNK_LU = 2, MNK_QR = 3, MNK_CHOLESKY = 4, MNK_LDL = 5, MNK_EVD = 6 };

typedef struct mnk_ctx mnk_ctx; /* device, stream(s), scratch */
typedef struct mnk_sc mnk_sc;   /* SparseCondensedKKTSystem device state */
typedef struct mnk_dc mnk_dc;   /* DenseCondensedKKTSystem / DenseKKTSystem device state */
typedef struct mnk_ls mnk_ls;   /* AbstractLinearSolver device state */

/* ---------------------------------------------------------------- context -- */
int mnk_version(void);
const char* mnk_last_error_string(void);
/* `stream`: an existing hipStream_t to enqueue on (e.g. the caller's current
 * stream), or NULL to let the context create its own. */
int mnk_ctx_create(int device, void* stream, mnk_ctx** out);
int mnk_ctx_destroy(mnk_ctx* ctx);
int mnk_ctx_synchronize(mnk_ctx* ctx);
void* mnk_ctx_stream(mnk_ctx* ctx);

/* ------------------------------------------- SparseCondensedKKTSystem (sc) -- */
/* Replaces the constructor `create_kkt_system(::Type{SparseCondensedKKTSystem}, ...)`
 * reference `src/KKT/Sparse/condensed.jl:55-133`:
 *   force_lower_triangular! (`src/matrixtools.jl:129-137`), coo_to_csc + get_mapping
 *   for J^T and tril(H) (`src/matrixtools.jl:55-95`) and the one-time symbolic
 *   analysis build_condensed_aug_symbolic (`condensed.jl:201-301`), all on the host
 *   in C++ (integer work), then uploads the maps.
 * jac_I/jac_J: COO pattern of the Jacobian (row = constraint, col = variable),
 * hess_I/hess_J: COO pattern of the Lagrangian Hessian (either triangle).
 * All m constraints are inequalities (the reference errors otherwise, `:68-70`).
 * ctx may be NULL: the handle is then host-only (symbolic analysis and the structure
 * getters work, nothing is allocated on a device) -- used by CPU-side tests. */
int mnk_sc_create(mnk_ctx* ctx, int64_t n, int64_t m,
                  int64_t nnzj, const int32_t* jac_I, const int32_t* jac_J,
                  int64_t nnzh, const int32_t* hess_I, const int32_t* hess_J,
                  int index_base, mnk_sc** out);
int mnk_sc_destroy(mnk_sc* sc);

/* Sizes of the derived structures: nnz(jt_csc), nnz(hess_com), nnz(aug_com), length(jptr). */
int mnk_sc_sizes(mnk_sc* sc, int64_t* nnz_jt, int64_t* nnz_hess, int64_t* nnz_aug, int64_t* len_jptr);
/* Derived CSC structures (0-based), so the host mirror can build `jt_csc`,
 * `hess_com`, `aug_com` (`condensed.jl:110-119`). which: 0 = jt_csc (n x m),
 * 1 = hess_com (n x n lower), 2 = aug_com (n x n lower). colptr has ncol+1 entries. */
enum { MNK_SC_JT = 0, MNK_SC_HESS = 1, MNK_SC_AUG = 2, MNK_SC_DIAGBUF = 3 };
int mnk_sc_get_structure(mnk_sc* sc, int which, int32_t* colptr, int32_t* rowval);
/* COO -> CSC slot maps `jt_csc_map` / `hess_csc_map` (0-based; which = MNK_SC_JT / MNK_SC_HESS). */
int mnk_sc_get_map(mnk_sc* sc, int which, int64_t* map);
/* dptr/hptr/jptr of `build_condensed_aug_symbolic` (0-based), in the reference's order:
 * dptr: dst[n], src[n]; hptr: dst[nnzH], src[nnzH]; jptr: dst[L], c[L], k[L], l[L].
 * Any output pointer may be NULL. */
int mnk_sc_get_ptrs(mnk_sc* sc, int32_t* d_dst, int32_t* d_src, int32_t* h_dst, int32_t* h_src,
                    int32_t* j_dst, int32_t* j_c, int32_t* j_k, int32_t* j_l);

/* compress_jacobian!(::SparseCondensedKKTSystem) `condensed.jl:145-148` = transfer!
 * (`src/matrixtools.jl:79-88`) of the nnzj COO values into jt_csc.nzval. */
int mnk_sc_compress_jacobian(mnk_sc* sc, const double* jac_coo, int loc);
/* compress_hessian!(::AbstractSparseKKTSystem) `src/KKT/Sparse/utils.jl:48-50`. */
int mnk_sc_compress_hessian(mnk_sc* sc, const double* hess_coo, int loc);
/* build_kkt!(::SparseCondensedKKTSystem) `condensed.jl:354-366` +
 * _build_condensed_aug_coord! `:328-345`.  pr_diag has n+m entries, du_diag m. */
int mnk_sc_build(mnk_sc* sc, const double* pr_diag, const double* du_diag, int loc);
/* Read back nzval of jt_csc / hess_com / aug_com, or diag_buffer (which = MNK_SC_*). */
int mnk_sc_get_values(mnk_sc* sc, int which, double* out, int loc);

/* Device-side pieces of solve_kkt!/mul! for the sparse condensed system (reference
 * `src/IPM/factorization.jl:143-167,278-299`): y = alpha*op(A)*x + beta*y with
 * A = jt_csc (which = MNK_SC_JT; trans = 0: n<-m, trans = 1: m<-n) or
 * Symmetric(hess_com, :L) (which = MNK_SC_HESS).  x, y in device memory. */
int mnk_sc_spmv(mnk_sc* sc, int which, int trans, double alpha, const double* x, double beta, double* y);

/* ------------------- DenseCondensedKKTSystem / DenseKKTSystem (dc) ---------- */
/* Replaces `create_kkt_system(::Type{DenseCondensedKKTSystem}, ...)`
 * `src/KKT/Dense/condensed.jl:52-111` (condensed = 1, order n + n_eq) and
 * `create_kkt_system(::Type{DenseKKTSystem}, ...)` `src/KKT/Dense/augmented.jl:41-94`
 * (condensed = 0, order n + ns + m).  ind_ineq has ns entries, ind_eq m - ns. */
int mnk_dc_create(mnk_ctx* ctx, int condensed, int64_t n, int64_t m,
                  int64_t ns, const int64_t* ind_ineq, const int64_t* ind_eq,
                  int index_base, mnk_dc** out);
int mnk_dc_destroy(mnk_dc* dc);
/* The callbacks write kkt.hess (n x n) / kkt.jac (m x n) in place in the reference
 * (`get_hessian/get_jacobian`); here they are uploaded (or copied device to device). */
int mnk_dc_set_hess(mnk_dc* dc, const double* hess, int64_t ld, int loc);
int mnk_dc_set_jac(mnk_dc* dc, const double* jac, int64_t ld, int loc);
/* build_kkt!(::DenseCondensedKKTSystem) `condensed.jl:157-186` (_build_ineq_jac!,
 * the J_i' D J_i product, _build_condensed_kkt_system!) or, for condensed = 0,
 * compress_hessian! + build_kkt!(::DenseKKTSystem) `augmented.jl:116-161`.
 * pr_diag has n+ns entries, du_diag m.  Both triangles of aug_com are written. */
int mnk_dc_build(mnk_dc* dc, const double* pr_diag, const double* du_diag, int loc);
int64_t mnk_dc_order(mnk_dc* dc);
/* Copy aug_com (order x order, column-major, ld = order) out. */
int mnk_dc_get_aug(mnk_dc* dc, double* out, int loc);

/* ------------------------------------------------- linear solver (ls) ------- */
/* Replaces `LapackCPUSolver(A; opt)` `src/LinearSolvers/lapack.jl:21-43` /
 * `LapackROCmSolver` `lib/MadNLPGPU/ext/MadNLPGPUAMDGPUExt/rocsolver.jl`:
 * allocates the private N x N factor buffer.  algo = MNK_CHOLESKY or MNK_LDL
 * (MNK_BUNCHKAUFMAN is accepted as an alias of MNK_LDL). */
int mnk_ls_create(mnk_ctx* ctx, int64_t N, int algo, mnk_ls** out);
int mnk_ls_destroy(mnk_ls* ls);
/* Options: "pivot_tol" (LDL: |d| <= pivot_tol counts as a zero pivot; default 0),
 * "outer_block" (outer panel width, multiple of 64; default 0 = by size: 512, and 1024 from 32 768 rows on),
 * "lookahead" (0/1; default 1: factor the next panel while the trailing update runs),
 * "probe" (0/1; default 1, effective while "early_reject" and "accept_only_pd" are set: after an early rejection that stopped in the
 * first half of the columns, a matrix from the same KKT handle that follows an ACCEPTED one is first factorized on its leading
 * principal block alone -- a child solver of that order; a block that is not positive definite settles the verdict at a few
 * percent of a factorization's work; statistics "probe_hits" / "probe_misses"). */
int mnk_ls_set_option(mnk_ls* ls, const char* key, double value);

/* factorize!(M) `src/LinearSolvers/lapack_common.jl:54-66` = transfer_matrix!
 * (`:28`; CSC -> dense zero-fill + scatter, device twin
 * `lib/MadNLPGPU/src/utils.jl:12-23`) + dpotrf('L') / dsytrf('L')
 * (`lapack.jl:145-148,164-167`).  The source is named explicitly:
 *   _sc    : aug_com of a sparse condensed system (CSC on device);
 *   _dc    : aug_com of a dense system (on device);
 *   _dense : caller's dense N x N matrix (host or device), lower triangle read;
 *   _csc   : caller's lower-triangular CSC (host), 0/1-based.
 * *info = 0 on success, k > 0 if pivot k (1-based) is not positive (CHOLESKY) or
 * is zero (LDL).  Synchronizes the stream (info is read back). */
int mnk_ls_factorize_sc(mnk_ls* ls, mnk_sc* sc, int* info);
int mnk_ls_factorize_dc(mnk_ls* ls, mnk_dc* dc, int* info);
int mnk_ls_factorize_dense(mnk_ls* ls, const double* A, int64_t lda, int loc, int* info);
int mnk_ls_factorize_csc(mnk_ls* ls, const int32_t* colptr, const int32_t* rowval,
                         const double* nzval, int index_base, int* info);
/* Asynchronous variants: enqueue only, info/inertia are fetched later with
 * mnk_ls_inertia (which synchronizes). */
int mnk_ls_factorize_sc_async(mnk_ls* ls, mnk_sc* sc);
int mnk_ls_factorize_dc_async(mnk_ls* ls, mnk_dc* dc);
/* Batches of INDEPENDENT factorizations (scenario batches, BASELINE config C5; the reference drives distinct solver
 * instances concurrently, `src/KKT/Schur/schur.jl:927-1001` `@blas_safe_threads for k in 1:ns`).  Between _begin and _end
 * the factorize! calls of the calling thread (any solver, any context of one device) transfer their matrices and are
 * queued; _end launches them together: the task queues of instances of the same order are merged into one persistent
 * launch beside two pivot chains, so that one instance's chain-bound ends are filled with its neighbours' trailing
 * updates (N = 11 192: ~10.7 -> ~9.x ms per instance).  Every instance's factor is bit-identical to the one a lone
 * factorize! produces.  Any call that needs a queued solver's factor (inertia, solve, another factorize!, destroy)
 * launches what is queued first, so a forgotten _end cannot give a stale answer.  Thread-local; _begin / _end pairs nest (the
 * outermost _end launches). */
int mnk_factorize_batch_begin(void);
int mnk_factorize_batch_end(void);
/* ... and of independent solves: between _begin and _end the mnk_ls_solve calls of the calling thread with ONE right-hand
 * side on a DEVICE vector are queued; _end runs them up to four systems per launch (one solve is bound by its chain of
 * hops, not by HBM: four side by side take hardly longer than one; N = 11 192: 0.45 -> ~0.2 ms per solve).  The vectors must
 * stay untouched until _end returns (asynchronously, like any device solve: mnk_ls_check_solve as usual).  Solves that do not
 * qualify run at once; several right-hand sides of one solver keep their order (one per launch).  Results are bit-identical
 * to lone solves.  Thread-local, like the factorization batch. */
int mnk_solve_batch_begin(void);
int mnk_solve_batch_end(void);
/* A process that goes idle next to OTHER GPU processes: destroys the CU-masked streams (= hardware queues) this library keeps
 * per device for its persistent schedules -- a device runs only so many queues side by side, all processes together, and a
 * process that merely holds a dozen idle ones can keep another process' pivot chain and bulk kernel from being scheduled
 * together.  Every stream is made again by the first operation that needs it (~ms, once).  The
 * last context of a device to be destroyed does the same.  No equivalent in the reference (its solvers hold no queues). */
int mnk_release_idle_streams(int device);
/* Array forms for n independent instances (one call per phase of an iteration; hosts with a per-call overhead):
 *   _step_batch    : per instance compress_jacobian! + compress_hessian! + build_kkt! + factorize! (asynchronous), the
 *                    factorizations as ONE batch (mnk_factorize_batch_begin / _end around the loop);
 *   _inertia_batch : inertia of every instance (each waits for its own factorization only);
 *   _solve_batch   : one right-hand side per instance, as ONE solve batch (x[i]: N_i entries, in place). */
int mnk_sc_step_batch(int n, mnk_sc* const* sc, mnk_ls* const* ls, const double* const* jac_coo, const double* const* hess_coo,
                      const double* const* pr_diag, const double* const* du_diag, int loc);
int mnk_ls_inertia_batch(int n, mnk_ls* const* ls, int64_t* num_pos, int64_t* num_zero, int64_t* num_neg);
int mnk_ls_solve_batch(int n, mnk_ls* const* ls, double* const* x, int loc);

/* inertia(M) `lapack_common.jl:96-109`, `lapack.jl:240-268`: (num_pos, num_zero, num_neg).
 * CHOLESKY: info == 0 ? (N,0,0) : (0,N,0).  LDL: signs of D. */
int mnk_ls_inertia(mnk_ls* ls, int64_t* num_pos, int64_t* num_zero, int64_t* num_neg);
/* solve_linear_system!(M, x) `lapack_common.jl:75-81` (dpotrs / dsytrs): in place,
 * nrhs right-hand sides with leading dimension ldx (the reference loops over
 * columns, `src/LinearSolvers/linearsolvers.jl:102-110`). */
int mnk_ls_solve(mnk_ls* ls, double* x, int64_t nrhs, int64_t ldx, int loc);
/* Status of the solves enqueued so far, for callers whose vectors live on the device (loc = MNK_DEVICE):
 * synchronizes the context stream; 0 if every solve completed, -3 if a one-launch ("persistent") solve gave
 * up waiting for a peer workgroup -- its result is invalid, the solver has switched to the stepwise solve and
 * the caller repeats the solve.  Host-resident callers never need it: mnk_ls_solve / mnk_*_solve_kkt with
 * loc = MNK_HOST detect the condition before copying anything back and redo the solve themselves.
 * (SolveException of the contract, reference src/LinearSolvers/linearsolvers.jl:133-137.) */
int mnk_ls_check_solve(mnk_ls* ls);
/* Debug / tests: copy the factor (N x N, ld = N; L in the lower triangle, for
 * LDL unit-lower L with D returned separately) and D (N entries, may be NULL). */
int mnk_ls_get_factor(mnk_ls* ls, double* L, double* D, int loc);

/* BUNCHKAUFMAN is served in two tiers: the static-pivot blocked LDL^T (fast path), and -- when that breaks down on
 * a matrix that is not quasi-definite in the given order -- a Bunch-Kaufman factorization with 1x1 / 2x2 pivots and
 * global partner search (dsytf2's strategy; option "bk_fallback", default 1), so that `inertia` follows the
 * reference rule `num_neg_ev` (`src/LinearSolvers/lapack.jl:240-268`) on such systems too.  This reports which tier
 * produced the current factor: *active = 1 if P A P^T = L D L^T with 2x2 blocks; *count = how many factorizations of
 * this solver took the pivoted tier; perm (N entries: row i of the permuted matrix is row perm[i] of A, 0-based) and
 * doff (N entries: sub-diagonal of D, non-zero at the first index of a 2x2 block) may be NULL. */
int mnk_ls_bk_info(mnk_ls* ls, int* active, int* count, int32_t* perm, double* doff);

/* Diagnostics of the last factorization (tests, tuning): key "panel_algo" = the panel algorithm that produced the
 * current factor (4: persistent panel kernel; 1: one launch per panel piece -- chosen automatically while more than
 * one context of this process lives on the device, or for good after a persistent panel gave up on a dependency),
 * "pp_fallbacks" = how many factorizations of this solver were redone for that reason.  Synchronizes when a
 * factorization is pending (the fallback is decided when `info` is read). */
int mnk_ls_get_stat(mnk_ls* ls, const char* key, double* value);

/* ----------------------------------------------------------- utilities ------ */
/* C (M x N) = / -= A (M x K) * B (N x K)^T on the fp64 MFMA tile kernel used by the
 * factorization's trailing update; exposed for unit tests and microbenchmarks.
 * mode 0: C -= A*B^T, 1: C = A*B^T, 2: as 0 but only tiles on/below the diagonal.
 * All pointers are device memory; M, N multiples of 64, K multiple of 16. */
int mnk_gemm_nt(mnk_ctx* ctx, int mode, int64_t M, int64_t N, int64_t K,
                const double* A, int64_t lda, const double* B, int64_t ldb,
                double* C, int64_t ldc);

/* ---- device-side solve_kkt! / mul! of the sparse condensed system -----------------------------
 * The primal-dual vector w = [x (n); s (m); z (m); zl (nlb); zu (nub)] (UnreducedKKTVector layout, reference
 * src/KKT/rhs.jl:119-129) stays on the device across reduce_rhs! / condensation / solve / expansion /
 * finish_aug_solve!, so one call replaces the host-side vector algebra and the per-solve round trips.
 *   mnk_sc_set_bounds         ind_lb / ind_ub: positions of the bounded entries in the primal block [0, n+m)
 *   mnk_sc_set_barrier_terms  reg (n+m), l_diag, u_diag, l_lower, u_lower of the current iterate
 *                             (pr_diag / du_diag come with mnk_sc_build)
 *   mnk_sc_solve_kkt          solve_kkt!(::SparseCondensedKKTSystem, w)  reference src/IPM/factorization.jl:143-167
 *   mnk_sc_mul                mul!(w, kkt, x, alpha, beta)               reference :289-308 + _kktmul! kernels.jl:161-180 */
int mnk_sc_set_bounds(mnk_sc* sc, int64_t nlb, const int64_t* ind_lb, int64_t nub, const int64_t* ind_ub,
                      int index_base);
int mnk_sc_set_barrier_terms(mnk_sc* sc, const double* reg, const double* l_diag, const double* u_diag,
                             const double* l_lower, const double* u_lower, int loc);
int mnk_sc_solve_kkt(mnk_sc* sc, mnk_ls* ls, double* w, int loc);
int mnk_sc_mul(mnk_sc* sc, double* w, const double* x, double alpha, double beta, int loc);

/* Device-side feeders of build_kkt! (first slice of 8(f).4): with these an iteration needs no host vector.
 *   mnk_sc_set_aug_diagonal    set_aug_diagonal!(kkt, solver)  reference src/IPM/kernels.jl:4-27: x, xl, xu, zl, zu are the
 *                              FULL primal-length vectors (n + m entries: variables and slacks) of the iterate; fills
 *                              reg = primal_reg, du_diag = -dual_reg, l_diag = xl_r - x_lr, u_diag = x_ur - xu_r,
 *                              l_lower = zl_r, u_lower = zu_r and pr_diag = reg - l_lower./l_diag - u_lower./u_diag
 *                              (scattered through ind_lb / ind_ub of mnk_sc_set_bounds) INSIDE the handle; the barrier
 *                              terms of mnk_sc_set_barrier_terms are thereby set too
 *   mnk_sc_regularize_diagonal regularize_diagonal!(kkt, primal, dual)  reference src/KKT/KKTsystem.jl:222-226
 *   mnk_sc_build(sc, NULL, NULL, loc) then builds from the handle's own pr_diag / du_diag
 *   mnk_sc_get_diagonals       host copies for tests (any pointer may be NULL) */
int mnk_sc_set_aug_diagonal(mnk_sc* sc, const double* x, const double* xl, const double* xu, const double* zl,
                            const double* zu, double primal_reg, double dual_reg, int loc);
int mnk_sc_regularize_diagonal(mnk_sc* sc, double primal, double dual);
int mnk_sc_get_diagonals(mnk_sc* sc, double* pr_diag, double* du_diag, double* reg, double* l_diag, double* u_diag,
                         double* l_lower, double* u_lower);
/* The bracket of a speculative trial of inertia_correction! (reference src/IPM/solver.jl:611-670; the speculation itself lives in
 * the caller -- madnlp_jl_amd/ipm_dev.py): reg, pr_diag and du_diag are copied into a buffer of the handle
 * before regularize_diagonal! perturbs them for the trial that is factorized AHEAD of the verdict on the unperturbed matrix, and
 * copied back -- bit for bit -- when that matrix is accepted after all. */
int mnk_sc_save_diagonals(mnk_sc* sc);
int mnk_sc_restore_diagonals(mnk_sc* sc);

/* The dense twins: w = [x (n); s (ns); y (m); zl; zu].  solve_kkt!(::DenseCondensedKKTSystem) reference
 * src/IPM/factorization.jl:190-229, the reduced solve of DenseKKTSystem :41-46, mul!(::AbstractDenseKKTSystem)
 * :310-330 (symv on the lower triangle of hess, two gemv with jac). */
int mnk_dc_set_bounds(mnk_dc* dc, int64_t nlb, const int64_t* ind_lb, int64_t nub, const int64_t* ind_ub,
                      int index_base);
int mnk_dc_set_barrier_terms(mnk_dc* dc, const double* reg, const double* l_diag, const double* u_diag,
                             const double* l_lower, const double* u_lower, int loc);
int mnk_dc_solve_kkt(mnk_dc* dc, mnk_ls* ls, double* w, int loc);
int mnk_dc_mul(mnk_dc* dc, double* w, const double* x, double alpha, double beta, int loc);
/* (vectors of n + ns entries; same semantics as the mnk_sc_* feeders above; mnk_dc_build(dc, NULL, NULL, loc) builds from
 * the handle's own diagonals) */
int mnk_dc_set_aug_diagonal(mnk_dc* dc, const double* x, const double* xl, const double* xu, const double* zl,
                            const double* zu, double primal_reg, double dual_reg, int loc);
int mnk_dc_regularize_diagonal(mnk_dc* dc, double primal, double dual);
int mnk_dc_get_diagonals(mnk_dc* dc, double* pr_diag, double* du_diag, double* reg, double* l_diag, double* u_diag,
                         double* l_lower, double* u_lower);


/* ---- IPM scalar reductions over DEVICE-resident iterates (second slice) ----------------------------------
 * The regular-phase functions of reference src/IPM/kernels.jl:263-388,675-695 (GPU twins: mapreduce calls in
 * lib/MadNLPGPU/src/IPM/kernels.jl:4-116).  Every vector argument is a DEVICE pointer; x, xl, xu, zl, zu, f, jacl, dx are
 * the FULL primal-length vectors (ntot = variables + slacks), the *_r views of the reference are taken through the
 * ind_lb / ind_ub given at creation; dzl / dzu (mnk_ipm_get_alpha_z) are the bound-length blocks dual_lb(d) / dual_ub(d)
 * of a KKT vector.  Each call enqueues one or two one-pass reductions on the context's stream, synchronizes and returns
 * the scalar(s) in *out (host).  max/min-type results are bit-identical to the reference's loops, sum-type results agree to
 * summation-order rounding. */
typedef struct mnk_ipm mnk_ipm;
int mnk_ipm_create(mnk_ctx* ctx, int64_t ntot, int64_t nlb, const int64_t* ind_lb, int64_t nub, const int64_t* ind_ub,
                   int index_base, mnk_ipm** out);
int mnk_ipm_destroy(mnk_ipm* ipm);
int mnk_ipm_get_varphi(mnk_ipm* ipm, double obj_val, const double* x, const double* xl, const double* xu, double mu,
                       double* out);                                                    /* kernels.jl:263-283 */
int mnk_ipm_get_inf_du(mnk_ipm* ipm, const double* f, const double* zl, const double* zu, const double* jacl, double sd,
                       double* out);                                                    /* :285-291 */
int mnk_ipm_get_inf_compl(mnk_ipm* ipm, const double* x, const double* xl, const double* xu, const double* zl,
                          const double* zu, double mu, double sc, double* out);         /* :293-303 */
int mnk_ipm_get_min_complementarity(mnk_ipm* ipm, const double* x, const double* xl, const double* xu, const double* zl,
                                    const double* zu, double* out);                     /* :322-332 */
int mnk_ipm_get_average_complementarity(mnk_ipm* ipm, const double* x, const double* xl, const double* xu,
                                        const double* zl, const double* zu, double* out); /* :305-313 */
int mnk_ipm_get_varphi_d(mnk_ipm* ipm, const double* f, const double* x, const double* xl, const double* xu,
                         const double* dx, double mu, double* out);                     /* :341-354 */
int mnk_ipm_get_alpha_max(mnk_ipm* ipm, const double* x, const double* xl, const double* xu, const double* dx, double tau,
                          double* out);                                                 /* :356-371 */
int mnk_ipm_get_alpha_z(mnk_ipm* ipm, const double* zl, const double* zu, const double* dzl, const double* dzu, double tau,
                        double* out);                                                   /* :373-388 */
int mnk_ipm_get_rel_search_norm(mnk_ipm* ipm, const double* x, const double* dx, double* out);   /* :675-682 */
int mnk_ipm_get_sd_sc(mnk_ipm* ipm, const double* l, int64_t m, const double* zl, const double* zu, double s_max,
                      double* out /* [sd, sc] */);                                      /* :684-695 */
int mnk_ipm_get_norms(mnk_ipm* ipm, const double* c, int64_t m, double* out /* [norm(c, Inf), norm(c, 1)] */);
/* Batch mode: between mnk_ipm_batch_begin and mnk_ipm_batch_end the mnk_ipm_get_* calls (regular and restoration phase) only
 * enqueue their reductions and return at once; batch_end synchronizes ONCE and then stores every result into the `out` pointer
 * its call was given (the pointers must stay valid until then; a scalar that divides a result -- sd, sc -- can be passed as 1
 * and applied by the caller).  At most 32 reductions per batch (each call uses one to four). */
int mnk_ipm_batch_begin(mnk_ipm* ipm);
int mnk_ipm_batch_end(mnk_ipm* ipm);
/* Elementwise pieces of the regular phase on device-resident vectors (asynchronous on the context's stream):
 *   mnk_ipm_set_aug_rhs            set_aug_rhs!  kernels.jl:113-131: px = -f + zl - zu - jacl, py = -c,
 *                                  pzl = (xl_r - x_lr) zl_r + mu, pzu = (xu_r - x_ur) zu_r - mu  (px ntot, py m, pzl nlb, pzu nub)
 *   mnk_ipm_set_perturbation_sets  ind_llb / ind_uub (entries bounded on one side only), once
 *   mnk_ipm_dual_inf_perturbation  dual_inf_perturbation!  :818-823
 *   mnk_ipm_adjust_boundary        adjust_boundary!  :656-673 (xl, xu full-length, in place)
 *   mnk_ipm_reset_bound_dual       the two reset_bound_dual! calls of an accepted step, :775-801 / solver.jl:280-291
 *                                  (full-length vectors; unbounded entries carry -Inf / +Inf bounds) */
int mnk_ipm_set_perturbation_sets(mnk_ipm* ipm, int64_t nllb, const int64_t* ind_llb, int64_t nuub, const int64_t* ind_uub,
                                  int index_base);
int mnk_ipm_set_aug_rhs(mnk_ipm* ipm, const double* f, const double* zl, const double* zu, const double* jacl,
                        const double* c, int64_t m, const double* x, const double* xl, const double* xu, double mu,
                        double* px, double* py, double* pzl, double* pzu);
int mnk_ipm_dual_inf_perturbation(mnk_ipm* ipm, double* px, double mu, double kappa_d);
int mnk_ipm_adjust_boundary(mnk_ipm* ipm, const double* x, double* xl, double* xu, double mu);
int mnk_ipm_reset_bound_dual(mnk_ipm* ipm, double* zl, double* zu, const double* x, const double* xl, const double* xu,
                             double mu, double kappa_sigma);

/* ---- restoration phase (robust restorer) on device-resident vectors (third slice) -------------------------
 * Reference src/IPM/kernels.jl:390-636 (GPU twins lib/MadNLPGPU/src/IPM/kernels.jl:117-462).  pp / nn / zp / zn / dpp / dnn /
 * dzp / dzn, c, l (= y) and dl are m-vectors (the RobustRestorer of src/IPM/restoration.jl:1-37 and the constraint block),
 * f_R / D_R / x_ref and x, xl, xu, zl, zu, jacl, dx full primal length; dzl / dzu bound-length blocks of a KKT vector.
 * Same conventions as the regular-phase reductions above (one synchronizing call per reference function). */
int mnk_ipm_get_obj_val_R(mnk_ipm* ipm, const double* p, const double* n, int64_t m, const double* D_R, const double* x,
                          const double* x_ref, double rho, double zeta, double* out);   /* kernels.jl:390-407 */
int mnk_ipm_get_theta_R(mnk_ipm* ipm, const double* c, const double* p, const double* n, int64_t m, double* out); /* :411-421 */
int mnk_ipm_get_inf_pr_R(mnk_ipm* ipm, const double* c, const double* p, const double* n, int64_t m, double* out); /* :423-433 */
int mnk_ipm_get_inf_du_R(mnk_ipm* ipm, const double* f_R, const double* l, const double* zl, const double* zu,
                         const double* jacl, const double* zp, const double* zn, int64_t m, double rho, double sd,
                         double* out);                                                  /* :435-454 */
int mnk_ipm_get_inf_compl_R(mnk_ipm* ipm, const double* x, const double* xl, const double* xu, const double* zl,
                            const double* zu, const double* pp, const double* zp, const double* nn, const double* zn,
                            int64_t m, double mu_R, double sc, double* out);            /* :456-484 */
int mnk_ipm_get_alpha_max_R(mnk_ipm* ipm, const double* x, const double* xl, const double* xu, const double* dx,
                            const double* pp, const double* dpp, const double* nn, const double* dnn, int64_t m,
                            double tau_R, double* out);                                 /* :486-515 */
int mnk_ipm_get_alpha_z_R(mnk_ipm* ipm, const double* zl, const double* zu, const double* dzl, const double* dzu,
                          const double* zp, const double* dzp, const double* zn, const double* dzn, int64_t m,
                          double tau_R, double* out);                                   /* :517-542 */
int mnk_ipm_get_varphi_R(mnk_ipm* ipm, double obj_val, const double* x, const double* xl, const double* xu, const double* pp,
                         const double* nn, int64_t m, double mu_R, double* out);        /* :544-570 */
/* get_F (soft restoration, :572-610).  The upper-bound term is the reference's own expression |(xu_r - xu_r) zu_r - mu|
 * (:606, the same in the GPU twin :407-410), restated as written */
int mnk_ipm_get_F(mnk_ipm* ipm, const double* c, int64_t m, const double* f, const double* zl, const double* zu,
                  const double* jacl, const double* x, const double* xl, const double* xu, double mu, double* out);
int mnk_ipm_get_varphi_d_R(mnk_ipm* ipm, const double* f_R, const double* x, const double* xl, const double* xu,
                           const double* dx, const double* pp, const double* nn, const double* dpp, const double* dnn,
                           int64_t m, double mu_R, double rho, double* out);            /* :612-636 */
/* Elementwise pieces (asynchronous on the context's stream):
 *   mnk_ipm_populate_RR_nn              populate_RR_nn!  :825-829
 *   mnk_ipm_initialize_robust_restorer  vector part of initialize_robust_restorer!  src/IPM/restoration.jl:46-70:
 *                                       x_ref = x, D_R = min(1, 1/|x_ref|), nn, pp = c + nn, zp = mu_R/pp, zn = mu_R/nn,
 *                                       zl_r / zu_r = min(rho, .) in place in the full-length zl / zu
 *   mnk_ipm_set_f_RR                    set_f_RR!  :106-110
 *   mnk_ipm_set_aug_rhs_RR              set_aug_rhs_RR!  :133-158
 *   mnk_ipm_finish_aug_solve_RR         finish_aug_solve_RR!  :251-257
 *   mnk_ipm_reset_bound_dual_1          reset_bound_dual!(z, x, mu, kappa_sigma)  :775-786
 *   mnk_ipm_set_initial_bounds          set_initial_bounds!  :206-218
 *   mnk_ipm_set_initial_rhs             set_initial_rhs!  :220-230
 *   mnk_ipm_set_aug_rhs_ifr             set_aug_rhs_ifr!  :233-240
 *   mnk_ipm_set_g_ifr                   set_g_ifr!  :242-248
 *   mnk_ipm_initialize_variables        initialize_variables!  :638-654 (in place)
 *   mnk_sc_set_aug_RR / mnk_dc_set_aug_RR   set_aug_RR!  :72-87 + _set_aug_diagonal!  :22-27 inside the KKT handle */
int mnk_ipm_populate_RR_nn(mnk_ipm* ipm, double* nn, const double* c, int64_t m, double mu, double rho);
int mnk_ipm_initialize_robust_restorer(mnk_ipm* ipm, const double* x, const double* c, int64_t m, double mu_R, double rho,
                                       double* x_ref, double* D_R, double* nn, double* pp, double* zp, double* zn,
                                       double* zl, double* zu);
int mnk_ipm_set_f_RR(mnk_ipm* ipm, double* f_R, const double* D_R, const double* x, const double* x_ref, double zeta);
int mnk_ipm_set_aug_rhs_RR(mnk_ipm* ipm, const double* f_R, const double* zl, const double* zu, const double* jacl,
                           const double* c, const double* y, const double* pp, const double* nn, const double* zp,
                           const double* zn, int64_t m, const double* x, const double* xl, const double* xu, double mu_R,
                           double rho, double* px, double* py, double* pzl, double* pzu);
int mnk_ipm_finish_aug_solve_RR(mnk_ipm* ipm, double* dpp, double* dnn, double* dzp, double* dzn, const double* l,
                                const double* dl, const double* pp, const double* nn, const double* zp, const double* zn,
                                int64_t m, double mu_R, double rho);
int mnk_ipm_reset_bound_dual_1(mnk_ipm* ipm, double* z, const double* x, int64_t n, double mu, double kappa_sigma);
int mnk_ipm_set_initial_bounds(mnk_ipm* ipm, double* xl, double* xu, int64_t n, double tol);
int mnk_ipm_set_initial_rhs(mnk_ipm* ipm, const double* f, const double* zl, const double* zu, double* px, double* py,
                            int64_t m, double* pzl, double* pzu);
int mnk_ipm_set_aug_rhs_ifr(mnk_ipm* ipm, const double* c, int64_t m, double* px, double* py, double* pzl, double* pzu);
int mnk_ipm_set_g_ifr(mnk_ipm* ipm, double* g, const double* f, const double* x, const double* xl, const double* xu,
                      const double* jacl, double mu);
int mnk_ipm_initialize_variables(mnk_ipm* ipm, double* x, const double* xl, const double* xu, int64_t n, double bound_push,
                                 double bound_fac);
int mnk_sc_set_aug_RR(mnk_sc* sc, const double* x, const double* xl, const double* xu, const double* zl, const double* zu,
                      const double* D_R, const double* pp, const double* zp, const double* nn, const double* zn, double zeta,
                      double primal_reg, double dual_reg);
int mnk_dc_set_aug_RR(mnk_dc* dc, const double* x, const double* xl, const double* xu, const double* zl, const double* zu,
                      const double* D_R, const double* pp, const double* zp, const double* nn, const double* zn, double zeta,
                      double primal_reg, double dual_reg);

/* ---- plain vector work of the solver loop on device vectors (callback half) ----------------------------
 * The copyto! / fill! / axpy! / dot / norm calls the reference's regular! / restore! / robust! loops make on the iterate
 * (src/IPM/solver.jl:236-291, :300-411, :413-545; src/IPM/line_search.jl:60-64, :170-176), the Richardson loop's axpy! and
 * norms (src/LinearSolvers/backsolve.jl:27-76) and the products a dense QP model's callbacks need.  Asynchronous on the
 * context's stream; mnk_ipm_get_* follow the batch protocol above.
 *   mnk_ipm_get_dot / _get_sum / _get_norm2     dot(x, y), sum(v), norm(v, 2)
 *   mnk_ipm_vec_copy / _fill / _axpby           copyto!, fill!, out = a x + b y (y = NULL: out = a x; aliasing allowed)
 *   mnk_ipm_vec_scatter_axpy / _gather          y[idx] .+= a .* x ;  out .= a .* x[idx]   (idx: device, 0-based)
 *   mnk_ipm_bound_dual_axpy / _fill             zl_r .+= a dzl, zu_r .+= a dzu ;  zl_r .= v, zu_r .= v
 *   mnk_ipm_gemv                                y = alpha op(A) x + beta y, A column-major m x n */
int mnk_ipm_get_dot(mnk_ipm* ipm, const double* x, const double* y, int64_t n, double* out);
int mnk_ipm_get_sum(mnk_ipm* ipm, const double* v, int64_t n, double* out);
int mnk_ipm_get_norm2(mnk_ipm* ipm, const double* v, int64_t n, double* out);
int mnk_ipm_vec_copy(mnk_ipm* ipm, double* dst, const double* src, int64_t n);
int mnk_ipm_vec_fill(mnk_ipm* ipm, double* v, int64_t n, double value);
int mnk_ipm_vec_axpby(mnk_ipm* ipm, double* out, double a, const double* x, double b, const double* y, int64_t n);
int mnk_ipm_vec_scatter_axpy(mnk_ipm* ipm, double* y, const int64_t* idx, double a, const double* x, int64_t n);
int mnk_ipm_vec_gather(mnk_ipm* ipm, double* out, double a, const double* x, const int64_t* idx, int64_t n);
int mnk_ipm_bound_dual_axpy(mnk_ipm* ipm, double* zl, double* zu, double a, const double* dzl, const double* dzu);
int mnk_ipm_bound_dual_fill(mnk_ipm* ipm, double* zl, double* zu, double v);
int mnk_ipm_gemv(mnk_ipm* ipm, int trans, int64_t m, int64_t n, double alpha, const double* A, int64_t lda, const double* x,
                 double beta, double* y);

/* ---- device evaluation of an NLP model's callbacks: polar AC optimal power flow (callback half) --------
 * What eval_f_wrapper / eval_grad_f_wrapper! / eval_cons_wrapper! / eval_jac_wrapper! / eval_lag_hess_wrapper!
 * (reference src/IPM/callbacks.jl:1-96) obtain from the model through NLPModels.obj / grad! / cons! / jac_coord! /
 * hess_coord!.  The model is the polar AC-OPF the reference's GPU benchmarks solve; variable / constraint / COO order are
 * those of `madnlp_jl_amd.problems.ACOPFModel` (see csrc/opf_eval.hip).  Index arrays are host, 0-based; arc_coef is
 * (2 nbranch) x 6 row-major (from sides first), bus_data nbus x 4 (pd, qd, gs, bs), gen_cost ngen x 3 (c2, c1, c0).
 * x, y and every output are device vectors; calls are asynchronous on the context's stream. */
typedef struct mnk_opf mnk_opf;
int mnk_opf_create(mnk_ctx* ctx, int64_t nbus, int64_t ngen, int64_t nbranch, const int32_t* fr, const int32_t* to,
                   const int32_t* gen_bus, const double* arc_coef, const double* bus_data, const double* gen_cost,
                   mnk_opf** out);
int mnk_opf_destroy(mnk_opf* opf);
int mnk_opf_sizes(mnk_opf* opf, int64_t* n, int64_t* m, int64_t* nnzj, int64_t* nnzh);
int mnk_opf_obj_terms(mnk_opf* opf, const double* x, double* terms /* ngen; obj = sum(terms) */);
int mnk_opf_grad(mnk_opf* opf, const double* x, double* g);
int mnk_opf_cons(mnk_opf* opf, const double* x, double* c);
int mnk_opf_jac_coord(mnk_opf* opf, const double* x, double* jac);
int mnk_opf_hess_coord(mnk_opf* opf, const double* x, const double* y, double obj_weight, double* hess);

/* ---- dense S stage of the Schur-complement KKT system ----------------------------------------------
 * Reference: `SchurComplementKKTSystem` src/KKT/Schur/schur.jl -- `build_kkt!` :927-1001 (phase 1: factor every
 * scenario block A_k and form A_k^-1 C_dk'; phase 2: S -= C_dk A_k^-1 C_dk'), `factorize_kkt!` :1003-1005, steps 3-5 of
 * `solve_kkt!` :1040-1058.  Scenario blocks arrive DENSE here (the reference factors them with a sparse solver per
 * scenario, which is outside this path); A_k (blk x blk, lower triangle read, symmetric indefinite), C_dk (nd x blk).
 * Scenarios are sharded over the ranks of a multi-GPU job: a handle holds the ns_local scenarios of ONE rank.
 *   mnk_schur_build_local  S_out = S0 - sum_{local k} C_dk A_k^-1 C_dk'   (S0 = H_dd + Sigma_dd + inequality terms on
 *                          the rank that owns it, NULL elsewhere); S_out is caller-owned DEVICE memory, nd x nd; the
 *                          caller sums the ranks' contributions with ONE all-reduce (RCCL), then
 *   mnk_schur_factorize_s  every rank factors S with the dense solver; mnk_schur_inertia_s: reference
 *                          `is_inertia_correct` :901-903 wants (nd, 0, 0)
 *   mnk_schur_forward      r_k <- A_k^-1 r_k (rhs_k: ns_local x blk, device) and contrib_d = -sum_k C_dk r_k (nd, device):
 *                          the caller adds r_d and all-reduces nd doubles
 *   mnk_schur_solve_s      S x_d = r_d
 *   mnk_schur_backward     x_k = r_k - (A_k^-1 C_dk') x_d */
typedef struct mnk_schur mnk_schur;
int mnk_schur_create(mnk_ctx* ctx, int64_t ns_local, int64_t blk, int64_t nd, int algo, mnk_schur** out);
int mnk_schur_destroy(mnk_schur* h);
int mnk_schur_set_block(mnk_schur* h, int64_t k, const double* A_kk, int64_t lda, const double* C_dk, int64_t ldc,
                        int loc);
int mnk_schur_build_local(mnk_schur* h, const double* S0, int64_t lds0, int loc_s0, double* S_out, int64_t lds_out);
int mnk_schur_factorize_s(mnk_schur* h, const double* S, int64_t lds, int loc, int* info);
int mnk_schur_inertia_s(mnk_schur* h, int64_t* num_pos, int64_t* num_zero, int64_t* num_neg);
int mnk_schur_scenario_inertia(mnk_schur* h, int64_t k, int64_t* num_pos, int64_t* num_zero, int64_t* num_neg);
int mnk_schur_forward(mnk_schur* h, double* rhs_k, double* contrib_d);
int mnk_schur_solve_s(mnk_schur* h, double* rhs_d);
int mnk_schur_backward(mnk_schur* h, double* rhs_k, const double* x_d);
/* Single-rank conveniences for callers that keep no device memory of their own (the host-driven KKT type of the Julia glue):
 *   mnk_schur_s_buffer  a device buffer of nd x nd doubles owned by the handle (S_out of _build_local, S of _factorize_s);
 *   mnk_schur_solve     steps 3-5 of solve_kkt! (reference :1078-1092) in one call, host (loc = MNK_HOST) or device vectors:
 *                       rhs_k = ns x blk (scenario k at k * blk), rhs_d = nd, both overwritten with the solution. */
void* mnk_schur_s_buffer(mnk_schur* h);
/* Device-side assembly of the scenario blocks (round 6).  Reference: the scatter of the callback values through precomputed index
 * maps in build_kkt! (src/KKT/Schur/schur.jl:935-972, maps built at :460-700) and its GPU twin's nine kernels
 * (lib/MadNLPGPU/ext/MadNLPGPUCUDAExt/kernels_schur.jl:14-174, @atomic adds).
 *   mnk_schur_set_structure  once: the COO patterns of the Lagrangian Hessian (any triangle: forced to the lower one) and of the
 *                            Jacobian (row = constraint, column = variable), the rows of the inequality constraints in slack order
 *                            (ind_ineq) and of the equality constraints (ind_eq); variable layout [v_1 .. v_ns (nv each), d (nd)],
 *                            constraint layout [c_1 .. c_ns (nc each)].  Checks what _build_schur_symbolic checks (:140-236: a Hessian
 *                            entry that couples two scenarios, a constraint that reaches another scenario, unequal row counts) and
 *                            builds, per touched entry of A_k (order nv + equality rows per scenario, both triangles), C_dk (nd x blk)
 *                            and S0 (nd x nd), the list of its sources in the reference's scatter order.  ns_global / local_scen:
 *                            the global scenario of every local block when the scenarios are sharded over ranks (NULL: all local);
 *                            own_design: this rank adds H_dd + Sigma_d to its S0 (exactly one rank does).
 *   mnk_schur_assemble       per build_kkt!: hess / jac (COO values), pr_diag (n + n_ineq), du_diag (m), host or device -> A_k, C_dk of
 *                            every local scenario and S0 (mnk_schur_s0_buffer, device) in one launch, every entry summed by one thread
 *                            in the reference's order (the bits of a sequential scatter-add, the same in every run); then
 *                            mnk_schur_build_local(h, mnk_schur_s0_buffer(h), nd, MNK_DEVICE, ...)
 *   mnk_schur_get_block      host copies of one assembled block pair and / or of S0 (tests; any pointer may be NULL) */
int mnk_schur_set_structure(mnk_schur* h, int64_t n, int64_t m, int64_t nv, int64_t nc, int64_t nnzh, const int32_t* hess_I,
                            const int32_t* hess_J, int64_t nnzj, const int32_t* jac_I, const int32_t* jac_J, int64_t n_ineq,
                            const int64_t* ind_ineq, int64_t n_eq, const int64_t* ind_eq, int index_base, int64_t ns_global,
                            const int64_t* local_scen, int own_design);
int mnk_schur_assemble(mnk_schur* h, const double* hess, const double* jac, const double* pr_diag, const double* du_diag, int loc);
void* mnk_schur_s0_buffer(mnk_schur* h);
int mnk_schur_get_block(mnk_schur* h, int64_t k, double* A_kk, double* C_dk, double* S0);
int mnk_schur_solve(mnk_schur* h, double* rhs_k, double* rhs_d, int loc);

/* Diagnostics: with option "solve_trace" = 1 the persistent solve kernel stamps the forward sweep's critical
 * path (8 x 100 MHz timer values per 64-row block); this copies them out (n = number of uint64 to copy). */
int mnk_ls_debug_solve_trace(mnk_ls* ls, unsigned long long* out, int64_t n);

/* Diagnostics: with option "dag_debug" = 1 a factorization of the task-DAG schedule that runs into its bounded waits keeps the
 * schedule's progress words (two queue counters | front[Np / 64] | af[ntile^2] | tprog[ntile^2]) and 8 words per pivot-chain
 * strip saying what it was waiting for ({site, strip-column, strip, word, target, value, spins >> 20, 0}) as the time-out left
 * them, before the factorization is redone.  Copies up to `nflags` / `nchain` ints; *have = 1 if a time-out was recorded. */
int mnk_ls_debug_dag_state(mnk_ls* ls, int* flags, int64_t nflags, int* chain, int64_t nchain, int* have);

/* Size of a persistent grid on the CU-mask bits [cu_first + first, cu_first + num_cu) with `per_cu` workgroups per CU: the
 * workgroups the hardware places AT LAUNCH (the dispatcher deals a grid out evenly per XCD and shader engine -- mask bit b is
 * CU b / 8 of XCD b % 8, shader engine (b / 8) % 4 --, so the engine the mask leaves the fewest CUs bounds it).  Pure host
 * arithmetic (no device needed); the task-DAG schedule sizes its bulk kernel with it: 672 beside the 16-CU chain of a 256-CU
 * part, not 3 x 240 (workgroups placed in mid-kernel were the schedule's rare time-out). */
int mnk_debug_grid_at_launch(int cu_first, int num_cu, int first, int per_cu);
/* Diagnostics (bench.py `clocks`): the shader clock the chip sustains under fp64 MFMA load right now, MHz (s_memtime against the
 * 100 MHz s_memrealtime over the second of two ~2.5 ms launches of MFMAs on every CU, on the context's stream). */
int mnk_debug_shader_clock(mnk_ctx* ctx, double* mhz);

/* Diagnostics / tests (host only): the task list of the task-DAG factorization schedule (csrc/dag.hip) for a matrix of `ntile`
 * 128-row tiles: 4 ints per task (flags | chunk index << 8, tile row I, tile column J, kbeg | kend << 16) in queue order, at most
 * `cap` tasks written; returns the number of tasks (negative: bad arguments). */
int mnk_debug_dag_tasks(int ntile, int chunk, int band_tiles, int js2, int taper0, int* out, int cap, int* first_phase);
/* ... and the merged queue of a batch of `ninst` instances shifted by `period` tile columns of chain position (with the
 * zero-fill tasks of the next factorization's buffer if `fill`): the instance index sits in bits 16.. of the second int. */
int mnk_debug_dag_merged_tasks(int ntile, int chunk, int band_tiles, int js2, int taper0, int fill, int ninst, int period,
                               int* out, int cap);

/* Diagnostics (tools/microbench_update.py): time `reps` lower-tile trailing updates C -= A*A^T under the
 * schedules the factorization uses (static tiling / tile queue; context, update, update+panel streams). */
int mnk_debug_update(mnk_ctx* ctx, int variant, int64_t M, int64_t K, const double* A, int64_t lda,
                     double* C, double* C2, int64_t ldc, int reps, double* ms);

#ifdef __cplusplus
}
#endif
#endif /* MADNLP_HIP_H */
