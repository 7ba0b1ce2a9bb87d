// Internal layout of the opaque handles (shared by factor.hip, solve.hip, ls.hip, kkt units).
#pragma once
#include <algorithm>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "common.h"

struct mnk_sc {
    mnk_ctx* ctx = nullptr;
    int64_t n = 0, m = 0, nnzj = 0, nnzh = 0;
    int64_t nnz_jt = 0, nnz_hess = 0, nnz_aug = 0, len_jptr = 0;
    // host copies of the derived structures (0-based)
    std::vector<int32_t> jt_colptr, jt_rowval, h_colptr, h_rowval, aug_colptr, aug_rowval;
    std::vector<int64_t> jt_map, h_map;
    std::vector<int32_t> d_dst, d_src, hp_dst, hp_src, j_dst, j_c, j_k, j_l;
    // device: COO -> CSC segmented transfer (sources grouped by destination slot)
    mnk::DevBuf<int32_t> jt_seg_ptr, jt_seg_src, h_seg_ptr, h_seg_src;
    // device: condensation lists grouped by aug_com slot
    mnk::DevBuf<int32_t> aug_hptr, aug_hsrc;   // per slot: range into hsrc (H.nz indices)
    mnk::DevBuf<int32_t> aug_dsrc;             // per slot: pr_diag index or -1
    mnk::DevBuf<int32_t> aug_jptr, aug_jc, aug_jk, aug_jl;
    mnk::DevBuf<int32_t> aug_row, aug_col;     // coordinates of every aug_com slot (densify)
    mnk::DevBuf<int32_t> d_jt_colptr, d_jt_rowval, d_jt_colidx, d_h_colptr, d_h_rowval, d_h_colidx;
    // device values
    mnk::DevBuf<double> jac_coo, hess_coo, jt_nz, h_nz, aug_nz, diag_buffer, pr_diag, du_diag;
    void* extra = nullptr;  // unit-private device structures (sparse_kkt.hip), owned by the handle
    // dies with the handle: a linear solver that keeps a way back to this handle's matrix (mnk_ls::retransfer) holds a
    // weak reference and refuses to touch a destroyed handle
    std::shared_ptr<int> alive = std::make_shared<int>(0);
};

struct mnk_dc {
    mnk_ctx* ctx = nullptr;
    int condensed = 1;
    bool mirror_pending = false;   // build_kkt! wrote the lower triangle only (all the factorization reads); mnk_dc_get_aug mirrors it on demand
    int64_t n = 0, m = 0, ns = 0, n_eq = 0, order = 0;
    std::vector<int64_t> ind_ineq, ind_eq;
    mnk::DevBuf<int64_t> d_ind_ineq, d_ind_eq;
    mnk::DevBuf<double> hess, jac, aug, pr_diag, du_diag, diag_buffer;
    mnk::DevBuf<double> jis;  // sqrt(D)-scaled, zero-padded inequality Jacobian^T workspace
    int64_t ld_jis = 0, kpad = 0, npad = 0;
    void* extra = nullptr;  // unit-private device structures (dense_kkt.hip), owned by the handle
    std::shared_ptr<int> alive = std::make_shared<int>(0);  // (see mnk_sc::alive)
};

struct mnk_ls {
    mnk_ctx* ctx = nullptr;
    int64_t N = 0, Np = 0, ld = 0, ldw = 0, nbo = 512;
    bool nbo_auto = true;   // outer_block = 0: the width of the outer panels by size (mnk_ls_effective_nbo)
    int algo = MNK_LDL;
    double pivot_tol = 0.0;
    int lookahead = 1;
    int share = 1;         // look-ahead only: the panel stream's CUs join the trailing update through a tile queue
    int small_tiles = 400;  // (a)-updates with fewer 128x128 tiles than this use 64x64 workgroup tiles
    int split_a = 2;          // 1: next panel delivered in two pieces by the update stream; 2: its first 64 columns by the panel stream itself
    int64_t single_rows = 2560;  // systems up to this (padded) order are factored as ONE outer panel on the whole chip (with the fused tail panels the look-ahead schedule wins from N = 3072 on: 2.12 vs 2.34 ms at 4096)
    int64_t tail_rows = 4096; // outer panels are tail_nbo wide once this many rows (or fewer) remain: one persistent launch per panel with the fused prologue, no inner update (C3: 11.80 -> 11.71 ms; 0 disables)
    int64_t tail_nbo = 256;
    int small_tiles_mid = 1000;  // same for the middle-level update inside an outer panel
    mnk::DevBuf<int> tile_ctr;  // one work-queue counter per outer step
    mnk::DevBuf<double> fact, wbuf[2], linv, dblk, inv16, linv256, linv256t, dvec, dinv, xwork;
    // Sparse sources are scattered into a ZEROED dense buffer (8 Np^2 / 2 bytes of zeros per factorization: 116 us at C3,
    // HBM-bound).  With `prefill` the zeros are written in the background into a second buffer while the current
    // factorization / its solves run; the next factorize! swaps the two (costs a second factor buffer, so only up to
    // prefill_max_rows).
    mnk::DevBuf<double> fact_spare;
    hipEvent_t ev_spare = nullptr, ev_free = nullptr;
    bool spare_zeroed = false, spare_pending = false;
    int prefill = 1;
    int64_t prefill_max_rows = 24576;
    int epoch = 0;    // value the hand-off flags of the current factorization carry
    mnk::DevBuf<int> flag_p;
    int panel0_whole = 1;  // look-ahead: the first outer panel is factored on the whole chip before the streams fork
    int algo_now = 1;    // the panel algorithm of the current factorization (panel_algo, or 1 where 4 is not safe)
    bool pp_blocked = false;
    int64_t fact_count = 0, pp_retry_at = 0, pp_backoff = 16;   // a schedule that timed out is tried again after 16, 64, 256, ... factorizations
    int pp_fallbacks = 0;
    int last_timeout_site = 0;   // diagnostics: which bounded device-side wait expired (get_stat "timeout_site")
    double stall_ms_total = 0.0; // host time between the launch of a factorization whose bounded wait expired and the moment the expiry was seen: what the fall-backs of this solver have cost (get_stat "stall_ms_total"; "stall_ms_process": all solvers)
    double t_fact_launch_ms = 0.0;   // host clock (ms) when the current factorization was enqueued
    int64_t pp_fuse_rows = 4096; // > 0: once this many rows (or fewer) remain, persistent panel launches apply the columns in front of them themselves
    int64_t own_cols = 128;   // split_a = 2: columns of the next panel that the panel stream updates itself
    // task-DAG schedule (panel_algo = 5, dag.hip)
    mnk::DevBuf<int> dag_tasks;   // 4 ints per task, built once per order
    int dag_ntasks = 0, dag_ntasks1 = 0, dag_js2 = 0;  // all tasks / tasks of the first phase / first strip-column of the second
    std::vector<int> dag_host_tasks, dag_host_ready;   // the list on the host (4 ints per task) and the chain position that makes each task ready: merged per batch (dag.hip)
    int dag_fill = 1;             // option: the zero-fill of the spare factor buffer (sparse sources) runs as DAG_FILL tasks of the bulk queue instead of on a side stream beside the solves
    bool dag_has_fill = false;    // the current task list contains them
    bool dag_filled = false;      // the factorization that was just queued zeroes the spare buffer itself (mnk_ls_prefill_spare has nothing to launch)
    int batch_period = 0;         // option: shift between consecutive instances of a batch in the merged queue, in tile columns of chain position (0: half the matrix, the minimum)
    hipEvent_t ev_info = nullptr;    // recorded behind finish_info_kernel: what mnk_ls_fetch_info waits for
    bool ev_info_recorded = false;
    hipEvent_t ev_defer = nullptr;   // batch: "matrix transferred" on this solver's stream / "batch done"
    bool deferred = false;        // a factorize! call of this solver is pending in an open batch
    mnk::DevBuf<int> dag_flags;   // [queue counter | front: Np/64 | af: 4 * Np/128], zeroed per factorization
    mnk::DevBuf<double> vfull;    // LDL^T: V = L D of every column, same layout as `fact` (B operand of the left-looking updates)
    mnk::DevBuf<unsigned long long> dag_trace;  // diagnostics (option dag_trace): time stamps per bulk task / chain strip
    bool dag_trace_on = false;
    int64_t inv_done = 0;         // strip-columns whose diagonal blocks the current factorization has already inverted for the solves
    int dag_band = 16;            // 64-row strips per band of the persistent pivot chain (8, 12 or 16; <= the chain's CUs)
    int dag_taper0 = 2;           // (1: -0.5 .. -1 % slower, alternating runs on two boxes) length of the last chunk in front of a tile's closing task; the chunks double from there (1, 2, 4, ...)
    long dag_spin_limit = 0;  // option: polls (~0.17 us each) a device-side wait of the schedule may take before it gives up
                              // (info = -7); 0 = by the order of the matrix (mnk_ls_dag_spin_limit)
    int dag_chunk = 64;           // tile columns (of 128) per bulk task behind the doubling taper 1, 2, 4, ... (every task ends with a read-modify-write of its tile; C3 at the end of round 3: 12 -> 9.58 ms, 48 / 64 / 88 / 128 / 1024 -> 9.30; N = 16 384: 26.3 -> 25.9 ms, N = 24 576: 82.0 / 82.5 ms; in the middle of the round, with slower closing tasks, 10-16 was the optimum)
    int64_t dag_min_rows = 1280;  // smaller systems keep the launch-per-panel schedules (round 6, LDL: N = 1024 0.441 (schedule 4) / 0.452 ms (this one), N = 1280 0.573 / 0.551; rounds 3-5: 1536)
    int64_t dag_deep_rows = 5376; // systems up to this order put every row into the chain's band (and at most 64 x MNK_DAG_CUS2 rows); larger ones: band + bulk kernel
    int64_t dag_max_rows = 30720; // larger ones too: their trailing updates already run at the update kernel's rate (round 6, LDL, schedule 4 / this one: N = 24 576 87.4 / 81.7 ms, 28 672 133.6 / 128.6, 32 768 188.2 / 190.2; rounds 3-5: 24 576)
    int panel_algo = 5;  // 5: task-DAG schedule (dag.hip: persistent pivot chain + persistent left-looking bulk kernel); 4: persistent panel kernel per 256 columns + one trailing update per outer panel (also what 5 uses outside [dag_min_rows, dag_max_rows]); 1: one launch per piece, the fallback of 4 and 5
    int persistent_solve = 1;  // both sweeps of a solve in one launch (solve.hip); 0: one launch per step
    int solve512 = 0;          // 1: the one-launch solve steps over 512 columns (32-row blocks, 512x512 explicit inverses) from solve512_min_rows on.  C3: solve! 0.46 -> 0.33 ms, but the extra inverses cost factorize! +0.37 ms (they finish 0.3 ms after the chain): worth it from ~4 solves per factorization
    int64_t solve512_min_rows = 4096;
    int dag_chain_inline = 1;  // task-DAG schedule, small systems: the pivot chain runs on the caller's stream (no fork / join around it)
    std::vector<std::string> env_keys;   // options fixed by MNK_OPTIONS for this process
    int dag_js2_override = -1;
    int linv_mfma = 1;         // 256x256 explicit inverses on the matrix cores (0: the scalar LDS kernel, ~70 us per workgroup)
    mnk::DevBuf<double> linv512, linv512t, linv512tmp;
    int pub_next = 0;            // the one-launch solve's publication buffers (two, used alternately): the next solve polls this one ...
    bool pub_clean[2] = {false, false};   // ... which must be all sentinel; every solve clears the other one in passing (solve.hip)
    int* solve_abort = nullptr;  // pinned host word the solve kernel raises when it gives up (host can read it without a sync)
    long ps_spin_limit = 6000000;  // polls (~0.5 us each) a persistent-solve wait may take before it gives up
    int debug_pp_missing = -1;     // tests only: this diagonal strip of every persistent panel launch never publishes
    int debug_ps_missing = -1;     // tests only: this workgroup of the persistent solve leaves at once (a peer that never became resident)
    mnk::DevBuf<unsigned long long> solve_trace;  // diagnostics: 8 time stamps per 64-row block (option solve_trace)
    mnk::DevBuf<int> info_dev;
    mnk::DevBuf<unsigned long long> inertia_dev;
    unsigned long long* pin = nullptr;  // 7 pinned, device-mapped host words: inertia counters, info, max|A|, growth numerator (bit patterns), sign changes are stored here by a kernel
    unsigned long long* pin_dev = nullptr;  // the same words as the device sees them
    // Bunch-Kaufman tier (bk.hip): taken when BUNCHKAUFMAN was requested and the static-pivot factorization broke down
    bool bk_requested = false;   // mnk_ls_create was called with MNK_BUNCHKAUFMAN (MNK_LDL: static pivoting only)
    int bk_fallback = 1;         // option: 0 = never take the pivoted tier (a breakdown is reported as num_zero)
    int early_reject = 0;        // option (with accept_only_pd): stop the static-pivot LDL^T at the first pivot that is not positive (leaf64.h); the factor
                                 // of a rejected matrix is then not usable and its inertia is a lower bound on num_neg
    int reject_on_device = -1;   // what info[2] on the device currently says
    bool src_persistent = false; // the source of the last factorize! call outlives the call (KKT handles): a rejected factorization can be completed later
    bool factor_invalid = false; // the last factorization was rejected early: solve / get_factor refuse
    int64_t early_rejects = 0, early_reject_col = -1, early_reject_redone = 0;   // statistics ("early_rejects", "early_reject_col")
    // LEADING-BLOCK PROBE (round 6; option "probe", effective while early rejection is armed): after a rejection that stopped in
    // the first half of the columns, a matrix that follows an ACCEPTED one is first probed through its leading principal block (a
    // child solver of that order on the same KKT handle) -- mnk_ls_factorize_sc_async
    int probe = 1;
    mnk_ls* probe_ls = nullptr;      // the child (order probe_order), owned
    int64_t probe_order = 0;
    int64_t probe_hint_col = -1;     // where the last early rejection of the first half stopped
    int probe_since_hint = 1 << 20;  // verdicts fetched since then
    bool probe_last_rejected = false;
    int64_t probe_hits = 0, probe_misses = 0;   // statistics: matrices rejected by the probe alone / probes that passed
    int accept_only_pd = 0;      // option: the caller accepts positive definite matrices only: "not PD" from the static tier is final
    // Growth guard of the static-pivot tier (BUNCHKAUFMAN only).  The pivots are entries of the successive Schur complements,
    // so max|d_k| / max|a_ij| is a lower bound of the element growth of the elimination; dsytrf's pivoting bounds the growth,
    // static pivoting does not on a matrix that is not quasi-definite (a pivot of 1e-14 is "not zero" and the next Schur
    // complement is 1e14 times the matrix).  Above bk_growth_tol the factor is discarded and the pivoted tier takes over.
    // SPD matrices never trip it (growth <= 1).  Matrices whose pivots come out "all positive, then all negative" have a
    // positive definite leading block and a negative definite Schur complement -- the quasi-definite structure of the KKT
    // systems, for which the unpivoted factorization is the intended one and whose growth |J|^2 / lambda_min(H) is a property
    // of the data, not of the pivot order: they get the lenient bound bk_growth_tol_qd.
    double bk_growth_tol = 64.0, bk_growth_tol_qd = 1e8;
    double last_growth = 0.0;    // max(|d_k|, |v_ik|) / max|a_ij| of the last static-pivot factorization (diagnostics, tests)
    int64_t last_sign_changes = 0;
    mnk::DevBuf<unsigned long long> amax_dev;  // [0] max|a_ij| as transferred (folded from the slots below when the info is published), [1] max(|d_k|, |v_ik|) (bit patterns), [2] sign changes of the pivots; [AMAX_SLOT0 + AMAX_STRIDE i]: partial max|a_ij| of the transfer kernels
    bool bk_active = false;      // the current factor is P A P^T = L D L^T with 2x2 blocks (solves use perm / dcoup)
    int bk_count = 0;            // how many factorizations took the pivoted tier (diagnostics, tests)
    std::function<int()> retransfer;  // puts the matrix of the last factorize! call back into `fact`
    mnk::DevBuf<int> bk_perm, bk_ptype;
    mnk::DevBuf<double> bk_doff, bk_dcoup, bk_work;  // (bk_work: the panel's W = L D and its zero-padded copy of L, 2 x Np x 72)
    mnk::DevBuf<char> bk_state;
    // multi-workgroup panel (bk.hip): W by stored row, staging of the per-panel permutation, message ring, rowof | lists
    mnk::DevBuf<double> bk_wv, bk_tmp;
    mnk::DevBuf<unsigned long long> bk_msg;
    mnk::DevBuf<int> bk_aux;
    int bk_panel_wgs = 0;        // option: 0 = a workgroup per 256 rows of the panel, 1 = one workgroup per panel (round 3)
    int bk_max_wgs = 0;              // option: cap on the workgroups of a multi-workgroup panel (0: one per CU); panels with more
                                     // than 256 x this many rows are factored by the one-workgroup kernel
    long bk_spin_limit = 1L << 21;   // option: polls one of its message rounds may take (~1 s) before the tier falls back
    int debug_bk_missing = -1;       // tests: a workgroup of the multi-workgroup panel that never takes part
    bool bk_multi_last = false, bk_mw_blocked = false;
    int bk_mw_fallbacks = 0;
    bool factorized = false, info_valid = false;
    int dag_debug = 0;           // option: keep the progress words of a factorization that timed out (mnk_ls_debug_dag_state)
    mnk::DevBuf<int> dag_dbg;    // 8 words per chain strip (PpDag::dbg)
    std::vector<int> dbg_flags, dbg_chain;
    bool spare_by_dag = false;   // the spare buffer's zero-fill was left to the DAG_FILL tasks of the factorization queued last: it
                                 // only happened if that factorization ran to its end (info == 0)
    int info = 0;
    int64_t npos = 0, nzero = 0, nneg = 0;
};

int64_t mnk_ls_effective_nbo(const mnk_ls* ls);
int mnk_ls_run_factorization(mnk_ls* ls);
int mnk_ls_run_factorization_now(mnk_ls* ls);      // factor.hip: the launch part (the schedule has been chosen; batches call it for leftovers)
int mnk_ls_launch_finish_info(mnk_ls* ls, hipStream_t s);   // factor.hip: inertia / growth words / info -> pinned host words
bool mnk_solve_defer(mnk_ls* ls, double* xuser);   // solve.hip: true if the calling thread has a solve batch open and queued this solve
int mnk_solve_sync_deferred(mnk_ls* ls);           // solve.hip: runs the queued solves if one of them belongs to this solver
int mnk_solve_batch_flush_pending(void);           // solve.hip: runs every solve the calling thread has queued (nested batches: schur.hip)
int mnk_ls_factorize_dense_dev_async(mnk_ls* ls, const double* Adev, int64_t lda);   // ls.hip
double mnk_host_ms();                               // factor.hip: steady host clock in ms
void mnk_add_process_stall_ms(double ms);           // factor.hip: time lost to expired device-side waits, all solvers of the process
double mnk_process_stall_ms();
bool mnk_ls_pending_elsewhere(const mnk_ls* ls);   // dag.hip: queued in a factorization batch of ANOTHER thread
bool mnk_batch_active();                           // dag.hip: the calling thread has a factorization batch open
size_t mnk_pchain_sys_bytes();                     // factor.hip: size of one record of mnk_launch_pchain_multi's table
int mnk_launch_pchain_multi(mnk_ls* const* v, int n, hipStream_t sp, hipStream_t fill_stream, void* table, int* const* front,
                            const int* const* af);   // factor.hip: the chains of n small systems in one launch
bool mnk_batch_defer(mnk_ls* ls);                  // dag.hip: true if the calling thread has a batch open and took the factorization into it
int mnk_ls_sync_deferred(mnk_ls* ls);              // dag.hip: runs what the open batches of this thread hold for this solver (queued solves, a pending factorization)
int mnk_ls_sync_deferred_fact(mnk_ls* ls);         // dag.hip: ... the pending factorization only
int mnk_ls_prefill_spare(mnk_ls* ls);   // ls.hip: queue the background zero-fill of the spare factor buffer (if one is due)
int mnk_ls_fetch_info(mnk_ls* ls);
int mnk_ls_run_factorization_dag(mnk_ls* ls);   // dag.hip: the task-DAG schedule (panel_algo = 5)
int mnk_ls_dag_prepare(mnk_ls* ls);             // dag.hip: its buffers (0: ready; otherwise the device cannot hold them -> schedule 4)
int mnk_launch_pchain(mnk_ls* ls, hipStream_t sp, const mnk::PpDag& dag, int js_begin, int js_end, unsigned strips);  // factor.hip
unsigned long long* mnk_ls_growth_word(mnk_ls* ls);  // factor.hip: where the kernels fold max(|d|, |v|) (NULL: guard off)
int mnk_ls_run_solve(mnk_ls* ls, double* xdev /* the solver's work vector */, double* xuser = nullptr /* N entries on the device, or NULL: xdev holds the padded rhs */);
int mnk_ls_build_inverses(mnk_ls* ls, hipStream_t s, int64_t sc0, int64_t sc1);  // 256x256 triangles of the strip-columns [sc0, sc1)
int mnk_ls_invert_blocks(mnk_ls* ls, hipStream_t s, int64_t sc0, int64_t sc1);   // 64x64 blocks + 256x256 triangles
// Right-side triangular solve of `nrows` free-standing rows (multiple of 16) against the factored diagonal block that
// starts at column j0: V = B L_jj^-T, X = V D^-1 (LDL) / X = B L_jj^-T (Cholesky).  B is read from / X written to
// Xrows(:, j0:j0+64), V written to Vrows(:, j0:j0+64) (LDL only); both row blocks have leading dimension ldr.
int mnk_ls_right_trsm_rows(mnk_ls* ls, hipStream_t s, int64_t j0, double* Xrows, double* Vrows, int64_t ldr, int64_t nrows);
int mnk_ls_run_bunchkaufman(mnk_ls* ls);                                  // bk.hip
long mnk_ls_dag_spin_limit(const mnk_ls* ls);   // dag.hip
namespace mnk {
// One system of a batch of SMALL factorizations (dag.hip: batch_run_group_small): what the kernels around the shared chain
// launch need, so that they are ONE launch per batch round instead of one per system (blockIdx.z / .y / .x = system; the
// host's launch rate was the bound: ~8 launches per system, 9 of the 10.9 ms of a 128-scenario Schur build).
struct SmallSysRec {
    int* flags; int64_t nflags; int* info;                 // progress words to zero, the info words (dag_reset_kernel)
    double* F; int64_t ld; const double* dblk; double* linv; double* linv256; double* linv256t; int64_t Np;   // inverses for the solves
    const double* dvec; int64_t ninertia; unsigned long long* pin_dev; const unsigned long long* amax;       // finish_info_kernel
};
}  // namespace mnk
int mnk_ls_invert_blocks_batch(hipStream_t s, bool ldl, const mnk::SmallSysRec* recs_dev, int n, int64_t Np);   // factor.hip (+ solve.hip)
int mnk_ls_build_inverses_batch(hipStream_t s, const mnk::SmallSysRec* recs_dev, int n, int64_t Np);           // solve.hip
int mnk_ls_finish_info_batch(hipStream_t s, const mnk::SmallSysRec* recs_dev, int n, int threads);             // factor.hip
void mnk_ls_fill_small_rec(mnk_ls* ls, mnk::SmallSysRec* rec);                                                  // factor.hip
void mnk_pchain_fill_sys(mnk_ls* ls, void* rec_host, int* front, const int* af);                               // factor.hip (one PcSys record)
int mnk_launch_trsm64_batch(hipStream_t s, bool ldl, const mnk::TrsmBatchRec* recs_dev, int nbatch, int64_t j0, int64_t nrows,
                            int64_t ldr);   // factor.hip
int mnk_ls_bk_permute(mnk_ls* ls, double* x, double* tmp, bool forward);  // x <- P x (forward) / P^T x
int mnk_ls_bk_dsolve(mnk_ls* ls, double* y);                              // y <- D^-1 y, 1x1 / 2x2 blocks
int mnk_ls_bk_inertia(mnk_ls* ls, unsigned long long* out_dev);
// after a stream synchronization: true (and the solver switched to the stepwise solve, abort word cleared) if a
// persistent solve gave up since the last check
bool mnk_ls_take_solve_abort(mnk_ls* ls);
