// Bunch-Kaufman LDL^T with 1x1 / 2x2 pivots and GLOBAL partner search on the device: the robust tier
// behind `factorize_bunchkaufman!` = dsytrf('L') (reference src/LinearSolvers/lapack.jl:164-167) and its
// inertia rule `num_neg_ev` (reference src/LinearSolvers/lapack.jl:240-268).
//
// Two tiers.  BUNCHKAUFMAN first runs the static-pivot blocked LDL^T of factor.hip (fp64 MFMA, the fast path:
// every KKT system that is quasi-definite in the given order -- all the condensed systems, the regularized
// augmented ones -- factors there, and by Sylvester's law its sign(D) inertia is the one dsytrf reports).
// When that factorization BREAKS DOWN (an exact zero / non-finite pivot in the given order, e.g. a zero
// diagonal block that needs a 2x2 pivot), the matrix is transferred again and factored here with the pivoting
// strategy of LAPACK's dsytf2 (alpha = (1 + sqrt(17)) / 8, partner = row of the largest off-diagonal entry of
// the pivot column, 1x1 or 2x2 pivot by the usual four tests), so the inertia equals the reference's on
// systems that are NOT quasi-definite in the given order instead of detouring through delta_c.
//
// This tier is BLOCKED (LAPACK dlasyf's scheme, lower variant): panels of NB = 64 columns are factored left-looking by ONE
// workgroup -- every column is brought up to date with the panel's previous columns just before its pivot search, the
// partner column likewise when the 1x1 test fails, so the search sees exactly the entries dsytf2 would -- and the trailing
// matrix receives the whole panel in one fp64-MFMA product A22 -= L21 (L21 D)^T (the tile k-loop of gemm_tile.h).  Two
// launches per panel, the panel's position and width (63 or 64 columns: a 2x2 pivot never straddles a panel end) live in
// device memory, so the host enqueues the whole factorization without a synchronization.  Unlike dsytf2 / dlasyf the
// interchanges are applied to the previous columns as well, so the result is a plain P A P^T = L D L^T with ONE
// permutation vector; the solve is gather, unit-lower sweeps (the same stepwise kernels as the static factor),
// block-diagonal D^-1, scatter.  (Round 2's tier was unblocked: four launches and an HBM-bound rank-1/2 update of the
// whole trailing triangle per pivot -- seconds at N ~ 1e4.)
#include <cfloat>
#include <cmath>

#include <atomic>

#include "gemm_tile.h"
#include "ls.h"

namespace mnk {

struct BkState {
    int k;      // first column that is not factored yet
    int p0;     // first column of the panel just factored
    int kb;     // its width (the trailing update applies columns p0 .. p0 + kb - 1)
    int info;   // LAPACK-style: 1-based index of the first exactly-zero pivot (0: none)
    int fail;   // multi-workgroup panel: a bounded wait expired (1) / internal inconsistency (2): the factor is void
};
// (Inside bkp_panel_mw_kernel this record and the list counters are written with agent-scope atomics ONLY: its workgroups
// sit on different XCDs, and a plain store leaves a dirty line in one XCD's L2 whose write-back at the kernel's end
// overwrites what the other XCDs' atomics did in memory -- seen as lost list entries of every workgroup but the first.)

constexpr double BK_ALPHA = 0.6403882032022076;  // (1 + sqrt(17)) / 8
constexpr int BK_NB = 64;    // panel width
constexpr int BK_NBW = 72;   // columns of the panel work space (NB + 1 working column, padded to the k-tile depth 8)
constexpr int BK_T = 1024;   // threads of the panel workgroup

template <bool WT>
__device__ __forceinline__ void put(double* p, double v) {
    if (WT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

__device__ __forceinline__ void block_argmax(double v, int idx, double* sval, int* sidx, double& outv, int& outi) {
    // largest |value|, smallest index among ties (idamax)
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_down(v, off);
        const int oi = __shfl_down(idx, off);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { sval[w] = v; sidx[w] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double bv = sval[0];
        int bi = sidx[0];
        for (int q = 1; q < (int)(blockDim.x >> 6); ++q)
            if (sval[q] > bv || (sval[q] == bv && sidx[q] < bi)) { bv = sval[q]; bi = sidx[q]; }
        sval[0] = bv;
        sidx[0] = bi;
    }
    __syncthreads();
    outv = sval[0];
    outi = sidx[0];
    __syncthreads();
}

// One panel, one workgroup.  W[:, c] = the panel's c-th column brought up to date = (L D)[:, c] once it is eliminated (what
// the trailing update multiplies with), LW[:, c] = L[:, p0 + c] (a zero-padded copy: the trailing update reads K = 72
// columns); both Np x BK_NBW, leading dimension ldw.  Pivoting: dsytf2 / dlasyf, column by column:
//   absakk = |a_kk|, colmax = max_{i>k} |a_ik| (row imax) on the UPDATED column;  zero column -> info, no elimination;
//   absakk >= alpha colmax -> 1x1, no interchange;  else with rowmax = largest off-diagonal of the UPDATED row/column imax:
//   absakk >= alpha colmax (colmax / rowmax) -> 1x1, no interchange;  |a_imax,imax| >= alpha rowmax -> 1x1, interchange k <->
//   imax;  else 2x2 pivot {k, imax}.
__global__ __launch_bounds__(BK_T) void bkp_panel_kernel(double* __restrict__ F, int64_t ld, int Np, BkState* st,
                                                         double* __restrict__ W, double* __restrict__ LW, int64_t ldw,
                                                         double* __restrict__ dvec, double* __restrict__ doff,
                                                         int* __restrict__ ptype, int* __restrict__ perm) {
    __shared__ double sval[16];
    __shared__ int sidx[16];
    __shared__ double s_wk[BK_NBW], s_wi[BK_NBW];
    const int t = threadIdx.x;
    const int p0 = st->k;
    __syncthreads();  // (everyone has read st->k before thread 0 rewrites the state at the end)
    if (p0 >= Np) {
        if (t == 0) { st->p0 = p0; st->kb = 0; }
        return;
    }
    for (int c = 0; c < BK_NBW; ++c)
        for (int i = p0 + t; i < Np; i += BK_T) { W[i + c * ldw] = 0.0; LW[i + c * ldw] = 0.0; }
    __syncthreads();
    int kb = 0;
    while (kb < BK_NB - 1 && p0 + kb < Np) {
        const int k = p0 + kb;
        double* Wk = W + (int64_t)kb * ldw;        // working column: the pivot column
        double* Wn = W + (int64_t)(kb + 1) * ldw;  // second working column: the partner
        // ---- 1. column k, brought up to date with the panel's columns
        if (t < kb) s_wk[t] = W[k + (int64_t)t * ldw];
        __syncthreads();
        for (int i = k + t; i < Np; i += BK_T) {
            double acc = F[i + (int64_t)k * ld];
            for (int c = 0; c < kb; ++c) acc -= LW[i + (int64_t)c * ldw] * s_wk[c];
            Wk[i] = acc;
        }
        __syncthreads();
        // ---- 2. pivot search
        const double absakk = fabs(Wk[k]);
        double v = -1.0;
        int vi = 0x7fffffff;
        for (int i = k + 1 + t; i < Np; i += BK_T) {
            const double a = fabs(Wk[i]);
            if (a > v || !(a <= DBL_MAX)) { v = !(a <= DBL_MAX) ? DBL_MAX : a; vi = i; }
        }
        double colmax;
        int imax;
        block_argmax(v, vi, sval, sidx, colmax, imax);
        if (colmax < 0.0) colmax = 0.0;  // k is the last column
        int kstep = 1, kp = k;
        bool zero = false;
        if (!(fmax(absakk, colmax) > 0.0) || !(absakk <= DBL_MAX) || colmax >= DBL_MAX) {
            zero = true;  // the column is exactly zero (or not finite): no elimination, info reports it
        } else if (absakk < BK_ALPHA * colmax) {
            // row / column imax of the trailing matrix, brought up to date
            if (t < kb) s_wi[t] = W[imax + (int64_t)t * ldw];
            __syncthreads();
            for (int i = k + t; i < Np; i += BK_T) {
                double acc = i < imax ? F[imax + (int64_t)i * ld] : F[i + (int64_t)imax * ld];
                for (int c = 0; c < kb; ++c) acc -= LW[i + (int64_t)c * ldw] * s_wi[c];
                Wn[i] = acc;
            }
            __syncthreads();
            double rv = -1.0;
            int ri = 0x7fffffff;
            for (int i = k + t; i < Np; i += BK_T) {
                if (i == imax) continue;
                const double a = fabs(Wn[i]);
                if (a > rv) { rv = a; ri = i; }
            }
            double rowmax;
            int jmax;
            block_argmax(rv, ri, sval, sidx, rowmax, jmax);
            if (absakk >= BK_ALPHA * colmax * (colmax / rowmax)) {
                kp = k;
            } else if (fabs(Wn[imax]) >= BK_ALPHA * rowmax) {
                kp = imax;  // 1x1 pivot on a_imax,imax: its column becomes the pivot column
                for (int i = k + t; i < Np; i += BK_T) Wk[i] = Wn[i];
                __syncthreads();
            } else {
                kp = imax;
                kstep = 2;
            }
        }
        const int kk = k + kstep - 1;
        // ---- 3. symmetric interchange kk <-> kp (kp > kk): the part of A that is not factored yet, the same rows of every
        // previous column (one permutation for the whole factorization), of the panel copies and of the working columns
        if (kp != kk) {
            auto swp = [&](double* a, double* b) { const double x = *a; *a = *b; *b = x; };
            for (int j = t; j < Np; j += BK_T) {
                if (j < k) {
                    swp(F + kk + (int64_t)j * ld, F + kp + (int64_t)j * ld);
                } else if (j > kp) {
                    swp(F + j + (int64_t)kk * ld, F + j + (int64_t)kp * ld);
                } else if (j > kk && j < kp) {
                    swp(F + j + (int64_t)kk * ld, F + kp + (int64_t)j * ld);
                } else if (j == kk) {
                    swp(F + kk + (int64_t)kk * ld, F + kp + (int64_t)kp * ld);
                    if (kstep == 2) swp(F + (k + 1) + (int64_t)k * ld, F + kp + (int64_t)k * ld);
                    const int p = perm[kk]; perm[kk] = perm[kp]; perm[kp] = p;
                }
            }
            if (t < kb + kstep) swp(W + kk + (int64_t)t * ldw, W + kp + (int64_t)t * ldw);
            if (t >= 64 && t - 64 < kb) swp(LW + kk + (int64_t)(t - 64) * ldw, LW + kp + (int64_t)(t - 64) * ldw);
            __syncthreads();
        }
        // ---- 4. eliminate
        if (zero) {
            if (t == 0) {
                if (st->info == 0) st->info = k + 1;
                dvec[k] = 0.0; doff[k] = 0.0; ptype[k] = 1;
            }
            for (int i = k + 1 + t; i < Np; i += BK_T) { F[i + (int64_t)k * ld] = 0.0; Wk[i] = 0.0; }
        } else if (kstep == 1) {
            const double d = Wk[k];
            const double r = 1.0 / d;
            __syncthreads();  // (everyone has read the pivot before the column is rewritten)
            for (int i = k + 1 + t; i < Np; i += BK_T) {
                const double l = Wk[i] * r;
                F[i + (int64_t)k * ld] = l;
                LW[i + (int64_t)kb * ldw] = l;
            }
            if (t == 0) { F[k + (int64_t)k * ld] = d; dvec[k] = d; doff[k] = 0.0; ptype[k] = 1; }
        } else {
            // [l1 l2] = [w1 w2] inv([[p11 p21], [p21 p22]]), scaled as dsytf2 does (no overflow from a tiny p21)
            const double p11 = Wk[k], p21 = Wk[k + 1], p22 = Wn[k + 1];
            const double d11 = p22 / p21, d22 = p11 / p21;
            const double tt = 1.0 / (d11 * d22 - 1.0) / p21;
            __syncthreads();
            for (int i = k + 2 + t; i < Np; i += BK_T) {
                const double a1 = Wk[i], a2 = Wn[i];
                const double l1 = tt * (d11 * a1 - a2), l2 = tt * (d22 * a2 - a1);
                F[i + (int64_t)k * ld] = l1;
                F[i + (int64_t)(k + 1) * ld] = l2;
                LW[i + (int64_t)kb * ldw] = l1;
                LW[i + (int64_t)(kb + 1) * ldw] = l2;
            }
            if (t == 0) {
                dvec[k] = p11; dvec[k + 1] = p22; doff[k] = p21; doff[k + 1] = 0.0;
                ptype[k] = 2; ptype[k + 1] = 3;
                F[k + (int64_t)k * ld] = p11; F[(k + 1) + (int64_t)(k + 1) * ld] = p22;
                F[(k + 1) + (int64_t)k * ld] = 0.0;  // L is unit lower: the 2x2 block's off-diagonal lives in doff
            }
        }
        kb += kstep;
        __syncthreads();
    }
    // the working column behind the panel is not part of it: the trailing update must not see it
    for (int c = kb; c < BK_NBW; ++c)
        for (int i = p0 + t; i < Np; i += BK_T) { W[i + c * ldw] = 0.0; LW[i + c * ldw] = 0.0; }
    if (t == 0) { st->p0 = p0; st->kb = kb; st->k = p0 + kb; }
}

// ---------------------------------------------------------------------------------------------------------------------
// Multi-workgroup panel (round 4).  One workgroup reads ~16 KB x Np of the panel copy per panel at the bandwidth of ONE
// CU (0.47 s at N = 11 192).  Here G workgroups share a panel, one ROW per thread: the row's L entries of the panel live
// in LDS, the pivot search is a reduction over G messages, and nothing moves while the panel is factored:
//   * VIRTUAL positions.  Thread (g, t) owns row p0 + 256 g + t of the matrix as it is stored when the panel starts, for
//     the whole panel; an interchange only changes `mypos` of the two rows and the replicated table prow[i] (row at panel
//     position p0 + i).  Columns are read through the symmetric access A0(r, q) = F[max, min].  At the panel's end the
//     rows are written out in position order (the operands of the trailing update), rowof[pos] and the lists of displaced
//     positions tell two small kernels how to permute the trailing matrix, the previous columns and `perm` -- dsytf2's
//     interchanges, applied once per panel instead of once per pivot.
//   * Hops.  Per column ONE all-to-all message round (A: local maximum of the updated pivot column with its row and
//     position, the diagonal entry, the column's entries at the rows of the next two panel positions), and a second one
//     when the 1x1 test fails (B: local maximum of the partner column, its entries at the partner row, the pivot row and
//     the next two positions).  Every workgroup reads all G messages and takes the same decision.  A message value is two
//     64-bit words {32 data bits, 32-bit sequence number}: a reader accepts a word when it carries the expected number,
//     so no fence orders payload against flag and a round costs one store + one load round trip.
//   * The pivot row's W entries.  Updating column q needs W[q, c] (c < kb) of the pivot row in every workgroup.  The
//     owner stores its rows' W entries to a global copy at every elimination and waits for its stores before its NEXT
//     message: what was stored in column k - 1 is visible after hop A of column k.  The entries produced after the last
//     hop travel in the messages (the next pivot row is always the pivot row, or the row at one of the next two positions,
//     of the column before); each workgroup keeps the W rows of these candidates in LDS.
//   tools/bk_mw_model.py is a host model of this data flow (one object per workgroup), checked against a plain dsytf2
//   restatement by tests/test_bk_multi_cpu.py.
// One leader wave per workgroup (messages, decisions, the cache of W rows) + four waves of row owners.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int BKM_ROWS = 256;        // rows per workgroup
constexpr int BKM_T = BKM_ROWS + 64; // threads: leader wave + row owners
constexpr int BKM_F = 8;             // values per message
constexpr int BKM_RING = 4;          // message slots in flight
constexpr int BKM_GMAX = 256;

struct BkMw {
    double* F; int64_t ld; int Np;
    BkState* st;
    double* Wp; double* LWp; int64_t ldw;   // position-ordered operands of the trailing update
    double* Wv;                              // W by row of the matrix as stored at the panel's start, Np x 64 column-major
    unsigned long long* msg;                 // [RING][GMAX][F][2]
    int* rowof; int* dlist;                  // dlist[0..63]: displaced trailing positions, [64..191]: all displaced positions
    int* cnt; int* cnt_next;                 // {nT, nS} of this panel / of the next one (zeroed here)
    double* dvec; double* doff; int* ptype;
    unsigned seq0;
    long spin_limit;   // polls a message round may take (option bk_spin_limit)
    int dbg_missing;   // tests: this workgroup never takes part (option debug_bk_missing; -1: off)
};

struct BkmCtl {
    int q, n1, n2, rmax, imax, decision, fail, slot_q;
    double colmax, akk, p21, p22, rowmax, wk_n1, wk_n2, wn_q, wn_n1, wn_n2;
};
// decision: 0 zero column, 1 1x1 without interchange, 2 phase B needed, 3 1x1 on the partner (interchange k <-> imax),
// 4 2x2 {k, imax} (interchange k + 1 <-> imax)

__device__ __forceinline__ void bkm_post(unsigned long long* msg, int g, unsigned seq, int f, double v) {
    unsigned long long* p = msg + (((size_t)(seq & (BKM_RING - 1)) * BKM_GMAX + g) * BKM_F + f) * 2;
    const unsigned long long b = (unsigned long long)__double_as_longlong(v), tag = (unsigned long long)seq << 32;
    __hip_atomic_store(p, (b & 0xffffffffull) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 1, (b >> 32) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// leader wave: all G messages of round `seq` into mv[g * 8 + f]; false: a bounded wait expired (or another workgroup said so)
__device__ __forceinline__ bool bkm_gather(const unsigned long long* msg, int G, unsigned seq, double* mv, int* fail_word,
                                           long spin_limit) {
    const int lane = threadIdx.x & 63;
    const unsigned long long* base = msg + (size_t)(seq & (BKM_RING - 1)) * BKM_GMAX * BKM_F * 2;
    bool ok = true;
    for (int e0 = 0; e0 < G * BKM_F && ok; e0 += 64 * 8) {
        bool done[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) done[j] = e0 + 64 * j + lane >= G * BKM_F;
        long spins = 0;
        for (;;) {
            unsigned long long u0[8], u1[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (!done[j]) {
                    const unsigned long long* p = base + 2 * (size_t)(e0 + 64 * j + lane);
                    u0[j] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    u1[j] = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            bool all = true;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (!done[j]) {
                    if ((unsigned)(u0[j] >> 32) == seq && (unsigned)(u1[j] >> 32) == seq) {
                        done[j] = true;
                        mv[e0 + 64 * j + lane] = __longlong_as_double((long long)((u0[j] & 0xffffffffull) | (u1[j] << 32)));
                    } else {
                        all = false;
                    }
                }
            if (__all(all)) break;
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 255) == 0) {
                if (spins > spin_limit || __hip_atomic_load(fail_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                    ok = false;
                    break;
                }
            }
        }
    }
    return ok;
}

__global__ __launch_bounds__(BKM_T) void bkp_panel_mw_kernel(BkMw a) {
    extern __shared__ __attribute__((aligned(16))) char bkm_smem[];
    double* LW = reinterpret_cast<double*>(bkm_smem);   // [64][256]: L entries of the panel's columns, by owned row
    double* wc = LW + 64 * BKM_ROWS;                    // [4][64]: W rows of the candidate pivot rows
    double* wtmp = wc + 4 * 64;                         // [64]: W row of the partner
    double* mv = wtmp + 64;                             // [GMAX * 8]: the messages of one round
    double* redv = mv + BKM_GMAX * BKM_F;               // [4][2] per-wave maxima (|v|, v)
    double* ownv = redv + 8;                            // [4]: entries of the working column at q / rmax / n1 / n2 (if owned here)
    int* redi = reinterpret_cast<int*>(ownv + 4);       // [4][2] (position, row) of the per-wave maxima
    int* prow = redi + 8;                               // [68]
    int* wc_row = prow + 68;                            // [4]
    BkmCtl* ctl = reinterpret_cast<BkmCtl*>(wc_row + 4);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bool leader = w == 0;
    const int t = tid - 64;                             // row owners: 0..255
    const int g = blockIdx.x, G = gridDim.x;
    const int Np = a.Np;
    const int64_t ld = a.ld;
    const double* __restrict__ F = a.F;
    const int p0 = a.st->k;
    int* fail_word = &a.st->fail;
    auto st_store = [](int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    if (g == 0 && tid == 0) { st_store(a.cnt_next, 0); st_store(a.cnt_next + 1, 0); }
    if (p0 >= Np) {
        if (g == 0 && tid == 0) { st_store(&a.st->p0, p0); st_store(&a.st->kb, 0); }
        return;
    }
    if (g == a.dbg_missing) return;   // (a peer that never became resident: everyone else must give up, not hang)
    if (__hip_atomic_load(fail_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;   // (an earlier panel gave up: the factor is void)
    for (int i = tid; i < 68; i += BKM_T) prow[i] = p0 + i;
    if (tid < 4) wc_row[tid] = tid == 0 ? p0 : -1;   // (the first pivot row has no W entries yet, but it is a candidate)
    const int row = p0 + g * BKM_ROWS + t;              // (row owners)
    const bool valid = !leader && row < Np;
    bool active = valid;
    int mypos = row;
    int pf_r1 = -1, pf_r2 = -1;                         // columns of the matrix held ahead of time (uniform)
    double pf_v1 = 0.0, pf_v2 = 0.0;
    unsigned seq = a.seq0;
    int kb = 0;
    auto a0 = [&](int r, int c) -> double { return r >= c ? F[r + (int64_t)c * ld] : F[c + (int64_t)r * ld]; };
    auto owner_here = [&](int r) { return r >= 0 && (r - p0) / BKM_ROWS == g; };
    auto find_slot = [&](int r) { int s = -1; for (int i = 0; i < 4; ++i) if (wc_row[i] == r) s = i; return s; };
    __syncthreads();

    while (kb < BK_NB - 1 && p0 + kb < Np) {
        const int k = p0 + kb;
        const int q = prow[kb], n1 = k + 1 < Np ? prow[kb + 1] : -1, n2 = k + 2 < Np ? prow[kb + 2] : -1;
        const int slot_q = kb > 0 ? find_slot(q) : 0;
        // ---- phase A: the pivot column on the owned rows
        double wk = 0.0;
        if (!leader) {
            if (active) {
                double acc = pf_r1 == q ? pf_v1 : (pf_r2 == q ? pf_v2 : a0(row, q));
                if (slot_q >= 0) {
                    const double* wq = wc + slot_q * 64;
                    for (int c = 0; c < kb; ++c) acc -= LW[c * BKM_ROWS + t] * wq[c];
                }
                wk = acc;
            }
            double v = -1.0, sv = 0.0;
            int vp = 0x7fffffff, vr = -1;
            if (active && row != q) {
                const double ab = fabs(wk);
                v = !(ab <= DBL_MAX) ? DBL_MAX : ab;
                sv = wk; vp = mypos; vr = row;
            }
            for (int off = 32; off > 0; off >>= 1) {
                const double ov = __shfl_down(v, off), os = __shfl_down(sv, off);
                const int op = __shfl_down(vp, off), orr = __shfl_down(vr, off);
                if (ov > v || (ov == v && op < vp)) { v = ov; sv = os; vp = op; vr = orr; }
            }
            if (lane == 0) { redv[2 * (w - 1)] = v; redv[2 * (w - 1) + 1] = sv; redi[2 * (w - 1)] = vp; redi[2 * (w - 1) + 1] = vr; }
            if (valid && row == q) ownv[0] = wk;
            if (valid && row == n1) ownv[2] = wk;
            if (valid && row == n2) ownv[3] = wk;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (this thread's stores to the global W are complete)
        }
        __syncthreads();
        if (leader) {
            double v = redv[0], sv = redv[1];
            int vp = redi[0], vr = redi[1];
            for (int i = 1; i < 4; ++i)
                if (redv[2 * i] > v || (redv[2 * i] == v && redi[2 * i] < vp)) { v = redv[2 * i]; sv = redv[2 * i + 1]; vp = redi[2 * i]; vr = redi[2 * i + 1]; }
            if (lane < BKM_F) {
                double val = 0.0;
                if (lane == 1) val = v;
                if (lane == 2) val = sv;
                if (lane == 3) val = __longlong_as_double(((long long)vp << 32) | (unsigned)vr);
                if (lane == 4 && owner_here(q)) val = ownv[0];
                if (lane == 5 && owner_here(n1)) val = ownv[2];
                if (lane == 6 && owner_here(n2)) val = ownv[3];
                bkm_post(a.msg, g, seq, lane, val);
            }
            bool ok = bkm_gather(a.msg, G, seq, mv, fail_word, a.spin_limit);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // (mv[] is read across lanes below: LDS is in order per wave)
            if (slot_q < 0) ok = false;   // (the pivot row's W entries are not in the cache: cannot happen, see the header)
            ++seq;
            // W rows of the next two positions that are not in the cache yet: everything stored before this hop is visible
            const int s1 = n1 >= 0 ? find_slot(n1) : 0, s2 = n2 >= 0 ? find_slot(n2) : 0;
            double f1 = 0.0, f2 = 0.0;
            if (s1 < 0 && lane < kb) f1 = __hip_atomic_load(a.Wv + n1 + (size_t)lane * Np, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (s2 < 0 && lane < kb) f2 = __hip_atomic_load(a.Wv + n2 + (size_t)lane * Np, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // the column's maximum over all workgroups: largest |v|, smallest position among ties
            double cv = -1.0, cs = 0.0;
            int cp = 0x7fffffff, cr = -1;
            for (int gg = lane; gg < G; gg += 64) {
                const double mvv = mv[gg * BKM_F + 1];
                const long long pr = __double_as_longlong(mv[gg * BKM_F + 3]);
                const int mp = (int)(pr >> 32), mr = (int)(pr & 0xffffffffll);
                if (mvv > cv || (mvv == cv && mp < cp)) { cv = mvv; cs = mv[gg * BKM_F + 2]; cp = mp; cr = mr; }
            }
            for (int off = 32; off > 0; off >>= 1) {
                const double ov = __shfl_down(cv, off), os = __shfl_down(cs, off);
                const int op = __shfl_down(cp, off), orr = __shfl_down(cr, off);
                if (ov > cv || (ov == cv && op < cp)) { cv = ov; cs = os; cp = op; cr = orr; }
            }
            cv = __shfl(cv, 0); cs = __shfl(cs, 0); cp = __shfl(cp, 0); cr = __shfl(cr, 0);
            const double colmax = cv < 0.0 ? 0.0 : cv;
            const double akk = mv[((q - p0) / BKM_ROWS) * BKM_F + 4];
            const double absakk = fabs(akk);
            int decision = 1;
            if (!(fmax(absakk, colmax) > 0.0) || !(absakk <= DBL_MAX) || colmax >= DBL_MAX) decision = 0;
            else if (absakk < BK_ALPHA * colmax) decision = 2;
            if (lane == 0) {
                ctl->q = q; ctl->n1 = n1; ctl->n2 = n2; ctl->rmax = cr; ctl->imax = cp; ctl->decision = decision;
                ctl->fail = ok ? 0 : 1; ctl->colmax = colmax; ctl->akk = akk; ctl->p21 = cs;
                ctl->wk_n1 = n1 >= 0 ? mv[((n1 - p0) / BKM_ROWS) * BKM_F + 5] : 0.0;
                ctl->wk_n2 = n2 >= 0 ? mv[((n2 - p0) / BKM_ROWS) * BKM_F + 6] : 0.0;
                ctl->wn_q = 0.0; ctl->wn_n1 = 0.0; ctl->wn_n2 = 0.0; ctl->p22 = 0.0; ctl->rowmax = 0.0;
                if (!ok) __hip_atomic_store(fail_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // (the fresh rows go into free slots -- the cache holds at most q, n1, n2 at this point; every lane derives both
            // slots from what it read before any of these writes)
            {
                int fs[2] = {-1, -1}, nf = 0;
                for (int i = 0; i < 4; ++i)
                    if (wc_row[i] < 0 && nf < 2) fs[nf++] = i;
                const int t1 = s1 < 0 ? fs[0] : -1, t2 = s2 < 0 ? fs[s1 < 0 ? 1 : 0] : -1;
                if ((s1 < 0 && t1 < 0) || (s2 < 0 && t2 < 0)) {
                    if (lane == 0) { ctl->fail = 2; __hip_atomic_store(fail_word, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                } else {
                    if (t1 >= 0) { if (lane < kb) wc[t1 * 64 + lane] = f1; if (lane == 0) wc_row[t1] = n1; }
                    if (t2 >= 0) { if (lane < kb) wc[t2 * 64 + lane] = f2; if (lane == 0) wc_row[t2] = n2; }
                }
            }
        } else {
            // the columns of the next two positions' rows, ahead of time (kept if they are held already)
            if (valid && active) {
                const double o1 = pf_v1, o2 = pf_v2;
                const int r1 = pf_r1, r2 = pf_r2;
                pf_v1 = n1 < 0 ? 0.0 : (r1 == n1 ? o1 : (r2 == n1 ? o2 : a0(row, n1)));
                pf_v2 = n2 < 0 ? 0.0 : (r1 == n2 ? o1 : (r2 == n2 ? o2 : a0(row, n2)));
            }
            pf_r1 = n1; pf_r2 = n2;
        }
        __syncthreads();
        if (ctl->fail) break;
        int decision = ctl->decision;
        const int rmax = ctl->rmax, imax = ctl->imax;
        double wn = 0.0;
        if (decision == 2) {
            // ---- phase B: the partner column (row rmax), brought up to date
            double a0n = 0.0;
            if (leader) {
                if (lane < kb) wtmp[lane] = __hip_atomic_load(a.Wv + rmax + (size_t)lane * Np, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (active) {
                a0n = a0(row, rmax);
            }
            __syncthreads();
            if (!leader) {
                if (active) {
                    double acc = a0n;
                    for (int c = 0; c < kb; ++c) acc -= LW[c * BKM_ROWS + t] * wtmp[c];
                    wn = acc;
                }
                double v = (active && row != rmax) ? fabs(wn) : -1.0;
                for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off));
                if (lane == 0) redv[2 * (w - 1)] = v;
                if (valid && row == q) ownv[0] = wn;
                if (valid && row == rmax) ownv[1] = wn;
                if (valid && row == n1) ownv[2] = wn;
                if (valid && row == n2) ownv[3] = wn;
            }
            __syncthreads();
            if (leader) {
                const double v = fmax(fmax(redv[0], redv[2]), fmax(redv[4], redv[6]));
                if (lane < BKM_F) {
                    double val = 0.0;
                    if (lane == 1) val = v;
                    if (lane == 2 && owner_here(rmax)) val = ownv[1];
                    if (lane == 3 && owner_here(q)) val = ownv[0];
                    if (lane == 4 && owner_here(n1)) val = ownv[2];
                    if (lane == 5 && owner_here(n2)) val = ownv[3];
                    bkm_post(a.msg, g, seq, lane, val);
                }
                const bool ok = bkm_gather(a.msg, G, seq, mv, fail_word, a.spin_limit);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                ++seq;
                double rm = -1.0;
                for (int gg = lane; gg < G; gg += 64) rm = fmax(rm, mv[gg * BKM_F + 1]);
                for (int off = 32; off > 0; off >>= 1) rm = fmax(rm, __shfl_down(rm, off));
                rm = __shfl(rm, 0);
                const double p22 = mv[((rmax - p0) / BKM_ROWS) * BKM_F + 2];
                const double absakk = fabs(ctl->akk), colmax = ctl->colmax;
                int d2 = 4;
                if (absakk >= BK_ALPHA * colmax * (colmax / rm)) d2 = 1;
                else if (fabs(p22) >= BK_ALPHA * rm) d2 = 3;
                if (lane == 0) {
                    ctl->decision = d2; ctl->p22 = p22; ctl->rowmax = rm; ctl->fail = ok ? 0 : 1;
                    ctl->wn_q = mv[((q - p0) / BKM_ROWS) * BKM_F + 3];
                    ctl->wn_n1 = n1 >= 0 ? mv[((n1 - p0) / BKM_ROWS) * BKM_F + 4] : 0.0;
                    ctl->wn_n2 = n2 >= 0 ? mv[((n2 - p0) / BKM_ROWS) * BKM_F + 5] : 0.0;
                    if (!ok) __hip_atomic_store(fail_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            __syncthreads();
            if (ctl->fail) break;
            decision = ctl->decision;
        }
        // ---- interchange (positions only) and elimination
        const int kstep = decision == 4 ? 2 : 1;
        const int swap_to = decision == 3 ? k : (decision == 4 ? k + 1 : -1);
        const int a_row = decision == 3 ? q : (decision == 4 ? n1 : -1);   // the row that leaves position swap_to (NOT read
                                                                           // from prow: the leader rewrites it in this stage)
        const bool swapped = swap_to >= 0 && swap_to != imax;
        const double akk = ctl->akk, p21 = ctl->p21, p22 = ctl->p22;
        if (!leader) {
            if (valid && swapped) {
                if (row == a_row) mypos = imax;
                else if (row == rmax) mypos = swap_to;
            }
            if (active) {
                if (decision == 0) {
                    LW[kb * BKM_ROWS + t] = 0.0;
                    put<true>(a.Wv + row + (size_t)kb * Np, 0.0);
                    if (row == q) {
                        active = false;
                        a.dvec[k] = 0.0; a.doff[k] = 0.0; a.ptype[k] = 1;
                        atomicCAS(&a.st->info, 0, k + 1);   // (agent-scope atomic)
                    }
                } else if (kstep == 1) {
                    const int pr = decision == 3 ? rmax : q;
                    const double d = decision == 3 ? p22 : akk, wv = decision == 3 ? wn : wk;
                    const double rd = 1.0 / d;
                    LW[kb * BKM_ROWS + t] = row == pr ? 0.0 : wv * rd;
                    put<true>(a.Wv + row + (size_t)kb * Np, wv);
                    if (row == pr) {
                        active = false;
                        a.dvec[k] = d; a.doff[k] = 0.0; a.ptype[k] = 1;
                    }
                } else {
                    // [l1 l2] = [w1 w2] inv([[p11 p21], [p21 p22]]), scaled as dsytf2 does (no overflow from a tiny p21)
                    const double p11 = akk;
                    const double d11 = p22 / p21, d22 = p11 / p21;
                    const double tt = 1.0 / (d11 * d22 - 1.0) / p21;
                    const bool piv = row == q || row == rmax;
                    LW[kb * BKM_ROWS + t] = piv ? 0.0 : tt * (d11 * wk - wn);
                    LW[(kb + 1) * BKM_ROWS + t] = piv ? 0.0 : tt * (d22 * wn - wk);
                    put<true>(a.Wv + row + (size_t)kb * Np, wk);
                    put<true>(a.Wv + row + (size_t)(kb + 1) * Np, wn);
                    if (row == q) { active = false; a.dvec[k] = p11; a.doff[k] = p21; a.ptype[k] = 2; }
                    if (row == rmax) { active = false; a.dvec[k + 1] = p22; a.doff[k + 1] = 0.0; a.ptype[k + 1] = 3; }
                }
            }
        } else {
            // leader: the newest W entries of the candidate rows (from the messages), the table of panel positions
            auto append = [&](int r, double v0, double v1) {
                const int s = r >= 0 ? find_slot(r) : -1;
                if (s >= 0 && lane == 0) { wc[s * 64 + kb] = v0; if (kstep == 2) wc[s * 64 + kb + 1] = v1; }
            };
            if (decision == 0) { append(n1, 0.0, 0.0); append(n2, 0.0, 0.0); }
            else if (decision == 1) { append(n1, ctl->wk_n1, 0.0); append(n2, ctl->wk_n2, 0.0); }
            else if (decision == 3) { append(n1, ctl->wn_n1, 0.0); append(n2, ctl->wn_n2, 0.0); append(q, ctl->wn_q, 0.0); }
            else { append(n1, ctl->wk_n1, ctl->wn_n1); append(n2, ctl->wk_n2, ctl->wn_n2); }
            // keep the rows of the next three positions only (the table as it will be after the interchange: every lane from
            // its own reads, the table itself is rewritten behind them)
            int nxt[3];
            {
                const int kn = kb + kstep;
                for (int i = 0; i < 3; ++i) {
                    const int ps = p0 + kn + i;
                    int r = ps < Np ? prow[kn + i] : -1;
                    if (swapped && ps == swap_to) r = rmax;
                    if (swapped && ps == imax) r = a_row;
                    nxt[i] = ps < Np ? r : -1;
                }
            }
            const int rc4 = lane < 4 ? wc_row[lane] : -1;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            if (lane < 4 && !(rc4 >= 0 && (rc4 == nxt[0] || rc4 == nxt[1] || rc4 == nxt[2]))) wc_row[lane] = -1;
            if (swapped && lane == 0) {
                prow[swap_to - p0] = rmax;
                if (imax - p0 < 68) prow[imax - p0] = a_row;
            }
        }
        kb += kstep;
        __syncthreads();
    }
    if (ctl->fail) return;
    // ---- the rows in position order: operands of the trailing update, rowof[], the lists of displaced positions
    if (valid) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int ps = mypos;
        a.rowof[ps] = row;
        const bool trailing = ps >= p0 + kb;
        const int ncol = trailing ? kb : ps - p0;   // (an eliminated row keeps the columns in front of its own)
        for (int c = 0; c < BK_NBW; ++c) {
            a.LWp[ps + (int64_t)c * a.ldw] = c < ncol ? LW[c * BKM_ROWS + t] : 0.0;
            a.Wp[ps + (int64_t)c * a.ldw] = (trailing && c < kb) ? __hip_atomic_load(a.Wv + row + (size_t)c * Np, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        }
        if (ps != row) {
            a.dlist[64 + atomicAdd(a.cnt + 1, 1)] = ps;
            if (trailing) a.dlist[atomicAdd(a.cnt, 1)] = ps;
        }
    }
    if (g == 0 && tid == 0) { st_store(&a.st->p0, p0); st_store(&a.st->kb, kb); st_store(&a.st->k, p0 + kb); }
}

// The interchanges of one panel, applied to everything that is stored in position order (two kernels: every source is read
// before any destination is written).  blockIdx.y: 0..63 displaced trailing positions x -- row / column x of the trailing
// matrix becomes A0(rowof[x], rowof[j]); 64..191 displaced positions -- their rows of the previous columns and of perm.
__global__ __launch_bounds__(256) void bkp_perm_gather_kernel(const double* __restrict__ F, int64_t ld, int Np, const BkState* st,
                                                              const int* __restrict__ rowof, const int* __restrict__ dlist,
                                                              double* __restrict__ tmp, const int* __restrict__ perm,
                                                              int* __restrict__ perm_tmp, const int* __restrict__ cnt) {
    const int y = blockIdx.y;
    const int p0 = st->p0, r0 = st->p0 + st->kb;
    if (st->kb <= 0 || st->fail != 0) return;
    if (y < 64) {
        if (y >= cnt[0]) return;
        const int x = dlist[y], rx = rowof[x];
        for (int j = r0 + blockIdx.x * blockDim.x + threadIdx.x; j < Np; j += gridDim.x * blockDim.x) {
            const int rj = rowof[j];
            tmp[(size_t)y * Np + j] = rx >= rj ? F[rx + (int64_t)rj * ld] : F[rj + (int64_t)rx * ld];
        }
    } else {
        const int si = y - 64;
        if (si >= cnt[1]) return;
        const int ps = dlist[64 + si], rs = rowof[ps];
        for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < p0; j += gridDim.x * blockDim.x)
            tmp[(size_t)y * Np + j] = F[rs + (int64_t)j * ld];
        if (blockIdx.x == 0 && threadIdx.x == 0) perm_tmp[si] = perm[rs];
    }
}

// ... and the panel's columns of L (the multi-workgroup panel writes nothing into F while other workgroups still read it)
__global__ __launch_bounds__(256) void bkp_perm_scatter_kernel(double* __restrict__ F, int64_t ld, int Np, const BkState* st,
                                                               const int* __restrict__ dlist, const double* __restrict__ tmp,
                                                               int* __restrict__ perm, const int* __restrict__ perm_tmp,
                                                               const double* __restrict__ LWp, int64_t ldw,
                                                               const double* __restrict__ dvec, const int* __restrict__ ptype,
                                                               const int* __restrict__ cnt) {
    const int y = blockIdx.y;
    const int p0 = st->p0, kb = st->kb, r0 = p0 + kb;
    if (kb <= 0 || st->fail != 0) return;
    if (y < 64) {
        if (y >= cnt[0]) return;
        const int x = dlist[y];
        for (int j = r0 + blockIdx.x * blockDim.x + threadIdx.x; j < Np; j += gridDim.x * blockDim.x) {
            const double v = tmp[(size_t)y * Np + j];
            if (x >= j) F[x + (int64_t)j * ld] = v; else F[j + (int64_t)x * ld] = v;
        }
    } else if (y < 192) {
        const int si = y - 64;
        if (si >= cnt[1]) return;
        const int ps = dlist[64 + si];
        for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < p0; j += gridDim.x * blockDim.x) F[ps + (int64_t)j * ld] = tmp[(size_t)y * Np + j];
        if (blockIdx.x == 0 && threadIdx.x == 0) perm[ps] = perm_tmp[si];
    } else {
        const int c = y - 192;   // panel column
        if (c >= kb) return;
        const int kc = p0 + c;
        for (int i = kc + blockIdx.x * blockDim.x + threadIdx.x; i < Np; i += gridDim.x * blockDim.x) {
            double v = LWp[i + (int64_t)c * ldw];
            if (i == kc) v = dvec[kc];
            if (i == kc + 1 && ptype[kc] == 2) v = 0.0;   // L is unit lower: the 2x2 block's off-diagonal lives in doff
            F[i + (int64_t)kc * ld] = v;
        }
    }
}

// Trailing update of one panel on the matrix cores: A[i, j] -= sum_c LW[i, c] W[j, c] for i >= j >= p0 + kb, 128 x 128
// tiles in absolute tile coordinates (tile (tm, tn) = rows 128 tm.., columns 128 tn..), K = BK_NBW zero-padded columns.
// The panel's position comes from device memory; the host launches the tiles of the largest region it can be.
__global__ __launch_bounds__(256, 3) void bkp_update_kernel(double* __restrict__ F, int64_t ld, int Np, const BkState* st,
                                                            const double* __restrict__ W, const double* __restrict__ LW,
                                                            int64_t ldw, int tile0) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int kb = st->kb;
    if (kb <= 0 || st->fail != 0) return;
    const int r0 = st->p0 + kb;
    // lower tiles of the square of `nt` tile rows that starts at tile0, enumerated row by row
    const int b = blockIdx.x;
    int tm = (int)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
    while (tm * (tm + 1) / 2 > b) --tm;
    while ((tm + 1) * (tm + 2) / 2 <= b) ++tm;
    const int tn = tile0 + (b - tm * (tm + 1) / 2);
    tm += tile0;
    if (128 * (tm + 1) <= r0 || 128 * (tn + 1) <= r0 || 128 * tm >= Np) return;  // nothing of this tile is in the trailing matrix
    int tid = threadIdx.x;
    v4f64 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = v4f64{0.0, 0.0, 0.0, 0.0};
    (void)gemm_nt_mainloop<2, 2, 4, 0, 8>(acc, LW + (int64_t)128 * tm, ldw, W + (int64_t)128 * tn, ldw, BK_NBW / 8, smem_raw, tid);
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1, l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int col = 128 * tn + 64 * wn + 16 * ni + l4 + 4 * r;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int row = 128 * tm + 64 * wm + 16 * mi + l15;
                if (col >= r0 && row >= col && row < Np) F[row + (int64_t)col * ld] -= acc[ni][mi][r];
            }
        }
}

__global__ void bkp_init_kernel(BkState* st, int* __restrict__ perm, int Np) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < Np) perm[t] = t;
    if (t == 0) { st->k = 0; st->p0 = 0; st->kb = 0; st->info = 0; st->fail = 0; }
}

// dinv / dcoup of the block-diagonal D^-1, and the factored diagonal 64x64 blocks for linv64_kernel
__global__ void bk_finish_kernel(const double* __restrict__ F, int64_t ld, int Np, const double* __restrict__ dvec,
                                 const double* __restrict__ doff, const int* __restrict__ ptype,
                                 double* __restrict__ dinv, double* __restrict__ dcoup, double* __restrict__ dblk) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < Np) {
        const int pt = ptype[t];
        if (pt == 1) {
            dinv[t] = dvec[t] != 0.0 ? 1.0 / dvec[t] : 0.0;
            dcoup[t] = 0.0;
        } else {
            const int f = pt == 2 ? t : t - 1;  // first index of the pair
            const double p11 = dvec[f], p22 = dvec[f + 1], p21 = doff[f];
            const double d11 = p22 / p21, d22 = p11 / p21;
            const double tt = 1.0 / (d11 * d22 - 1.0) / p21;
            // inv = tt * [[d11, -1], [-1, d22]]
            dinv[t] = pt == 2 ? tt * d11 : tt * d22;
            dcoup[t] = -tt;
        }
    }
    // diagonal blocks (column-major 64x64, lower part; the diagonal entry is irrelevant for the unit-lower inverse)
    for (int64_t e = t; e < (int64_t)(Np / 64) * 4096; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = e >> 12;
        const int r = (int)(e & 63), c = (int)((e >> 6) & 63);
        dblk[e] = r > c ? F[(b * 64 + r) + (b * 64 + c) * ld] : (r == c ? 1.0 : 0.0);
    }
}

// reference `num_neg_ev` (src/LinearSolvers/lapack.jl:247-268) over the first N pivots: out[0] = #negative,
// out[1] = #(d == 0) hits
__global__ void bk_inertia_kernel(const double* __restrict__ dvec, const double* __restrict__ doff,
                                  const int* __restrict__ ptype, int64_t N, unsigned long long* out) {
    unsigned long long neg = 0, zer = 0;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < N; k += (int64_t)gridDim.x * blockDim.x) {
        const int pt = ptype[k];
        double d;
        if (pt == 1) d = dvec[k];
        else if (pt == 2) { const double tt = fabs(doff[k]); d = (dvec[k] / tt) * dvec[k + 1] - tt; }
        else d = 1.0;  // second index of a pair: the reference counts it as positive (d = t > 0)
        if (d < 0.0) ++neg;
        if (d == 0.0) ++zer;
    }
    for (int off = 32; off > 0; off >>= 1) { neg += __shfl_down(neg, off); zer += __shfl_down(zer, off); }
    if ((threadIdx.x & 63) == 0) {
        if (neg) atomicAdd(&out[0], neg);
        if (zer) atomicAdd(&out[1], zer);
    }
}

__global__ void bk_gather_kernel(double* __restrict__ dst, const double* __restrict__ src, const int* __restrict__ perm, int Np) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < Np) dst[t] = src[perm[t]];
}
__global__ void bk_scatter_kernel(double* __restrict__ dst, const double* __restrict__ src, const int* __restrict__ perm, int Np) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < Np) dst[perm[t]] = src[t];
}
// y <- D^-1 y with 1x1 / 2x2 blocks, in place (the first index of a pair writes both entries)
__global__ void bk_dsolve_kernel(double* __restrict__ y, const double* __restrict__ dinv, const double* __restrict__ dcoup,
                                 const int* __restrict__ ptype, int Np) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Np) return;
    const int pt = ptype[t];
    if (pt == 1) y[t] *= dinv[t];
    else if (pt == 2) {
        const double y1 = y[t], y2 = y[t + 1];
        y[t] = dinv[t] * y1 + dcoup[t] * y2;
        y[t + 1] = dcoup[t + 1] * y1 + dinv[t + 1] * y2;
    }
}

}  // namespace mnk

using namespace mnk;

// Factor the matrix currently in ls->fact (lower triangle, padded with a unit diagonal) by Bunch-Kaufman.
// `multi`: panels by bkp_panel_mw_kernel (all its workgroups resident: the whole factorization holds the device
// arbiter's turn) wherever a workgroup per 256 rows fits the device; mnk_ls_bk_failed tells afterwards whether one of
// its bounded waits expired (the caller then transfers the matrix again and comes back with multi = false).
static int run_bunchkaufman(mnk_ls* ls, bool multi) {
    hipStream_t s = ls->ctx->stream;
    const int Np = (int)ls->Np;
    const int64_t ld = ls->ld;
    double* F = ls->fact.p;
    int rc = 0;
    if (!ls->bk_perm.p) {
        rc |= ls->bk_perm.alloc(Np);
        rc |= ls->bk_ptype.alloc(Np);
        rc |= ls->bk_doff.alloc(Np);
        rc |= ls->bk_dcoup.alloc(Np);
        rc |= ls->bk_state.alloc(sizeof(BkState));
        rc |= ls->bk_work.alloc((size_t)2 * Np * BK_NBW + SLACK);
        if (rc) return -2;
    }
    if (multi && !ls->bk_wv.p) {
        rc |= ls->bk_wv.alloc((size_t)Np * BK_NB + SLACK);
        rc |= ls->bk_tmp.alloc((size_t)192 * Np + SLACK);
        rc |= ls->bk_msg.alloc((size_t)BKM_RING * BKM_GMAX * BKM_F * 2);
        rc |= ls->bk_aux.alloc((size_t)Np + 192 + 128 + 64);
        if (rc) {   // (not enough memory for the staging buffers: the one-workgroup panel needs none)
            (void)hipGetLastError();
            ls->bk_wv.release(); ls->bk_tmp.release(); ls->bk_msg.release(); ls->bk_aux.release();
            multi = false;
            rc = 0;
        }
    }
    BkState* st = reinterpret_cast<BkState*>(ls->bk_state.p);
    double* W = ls->bk_work.p;
    double* LW = W + (size_t)Np * BK_NBW;
    const size_t smem = 2 * 8 * ((128 + 16) + (128 + 16)) * sizeof(double);
    const size_t smem_mw = (size_t)(64 * BKM_ROWS + 4 * 64 + 64 + BKM_GMAX * BKM_F + 8 + 4) * sizeof(double) + (8 + 68 + 4) * sizeof(int) + sizeof(BkmCtl) + 64;
    {
        static std::atomic<uint64_t> attr_devs{0};
        int dev = 0;
        MNK_HIP(hipGetDevice(&dev));
        if (!(attr_devs.load(std::memory_order_relaxed) >> (dev & 63) & 1)) {
            MNK_HIP(hipFuncSetAttribute((const void*)bkp_update_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            MNK_HIP(hipFuncSetAttribute((const void*)bkp_panel_mw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_mw));
            attr_devs.fetch_or(1ull << (dev & 63), std::memory_order_relaxed);
        }
    }
    int gcap = std::min(BKM_GMAX, ls->ctx->num_cu);
    if (ls->bk_max_wgs > 0) gcap = std::min(gcap, ls->bk_max_wgs);
    if (multi) {
        rc = mnk_persist_begin(ls->ctx, s);
        if (rc) return rc;
        (void)hipMemsetAsync(ls->bk_msg.p, 0, ls->bk_msg.n * sizeof(unsigned long long), s);
    }
    auto body = [&]() -> int {
        MNK_HIP(hipMemsetAsync(ls->info_dev.p, 0, sizeof(int), s));
        hipLaunchKernelGGL(bkp_init_kernel, dim3((Np + 255) / 256), dim3(256), 0, s, st, ls->bk_perm.p, Np);
        // a panel takes 63 or 64 columns: the p-th panel starts at column >= 63 p, so its update touches no tile row above
        // floor(63 (p + 1) / 128); one more (empty) round covers the case that every panel was 63 wide
        const int npanel = (Np + BK_NB - 2) / (BK_NB - 1) + 1;
        int* rowof = ls->bk_aux.p;
        int* dlist = rowof ? rowof + Np : nullptr;
        int* perm_tmp = rowof ? dlist + 192 : nullptr;
        int* cnt = rowof ? perm_tmp + 128 : nullptr;   // two {nT, nS} pairs, used alternately
        if (multi) MNK_HIP(hipMemsetAsync(cnt, 0, 4 * sizeof(int), s));
        for (int p = 0; p < npanel; ++p) {
            if ((int64_t)(BK_NB - 1) * p >= Np) break;
            const int rows_max = Np - (BK_NB - 1) * p;   // (the panel starts at column >= 63 p)
            const int G = (rows_max + BKM_ROWS - 1) / BKM_ROWS;
            if (multi && G <= gcap) {
                BkMw a{F, ld, Np, st, W, LW, (int64_t)Np, ls->bk_wv.p, ls->bk_msg.p, rowof, dlist, nullptr, nullptr, ls->dvec.p, ls->bk_doff.p,
                       ls->bk_ptype.p, (unsigned)(1 + 130 * p), ls->bk_spin_limit, ls->debug_bk_missing};
                a.cnt = cnt + 2 * (p & 1);
                a.cnt_next = cnt + 2 * ((p + 1) & 1);
                hipLaunchKernelGGL(bkp_panel_mw_kernel, dim3(G), dim3(BKM_T), smem_mw, s, a);
                hipLaunchKernelGGL(bkp_perm_gather_kernel, dim3(8, 192), dim3(256), 0, s, F, ld, Np, st, rowof, dlist, ls->bk_tmp.p,
                                   ls->bk_perm.p, perm_tmp, a.cnt);
                hipLaunchKernelGGL(bkp_perm_scatter_kernel, dim3(8, 256), dim3(256), 0, s, F, ld, Np, st, dlist, ls->bk_tmp.p,
                                   ls->bk_perm.p, perm_tmp, LW, (int64_t)Np, ls->dvec.p, ls->bk_ptype.p, a.cnt);
            } else {
                hipLaunchKernelGGL(bkp_panel_kernel, dim3(1), dim3(BK_T), 0, s, F, ld, Np, st, W, LW, (int64_t)Np, ls->dvec.p, ls->bk_doff.p,
                                   ls->bk_ptype.p, ls->bk_perm.p);
            }
            const int tile0 = (int)(((int64_t)(BK_NB - 1) * (p + 1)) / 128);
            const int nt = Np / 128 - tile0;
            if (nt > 0)
                hipLaunchKernelGGL(bkp_update_kernel, dim3((unsigned)(nt * (nt + 1) / 2)), dim3(256), smem, s, F, ld, Np, st, W, LW,
                                   (int64_t)Np, tile0);
        }
        hipLaunchKernelGGL(bk_finish_kernel, dim3((Np + 255) / 256), dim3(256), 0, s, F, ld, Np, ls->dvec.p, ls->bk_doff.p,
                           ls->bk_ptype.p, ls->dinv.p, ls->bk_dcoup.p, ls->dblk.p);
        MNK_HIP(hipGetLastError());
        // LAPACK-style info -> the solver's info word
        MNK_HIP(hipMemcpyAsync(ls->info_dev.p, &st->info, sizeof(int), hipMemcpyDeviceToDevice, s));
        return 0;
    };
    rc = body();
    if (multi) rc = mnk_persist_end(ls->ctx, s, rc);
    if (rc) return rc;
    ls->bk_active = true;
    ls->bk_multi_last = multi;
    return 0;
}

int mnk_ls_run_bunchkaufman(mnk_ls* ls) {
    const bool multi = ls->bk_panel_wgs != 1 && !ls->bk_mw_blocked;
    int rc = run_bunchkaufman(ls, multi);
    if (rc || !ls->bk_multi_last) return rc;
    // the multi-workgroup panels wait for each other's messages with a bound: did one expire?  (This tier is entered from
    // mnk_ls_fetch_info, which waits for the stream anyway.)
    MNK_HIP(mnk::stream_wait(ls->ctx->stream));
    int fail = 0;
    MNK_HIP(mnk::d2h_copy(&fail, &reinterpret_cast<BkState*>(ls->bk_state.p)->fail, sizeof(int), ls->ctx->stream));
    if (fail == 0) return 0;
    // (another process' kernels on the CUs, as for the persistent schedules of factor.hip: redo with one workgroup per panel
    // and stay there)
    ++ls->bk_mw_fallbacks;
    ls->bk_mw_blocked = true;
    MNK_REQUIRE((bool)ls->retransfer, "Bunch-Kaufman tier: a panel hand-off timed out and the matrix cannot be transferred again");
    rc = ls->retransfer();
    if (rc) return rc;
    return run_bunchkaufman(ls, false);
}

int mnk_ls_bk_permute(mnk_ls* ls, double* x, double* tmp, bool forward) {
    hipStream_t s = ls->ctx->stream;
    const int Np = (int)ls->Np;
    if (forward) hipLaunchKernelGGL(bk_gather_kernel, dim3((Np + 255) / 256), dim3(256), 0, s, tmp, x, ls->bk_perm.p, Np);
    else hipLaunchKernelGGL(bk_scatter_kernel, dim3((Np + 255) / 256), dim3(256), 0, s, tmp, x, ls->bk_perm.p, Np);
    MNK_HIP(hipMemcpyAsync(x, tmp, (size_t)Np * sizeof(double), hipMemcpyDeviceToDevice, s));
    return 0;
}

int mnk_ls_bk_dsolve(mnk_ls* ls, double* y) {
    const int Np = (int)ls->Np;
    hipLaunchKernelGGL(bk_dsolve_kernel, dim3((Np + 255) / 256), dim3(256), 0, ls->ctx->stream, y, ls->dinv.p,
                       ls->bk_dcoup.p, ls->bk_ptype.p, Np);
    MNK_HIP(hipGetLastError());
    return 0;
}

int mnk_ls_bk_inertia(mnk_ls* ls, unsigned long long* out_dev) {
    const int blocks = (int)std::min<int64_t>(256, (ls->N + 255) / 256);
    hipLaunchKernelGGL(bk_inertia_kernel, dim3(blocks), dim3(256), 0, ls->ctx->stream, ls->dvec.p, ls->bk_doff.p,
                       ls->bk_ptype.p, ls->N, out_dev);
    MNK_HIP(hipGetLastError());
    return 0;
}
