// The 64x64 diagonal-block factorization of the pivot chain (potrf64w_core: one wave; potrf64q_core: the four waves of a
// strip's workgroup) and its scalar pieces -- shared by factor.hip and the leaf laboratory tools/hip/leaf_lab.hip.
// Reference operation: the unblocked part of dpotrf / dsytrf on a diagonal block (src/LinearSolvers/lapack.jl:145-148,164-167).
#pragma once
#include <cfloat>
#include <cmath>
#include <type_traits>
#include "common.h"

namespace mnk {

__device__ __forceinline__ double fast_rsqrt(double x) {
    // v_rsq_f64 seed + two Newton steps in fma form: ~1 ulp, no division / software sqrt on
    // the pivot chain
    double y = __builtin_amdgcn_rsq(x);
    double e = fma(-(x * y), y, 1.0);
    y = fma(0.5 * y, e, y);
    e = fma(-(x * y), y, 1.0);
    y = fma(0.5 * y, e, y);
    return y;
}
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

struct Piv4 {
    double c10, c20, c30, c21, c31, c32, s0, s1, s2, s3;
};

typedef double v4d __attribute__((ext_vector_type(4)));
#ifndef MNK_DIAG_FAST_LEAF
#define MNK_DIAG_FAST_LEAF 0
#endif

// Store of a result another CU will read after a flag.  WT (the persistent chain of the task-DAG schedule): write-through
// (sc1), so that the publishing workgroup needs no agent-scope release fence -- that fence (buffer_wbl2) writes back EVERY
// dirty line of the XCD's L2, and with a dozen strips per XCD storing their rows it grew the pivot chain's step from 24 to
// 38 us per block (measured: the step time followed the number of resident strips).
template <bool WT>
__device__ __forceinline__ void put(double* p, double v) {
    if (WT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// Growth monitor of the static-pivot LDL^T (the guard of BUNCHKAUFMAN's first tier, ls.h): the entries of V = L D are the
// entries of the successive Schur complements at the moment their column is eliminated, so max|v_ik| / max|a_ij| is the
// element growth of the elimination as far as it can be seen without extra passes.  Every kernel that produces V folds
// |v| into one word: max over the wave, one atomicMax on the bit pattern (NaN / Inf -> +Inf).
__device__ __forceinline__ void growth_fold(unsigned long long* word, double vm) {
    if (word == nullptr) return;
    if (!(vm <= DBL_MAX)) vm = __longlong_as_double(0x7ff0000000000000LL);
    for (int off = 32; off > 0; off >>= 1) vm = fmax(vm, __shfl_xor(vm, off));
    if ((threadIdx.x & 63) == 0 && vm > 0.0) {   // (look first: the word saturates early, the atomic is then skipped)
        const unsigned long long bits = (unsigned long long)__double_as_longlong(vm);
        if (bits > __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(word, bits);
    }
}

__device__ __forceinline__ double readlane_f64(double x, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}

// ---------------------------------------------------------------------------------------
// potrf64 on the matrix cores, ONE wave, no LDS, no barriers (the default diagonal-block kernel).
// The lower triangle of the 64x64 block lives in accumulator registers as ten 16x16 blocks in the
// transposed ("C^T") layout of gemm_f64.hip: register r of block (cb, b) at lane (l15, l4) holds
// A[16 cb + l15][16 b + l4 + 4 r].  In this layout register s of a block IS both the A operand
// (lane (i, k) <-> M[i][4 s + k]) and the B operand (lane (j, k) <-> M^T[4 s + k][j]) of a K = 4
// v_mfma_f64_16x16x4 product, so the whole factorization proceeds in 16 steps of 4 pivots without
// ever shuffling data between lanes:
//   1. the 4x4 pivot block is broadcast (v_readlane, 10 values) and factored redundantly by every
//      lane -- the same dependent rsqrt/rcp chain as the fused elimination kernel, the only serial part;
//   2. X_t^T = inv(L44) A_t^T for every block of the block column: one MFMA per block, the 4x4 inverse
//      supplied as the A operand on the rows of the pivot group;
//   3. rank-4 update of every trailing block: acc(cb2, cb1) -= X_t[cb1] X_t[cb2]^T, one MFMA per block;
// and after the four steps of a block column the inverse of its 16x16 diagonal block (needed by
// trsm64_mfma_kernel) by block forward substitution, again on MFMA (7 products).
// Measured against the 256-thread LDS/barrier kernel (potrf64_kernel): see DESIGN.md section 5.
// ---------------------------------------------------------------------------------------
template <bool LDL>
__device__ __forceinline__ void factor_piv4_vals(const double p00, const double p10, const double p11,
                                                 const double p20, const double p21, const double p22,
                                                 const double p30, const double p31, const double p32,
                                                 const double p33, const double pivot_tol, Piv4& P, double (&dg)[4],
                                                 int& fail) {
    fail = 0;
    if (LDL) {
        // (the reciprocal is started on the pivot as it is and the harmless value 1 is selected at the END -- fast_rcp(1.0) is
        // exactly 1.0, so this is what fast_rcp(zero ? 1.0 : d) returns, bit for bit, without the comparison -> scalar OR ->
        // selection hops in FRONT of every reciprocal of this dependent chain: ~70 of ~260 cycles per pivot, tools/hip/leaf_lab.hip)
        auto piv = [&](double d, double& sc, double& rec) {
            const bool zero = !(fabs(d) > pivot_tol) || !(fabs(d) <= DBL_MAX);
            const double r = fast_rcp(d);
            sc = zero ? 1.0 : r;  // harmless pivot; dvec records the zero
            rec = zero ? 0.0 : d;
        };
        piv(p00, P.s0, dg[0]);
        P.c10 = p10; P.c20 = p20; P.c30 = p30;
        const double x10 = p10 * P.s0, x20 = p20 * P.s0, x30 = p30 * P.s0;
        piv(fma(-x10, P.c10, p11), P.s1, dg[1]);
        P.c21 = fma(-x20, P.c10, p21);
        P.c31 = fma(-x30, P.c10, p31);
        const double x21 = P.c21 * P.s1, x31 = P.c31 * P.s1;
        piv(fma(-x21, P.c21, fma(-x20, P.c20, p22)), P.s2, dg[2]);
        P.c32 = fma(-x31, P.c21, fma(-x30, P.c20, p32));
        const double x32 = P.c32 * P.s2;
        piv(fma(-x32, P.c32, fma(-x31, P.c31, fma(-x30, P.c30, p33))), P.s3, dg[3]);
    } else {
        auto piv = [&](double t, double& sc, double& rec, int k) {
            const bool bad = !(t > 0.0) || !(t <= DBL_MAX);  // not positive definite / NaN / Inf
            fail = (bad && fail == 0) ? k + 1 : fail;
            const double r = fast_rsqrt(t);   // (as above: fast_rsqrt(1.0) is exactly 1.0)
            sc = bad ? 1.0 : r;
            rec = bad ? 1.0 : t * sc;
        };
        piv(p00, P.s0, dg[0], 0);
        P.c10 = p10 * P.s0; P.c20 = p20 * P.s0; P.c30 = p30 * P.s0;
        piv(fma(-P.c10, P.c10, p11), P.s1, dg[1], 1);
        P.c21 = fma(-P.c20, P.c10, p21) * P.s1;
        P.c31 = fma(-P.c30, P.c10, p31) * P.s1;
        piv(fma(-P.c21, P.c21, fma(-P.c20, P.c20, p22)), P.s2, dg[2], 2);
        P.c32 = fma(-P.c31, P.c21, fma(-P.c30, P.c20, p32)) * P.s2;
        piv(fma(-P.c32, P.c32, fma(-P.c31, P.c31, fma(-P.c30, P.c30, p33))), P.s3, dg[3], 3);
    }
}

// (one wave; `Lsh` / `Ish`: optional LDS copies of the factored block and of its four 16x16 inverses)
// potrf64w_core: the block is already in registers (Lt[cb][b], b <= cb, strict upper triangle of the diagonal
// 16x16 blocks zeroed); potrf64w_body loads it from the factor matrix first.
template <bool LDL, bool WT = false>
__device__ __forceinline__ void potrf64w_core(v4d (&Lt)[4][4], int64_t j0, double* __restrict__ Dout,
                                              double* __restrict__ inv16, double* __restrict__ dvec,
                                              double* __restrict__ dinv, int* __restrict__ info, double pivot_tol,
                                              double* Lsh, double* Ish, unsigned long long* __restrict__ vmax = nullptr) {
    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, l4 = lane >> 4;
    const v4d zero4 = {0.0, 0.0, 0.0, 0.0};
    double vm = 0.0;
    // EARLY REJECTION (round 5, option early_reject; info[2] != 0): the caller accepts POSITIVE DEFINITE matrices only (the sparse
    // condensed KKT system: reference src/KKT/Sparse/condensed.jl:138-141), so the first pivot that is not positive settles
    // `is_inertia_correct` -- info = -9, and every kernel behind this block drops its work (they all look at `info`): the
    // interior-point loop's rejected trials of inertia_correction! (19 of the 39 factorizations of the AC-OPF run of the bench
    // line) stop where LAPACK's dpotrf would, instead of running to the end as dsytrf does.  info[1] = last pivot whose D is valid.
    const int reject = LDL ? __hip_atomic_load(info + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        double aopinv[4] = {0.0, 0.0, 0.0, 0.0};
        // the finished columns of block column b (Xf[cb][tt]: pivot group tt of block (cb, b)) live OUTSIDE the accumulators:
        // a kernel with more than 256 registers gets its MFMA accumulators in AGPRs, where overwriting one component of a
        // block costs a round trip of the whole block through VGPRs (~50 v_accvgpr moves per pivot group, 18 % of the
        // kernel's instructions); the finished components of Lt are dead from here on (later updates add zeros to them)
        v4d Xf[4] = {zero4, zero4, zero4, zero4};
        // (-DMNK_DIAG_FAST_LEAF=1: a timing-only build whose pivot kernel does no arithmetic -- results void -- to see how much
        // of a factorization's time is the leaf's; tools/diag_fast_leaf.sh)
#pragma unroll
        for (int tt = 0; tt < (MNK_DIAG_FAST_LEAF ? 0 : 4); ++tt) {
            const int t = 4 * b + tt;
            // ---- 1. pivot block: A[16b + 4tt + jj][16b + 4tt + kk] sits in register tt of lane (4tt + jj) + 16 kk
            const double dsrc = Lt[b][b][tt];
            const double p00 = readlane_f64(dsrc, 4 * tt + 0);
            const double p10 = readlane_f64(dsrc, 4 * tt + 1), p11 = readlane_f64(dsrc, 4 * tt + 1 + 16);
            const double p20 = readlane_f64(dsrc, 4 * tt + 2), p21 = readlane_f64(dsrc, 4 * tt + 2 + 16),
                         p22 = readlane_f64(dsrc, 4 * tt + 2 + 32);
            const double p30 = readlane_f64(dsrc, 4 * tt + 3), p31 = readlane_f64(dsrc, 4 * tt + 3 + 16),
                         p32 = readlane_f64(dsrc, 4 * tt + 3 + 32), p33 = readlane_f64(dsrc, 4 * tt + 3 + 48);
            Piv4 P;
            double dg[4];
            int fail;
            factor_piv4_vals<LDL>(p00, p10, p11, p20, p21, p22, p30, p31, p32, p33, pivot_tol, P, dg, fail);
            if (!LDL && fail != 0 && lane == 0) atomicCAS(info, 0, (int)(j0 + 4 * t + fail));
            if (LDL && reject != 0) {   // (uniform: dg are the recorded pivots, 0 for a zero / non-finite one)
                const bool bad = !(dg[0] > 0.0) | !(dg[1] > 0.0) | !(dg[2] > 0.0) | !(dg[3] > 0.0);
                if (bad && lane == 0 && atomicCAS(info, 0, -9) == 0) info[1] = (int)(j0 + 63);
            }
            // entries of the 4x4 factor: l = L (unit lower for LDL), cv = d_k L (LDL) / L (Cholesky)
            const double l10 = LDL ? P.c10 * P.s0 : P.c10, l20 = LDL ? P.c20 * P.s0 : P.c20,
                         l30 = LDL ? P.c30 * P.s0 : P.c30, l21 = LDL ? P.c21 * P.s1 : P.c21,
                         l31 = LDL ? P.c31 * P.s1 : P.c31, l32 = LDL ? P.c32 * P.s2 : P.c32;
            const double rd0 = LDL ? 1.0 : P.s0, rd1 = LDL ? 1.0 : P.s1, rd2 = LDL ? 1.0 : P.s2, rd3 = LDL ? 1.0 : P.s3;
            // inverse of the 4x4 factor (forward substitution, uniform values)
            const double y00 = rd0, y11 = rd1, y22 = rd2, y33 = rd3;
            const double y10 = -(l10 * y00) * rd1;
            const double y20 = -fma(l21, y10, l20 * y00) * rd2;
            const double y30 = -fma(l32, y20, fma(l31, y10, l30 * y00)) * rd3;
            const double y21 = -(l21 * y11) * rd2;
            const double y31 = -fma(l32, y21, l31 * y11) * rd3;
            const double y32 = -(l32 * y22) * rd3;
            // A operand of step 2: lane (i, k) holds inv(L44)[i - 4tt][k] on the rows of the pivot group, else 0
            const int ii = l15 - 4 * tt;
            // entry (ii, l4) of a lower-triangular 4x4 matrix of uniform values for lane (ii, l4): the strictly lower part
            // by column then row (5 selections), the diagonal by column (3), zero elsewhere (2) -- instead of four row
            // vectors and a 4-way choice between them (13-14)
            const bool r1 = ii == 1, r2 = ii == 2, on_diag = ii == l4, below = (l4 < ii) & (ii < 4);  // (& not &&: no branch)
            auto sel44 = [&](double d0, double d1, double d2, double d3, double e10, double e20, double e30, double e21,
                             double e31, double e32, bool unit) {
                const double c0v = r1 ? e10 : (r2 ? e20 : e30), c1v = r2 ? e21 : e31;
                const double off = l4 == 0 ? c0v : (l4 == 1 ? c1v : e32);
                const double dia = unit ? 1.0 : (l4 == 0 ? d0 : (l4 == 1 ? d1 : (l4 == 2 ? d2 : d3)));
                const double lo = below ? off : 0.0;
                return on_diag ? dia : lo;
            };
            const double aop = sel44(y00, y11, y22, y33, y10, y20, y30, y21, y31, y32, LDL);
            aopinv[tt] = aop;
            const double ssel = l4 == 0 ? P.s0 : (l4 == 1 ? P.s1 : (l4 == 2 ? P.s2 : P.s3));
            // exact entries of the pivot rows of the diagonal block (from the scalar factorization)
            // (Cholesky: c IS l, one selection tree.  LDL^T: v = c by selection; l_ik = c_ik * s_k is the very product the
            // scalar factorization forms, so l comes from v with one multiplication -- bit-identical, 11 selections fewer)
            const double vpiv = sel44(dg[0], dg[1], dg[2], dg[3], P.c10, P.c20, P.c30, P.c21, P.c31, P.c32, false);
            const double lpiv = LDL ? (l4 < ii ? vpiv * ssel : vpiv) : vpiv;
            // ---- 2. X_t^T = inv(L44) A_t^T for every block of block column b
            double X[4], V[4];
#pragma unroll
            for (int cb = b; cb < 4; ++cb) {
                const v4d out = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, Lt[cb][b][tt], zero4, 0, 0, 0);
                double v = out[tt];
                double x = LDL ? v * ssel : v;
                if (cb == b) {
                    // rows above the pivot group: not part of the lower triangle; the pivot rows: exact values
                    // (lpiv / vpiv are already zero on the rows above the group: one selection each)
                    x = ii < 4 ? lpiv : x;
                    v = ii < 4 ? (LDL ? vpiv : lpiv) : v;
                }
                X[cb] = x;
                V[cb] = v;
                if (LDL) vm = fmax(vm, fabs(v));
                Xf[cb][tt] = x;
            }
            // ---- 3. rank-4 update of the trailing blocks: acc(cb2, cb1) -= X[cb1] (V|X)[cb2]^T
#pragma unroll
            for (int cb1 = b; cb1 < 4; ++cb1) {
                // columns up to the pivot group of block column b are final: no update (zero rows of the A operand)
                const double na = (cb1 == b && l15 < 4 * tt + 4) ? 0.0 : -X[cb1];
#pragma unroll
                for (int cb2 = cb1; cb2 < 4; ++cb2)
                    Lt[cb2][cb1] = __builtin_amdgcn_mfma_f64_16x16x4f64(na, LDL ? V[cb2] : X[cb2], Lt[cb2][cb1], 0, 0, 0);
            }
        }
        // ---- inverse of the 16x16 diagonal block (unit diagonal for LDL): Y = inv(L16), block forward substitution
        {
            v4d T, Y;
#pragma unroll
            for (int r = 0; r < 4; ++r) T[r] = (l15 == l4 + 4 * r) ? 1.0 : 0.0;
#pragma unroll
            for (int pg = 0; pg < 4; ++pg) {
                const v4d out = __builtin_amdgcn_mfma_f64_16x16x4f64(aopinv[pg], T[pg], zero4, 0, 0, 0);
                Y[pg] = out[pg];
                if (pg < 3) T = __builtin_amdgcn_mfma_f64_16x16x4f64(-Xf[b][pg], Y[pg], T, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                put<WT>(inv16 + b * 256 + (l4 + 4 * r) + 16 * l15, Y[r]);
                if (Ish != nullptr) Ish[b * 256 + (l4 + 4 * r) + 16 * l15] = Y[r];
            }
        }
        // ---- D and D^-1 of the 16 pivots of this block column, once per block column instead of once per pivot group by
        // lane 0: the pivots sit on the diagonal of the factored block -- entry (i, i) in register i >> 2 of lane
        // (l15 = i, l4 = i & 3) -- and D^-1 is the same fast_rcp of the same recorded pivot (0 recorded -> harmless pivot 1)
        {
            const int rsel = l15 >> 2;
            const double dsel = rsel == 0 ? Xf[b][0] : (rsel == 1 ? Xf[b][1] : (rsel == 2 ? Xf[b][2] : Xf[b][3]));
            if ((l15 & 3) == l4) {
                put<WT>(dvec + j0 + 16 * b + l15, dsel);
                put<WT>(dinv + j0 + 16 * b + l15, LDL ? fast_rcp(dsel == 0.0 ? 1.0 : dsel) : 1.0);
            }
        }
        // ---- store block column b of the factored block (column-major 64x64, lower part)
#pragma unroll
        for (int cb = b; cb < 4; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double v = (cb == b && l15 < l4 + 4 * r) ? 0.0 : Xf[cb][r];
                put<WT>(Dout + (16 * cb + l15) + 64 * (16 * b + l4 + 4 * r), v);
                if (Lsh != nullptr) Lsh[(16 * cb + l15) + 64 * (16 * b + l4 + 4 * r)] = v;
            }
    }
    if (LDL) growth_fold(vmax, vm);
}

// ---------------------------------------------------------------------------------------
// potrf64v_core (round 6): the one-wave leaf with its instruction count halved -- same recurrence, same layout, and (flags
// RCP3 = false, M44 = false) the same bits as potrf64w_core whenever no pivot of the block is zero / non-finite.
// tools/hip/leaf_lab.hip had the round-5 leaf at 28 780 cycles for ~4500 instructions, of which 876 are v_cndmask_b32: every
// lane factors the 4x4 pivot block redundantly (uniform values) and then SELECTS its own entry of inv(L44), of the pivot
// rows of V = L D and of D^-1 by comparisons (two selections of ten values + one of four per group: ~55 v_cndmask_b32 and
// their compares), and every pivot carries its zero / non-finite test in front of its reciprocal's use.  Here:
//   * one indicator vector per entry of the 4x4 pattern, set up once per call -- I_e(lane) = 1 on the lanes (l15 & 3, l4)
//     that own entry e, else 0 -- turns every selection into a sum of products e_0 I_0 + e_1 I_1 + ...: ONE v_fma_f64 per
//     candidate instead of two v_cndmask_b32 (+ compares), and the sums are exact (all terms but one are +-0).  The
//     pattern repeats in every 4-row group of the 16 rows: the A operand of the block solve needs no zeros outside the
//     pivot group (those rows of the product land in the three result registers nobody reads), and above the group the
//     diagonal block's rows of X / V are strictly upper triangle -- never stored (the stores mask it), never an operand
//     of anything that is (rows of the A operand that `na` zeroes; columns of results in the upper triangle);
//   * the pivots' zero / non-finite tests leave the chain: the four pivots are tested TOGETHER behind it (min |d| against
//     the tolerance, a sum that is finite iff all four are), and only a group that fails repeats its scalar factorization
//     with the guarded code of potrf64w_core (factor_piv4_vals) -- a wave-uniform branch nobody takes on the matrices the
//     interior-point loop accepts;
//   * M44: the block solves X_t^T = inv(L44) A_t^T and the 4x4 steps of the 16x16 inverses on v_mfma_f64_4x4x4_4b (four 4x4x4
//     products, 4 passes) instead of v_mfma_f64_16x16x4 (16 passes) three quarters of whose result was thrown away;
//   * RCP3: reciprocal = v_rcp_f64 (2^-24) + ONE third-order step r (1 + e + e^2), e = 1 - d r: three dependent fma
//     instead of four, <= 1 ulp.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double fast_rcp3(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r, 1.0);
    return fma(r, fma(e, e, e), r);
}
__device__ __forceinline__ double fast_rsqrt3(double x) {
    // v_rsq_f64 (2^-24) + one third-order step: y (1 + e/2 + 3 e^2 / 8), e = 1 - x y^2
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-(x * y), y, 1.0);
    return fma(y * e, fma(0.375, e, 0.5), y);
}
__device__ __forceinline__ double min_abs(double a, double b) {   // min(|a|, |b|) of the hardware (a NaN operand loses: callers test for NaN apart)
    double r;
    asm("v_min_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ int or3(int a, int b, int c) {
    int r;
    asm("v_or3_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// (keeps a lane constant in its register: no rematerialization by compare + select; `tie`: a value of the call -- without it the
// constants of potrf64v_core are hoisted to the top of the persistent chain kernel, spilled to scratch there and reloaded in front
// of every leaf)
__device__ __forceinline__ double opaque(double x, int tie = 0) {
    asm volatile("" : "+v"(x) : "s"(tie));
    return x;
}

template <bool LDL, bool WT = false, bool M44 = true, bool RCP3 = true, bool LOOK = false>
__device__ __forceinline__ void potrf64v_core(v4d (&Lt)[4][4], int64_t j0, double* __restrict__ Dout,
                                              double* __restrict__ inv16, double* __restrict__ dvec,
                                              double* __restrict__ dinv, int* __restrict__ info, double pivot_tol,
                                              double* Lsh, double* Ish, unsigned long long* __restrict__ vmax = nullptr) {
    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, l4 = lane >> 4, i4 = l15 & 3;
    const v4d zero4 = {0.0, 0.0, 0.0, 0.0};
    double vm = 0.0;
    const int reject = LDL ? __hip_atomic_load(info + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    // indicator vectors of the 4x4 pattern: entry (i, k) lives on the lanes with (l15 & 3) == i, l4 == k
    const int tie = (int)j0;
    auto ind = [&](int i, int k) { return opaque((i4 == i && l4 == k) ? 1.0 : 0.0, tie); };
    const double I10 = ind(1, 0), I20 = ind(2, 0), I30 = ind(3, 0), I21 = ind(2, 1), I31 = ind(3, 1), I32 = ind(3, 2);
    const double D0 = ind(0, 0), D1 = ind(1, 1), D2 = ind(2, 2), D3 = ind(3, 3);
    const double J0 = opaque(l4 == 0 ? 1.0 : 0.0, tie), J1 = opaque(l4 == 1 ? 1.0 : 0.0, tie), J2 = opaque(l4 == 2 ? 1.0 : 0.0, tie),
                 J3 = opaque(l4 == 3 ? 1.0 : 0.0, tie);
    const double Idg = opaque(i4 == l4 ? 1.0 : 0.0, tie), Ibelow = opaque(l4 < i4 ? 1.0 : 0.0, tie);
    auto rcp = [&](double x) { return RCP3 ? fast_rcp3(x) : fast_rcp(x); };
    auto rsq = [&](double x) { return RCP3 ? fast_rsqrt3(x) : fast_rsqrt(x); };
    // one 4-row group of inv(L44) A^T: lane (l15, l4) <- sum_k aop(l15 & 3 .., k) a(l15, k)
    auto solve44 = [&](double aop, double a, int tt) -> double {
        if (M44) return __builtin_amdgcn_mfma_f64_4x4x4f64(aop, a, 0.0, 0, 0, 0);
        const v4d out = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, a, zero4, 0, 0, 0);
        return out[tt];
    };
    // LOOK: the NEXT pivot block, updated ahead of the diagonal block it belongs to by one 4x4x4 product (the diagonal 4x4
    // blocks of X V^T are exactly what v_mfma_f64_4x4x4_4b computes from the update's own operands): the scalar chain of the
    // next group starts 45 cycles behind this group's X instead of 96 + behind the 16x16x4 update
    double pnext = 0.0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        double aopinv[4] = {0.0, 0.0, 0.0, 0.0};
        v4d Xf[4] = {zero4, zero4, zero4, zero4};
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int t = 4 * b + tt;
            const double dsrc = (LOOK && t > 0) ? pnext : Lt[b][b][tt];
            const double p00 = readlane_f64(dsrc, 4 * tt + 0);
            const double p10 = readlane_f64(dsrc, 4 * tt + 1), p11 = readlane_f64(dsrc, 4 * tt + 1 + 16);
            const double p20 = readlane_f64(dsrc, 4 * tt + 2), p21 = readlane_f64(dsrc, 4 * tt + 2 + 16),
                         p22 = readlane_f64(dsrc, 4 * tt + 2 + 32);
            const double p30 = readlane_f64(dsrc, 4 * tt + 3), p31 = readlane_f64(dsrc, 4 * tt + 3 + 16),
                         p32 = readlane_f64(dsrc, 4 * tt + 3 + 32), p33 = readlane_f64(dsrc, 4 * tt + 3 + 48);
            // the unguarded chain: operation for operation that of factor_piv4_vals (LDL^T: x_ik = c_ik s_k IS l_ik)
            double s0, s1, s2, s3, c10, c20, c30, c21, c31, c32, g0, g1, g2, g3;   // g: the recorded pivots (LDL^T: d, Cholesky: sqrt)
            double x10 = 0.0, x20 = 0.0, x30 = 0.0, x21 = 0.0, x31 = 0.0, x32 = 0.0;   // LDL^T: l_ik = c_ik s_k
            bool ok;
            if (LDL) {
                s0 = rcp(p00);
                x10 = p10 * s0; x20 = p20 * s0; x30 = p30 * s0;
                g1 = fma(-x10, p10, p11);
                c21 = fma(-x20, p10, p21); c31 = fma(-x30, p10, p31);
                s1 = rcp(g1);
                x21 = c21 * s1; x31 = c31 * s1;
                g2 = fma(-x21, c21, fma(-x20, p20, p22));
                c32 = fma(-x31, c21, fma(-x30, p20, p32));
                s2 = rcp(g2);
                x32 = c32 * s2;
                g3 = fma(-x32, c32, fma(-x31, c31, fma(-x30, p30, p33)));
                s3 = rcp(g3);
                g0 = p00; c10 = p10; c20 = p20; c30 = p30;
                // the four pivots' tests, together: min |d| > tol; |d0| + .. + |d3| finite (NaN / Inf iff one of them is -- or the sum
                // overflows: guarded path); early rejection: no sign bit set
                const double amin = min_abs(min_abs(g0, g1), min_abs(g2, g3));
                const double asum = (fabs(g0) + fabs(g1)) + (fabs(g2) + fabs(g3));
                const int signs = or3(__double2hiint(g0), __double2hiint(g1), __double2hiint(g2)) | __double2hiint(g3);
                ok = (amin > pivot_tol) & (asum <= DBL_MAX) & ((reject == 0) | (signs >= 0));
            } else {
                s0 = rsq(p00);
                c10 = p10 * s0; c20 = p20 * s0; c30 = p30 * s0;
                const double t1 = fma(-c10, c10, p11);
                s1 = rsq(t1);
                c21 = fma(-c20, c10, p21) * s1; c31 = fma(-c30, c10, p31) * s1;
                const double t2 = fma(-c21, c21, fma(-c20, c20, p22));
                s2 = rsq(t2);
                c32 = fma(-c31, c21, fma(-c30, c20, p32)) * s2;
                const double t3 = fma(-c32, c32, fma(-c31, c31, fma(-c30, c30, p33)));
                s3 = rsq(t3);
                g0 = p00 * s0; g1 = t1 * s1; g2 = t2 * s2; g3 = t3 * s3;
                // (every t > 0 and finite: no sign bit, min > 0, finite sum)
                const int signs = or3(__double2hiint(p00), __double2hiint(t1), __double2hiint(t2)) | __double2hiint(t3);
                ok = (min_abs(min_abs(p00, t1), min_abs(t2, t3)) > 0.0) & ((fabs(p00) + fabs(t1)) + (fabs(t2) + fabs(t3)) <= DBL_MAX) & (signs >= 0);
            }
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(!ok) != 0, 0)) {   // (uniform values: all lanes or none)
                // ---- a zero / non-finite / (Cholesky; early rejection) non-positive pivot in this group: the guarded scalar
                // factorization of potrf64w_core (harmless pivot 1, the zero recorded); the selections below are the same
                Piv4 P;
                double dg[4];
                int fail;
                factor_piv4_vals<LDL>(p00, p10, p11, p20, p21, p22, p30, p31, p32, p33, pivot_tol, P, dg, fail);
                if (!LDL && fail != 0 && lane == 0) atomicCAS(info, 0, (int)(j0 + 4 * t + fail));
                if (LDL && reject != 0) {
                    const bool bad = !(dg[0] > 0.0) | !(dg[1] > 0.0) | !(dg[2] > 0.0) | !(dg[3] > 0.0);
                    if (bad && lane == 0 && atomicCAS(info, 0, -9) == 0) info[1] = (int)(j0 + 63);
                }
                s0 = P.s0; s1 = P.s1; s2 = P.s2; s3 = P.s3;
                c10 = P.c10; c20 = P.c20; c30 = P.c30; c21 = P.c21; c31 = P.c31; c32 = P.c32;
                g0 = dg[0]; g1 = dg[1]; g2 = dg[2]; g3 = dg[3];
                if (LDL) { x10 = c10 * s0; x20 = c20 * s0; x30 = c30 * s0; x21 = c21 * s1; x31 = c31 * s1; x32 = c32 * s2; }
            }
            double aop, ssel, vpiv, lpiv;
            // V = L D (LDL^T) / L (Cholesky) on the pivot rows: one indicator sum
            vpiv = fma(c32, I32, fma(c31, I31, fma(c21, I21, fma(c30, I30, fma(c20, I20, fma(c10, I10,
                   fma(g3, D3, fma(g2, D2, fma(g1, D1, g0 * D0)))))))));
            if (LDL) {
                // inverse of the unit lower triangular 4x4 factor: y10 = -l10, y21 = -l21, y32 = -l32, y20 = l21 l10 - l20,
                // y31 = l32 l21 - l31, y30 = -(l32 y20 + (l30 - l31 l10)).  The entries of row 3 are linear in l32, the last value of
                // the chain: aop = l32 G + H with G, H complete before l32 is -- ONE fma between the chain and the block solve, and
                // in every lane the very fma that defines its entry (the same bits)
                const double y20 = fma(x21, x10, -x20), e30 = fma(x31, -x10, x30);
                const double G = fma(-y20, I30, fma(x21, I31, -I32));
                const double H = fma(-e30, I30, fma(-x31, I31, fma(-x21, I21, fma(y20, I20, fma(-x10, I10, Idg)))));
                aop = fma(x32, G, H);
                ssel = fma(s3, J3, fma(s2, J2, fma(s1, J1, s0 * J0)));
                lpiv = vpiv * fma(ssel, Ibelow, Idg);
            } else {
                // inverse of the 4x4 Cholesky factor (forward substitution as in potrf64w_core: the same products in the same order)
                const double y10 = -(c10 * s0) * s1;
                const double y20 = -fma(c21, y10, c20 * s0) * s2;
                const double y30 = -fma(c32, y20, fma(c31, y10, c30 * s0)) * s3;
                const double y21 = -(c21 * s1) * s2;
                const double y31 = -fma(c32, y21, c31 * s1) * s3;
                const double y32 = -(c32 * s2) * s3;
                aop = fma(y32, I32, fma(y31, I31, fma(y21, I21, fma(y30, I30, fma(y20, I20, fma(y10, I10,
                      fma(s3, D3, fma(s2, D2, fma(s1, D1, s0 * D0)))))))));
                ssel = 1.0;
                lpiv = vpiv;
            }
            aopinv[tt] = aop;
            const bool done_row = l15 < 4 * tt + 4;   // rows of the diagonal block up to and including the pivot group
            // ---- 2. X_t^T = inv(L44) A_t^T for every block of block column b
            double X[4], V[4];
#pragma unroll
            for (int cb = b; cb < 4; ++cb) {
                double v = solve44(aop, Lt[cb][b][tt], tt);
                double x = LDL ? v * ssel : v;
                if (cb == b) {
                    // the pivot rows: exact values from the scalar factorization (above them: the strict upper triangle, see the header)
                    x = done_row ? lpiv : x;
                    v = done_row ? (LDL ? vpiv : lpiv) : v;
                }
                X[cb] = x;
                V[cb] = v;
                if (LDL) vm = fmax(vm, fabs(v));
                Xf[cb][tt] = x;
            }
            if (LOOK) {
                if (tt < 3) pnext = __builtin_amdgcn_mfma_f64_4x4x4f64(-X[b], LDL ? V[b] : X[b], Lt[b][b][tt + 1], 0, 0, 0);
                else if (b < 3) pnext = __builtin_amdgcn_mfma_f64_4x4x4f64(-X[b + 1], LDL ? V[b + 1] : X[b + 1], Lt[b + 1][b + 1][0], 0, 0, 0);
            }
            // ---- 3. rank-4 update of the trailing blocks: acc(cb2, cb1) -= X[cb1] (V|X)[cb2]^T
#pragma unroll
            for (int cb1 = b; cb1 < 4; ++cb1) {
                const double na = (cb1 == b && done_row) ? 0.0 : -X[cb1];
#pragma unroll
                for (int cb2 = cb1; cb2 < 4; ++cb2)
                    Lt[cb2][cb1] = __builtin_amdgcn_mfma_f64_16x16x4f64(na, LDL ? V[cb2] : X[cb2], Lt[cb2][cb1], 0, 0, 0);
            }
        }
        // ---- inverse of the 16x16 diagonal block (unit diagonal for LDL): Y = inv(L16), block forward substitution
        {
            v4d T, Y;
#pragma unroll
            for (int r = 0; r < 4; ++r) T[r] = (l15 == l4 + 4 * r) ? 1.0 : 0.0;
#pragma unroll
            for (int pg = 0; pg < 4; ++pg) {
                Y[pg] = solve44(aopinv[pg], T[pg], pg);
                if (pg < 3) T = __builtin_amdgcn_mfma_f64_16x16x4f64(-Xf[b][pg], Y[pg], T, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                put<WT>(inv16 + b * 256 + (l4 + 4 * r) + 16 * l15, Y[r]);
                if (Ish != nullptr) Ish[b * 256 + (l4 + 4 * r) + 16 * l15] = Y[r];
            }
        }
        {
            const int rsel = l15 >> 2;
            const double dsel = rsel == 0 ? Xf[b][0] : (rsel == 1 ? Xf[b][1] : (rsel == 2 ? Xf[b][2] : Xf[b][3]));
            if ((l15 & 3) == l4) {
                put<WT>(dvec + j0 + 16 * b + l15, dsel);
                put<WT>(dinv + j0 + 16 * b + l15, LDL ? fast_rcp(dsel == 0.0 ? 1.0 : dsel) : 1.0);
            }
        }
#pragma unroll
        for (int cb = b; cb < 4; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double v = (cb == b && l15 < l4 + 4 * r) ? 0.0 : Xf[cb][r];
                put<WT>(Dout + (16 * cb + l15) + 64 * (16 * b + l4 + 4 * r), v);
                if (Lsh != nullptr) Lsh[(16 * cb + l15) + 64 * (16 * b + l4 + 4 * r)] = v;
            }
    }
    if (LDL) growth_fold(vmax, vm);
}

// ---------------------------------------------------------------------------------------
// potrf64w_core with the instruction stream SOFTWARE-PIPELINED by hand (round 5): the same operations on the same operands,
// so the same bits -- issued in a different order.  A wave issues in order, and an MFMA that finds the matrix pipe busy
// (16 passes = 64 cycles per v_mfma_f64_16x16x4) holds back everything behind it, so the one-wave leaf above runs its two
// halves one after the other (tools/hip/leaf_lab.hip: 29 300 cycles = 16 300 for the broadcasts, the scalar 4x4 factorizations
// and the operand selections -- ~230 VALU instructions per pivot group, issue-bound at ~4.4 cycles each -- plus 13 000 for the
// block solves and rank-4 updates).  Only ONE chain of a pivot group is critical: pivot block -> 4x4 factorization -> block
// solve of the diagonal block (b, b) -> its own rank-4 update -> next pivot block.  Here every group g issues that chain [A]
// and, interleaved with it one MFMA per ~VALU_PER_MFMA vector instructions (sched_group_barrier), the block solves and updates
// of the OTHER blocks that group g - 1 left behind [B]: the matrix pipe works under the vector chain instead of in front of it.
// ---------------------------------------------------------------------------------------
template <bool LDL, bool WT = false, int VALU_PER_MFMA = 14>
__device__ __forceinline__ void potrf64s_core(v4d (&Lt)[4][4], int64_t j0, double* __restrict__ Dout,
                                              double* __restrict__ inv16, double* __restrict__ dvec,
                                              double* __restrict__ dinv, int* __restrict__ info, double pivot_tol,
                                              unsigned long long* __restrict__ vmax = nullptr) {
    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, l4 = lane >> 4;
    const v4d zero4 = {0.0, 0.0, 0.0, 0.0};
    double vm = 0.0;
    v4d Xf[4][4];        // finished columns: Xf[cb][b] = block (cb, b) (only b <= cb used)
    double aopinv[4][4]; // [b][tt]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { Xf[i][j] = zero4; aopinv[i][j] = 0.0; }
    // what group g - 1 leaves for the next group's shadow: its operands for the other blocks
    double p_aop = 0.0, p_ssel = 0.0, p_Xb = 0.0, p_Vb = 0.0;
    auto groupB = [&](auto Bc, auto Tc) __attribute__((always_inline)) {   // the non-critical part of group (b, tt)
        constexpr int b = decltype(Bc)::value, tt = decltype(Tc)::value;
        double X[4], V[4];
        X[b] = p_Xb;
        V[b] = p_Vb;
#pragma unroll
        for (int cb = b + 1; cb < 4; ++cb) {
            const v4d out = __builtin_amdgcn_mfma_f64_16x16x4f64(p_aop, Lt[cb][b][tt], zero4, 0, 0, 0);
            const double v = out[tt];
            const double x = LDL ? v * p_ssel : v;
            X[cb] = x;
            V[cb] = v;
            if (LDL) vm = fmax(vm, fabs(v));
            Xf[cb][b][tt] = x;
        }
        // the updates in the order the NEXT critical chain needs them: the next diagonal block first (a block column's last group)
#pragma unroll
        for (int cb1 = b; cb1 < 4; ++cb1) {
            const double na = (cb1 == b && l15 < 4 * tt + 4) ? 0.0 : -X[cb1];
#pragma unroll
            for (int cb2 = cb1; cb2 < 4; ++cb2) {
                if (cb1 == b && cb2 == b) continue;   // (done by the critical part)
                Lt[cb2][cb1] = __builtin_amdgcn_mfma_f64_16x16x4f64(na, LDL ? V[cb2] : X[cb2], Lt[cb2][cb1], 0, 0, 0);
            }
        }
    };
    auto groupA = [&](auto Bc, auto Tc) __attribute__((always_inline)) {   // the critical chain of group (b, tt)
        constexpr int b = decltype(Bc)::value, tt = decltype(Tc)::value;
        constexpr int t = 4 * b + tt;
        const double dsrc = Lt[b][b][tt];
        const double p00 = readlane_f64(dsrc, 4 * tt + 0);
        const double p10 = readlane_f64(dsrc, 4 * tt + 1), p11 = readlane_f64(dsrc, 4 * tt + 1 + 16);
        const double p20 = readlane_f64(dsrc, 4 * tt + 2), p21 = readlane_f64(dsrc, 4 * tt + 2 + 16),
                     p22 = readlane_f64(dsrc, 4 * tt + 2 + 32);
        const double p30 = readlane_f64(dsrc, 4 * tt + 3), p31 = readlane_f64(dsrc, 4 * tt + 3 + 16),
                     p32 = readlane_f64(dsrc, 4 * tt + 3 + 32), p33 = readlane_f64(dsrc, 4 * tt + 3 + 48);
        Piv4 P;
        double dg[4];
        int fail;
        factor_piv4_vals<LDL>(p00, p10, p11, p20, p21, p22, p30, p31, p32, p33, pivot_tol, P, dg, fail);
        if (!LDL && fail != 0 && lane == 0) atomicCAS(info, 0, (int)(j0 + 4 * t + fail));
        const double l10 = LDL ? P.c10 * P.s0 : P.c10, l20 = LDL ? P.c20 * P.s0 : P.c20,
                     l30 = LDL ? P.c30 * P.s0 : P.c30, l21 = LDL ? P.c21 * P.s1 : P.c21,
                     l31 = LDL ? P.c31 * P.s1 : P.c31, l32 = LDL ? P.c32 * P.s2 : P.c32;
        const double rd0 = LDL ? 1.0 : P.s0, rd1 = LDL ? 1.0 : P.s1, rd2 = LDL ? 1.0 : P.s2, rd3 = LDL ? 1.0 : P.s3;
        const double y00 = rd0, y11 = rd1, y22 = rd2, y33 = rd3;
        const double y10 = -(l10 * y00) * rd1;
        const double y20 = -fma(l21, y10, l20 * y00) * rd2;
        const double y30 = -fma(l32, y20, fma(l31, y10, l30 * y00)) * rd3;
        const double y21 = -(l21 * y11) * rd2;
        const double y31 = -fma(l32, y21, l31 * y11) * rd3;
        const double y32 = -(l32 * y22) * rd3;
        const int ii = l15 - 4 * tt;
        const bool r1 = ii == 1, r2 = ii == 2, on_diag = ii == l4, below = (l4 < ii) & (ii < 4);
        auto sel44 = [&](double d0, double d1, double d2, double d3, double e10, double e20, double e30, double e21,
                         double e31, double e32, bool unit) {
            const double c0v = r1 ? e10 : (r2 ? e20 : e30), c1v = r2 ? e21 : e31;
            const double off = l4 == 0 ? c0v : (l4 == 1 ? c1v : e32);
            const double dia = unit ? 1.0 : (l4 == 0 ? d0 : (l4 == 1 ? d1 : (l4 == 2 ? d2 : d3)));
            const double lo = below ? off : 0.0;
            return on_diag ? dia : lo;
        };
        const double aop = sel44(y00, y11, y22, y33, y10, y20, y30, y21, y31, y32, LDL);
        aopinv[b][tt] = aop;
        const double ssel = l4 == 0 ? P.s0 : (l4 == 1 ? P.s1 : (l4 == 2 ? P.s2 : P.s3));
        const double vpiv = sel44(dg[0], dg[1], dg[2], dg[3], P.c10, P.c20, P.c30, P.c21, P.c31, P.c32, false);
        const double lpiv = LDL ? (l4 < ii ? vpiv * ssel : vpiv) : vpiv;
        const v4d out = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, Lt[b][b][tt], zero4, 0, 0, 0);
        double v = out[tt];
        double x = LDL ? v * ssel : v;
        x = ii < 4 ? lpiv : x;
        v = ii < 4 ? (LDL ? vpiv : lpiv) : v;
        if (LDL) vm = fmax(vm, fabs(v));
        Xf[b][b][tt] = x;
        const double na = (l15 < 4 * tt + 4) ? 0.0 : -x;
        Lt[b][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(na, LDL ? v : x, Lt[b][b], 0, 0, 0);
        p_aop = aop; p_ssel = ssel; p_Xb = x; p_Vb = v;
    };
    auto interleave = [&](int nmfma) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 12; ++i)
            if (i < nmfma) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);               // one MFMA of [B]
                __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER_MFMA, 0);   // a stretch of [A]'s vector instructions
            }
    };
    // group (0, 0): nothing behind it
    groupA(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    __builtin_amdgcn_sched_barrier(0);
    auto step = [&](auto Bc, auto Tc, auto Pb, auto Pt) __attribute__((always_inline)) {
        constexpr int pb = decltype(Pb)::value;
        groupB(Pb, Pt);      // the previous group's other blocks ...
        groupA(Bc, Tc);      // ... under this group's chain
        interleave((3 - pb) + (4 - pb) * (5 - pb) / 2 - 1);
        __builtin_amdgcn_sched_barrier(0);
    };
#define MNK_I(x) std::integral_constant<int, x>{}
    step(MNK_I(0), MNK_I(1), MNK_I(0), MNK_I(0)); step(MNK_I(0), MNK_I(2), MNK_I(0), MNK_I(1)); step(MNK_I(0), MNK_I(3), MNK_I(0), MNK_I(2));
    step(MNK_I(1), MNK_I(0), MNK_I(0), MNK_I(3)); step(MNK_I(1), MNK_I(1), MNK_I(1), MNK_I(0)); step(MNK_I(1), MNK_I(2), MNK_I(1), MNK_I(1));
    step(MNK_I(1), MNK_I(3), MNK_I(1), MNK_I(2)); step(MNK_I(2), MNK_I(0), MNK_I(1), MNK_I(3)); step(MNK_I(2), MNK_I(1), MNK_I(2), MNK_I(0));
    step(MNK_I(2), MNK_I(2), MNK_I(2), MNK_I(1)); step(MNK_I(2), MNK_I(3), MNK_I(2), MNK_I(2)); step(MNK_I(3), MNK_I(0), MNK_I(2), MNK_I(3));
    step(MNK_I(3), MNK_I(1), MNK_I(3), MNK_I(0)); step(MNK_I(3), MNK_I(2), MNK_I(3), MNK_I(1)); step(MNK_I(3), MNK_I(3), MNK_I(3), MNK_I(2));
#undef MNK_I
    // (group (3, 3) leaves nothing: block column 3 has one block)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        {   // inverse of the 16x16 diagonal block (unit diagonal for LDL): Y = inv(L16), block forward substitution
            v4d T, Y;
#pragma unroll
            for (int r = 0; r < 4; ++r) T[r] = (l15 == l4 + 4 * r) ? 1.0 : 0.0;
#pragma unroll
            for (int pg = 0; pg < 4; ++pg) {
                const v4d out = __builtin_amdgcn_mfma_f64_16x16x4f64(aopinv[b][pg], T[pg], zero4, 0, 0, 0);
                Y[pg] = out[pg];
                if (pg < 3) T = __builtin_amdgcn_mfma_f64_16x16x4f64(-Xf[b][b][pg], Y[pg], T, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) put<WT>(inv16 + b * 256 + (l4 + 4 * r) + 16 * l15, Y[r]);
        }
        {
            const int rsel = l15 >> 2;
            const double dsel = rsel == 0 ? Xf[b][b][0] : (rsel == 1 ? Xf[b][b][1] : (rsel == 2 ? Xf[b][b][2] : Xf[b][b][3]));
            if ((l15 & 3) == l4) {
                put<WT>(dvec + j0 + 16 * b + l15, dsel);
                put<WT>(dinv + j0 + 16 * b + l15, LDL ? fast_rcp(dsel == 0.0 ? 1.0 : dsel) : 1.0);
            }
        }
#pragma unroll
        for (int cb = b; cb < 4; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double v = (cb == b && l15 < l4 + 4 * r) ? 0.0 : Xf[cb][b][r];
                put<WT>(Dout + (16 * cb + l15) + 64 * (16 * b + l4 + 4 * r), v);
            }
    }
    if (LDL) growth_fold(vmax, vm);
}

// ---------------------------------------------------------------------------------------
// The same factorization of a 64x64 diagonal block by the FOUR waves of a workgroup (the pivot chain's strips, pp_strip):
// wave w owns block row w -- the 16x16 blocks (w, b), b <= w, in the layout above -- which is where pp_strip's waves hold
// the diagonal block anyway (round 2-4 copied the block through LDS to wave 0 and the other three waves left).  The
// arithmetic of every entry is that of potrf64w_core, operation for operation: the same bits.
//
// Why: the one-wave leaf is bound by its instruction stream (~4100 instructions, 13 us), of which the 120 MFMAs of the
// rank-4 updates and block solves and their operand selections can be split by block row, while the 4x4 pivot
// factorization -- a dependent chain of reciprocals / reciprocal square roots -- cannot: every wave repeats it (same
// inputs, same instructions: same values), which costs nothing since the chain is the critical path either way.
// Per pivot group (16 of them) ONE workgroup barrier:
//   before it   every active wave (w >= b) has factored the pivot block (values read from LDS), solved its own block
//               X_w = A_w inv(L44)^T, applied the update that needs only its OWN X -- its diagonal block (w, w) -- and
//               written X_w (V_w for LDL^T) to LDS; the owner of the NEXT pivot block (wave b inside a block column, wave
//               b + 1 at its end) has written that block to LDS as well;
//   behind it   the updates of the blocks (w, cb1), cb1 < w, with the X of the other waves.
// LDS buffers alternate between consecutive groups (a fast wave writes group g + 1 while a slow one still reads group g).
// At the end every wave inverts its own 16x16 diagonal block and stores its block row.
// `ex`: 2 x PQ_EX doubles of LDS.  Every thread of the workgroup must call it.
// ---------------------------------------------------------------------------------------
constexpr int PQ_EX = 64 + 4 * 64 + 4 * 64;   // pivot block | X of the four waves | V of the four waves (doubles per buffer)
template <bool LDL, bool WT = false>
__device__ __forceinline__ void potrf64q_core(v4d (&Lt)[4], const int w, int64_t j0, double* __restrict__ Dout,
                                              double* __restrict__ inv16, double* __restrict__ dvec,
                                              double* __restrict__ dinv, int* __restrict__ info, double pivot_tol,
                                              double* ex, unsigned long long* __restrict__ vmax = nullptr) {
    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, l4 = lane >> 4;
    const v4d zero4 = {0.0, 0.0, 0.0, 0.0};
    double vm = 0.0;
    v4d Xf[4] = {zero4, zero4, zero4, zero4};   // finished columns of the blocks (w, b)
    double aopinv[4] = {0.0, 0.0, 0.0, 0.0};    // (block column w: this wave's diagonal block)
    if (w == 0) ex[lane] = Lt[0][0];
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 4; ++b) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int g = 4 * b + tt;
            double* exr = ex + (g & 1) * PQ_EX;          // this group's pivot block / the previous group's X
            double* exw = ex + ((g + 1) & 1) * PQ_EX;    // this group's X / the next group's pivot block
            double X = 0.0, V = 0.0;
            if (w >= b) {   // (wave-uniform)
                // ---- 1. pivot block: A[16b + 4tt + jj][16b + 4tt + kk] is entry (4tt + jj) + 16 kk of the owner's register
                const double p00 = exr[4 * tt + 0];
                const double p10 = exr[4 * tt + 1], p11 = exr[4 * tt + 1 + 16];
                const double p20 = exr[4 * tt + 2], p21 = exr[4 * tt + 2 + 16], p22 = exr[4 * tt + 2 + 32];
                const double p30 = exr[4 * tt + 3], p31 = exr[4 * tt + 3 + 16], p32 = exr[4 * tt + 3 + 32],
                             p33 = exr[4 * tt + 3 + 48];
                Piv4 P;
                double dg[4];
                int fail;
                factor_piv4_vals<LDL>(p00, p10, p11, p20, p21, p22, p30, p31, p32, p33, pivot_tol, P, dg, fail);
                if (!LDL && fail != 0 && lane == 0 && w == b) atomicCAS(info, 0, (int)(j0 + 4 * g + fail));
                const double l10 = LDL ? P.c10 * P.s0 : P.c10, l20 = LDL ? P.c20 * P.s0 : P.c20,
                             l30 = LDL ? P.c30 * P.s0 : P.c30, l21 = LDL ? P.c21 * P.s1 : P.c21,
                             l31 = LDL ? P.c31 * P.s1 : P.c31, l32 = LDL ? P.c32 * P.s2 : P.c32;
                const double rd0 = LDL ? 1.0 : P.s0, rd1 = LDL ? 1.0 : P.s1, rd2 = LDL ? 1.0 : P.s2, rd3 = LDL ? 1.0 : P.s3;
                const double y00 = rd0, y11 = rd1, y22 = rd2, y33 = rd3;
                const double y10 = -(l10 * y00) * rd1;
                const double y20 = -fma(l21, y10, l20 * y00) * rd2;
                const double y30 = -fma(l32, y20, fma(l31, y10, l30 * y00)) * rd3;
                const double y21 = -(l21 * y11) * rd2;
                const double y31 = -fma(l32, y21, l31 * y11) * rd3;
                const double y32 = -(l32 * y22) * rd3;
                const int ii = l15 - 4 * tt;
                const bool r1 = ii == 1, r2 = ii == 2, on_diag = ii == l4, below = (l4 < ii) & (ii < 4);
                auto sel44 = [&](double d0, double d1, double d2, double d3, double e10, double e20, double e30, double e21,
                                 double e31, double e32, bool unit) {
                    const double c0v = r1 ? e10 : (r2 ? e20 : e30), c1v = r2 ? e21 : e31;
                    const double off = l4 == 0 ? c0v : (l4 == 1 ? c1v : e32);
                    const double dia = unit ? 1.0 : (l4 == 0 ? d0 : (l4 == 1 ? d1 : (l4 == 2 ? d2 : d3)));
                    const double lo = below ? off : 0.0;
                    return on_diag ? dia : lo;
                };
                const double aop = sel44(y00, y11, y22, y33, y10, y20, y30, y21, y31, y32, LDL);
                const double ssel = l4 == 0 ? P.s0 : (l4 == 1 ? P.s1 : (l4 == 2 ? P.s2 : P.s3));
                // ---- 2. X^T = inv(L44) A^T for this wave's block (w, b)
                const v4d out = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, Lt[b][tt], zero4, 0, 0, 0);
                double v = out[tt];
                double x = LDL ? v * ssel : v;
                if (w == b) {   // the diagonal block: exact entries of the pivot rows, zeros above them
                    aopinv[tt] = aop;
                    const double vpiv = sel44(dg[0], dg[1], dg[2], dg[3], P.c10, P.c20, P.c30, P.c21, P.c31, P.c32, false);
                    const double lpiv = LDL ? (l4 < ii ? vpiv * ssel : vpiv) : vpiv;
                    x = ii < 4 ? lpiv : x;
                    v = ii < 4 ? (LDL ? vpiv : lpiv) : v;
                }
                X = x;
                V = v;
                if (LDL) vm = fmax(vm, fabs(v));
                Xf[b][tt] = x;
                // ---- 3a. the update of this wave's own diagonal-side block that needs nothing from the others: (w, w)
                {
                    const double na = (w == b && l15 < 4 * tt + 4) ? 0.0 : -X;
                    Lt[w] = __builtin_amdgcn_mfma_f64_16x16x4f64(na, LDL ? V : X, Lt[w], 0, 0, 0);
                }
                exw[64 + 64 * w + lane] = X;
                if (LDL) exw[64 + 256 + 64 * w + lane] = V;
                // the next pivot block: component tt + 1 of the owner's diagonal block, or component 0 of the next owner's
                if (tt < 3 ? w == b : w == b + 1) exw[lane] = Lt[w][tt < 3 ? tt + 1 : 0];
            }
            __syncthreads();
            if (w > b) {
                // ---- 3b. acc(w, cb1) -= X[cb1] (V|X)[w]^T for the block rows above this one
#pragma unroll
                for (int cb1 = b; cb1 < 4; ++cb1) {
                    if (cb1 >= w) break;
                    const double xo = exw[64 + 64 * cb1 + lane];
                    const double na = (cb1 == b && l15 < 4 * tt + 4) ? 0.0 : -xo;
                    Lt[cb1] = __builtin_amdgcn_mfma_f64_16x16x4f64(na, LDL ? V : X, Lt[cb1], 0, 0, 0);
                }
            }
        }
    }
    // ---- inverse of this wave's 16x16 diagonal block (unit diagonal for LDL): block forward substitution (as in potrf64w_core)
    {
        v4d T, Y;
#pragma unroll
        for (int r = 0; r < 4; ++r) T[r] = (l15 == l4 + 4 * r) ? 1.0 : 0.0;
#pragma unroll
        for (int pg = 0; pg < 4; ++pg) {
            const v4d out = __builtin_amdgcn_mfma_f64_16x16x4f64(aopinv[pg], T[pg], zero4, 0, 0, 0);
            Y[pg] = out[pg];
            if (pg < 3) T = __builtin_amdgcn_mfma_f64_16x16x4f64(-Xf[w][pg], Y[pg], T, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) put<WT>(inv16 + w * 256 + (l4 + 4 * r) + 16 * l15, Y[r]);
    }
    {   // D and D^-1 of the 16 pivots of block column w
        const int rsel = l15 >> 2;
        const double dsel = rsel == 0 ? Xf[w][0] : (rsel == 1 ? Xf[w][1] : (rsel == 2 ? Xf[w][2] : Xf[w][3]));
        if ((l15 & 3) == l4) {
            put<WT>(dvec + j0 + 16 * w + l15, dsel);
            put<WT>(dinv + j0 + 16 * w + l15, LDL ? fast_rcp(dsel == 0.0 ? 1.0 : dsel) : 1.0);
        }
    }
    // ---- block row w of the factored block (column-major 64x64, lower part)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        if (b > w) break;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double v = (w == b && l15 < l4 + 4 * r) ? 0.0 : Xf[b][r];
            put<WT>(Dout + (16 * w + l15) + 64 * (16 * b + l4 + 4 * r), v);
        }
    }
    if (LDL) growth_fold(vmax, vm);
}

}  // namespace mnk
