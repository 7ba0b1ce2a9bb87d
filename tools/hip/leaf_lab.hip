// Leaf laboratory: the 64x64 diagonal-block factorization of the pivot chain (csrc/leaf64.h) on its own -- one workgroup, the
// block in registers, repeated; shader-clock cycles per factorization for the one-wave leaf (potrf64w_core) and the four-wave
// leaf (potrf64q_core), LDL^T and Cholesky, and for timing-only variants with pieces left out (results void) that say where
// the cycles go.  build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I madnlp.jl_amd/csrc tools/hip/leaf_lab.hip -o tools/hip/leaf_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "leaf64.h"
namespace mnk { void set_error(const char*, ...) {} }
using namespace mnk;

// VAR: 0 = the shipped leaf; timing-only variants of the ONE-wave leaf: 1 = no rank-4 updates (step 3), 2 = no block solves and
// no updates (steps 2-3), 3 = steps 2-3 with a constant pivot factorization (no scalar chain, no broadcasts)
template <bool LDL, int VAR>
__device__ __forceinline__ void leaf_variant(v4d (&Lt)[4][4], double* out, double pivot_tol) {
    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, l4 = lane >> 4;
    const v4d zero4 = {0.0, 0.0, 0.0, 0.0};
    double sink = 0.0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        v4d Xf[4] = {zero4, zero4, zero4, zero4};
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            Piv4 P;
            double dg[4];
            int fail = 0;
            if (VAR != 3) {
                const double dsrc = Lt[b][b][tt];
                const double p00 = readlane_f64(dsrc, 4 * tt + 0);
                const double p10 = readlane_f64(dsrc, 4 * tt + 1), p11 = readlane_f64(dsrc, 4 * tt + 1 + 16);
                const double p20 = readlane_f64(dsrc, 4 * tt + 2), p21 = readlane_f64(dsrc, 4 * tt + 2 + 16), p22 = readlane_f64(dsrc, 4 * tt + 2 + 32);
                const double p30 = readlane_f64(dsrc, 4 * tt + 3), p31 = readlane_f64(dsrc, 4 * tt + 3 + 16), p32 = readlane_f64(dsrc, 4 * tt + 3 + 32),
                             p33 = readlane_f64(dsrc, 4 * tt + 3 + 48);
                factor_piv4_vals<LDL>(p00, p10, p11, p20, p21, p22, p30, p31, p32, p33, pivot_tol, P, dg, fail);
            } else {
                P = Piv4{0.1, 0.2, 0.3, 0.15, 0.25, 0.35, 0.5, 0.6, 0.7, 0.8};
                dg[0] = 2.0; dg[1] = 1.7; dg[2] = 1.4; dg[3] = 1.2;
            }
            const double l10 = LDL ? P.c10 * P.s0 : P.c10, l20 = LDL ? P.c20 * P.s0 : P.c20, l30 = LDL ? P.c30 * P.s0 : P.c30,
                         l21 = LDL ? P.c21 * P.s1 : P.c21, l31 = LDL ? P.c31 * P.s1 : P.c31, l32 = LDL ? P.c32 * P.s2 : P.c32;
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
            auto sel44 = [&](double d0, double d1, double d2, double d3, double e10, double e20, double e30, double e21, double e31, double e32, bool unit) {
                const double c0v = r1 ? e10 : (r2 ? e20 : e30), c1v = r2 ? e21 : e31;
                const double off = l4 == 0 ? c0v : (l4 == 1 ? c1v : e32);
                const double dia = unit ? 1.0 : (l4 == 0 ? d0 : (l4 == 1 ? d1 : (l4 == 2 ? d2 : d3)));
                const double lo = below ? off : 0.0;
                return on_diag ? dia : lo;
            };
            const double aop = sel44(y00, y11, y22, y33, y10, y20, y30, y21, y31, y32, LDL);
            const double ssel = l4 == 0 ? P.s0 : (l4 == 1 ? P.s1 : (l4 == 2 ? P.s2 : P.s3));
            const double vpiv = sel44(dg[0], dg[1], dg[2], dg[3], P.c10, P.c20, P.c30, P.c21, P.c31, P.c32, false);
            const double lpiv = LDL ? (l4 < ii ? vpiv * ssel : vpiv) : vpiv;
            sink += aop + lpiv + (double)fail;
            if (VAR == 2) {   // the diagonal block still has to change, or the next pivots are the same registers
                Lt[b][b][(tt + 1) & 3] += aop;
                continue;
            }
            double X[4], V[4];
#pragma unroll
            for (int cb = b; cb < 4; ++cb) {
                const v4d o = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, Lt[cb][b][tt], zero4, 0, 0, 0);
                double v = o[tt];
                double x = LDL ? v * ssel : v;
                if (cb == b) { x = ii < 4 ? lpiv : x; v = ii < 4 ? (LDL ? vpiv : lpiv) : v; }
                X[cb] = x; V[cb] = v;
                Xf[cb][tt] = x;
            }
            if (VAR == 1) { Lt[b][b][(tt + 1) & 3] += X[b]; continue; }
#pragma unroll
            for (int cb1 = b; cb1 < 4; ++cb1) {
                const double na = (cb1 == b && l15 < 4 * tt + 4) ? 0.0 : -X[cb1];
#pragma unroll
                for (int cb2 = cb1; cb2 < 4; ++cb2)
                    Lt[cb2][cb1] = __builtin_amdgcn_mfma_f64_16x16x4f64(na, LDL ? V[cb2] : X[cb2], Lt[cb2][cb1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int cb = b; cb < 4; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(16 * cb + l15) + 64 * (16 * b + l4 + 4 * r)] = Xf[cb][r];
    }
    if (sink == 1.2345e300) out[lane] = sink;
}

// MODE 0: one-wave shipped leaf; 1: four-wave shipped leaf; 10 + VAR: one-wave timing variant
template <bool LDL, int MODE>
__global__ __launch_bounds__(256) void lab(const double* A, double* scratch, unsigned long long* cyc, int reps, int* info) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    unsigned long long best = ~0ull, sum = 0;
    for (int rep = 0; rep < reps; ++rep) {
        double* Dout = scratch;
        double* inv16 = scratch + 4096;
        double* dvec = scratch + 4096 + 1024;
        double* dinv = dvec + 64;
        __syncthreads();
        const unsigned long long t0 = __builtin_readcyclecounter();
        if (MODE == 1) {
            v4d Lq[4];
            for (int b = 0; b < 4; ++b)
                for (int r = 0; r < 4; ++r) {
                    const double v = b <= w ? A[(16 * w + l15) + 64 * (16 * b + l4 + 4 * r)] : 0.0;
                    Lq[b][r] = (b == w && l15 < l4 + 4 * r) ? 0.0 : v;
                }
            potrf64q_core<LDL, true>(Lq, w, 0, Dout, inv16, dvec, dinv, info, 1e-300, reinterpret_cast<double*>(smem), nullptr);
        } else if (w == 0) {
            v4d Lt[4][4];
            for (int cb = 0; cb < 4; ++cb)
                for (int b = 0; b <= cb; ++b)
                    for (int r = 0; r < 4; ++r) {
                        const double v = A[(16 * cb + l15) + 64 * (16 * b + l4 + 4 * r)];
                        Lt[cb][b][r] = (cb == b && l15 < l4 + 4 * r) ? 0.0 : v;
                    }
            if (MODE == 0) potrf64w_core<LDL, true>(Lt, 0, Dout, inv16, dvec, dinv, info, 1e-300, nullptr, nullptr, nullptr);
            else if (MODE == 2) potrf64s_core<LDL, true, 14>(Lt, 0, Dout, inv16, dvec, dinv, info, 1e-300, nullptr);
            else if (MODE == 3) potrf64s_core<LDL, true, 10>(Lt, 0, Dout, inv16, dvec, dinv, info, 1e-300, nullptr);
            else if (MODE == 4) potrf64s_core<LDL, true, 18>(Lt, 0, Dout, inv16, dvec, dinv, info, 1e-300, nullptr);
            else if (MODE == 5) potrf64s_core<LDL, true, 24>(Lt, 0, Dout, inv16, dvec, dinv, info, 1e-300, nullptr);
            else if (MODE == 6) potrf64v_core<LDL, true, false, false>(Lt, 0, Dout, inv16, dvec, dinv, info, 1e-300, nullptr, nullptr, nullptr);
            else if (MODE == 7) potrf64v_core<LDL, true, true, false>(Lt, 0, Dout, inv16, dvec, dinv, info, 1e-300, nullptr, nullptr, nullptr);
            else if (MODE == 20) potrf64v_core<LDL, true, true, true, false>(Lt, 0, Dout, inv16, dvec, dinv, info, 1e-300, nullptr, nullptr, nullptr);
            else if (MODE == 8) potrf64v_core<LDL, true, true, true>(Lt, 0, Dout, inv16, dvec, dinv, info, 1e-300, nullptr, nullptr, nullptr);
            else if (MODE == 9) potrf64v_core<LDL, true, false, true>(Lt, 0, Dout, inv16, dvec, dinv, info, 1e-300, nullptr, nullptr, nullptr);
            else leaf_variant<LDL, MODE - 10>(Lt, Dout, 1e-300);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const unsigned long long t1 = __builtin_readcyclecounter();
        if (rep > 0) { best = t1 - t0 < best ? t1 - t0 : best; sum += t1 - t0; }
    }
    if (threadIdx.x == 0) { cyc[0] = best; cyc[1] = sum / (reps - 1); }
}

// Do fp64 MFMAs and fp64 vector instructions overlap?  One wave per SIMD (256 threads) or two (512): KIND 0 = MFMAs only, 1 = FMAs
// only, 2 = both in ONE wave (interleaved in program order: 1 MFMA, 12 independent FMAs), 3 = waves 0-3 MFMAs, waves 4-7 FMAs.
template <int KIND>
__global__ __launch_bounds__(512) void overlap(double* out, unsigned long long* cyc, int iters) {
    const int w = threadIdx.x >> 6;
    v4d acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    double f[12];
    for (int i = 0; i < 12; ++i) f[i] = 1e-3 * (threadIdx.x + i);
    const double a = 1.0 + 1e-9 * threadIdx.x, b = 1e-7;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    const bool do_m = KIND == 0 || KIND == 2 || (KIND == 3 && w < 4), do_f = KIND == 1 || KIND == 2 || (KIND == 3 && w >= 4);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (do_m) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
            if (do_f) {
#pragma unroll
                for (int i = 0; i < 12; ++i) f[i] = fma(f[i], a, b);
            }
        }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    double sm = 0;
    for (int u = 0; u < 4; ++u) sm += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
    for (int i = 0; i < 12; ++i) sm += f[i];
    if (sm == 1.2345e300) out[threadIdx.x] = sm;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int KIND>
static void run_overlap(const char* name, int threads, double* scratch, unsigned long long* cyc) {
    const int iters = 256;
    hipLaunchKernelGGL((overlap<KIND>), dim3(1), dim3(threads), 0, 0, scratch, cyc, iters);
    hipLaunchKernelGGL((overlap<KIND>), dim3(1), dim3(threads), 0, 0, scratch, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h = 0;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-72s: %8llu cycles = %.1f per {1 MFMA + 12 FMA} slot\n", name, h, (double)h / (iters * 4.0));
}


// Lane layout of v_mfma_f64_4x4x4_4b: A = indicator of lane la, B = indicator of lane lb -> which lane of D is 1
__global__ void probe44(int* map) {
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(lane == la ? 1.0 : 0.0, lane == lb ? 1.0 : 0.0, 0.0, 0, 0, 0);
            if (d != 0.0) map[la * 64 + lb] = lane;
        }
}
static void run_probe44() {
    int* map; hipMalloc(&map, 4096 * 4); hipMemset(map, 0xff, 4096 * 4);
    hipLaunchKernelGGL(probe44, dim3(1), dim3(64), 0, 0, map);
    std::vector<int> h(4096);
    hipMemcpy(h.data(), map, 4096 * 4, hipMemcpyDeviceToHost);
    // the layout potrf64v_core assumes: A lane (i + 4 blk) + 16 k, B lane (j + 4 blk) + 16 k, D lane (j + 4 blk) + 16 i
    int bad = 0, shown = 0;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const int ia = la & 3, ba = (la >> 2) & 3, ka = la >> 4, jb = lb & 3, bb = (lb >> 2) & 3, kb = lb >> 4;
            const int want = (ba == bb && ka == kb) ? (jb + 4 * bb) + 16 * ia : -1;
            if (h[la * 64 + lb] != want) {
                ++bad;
                if (shown++ < 24) printf("  probe44: A lane %2d x B lane %2d -> D lane %2d (assumed %2d)\n", la, lb, h[la * 64 + lb], want);
            }
        }
    printf("v_mfma_f64_4x4x4_4b lane layout vs the assumed one (A (i+4blk)+16k, B (j+4blk)+16k, D (j+4blk)+16i): %d mismatches\n", bad);
    hipFree(map);
}

template <bool LDL, int MODE>
static void run(const char* name, const double* A, double* scratch, unsigned long long* cyc, int* info) {
    hipFuncSetAttribute((const void*)lab<LDL, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 16384);
    hipLaunchKernelGGL((lab<LDL, MODE>), dim3(1), dim3(256), 16384, 0, A, scratch, cyc, 50, info);
    hipDeviceSynchronize();
    unsigned long long h[2];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-64s %s: best %6llu cycles (%.2f us at 2.4 GHz), mean %6llu  %s\n", name, LDL ? "LDL' " : "Chol.", h[0], h[0] / 2400.0, h[1],
           hipGetErrorString(hipGetLastError()));
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    std::vector<double> h(4096);
    srand(5);
    for (int j = 0; j < 64; ++j)
        for (int i = 0; i < 64; ++i) h[i + 64 * j] = (i == j ? 80.0 : 0.0) + ((rand() % 2001) - 1000) * 1e-3;
    for (int j = 0; j < 64; ++j)
        for (int i = 0; i < j; ++i) h[i + 64 * j] = h[j + 64 * i];
    double *A, *scratch; unsigned long long* cyc; int* info;
    hipMalloc(&A, 4096 * 8); hipMalloc(&scratch, 8192 * 8); hipMalloc(&cyc, 64); hipMalloc(&info, 64);
    hipMemcpy(A, h.data(), 4096 * 8, hipMemcpyHostToDevice);
    hipMemset(info, 0, 64);
    run_overlap<0>("fp64 MFMA 16x16x4 only, one wave per SIMD", 256, scratch, cyc);
    run_overlap<1>("12 independent fp64 FMAs only, one wave per SIMD", 256, scratch, cyc);
    run_overlap<2>("1 MFMA + 12 FMAs interleaved in ONE wave per SIMD", 256, scratch, cyc);
    run_overlap<3>("two waves per SIMD: one issues the MFMAs, the other the FMAs", 512, scratch, cyc);
    run_overlap<0>("fp64 MFMA only, two waves per SIMD", 512, scratch, cyc);
    run_overlap<1>("FMAs only, two waves per SIMD", 512, scratch, cyc);
    // the two shipped leaves produce the same block
    std::vector<double> r0(4096 + 1024 + 128), r1(4096 + 1024 + 128);
    run<true, 0>("one-wave leaf (potrf64w_core)", A, scratch, cyc, info);
    hipMemcpy(r0.data(), scratch, r0.size() * 8, hipMemcpyDeviceToHost);
    hipMemset(scratch, 0, 8192 * 8);
    run<true, 1>("four-wave leaf (potrf64q_core)", A, scratch, cyc, info);
    hipMemcpy(r1.data(), scratch, r1.size() * 8, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (int j = 0; j < 64; ++j)
        for (int i = j; i < 64; ++i) bad += r0[i + 64 * j] != r1[i + 64 * j];
    for (size_t i = 4096; i < r0.size(); ++i) bad += r0[i] != r1[i];
    printf("entries of (L, inverses, D, 1/D) that differ between the two leaves: %zu\n", bad);
    for (int m = 2; m <= 5; ++m) {
        hipMemset(scratch, 0, 8192 * 8);
        if (m == 2) run<true, 2>("one-wave leaf, software-pipelined (14 VALU per MFMA)", A, scratch, cyc, info);
        if (m == 3) run<true, 3>("one-wave leaf, software-pipelined (10 VALU per MFMA)", A, scratch, cyc, info);
        if (m == 4) run<true, 4>("one-wave leaf, software-pipelined (18 VALU per MFMA)", A, scratch, cyc, info);
        if (m == 5) run<true, 5>("one-wave leaf, software-pipelined (24 VALU per MFMA)", A, scratch, cyc, info);
        hipMemcpy(r1.data(), scratch, r1.size() * 8, hipMemcpyDeviceToHost);
        bad = 0;
        for (int j = 0; j < 64; ++j)
            for (int i = j; i < 64; ++i) bad += r0[i + 64 * j] != r1[i + 64 * j];
        for (size_t i = 4096; i < r0.size(); ++i) bad += r0[i] != r1[i];
        printf("entries that differ from the one-wave leaf: %zu\n", bad);
    }
    run<false, 0>("one-wave leaf (potrf64w_core)", A, scratch, cyc, info);
    run<false, 2>("one-wave leaf, software-pipelined (14 VALU per MFMA)", A, scratch, cyc, info);
    run<false, 1>("four-wave leaf (potrf64q_core)", A, scratch, cyc, info);
    run<true, 10>("one-wave copy, everything (no inverses / D stores)", A, scratch, cyc, info);
    run<true, 11>("one-wave copy without the rank-4 updates", A, scratch, cyc, info);
    run<true, 12>("one-wave copy: broadcasts + scalar chain + selections only", A, scratch, cyc, info);
    run<true, 13>("one-wave copy: block solves + updates, constant pivots", A, scratch, cyc, info);
    run<false, 10>("one-wave copy, everything (no inverses / D stores)", A, scratch, cyc, info);
    run<false, 13>("one-wave copy: block solves + updates, constant pivots", A, scratch, cyc, info);
    // ---- round 6: potrf64v_core (indicator sums instead of selections, pivot tests behind the chain, 4x4x4 block solves, third-order reciprocal)
    run_probe44();
    auto fetch = [&](std::vector<double>& r) { hipMemcpy(r.data(), scratch, r.size() * 8, hipMemcpyDeviceToHost); hipMemset(scratch, 0, 8192 * 8); };
    auto cmp = [&](const char* what, const std::vector<double>& a, const std::vector<double>& b2) {
        size_t nd = 0; double mx = 0.0;
        auto one = [&](double x, double y) { if (x != y && !(x != x && y != y)) { ++nd; const double e = fabs(x - y) / fmax(fabs(x), 1e-300); mx = e > mx ? e : mx; } };
        for (int j = 0; j < 64; ++j) for (int i = j; i < 64; ++i) one(a[i + 64 * j], b2[i + 64 * j]);
        for (size_t i = 4096; i < a.size(); ++i) one(a[i], b2[i]);
        printf("    %-52s: %zu entries differ, max relative difference %.2e\n", what, nd, mx);
    };
    std::vector<double> rw(4096 + 1024 + 128), rv(4096 + 1024 + 128);
    hipMemset(scratch, 0, 8192 * 8);
    run<true, 0>("one-wave leaf (potrf64w_core)", A, scratch, cyc, info); fetch(rw);
    run<true, 6>("potrf64v_core (16x16x4 solves, two Newton steps)", A, scratch, cyc, info); fetch(rv); cmp("vs potrf64w_core", rw, rv);
    run<true, 7>("potrf64v_core (4x4x4 solves, two Newton steps)", A, scratch, cyc, info); fetch(rv); cmp("vs potrf64w_core", rw, rv);
    run<true, 9>("potrf64v_core (16x16x4 solves, third-order reciprocal)", A, scratch, cyc, info); fetch(rv); cmp("vs potrf64w_core", rw, rv);
    run<true, 20>("potrf64v_core (4x4x4, third-order, no look-ahead)", A, scratch, cyc, info); fetch(rv); cmp("vs potrf64w_core", rw, rv);
    run<true, 8>("potrf64v_core (4x4x4 solves, third-order reciprocal)", A, scratch, cyc, info); fetch(rv); cmp("vs potrf64w_core", rw, rv);
    {   // backward error of both against the matrix: max |A - L D L'| / max |A|
        auto berr = [&](const std::vector<double>& r) {
            double mx = 0.0;
            for (int i = 0; i < 64; ++i)
                for (int j = 0; j <= i; ++j) {
                    long double sacc = 0.0L;
                    for (int k = 0; k <= j; ++k) {
                        const long double lik = i == k ? 1.0L : (long double)r[i + 64 * k], ljk = j == k ? 1.0L : (long double)r[j + 64 * k];
                        sacc += lik * (long double)r[4096 + 1024 + k] * ljk;
                    }
                    const double e = fabs((double)(sacc - (long double)h[i + 64 * j]));
                    mx = e > mx ? e : mx;
                }
            return mx / 81.0;
        };
        printf("    backward error max|A - L D L'| / max|A|: potrf64w_core %.3e, potrf64v_core (4x4x4, third-order) %.3e\n", berr(rw), berr(rv));
    }
    run<false, 0>("one-wave leaf (potrf64w_core)", A, scratch, cyc, info); fetch(rw);
    run<false, 6>("potrf64v_core (16x16x4 solves, two Newton steps)", A, scratch, cyc, info); fetch(rv); cmp("vs potrf64w_core", rw, rv);
    run<false, 8>("potrf64v_core (4x4x4 solves, third-order rsqrt)", A, scratch, cyc, info); fetch(rv); cmp("vs potrf64w_core", rw, rv);
    {   // the guarded path: a matrix with an exactly zero pivot (column 21 decoupled, zero diagonal), a negative pivot (column 40) and, for
        // Cholesky, the breakdown it causes; early rejection armed in a second pass (info[2] = 1)
        std::vector<double> h2 = h;
        for (int i = 0; i < 64; ++i) { h2[i + 64 * 21] = 0.0; h2[21 + 64 * i] = 0.0; }
        h2[40 + 64 * 40] = -3.0;
        double* A2; hipMalloc(&A2, 4096 * 8); hipMemcpy(A2, h2.data(), 4096 * 8, hipMemcpyHostToDevice);
        int hi[4];
        for (int pass = 0; pass < 2; ++pass) {
            const int arm[4] = {0, 0, pass, 0};
            hipMemcpy(info, arm, 16, hipMemcpyHostToDevice);
            run<true, 0>(pass ? "zero + negative pivot, early rejection armed: w" : "zero + negative pivot: potrf64w_core", A2, scratch, cyc, info); fetch(rw);
            hipMemcpy(hi, info, 16, hipMemcpyDeviceToHost); printf("    info = %d %d\n", hi[0], hi[1]);
            hipMemcpy(info, arm, 16, hipMemcpyHostToDevice);
            run<true, 8>(pass ? "zero + negative pivot, early rejection armed: v" : "zero + negative pivot: potrf64v_core", A2, scratch, cyc, info); fetch(rv);
            hipMemcpy(hi, info, 16, hipMemcpyDeviceToHost); printf("    info = %d %d\n", hi[0], hi[1]);
            cmp("vs potrf64w_core", rw, rv);
            hipMemcpy(info, arm, 16, hipMemcpyHostToDevice);
            run<true, 6>("  ... potrf64v_core with the round-5 arithmetic", A2, scratch, cyc, info); fetch(rv); cmp("vs potrf64w_core", rw, rv);
        }
        hipMemset(info, 0, 64);
        run<false, 0>("Cholesky of the same matrix: potrf64w_core", A2, scratch, cyc, info); fetch(rw);
        hipMemcpy(hi, info, 16, hipMemcpyDeviceToHost); printf("    info = %d\n", hi[0]);
        hipMemset(info, 0, 64);
        run<false, 6>("Cholesky of the same matrix: potrf64v_core, round-5 arithmetic", A2, scratch, cyc, info); fetch(rv);
        hipMemcpy(hi, info, 16, hipMemcpyDeviceToHost); printf("    info = %d\n", hi[0]);
        cmp("vs potrf64w_core", rw, rv);
        hipMemset(info, 0, 64);
        hipFree(A2);
    }
    return 0;
}
