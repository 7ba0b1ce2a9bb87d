// K-loop laboratory: the operand-streaming regime of the task-DAG bulk kernel (dag.hip) without the schedule.  Every workgroup
// runs a private list of left-looking tile tasks -- (I, J, kbeg, klen): acc = L(I, k-range) V(J, k-range)^T over klen tile
// columns, then the read-modify-write of the tile -- drawn at random from an N = 11 264 factor, so that the operand streams of
// the workgroups in flight are unrelated (as in the schedule: everything misses the L2s) and the tasks end at unrelated times.
// Variants: the 128 x 128 two-buffer loop of the shipped kernel (three workgroups per CU), its three-buffer form, and the
// macro tiles of gemm_macro.h (RT = 2 / 3 tiles per workgroup, NS = 3 / 4 stages, one workgroup per CU), each on a
// column-major and on a tile-major operand layout.  Reports TFLOP/s of the whole chip and the equivalent time of one
// 128^3 k-step of a CU with three 128 x 128 workgroups (the unit of DESIGN.md 5c: 41.2 us = MFMA-bound at 2.4 GHz).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -I madnlp.jl_amd/csrc tools/hip/kloop_lab.hip -o tools/hip/kloop_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <algorithm>
#include <random>
#include <vector>
#include "gemm_macro.h"
namespace mnk { void set_error(const char*, ...) {} }
using namespace mnk;

constexpr int NTILE = 88;
constexpr int64_t LD = 128 * NTILE;

__global__ void fill_kernel(double* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = ((double)(h & 0xffff) - 32768.0) * (1.0 / 65536.0);
    }
}

// operand addressing: column-major factor (tile (I, k) at rows 128 I, columns 128 k, ld = LD) or tile-major (the 128 x 128
// tiles of a tile row are consecutive 128-KB blocks, ld = 128)
struct Layout { int64_t ld, rowblk, colblk; };   // element (128 I + r, 128 k + c) = base + I * rowblk + k * colblk + r + c * ld
__host__ __device__ inline Layout layout(int tilemajor) {
    return tilemajor ? Layout{128, (int64_t)NTILE * 16384, 16384} : Layout{LD, 128, 128 * LD};
}

// reference: MFMAs from registers only (no LDS, no memory), three workgroups per CU -- the sustained rate of the matrix cores and
// the clock they run at under that load
__global__ __launch_bounds__(256, 3) void lab_mfma_only(double* Cw, int iters, unsigned long long* clk) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter(); clk[1] = wall_clock64(); }
    v4f64 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = v4f64{0, 0, 0, 0};
    double af[4], bf[4];
    for (int i = 0; i < 4; ++i) { af[i] = 1e-3 * (threadIdx.x + i); bf[i] = 1e-3 * (threadIdx.x * 3 + i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[ni], af[mi], acc[ni][mi], 0, 0, 0);
    }
    v4f64 sm = v4f64{0, 0, 0, 0};
    for (auto& row : acc) for (auto& v : row) sm += v;
    if (sm[0] + sm[1] + sm[2] + sm[3] == 12345.678) Cw[threadIdx.x] = sm[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[2] = __builtin_readcyclecounter(); clk[3] = wall_clock64(); }
}

template <int V>   // 2: two-buffer 128 x 128, 3: three-buffer 128 x 128
__global__ __launch_bounds__(256, 3) void lab_base(const double* F, const double* Vb, double* Cw, const int4* tasks, int ntask, int epi, int tilemajor, int* qctr, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_t;
    const Layout lo = layout(tilemajor);
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter(); clk[1] = wall_clock64(); }
    for (;;) {
        if (threadIdx.x == 0) s_t = qctr ? atomicAdd(qctr, 1) : -1;
        __syncthreads();
        const int t = s_t;
        __syncthreads();
        if (t < 0 || t >= ntask) {
            if (blockIdx.x == 0 && threadIdx.x == 0) { clk[2] = __builtin_readcyclecounter(); clk[3] = wall_clock64(); }
            return;
        }
        const int4 tk = tasks[t];
        const int I = __builtin_amdgcn_readfirstlane(tk.x), J = __builtin_amdgcn_readfirstlane(tk.y);
        const int kb = __builtin_amdgcn_readfirstlane(tk.z), kl = __builtin_amdgcn_readfirstlane(tk.w);
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const double* Ag = F + I * lo.rowblk + kb * lo.colblk;
        const double* Bg = Vb + J * lo.rowblk + kb * lo.colblk;
        v4f64 acc[4][4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = v4f64{0, 0, 0, 0};
        if (V == 2) gemm_nt_mainloop<2, 2, 4, 0, 8>(acc, Ag, lo.ld, Bg, lo.ld, kl * 16, smem, tid);
        else gemm_nt_mainloop3<4, 4>(acc, Ag, lo.ld, Bg, lo.ld, kl * 16, smem, tid);
        if (epi) gemm_nt_epilogue<2, 2, 4, 2, false, true>(acc, (int64_t)128 * I, (int64_t)128 * J, (int64_t)1 << 40, (int64_t)1 << 40, Cw, LD, nullptr, nullptr, 0, tid);
        else { v4f64 sm = v4f64{0, 0, 0, 0}; for (auto& row : acc) for (auto& v : row) sm += v; if (sm[0] + sm[1] + sm[2] + sm[3] == 12345.678) Cw[tid] = sm[0]; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

template <int RT, int NS>
__global__ __launch_bounds__(MacroCfg<RT>::NT, 1) void lab_macro(const double* F, const double* Vb, double* Cw, const int4* tasks, int ntask, int epi, int tilemajor, int* qctr, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_t;
    const Layout lo = layout(tilemajor);
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter(); clk[1] = wall_clock64(); }
    for (;;) {
        if (threadIdx.x == 0) s_t = qctr ? atomicAdd(qctr, 1) : -1;
        __syncthreads();
        const int t = s_t;
        __syncthreads();
        if (t < 0 || t >= ntask) {
            if (blockIdx.x == 0 && threadIdx.x == 0) { clk[2] = __builtin_readcyclecounter(); clk[3] = wall_clock64(); }
            return;
        }
        const int4 tk = tasks[t];
        const int I = __builtin_amdgcn_readfirstlane(tk.x), J = __builtin_amdgcn_readfirstlane(tk.y);
        const int kb = __builtin_amdgcn_readfirstlane(tk.z), kl = __builtin_amdgcn_readfirstlane(tk.w);
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const double* Ag = F + I * lo.rowblk + kb * lo.colblk;
        const double* Bg = Vb + J * lo.rowblk + kb * lo.colblk;
        v4f64 acc[4][4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = v4f64{0, 0, 0, 0};
        macro_mainloop<RT, NS>(acc, Ag, lo.ld, lo.rowblk, Bg, lo.ld, kl * 16, RT, smem, tid);
        if (epi) gemm_nt_epilogue<2, 2, 4, 2, false, true>(acc, (int64_t)128 * (I + (tid >> 8)), (int64_t)128 * J, (int64_t)1 << 40, (int64_t)1 << 40, Cw, LD, nullptr, nullptr, 0, tid & 255);
        else { v4f64 sm = v4f64{0, 0, 0, 0}; for (auto& row : acc) for (auto& v : row) sm += v; if (sm[0] + sm[1] + sm[2] + sm[3] == 12345.678) Cw[tid] = sm[0]; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

static double *F, *Vb, *Cw;
static int4* dtasks;
static int* dq;
static unsigned long long* dclk;   // {shader cycles, 100 MHz ticks} at start and end of workgroup 0
static int g_cus = 256;

// tasks of `rt` tiles each: J in [8, 80), klen in [klo, khi], kbeg + klen <= J, I (first tile row) in (J, NTILE - rt]
static double make_tasks(int nwg, int ntask, int rt, int klo, int khi, unsigned seed) {
    std::mt19937 g(seed);
    std::vector<int4> h((size_t)nwg * ntask);
    double ksteps = 0;
    for (auto& t : h) {
        const int J = 8 + (int)(g() % 72);
        int kl = klo + (int)(g() % (khi - klo + 1));
        if (kl > J) kl = J;
        const int kb = (int)(g() % (J - kl + 1));
        const int I = J + 1 + (int)(g() % (NTILE - rt - J));
        t = make_int4(I, J, kb, kl);
        ksteps += (double)kl * rt;
    }
    // (the last tenth of the queue in order of decreasing length: the kernel ends within a short task of its last pop)
    std::sort(h.begin() + (h.size() - h.size() / 10), h.end(), [](const int4& a, const int4& b) { return a.w > b.w; });
    hipMemcpy(dtasks, h.data(), h.size() * sizeof(int4), hipMemcpyHostToDevice);
    return ksteps;
}

template <class K>
static void timeit(const char* name, K kern, int nthreads, int nwg, size_t lds, int rt, int ntask, int epi, int tilemajor, int klo, int khi, int cus_used = 0) {
    if (cus_used <= 0) cus_used = g_cus;
    const double ksteps = make_tasks(nwg, ntask, rt, klo, khi, 12345u);
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemsetAsync(dq, 0, 4, 0);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(nthreads), lds, 0, F, Vb, Cw, dtasks, nwg * ntask, epi, tilemajor, dq, dclk);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    unsigned long long hc[4];
    hipMemcpy(hc, dclk, sizeof(hc), hipMemcpyDeviceToHost);
    const double mhz = (double)(hc[2] - hc[0]) / (double)(hc[3] - hc[1]) * 100.0;
    hipError_t err = hipGetLastError();
    const double flops = ksteps * 2.0 * 128.0 * 128.0 * 128.0;
    const double tf = flops / (best * 1e-3) / 1e12;
    // one k-step of a CU's three 128 x 128 workgroups = 3 x 128^3 x 2 flop at the chip's rate / #CUs
    const double us_kstep = 3.0 * 2.0 * 128.0 * 128.0 * 128.0 / (tf * 1e12 / cus_used) * 1e6;
    printf("%-44s wgs %4d epi %d tm %d klen %2d-%2d: %8.3f ms  %6.2f TFLOP/s  (%.3f of the %d CUs' peak)  k-step %.1f us  shader clock %.0f MHz -> %.3f of the MFMA rate at that clock %s\n", name, nwg, epi, tilemajor, klo, khi, best, tf,
           tf / (78.6 * cus_used / 256.0), cus_used, us_kstep, mhz, tf / (cus_used * 4 * 32.0 * mhz * 1e6 / 1e12), err == hipSuccess ? "" : hipGetErrorString(err));
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int ntask = argc > 1 ? atoi(argv[1]) : 96;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    g_cus = prop.multiProcessorCount;
    const size_t n = (size_t)LD * LD + 4096;
    hipMalloc(&F, n * 8); hipMalloc(&Vb, n * 8); hipMalloc(&Cw, n * 8);
    hipMalloc(&dtasks, (size_t)4096 * 256 * sizeof(int4));
    hipMalloc(&dq, 64);
    hipMalloc(&dclk, 64);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, F, n, 1u);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, Vb, n, 2u);
    hipMemset(Cw, 0, n * 8);
    hipDeviceSynchronize();
    printf("CUs %d, tasks per workgroup %d\n", g_cus, ntask);
    {   // the macro loops against the shipped loop on the same tiles: same k order per accumulator, so the same bits
        auto check = [&](auto kern, int nthreads, size_t lds, int rt, int tm, const char* name) {
            const int I = 40, J = 20, kb = 3, kl = 5;
            std::vector<int4> one{make_int4(I, J, kb, kl)};
            std::vector<double> got((size_t)rt * 16384), ref((size_t)rt * 16384);
            auto grab = [&](std::vector<double>& out) {
                for (int b = 0; b < rt; ++b)
                    hipMemcpy2D(out.data() + (size_t)b * 16384, 128 * 8, Cw + (size_t)128 * (I + b) + (size_t)128 * J * LD, LD * 8, 128 * 8, 128, hipMemcpyDeviceToHost);
            };
            hipMemset(Cw, 0, n * 8);
            hipMemcpy(dtasks, one.data(), sizeof(int4), hipMemcpyHostToDevice);
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipMemset(dq, 0, 4);
            hipLaunchKernelGGL(kern, dim3(1), dim3(nthreads), lds, 0, F, Vb, Cw, dtasks, 1, 1, tm, dq, dclk);
            hipDeviceSynchronize();
            grab(got);
            hipMemset(Cw, 0, n * 8);
            std::vector<int4> per;
            for (int b = 0; b < rt; ++b) per.push_back(make_int4(I + b, J, kb, kl));
            hipMemcpy(dtasks, per.data(), per.size() * sizeof(int4), hipMemcpyHostToDevice);
            hipFuncSetAttribute((const void*)lab_base<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 8 * 288 * 8);
            hipMemset(dq, 0, 4);
            hipLaunchKernelGGL(lab_base<2>, dim3(rt), dim3(256), 2 * 8 * 288 * 8, 0, F, Vb, Cw, dtasks, rt, 1, tm, dq, dclk);
            hipDeviceSynchronize();
            grab(ref);
            size_t bad = 0; double mx = 0;
            for (size_t i = 0; i < got.size(); ++i) { if (got[i] != ref[i]) ++bad; mx = std::max(mx, std::fabs(ref[i])); }
            printf("check %-28s tm %d: %zu of %zu entries differ (max |ref| %.3g) %s\n", name, tm, bad, got.size(), mx, hipGetErrorString(hipGetLastError()));
        };
        for (int tm = 0; tm < 2; ++tm) {
            check(lab_macro<2, 3>, 512, macro_lds_bytes<2, 3>(), 2, tm, "macro 256x128, 3 stages");
            check(lab_macro<2, 4>, 512, macro_lds_bytes<2, 4>(), 2, tm, "macro 256x128, 4 stages");
            check(lab_macro<3, 3>, 768, macro_lds_bytes<3, 3>(), 3, tm, "macro 384x128, 3 stages");
            check(lab_macro<3, 4>, 768, macro_lds_bytes<3, 4>(), 3, tm, "macro 384x128, 4 stages");
        }
        hipMemset(Cw, 0, n * 8);
    }
    const int C = g_cus;
    for (int nw : {3 * C, 2 * C, C}) {   // MFMAs only
        const int iters = 40000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(lab_mfma_only, dim3(nw), dim3(256), 0, 0, Cw, iters, dclk);
            hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        unsigned long long hc[4];
        hipMemcpy(hc, dclk, sizeof(hc), hipMemcpyDeviceToHost);
        const double mhz = (double)(hc[2] - hc[0]) / (double)(hc[3] - hc[1]) * 100.0;
        const double tf = (double)nw * 4 * iters * 16 * 2048.0 / (ms * 1e-3) / 1e12;
        printf("MFMA from registers only, %4d workgroups: %8.3f ms  %6.2f TFLOP/s  shader clock %.0f MHz -> %.3f of the MFMA rate at that clock\n", nw, ms, tf, mhz,
               tf / (C * 4 * 32.0 * mhz * 1e6 / 1e12));
    }
    for (int pass = 0; pass < 2; ++pass) {
        const int klo = pass == 0 ? 2 : 8, khi = pass == 0 ? 16 : 40;
        for (int mode = 0; mode < 3; ++mode) {   // (epilogue, layout): (1, column-major), (1, tile-major), (0, column-major)
            const int epi = mode < 2, tm = mode == 1;
            timeit("128x128 two buffers, 3 wg/CU", lab_base<2>, 256, 3 * C, 2 * 8 * 288 * 8, 1, ntask, epi, tm, klo, khi);
            timeit("128x128 three buffers, 3 wg/CU", lab_base<3>, 256, 3 * C, TILE3_LDS_BYTES, 1, ntask, epi, tm, klo, khi);
            timeit("macro 256x128 (8 waves), 3 stages", lab_macro<2, 3>, 512, C, macro_lds_bytes<2, 3>(), 2, ntask * 3 / 2, epi, tm, klo, khi);
            timeit("macro 256x128 (8 waves), 4 stages", lab_macro<2, 4>, 512, C, macro_lds_bytes<2, 4>(), 2, ntask * 3 / 2, epi, tm, klo, khi);
            timeit("macro 384x128 (12 waves), 3 stages", lab_macro<3, 3>, 768, C, macro_lds_bytes<3, 3>(), 3, ntask, epi, tm, klo, khi);
            timeit("macro 384x128 (12 waves), 4 stages", lab_macro<3, 4>, 768, C, macro_lds_bytes<3, 4>(), 3, ntask, epi, tm, klo, khi);
        }
    }
    // the same loops on a grid that leaves 16 CUs out (224 CUs' worth, as beside the pivot chain): per-CU rates should not move
    for (int nw : {2 * C, 2 * C + C / 4, 2 * C + C / 2})
        timeit("128x128 two buffers, fewer workgroups", lab_base<2>, 256, nw, 2 * 8 * 288 * 8, 1, ntask, 1, 0, 2, 16);
    timeit("128x128 two buffers, 672 wgs", lab_base<2>, 256, 672, 2 * 8 * 288 * 8, 1, ntask, 1, 0, 2, 16, 224);
    timeit("macro 384x128, 4 stages, 224 wgs", lab_macro<3, 4>, 768, 224, macro_lds_bytes<3, 4>(), 3, ntask, 1, 0, 2, 16, 224);
    timeit("macro 256x128, 4 stages, 224 wgs", lab_macro<2, 4>, 512, 224, macro_lds_bytes<2, 4>(), 2, ntask, 1, 0, 2, 16, 224);
    return 0;
}
