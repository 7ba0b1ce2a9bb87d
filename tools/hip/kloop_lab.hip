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

template <int V>   // 2: two-buffer 128 x 128, 3: three-buffer 128 x 128
__global__ __launch_bounds__(256, 3) void lab_base(const double* F, const double* Vb, double* Cw, const int4* tasks, int ntask, int epi, int tilemajor) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Layout lo = layout(tilemajor);
    for (int t = 0; t < ntask; ++t) {
        const int4 tk = tasks[(size_t)blockIdx.x * ntask + t];
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
__global__ __launch_bounds__(MacroCfg<RT>::NT, 1) void lab_macro(const double* F, const double* Vb, double* Cw, const int4* tasks, int ntask, int epi, int tilemajor) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Layout lo = layout(tilemajor);
    for (int t = 0; t < ntask; ++t) {
        const int4 tk = tasks[(size_t)blockIdx.x * ntask + t];
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
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(nthreads), lds, 0, F, Vb, Cw, dtasks, ntask, epi, tilemajor);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    hipError_t err = hipGetLastError();
    const double flops = ksteps * 2.0 * 128.0 * 128.0 * 128.0;
    const double tf = flops / (best * 1e-3) / 1e12;
    // one k-step of a CU's three 128 x 128 workgroups = 3 x 128^3 x 2 flop at the chip's rate / #CUs
    const double us_kstep = 3.0 * 2.0 * 128.0 * 128.0 * 128.0 / (tf * 1e12 / cus_used) * 1e6;
    printf("%-44s wgs %4d epi %d tm %d klen %2d-%2d: %8.3f ms  %6.2f TFLOP/s  (%.3f of the %d CUs' peak)  k-step %.1f us %s\n", name, nwg, epi, tilemajor, klo, khi, best, tf,
           tf / (78.6 * cus_used / 256.0), cus_used, us_kstep, err == hipSuccess ? "" : hipGetErrorString(err));
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int ntask = argc > 1 ? atoi(argv[1]) : 24;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    g_cus = prop.multiProcessorCount;
    const size_t n = (size_t)LD * LD + 4096;
    hipMalloc(&F, n * 8); hipMalloc(&Vb, n * 8); hipMalloc(&Cw, n * 8);
    hipMalloc(&dtasks, (size_t)4096 * 256 * sizeof(int4));
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
            hipLaunchKernelGGL(kern, dim3(1), dim3(nthreads), lds, 0, F, Vb, Cw, dtasks, 1, 1, tm);
            hipDeviceSynchronize();
            grab(got);
            hipMemset(Cw, 0, n * 8);
            std::vector<int4> per;
            for (int b = 0; b < rt; ++b) per.push_back(make_int4(I + b, J, kb, kl));
            hipMemcpy(dtasks, per.data(), per.size() * sizeof(int4), hipMemcpyHostToDevice);
            hipFuncSetAttribute((const void*)lab_base<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 8 * 288 * 8);
            hipLaunchKernelGGL(lab_base<2>, dim3(rt), dim3(256), 2 * 8 * 288 * 8, 0, F, Vb, Cw, dtasks, 1, 1, tm);
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
    for (int pass = 0; pass < 2; ++pass) {
        const int klo = pass == 0 ? 2 : 8, khi = pass == 0 ? 16 : 40;
        for (int epi = 1; epi >= 0; --epi)
            for (int tm = 0; tm < 2; ++tm) {
                if (epi == 0 && tm == 1) continue;
                timeit("128x128 two buffers, 3 wg/CU", lab_base<2>, 256, 3 * C, 2 * 8 * 288 * 8, 1, ntask, epi, tm, klo, khi);
                timeit("128x128 three buffers, 3 wg/CU", lab_base<3>, 256, 3 * C, TILE3_LDS_BYTES, 1, ntask, epi, tm, klo, khi);
                timeit("macro 256x128 (8 waves), 3 stages", lab_macro<2, 3>, 512, C, macro_lds_bytes<2, 3>(), 2, ntask, epi, tm, klo, khi);
                timeit("macro 256x128 (8 waves), 4 stages", lab_macro<2, 4>, 512, C, macro_lds_bytes<2, 4>(), 2, ntask, epi, tm, klo, khi);
                timeit("macro 384x128 (12 waves), 3 stages", lab_macro<3, 3>, 768, C, macro_lds_bytes<3, 3>(), 3, ntask, epi, tm, klo, khi);
                timeit("macro 384x128 (12 waves), 4 stages", lab_macro<3, 4>, 768, C, macro_lds_bytes<3, 4>(), 3, ntask, epi, tm, klo, khi);
            }
    }
    // the same loops on a grid that leaves 16 CUs out (224 CUs' worth, as beside the pivot chain): per-CU rates should not move
    timeit("128x128 two buffers, 672 wgs", lab_base<2>, 256, 672, 2 * 8 * 288 * 8, 1, ntask, 1, 0, 2, 16, 224);
    timeit("macro 384x128, 4 stages, 224 wgs", lab_macro<3, 4>, 768, 224, macro_lds_bytes<3, 4>(), 3, ntask, 1, 0, 2, 16, 224);
    timeit("macro 256x128, 4 stages, 224 wgs", lab_macro<2, 4>, 512, 224, macro_lds_bytes<2, 4>(), 2, ntask, 1, 0, 2, 16, 224);
    return 0;
}
