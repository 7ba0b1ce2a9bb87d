// Sustained fp64 MFMA rate and shader clock of the whole chip: every wave issues v_mfma_f64_16x16x4 back to back from
// registers (16 independent accumulators, no memory traffic) for tens of milliseconds; the kernel reports flop/s against the
// 100 MHz wall clock and the shader-clock / wall-clock ratio (s_memtime vs s_memrealtime) seen by the waves.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form tools/hip/mfma_clock.hip -o tools/hip/mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(double* sink, unsigned long long* out, int iters, double seed) {
    v4d acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = v4d{0, 0, 0, 0};
    double a = seed * (1.0 + threadIdx.x * 0.37), b = seed * (0.7 - threadIdx.x * 0.011);
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        a = -a;   // keep the sums bounded
    }
    const unsigned long long w1 = wall_clock64(), c1 = clock64();
    v4d s = v4d{0, 0, 0, 0};
    for (int i = 0; i < 16; ++i) s += acc[i];
    if (s[0] + s[1] + s[2] + s[3] == 12345.678) sink[threadIdx.x] = s[0];
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * 4 + threadIdx.x / 64;
        out[2 * w] = w1 - w0;
        out[2 * w + 1] = c1 - c0;
    }
}
int main() {
    double* sink; unsigned long long* out;
    const int maxwg = 256 * 3;
    hipMalloc(&sink, 4096); hipMalloc(&out, (size_t)maxwg * 4 * 2 * 8);
    for (double seed : {0.0, 1.2345678901234567}) {
        for (int wgs : {256, 768}) {
            for (int iters : {20000, 200000}) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, sink, out, iters, seed);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms = 0; hipEventElapsedTime(&ms, e0, e1);
                std::vector<unsigned long long> h((size_t)wgs * 4 * 2);
                hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
                double wsum = 0, csum = 0;
                for (size_t i = 0; i < h.size(); i += 2) { wsum += h[i]; csum += h[i + 1]; }
                const double flop = (double)wgs * 4 * iters * 16 * 2048.0;
                printf("operands %s, %3d workgroups x 4 waves, %6d x 16 MFMA per wave: %.2f ms, %.1f TFLOP/s; shader clock / 100 MHz clock = %.2f -> %.0f MHz\n",
                       seed == 0.0 ? "zero   " : "nonzero", wgs, iters, ms, flop / ms * 1e-9, csum / wsum, csum / wsum * 100.0);
            }
        }
    }
    return 0;
}
