// Sustained fp64 MFMA rate of the device (no memory traffic): what the "peak" in the roofline is
// worth under sustained load (clock/power management included).  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v4f64 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(1024) void mfma_loop(double* out, const double* in, long iters) {
    v4f64 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
    double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
    double xs[8], ys[8];
    if (in != nullptr) {  // random operands (data toggling as in a real product)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            xs[j] = in[(threadIdx.x * 8 + j) % 4096];
            ys[j] = in[(threadIdx.x * 8 + j + 2048) % 4096];
        }
        for (long i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xs[0], ys[0], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xs[1], ys[1], a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(xs[2], ys[2], a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(xs[3], ys[3], a3, 0, 0, 0);
            a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(xs[4], ys[4], a4, 0, 0, 0);
            a5 = __builtin_amdgcn_mfma_f64_16x16x4f64(xs[5], ys[5], a5, 0, 0, 0);
            a6 = __builtin_amdgcn_mfma_f64_16x16x4f64(xs[6], ys[6], a6, 0, 0, 0);
            a7 = __builtin_amdgcn_mfma_f64_16x16x4f64(xs[7], ys[7], a7, 0, 0, 0);
        }
    } else
    for (long i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
        a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a4, 0, 0, 0);
        a5 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a5, 0, 0, 0);
        a6 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a6, 0, 0, 0);
        a7 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a7, 0, 0, 0);
    }
    v4f64 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s[0] == 123.456) out[threadIdx.x] = s[1];
}

int main(int argc, char** argv) {
    double* out;
    hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    double* in;
    hipMalloc(&in, 4096 * 8);
    {
        double h[4096];
        srand(1);
        for (int i = 0; i < 4096; ++i) h[i] = (rand() / (double)RAND_MAX - 0.5) * 1e-3;
        hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    }
    const int grids[] = {64, 128, 192, 256, 512};
    const int blocks[] = {256, 512};
    const long it = argc > 1 ? atol(argv[1]) : 100000;
    for (int rnd = 0; rnd < 2; ++rnd)
    for (int b : blocks)
        for (int g : grids) {
            if ((long)g * b > 1024L * 1024) continue;
            hipEventRecord(e0);
            hipLaunchKernelGGL(mfma_loop, dim3(g), dim3(b), 0, 0, out, rnd ? in : nullptr, it);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)g * (b / 64) * it * 8 * 2048.0;
            // cycles per MFMA per SIMD at 2.4 GHz if the workgroups are spread one per CU (g <= 256)
            const double wgs_per_cu = g <= 256 ? 1.0 : g / 256.0;
            const double waves_per_simd = wgs_per_cu * (b / 256.0);
            const double cyc = ms * 1e-3 * 2.4e9 / (it * 8 * waves_per_simd);
            printf("%s grid %4d x %4d thr  (%.0f waves/SIMD)  %9.3f ms  %7.2f TFLOP/s  %.1f cycles@2.4GHz per MFMA per SIMD\n", rnd ? "random" : "const ", g, b,
                   waves_per_simd, ms, flops / ms / 1e9, cyc);
        }
    return 0;
}
