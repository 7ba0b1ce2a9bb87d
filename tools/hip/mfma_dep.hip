// fp64 MFMA 16x16x4: 64 products into ONE accumulator back to back (the order of the chain strips' tile steps: 16 dependent products
// per accumulator) against the same 64 products rotating over 2 / 4 accumulators.  One wave per SIMD (256 threads), cycles per product.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/hip/mfma_dep.hip -o tools/hip/mfma_dep
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, unsigned long long* cyc, double a0) {
    v4d acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    double a[16], b[16];
    for (int i = 0; i < 16; ++i) { a[i] = a0 + 1e-3 * i + 1e-6 * threadIdx.x; b[i] = 1e-4 * i; }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    asm volatile("" ::: "memory");
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int q = 0; q < 64; ++q) {
            const int u = NACC == 1 ? (q >> 4) : (NACC == 4 ? (q & 3) : ((q >> 4) & 2) + (q & 1));   // 1: 16 in a row per accumulator; 4: rotate; 2: pairs
            acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q & 15], b[(q >> 2) & 15], acc[u], 0, 0, 0);
        }
        asm volatile("" : "+v"(a[0]));
    }
    asm volatile("" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    double sm = 0;
    for (int u = 0; u < 4; ++u) sm += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
    out[threadIdx.x] = sm;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC>
static void run(const char* name, double* out, unsigned long long* cyc) {
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<NACC>), dim3(1), dim3(256), 0, 0, out, cyc, 0.5);
    hipDeviceSynchronize();
    unsigned long long h = 0;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-70s: %6.1f cycles per product\n", name, (double)h / (64.0 * 64.0));
}
int main() {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 8192); hipMalloc(&cyc, 64);
    run<1>("16 dependent products per accumulator, four accumulators in turn", out, cyc);
    run<2>("products alternating between two accumulators", out, cyc);
    run<4>("products rotating over four accumulators", out, cyc);
    return 0;
}
