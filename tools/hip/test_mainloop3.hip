// Stand-alone check of gemm_nt_mainloop3 (three LDS buffers) against gemm_nt_mainloop (two): same 128x128 tile product.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I madnlp.jl_amd/csrc tools/hip/test_mainloop3.hip -o /tmp/t3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include "gemm_tile.h"
namespace mnk { void set_error(const char*, ...) {} }
using namespace mnk;
template <int V>
__global__ __launch_bounds__(256, 3) void k(const double* A, const double* B, double* C, int K, int ld) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    v4f64 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = v4f64{0, 0, 0, 0};
    const double* Ag = A + 128 * blockIdx.x;
    const double* Bg = B + 128 * blockIdx.y;
    if (V == 2) gemm_nt_mainloop<2, 2, 4, 0, 8>(acc, Ag, ld, Bg, ld, K / 8, smem, threadIdx.x);
    else gemm_nt_mainloop3(acc, Ag, ld, Bg, ld, K / 8, smem, threadIdx.x);
    gemm_nt_epilogue<2, 2, 4, 1, false>(acc, 128 * blockIdx.x, 128 * blockIdx.y, 1 << 30, 1 << 30, C, ld, nullptr, nullptr, 0, threadIdx.x);
}
int main() {
    const int M = 512, ld = M;
    for (int K : {8, 16, 24, 128, 1000 / 8 * 8}) {
        std::vector<double> hA((size_t)ld * K), hB((size_t)ld * K), c2((size_t)ld * M), c3((size_t)ld * M);
        for (size_t i = 0; i < hA.size(); ++i) { hA[i] = sin(0.37 * i) ; hB[i] = cos(0.11 * i); }
        double *A, *B, *C2, *C3;
        hipMalloc(&A, hA.size() * 8); hipMalloc(&B, hB.size() * 8); hipMalloc(&C2, c2.size() * 8); hipMalloc(&C3, c3.size() * 8);
        hipMemcpy(A, hA.data(), hA.size() * 8, hipMemcpyHostToDevice); hipMemcpy(B, hB.data(), hB.size() * 8, hipMemcpyHostToDevice);
        hipFuncSetAttribute((const void*)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, TILE3_LDS_BYTES);
        hipLaunchKernelGGL(k<2>, dim3(M / 128, M / 128), dim3(256), 2 * 8 * 288 * 8, 0, A, B, C2, K, ld);
        hipLaunchKernelGGL(k<3>, dim3(M / 128, M / 128), dim3(256), TILE3_LDS_BYTES, 0, A, B, C3, K, ld);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(c2.data(), C2, c2.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(c3.data(), C3, c3.size() * 8, hipMemcpyDeviceToHost);
        double md = 0, mx = 0; size_t bad = 0;
        for (size_t i = 0; i < c2.size(); ++i) { double d = fabs(c2[i] - c3[i]); if (d > md) md = d; if (fabs(c2[i]) > mx) mx = fabs(c2[i]); if (d != 0) ++bad; }
        printf("K=%d: %s max|C2-C3| = %.3e (max|C| %.3e), %zu entries differ\n", K, hipGetErrorString(e), md, mx, bad);
    }
    return 0;
}
