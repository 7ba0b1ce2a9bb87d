// Time of ONE 128-column K-step (16 k-tiles of 8) of a lone workgroup: two-buffer loop vs three-buffer loop, 64x64 and
// 32x128 wave tiles, operands cold (never touched) or warm (second pass over the same operands).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I madnlp.jl_amd/csrc tools/hip/time_kstep.hip -o tools/hip/time_kstep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "gemm_tile.h"
namespace mnk { void set_error(const char*, ...) {} }
using namespace mnk;
template <int V>
__global__ __launch_bounds__(256, 3) void k(const double* A, const double* B, double* C, int ld, unsigned long long* out, int reps, int share) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int rep = 0; rep < reps; ++rep) {
        // share > 0: only `share` distinct row blocks per operand (everything after the first touch is an L2 hit)
        const size_t rb = share > 0 ? blockIdx.x % share : blockIdx.x;
        const double* Ag = A + 128 * rb + (size_t)rep * 128 * ld;   // fresh 128 columns per repetition
        const double* Bg = B + 128 * rb + (size_t)rep * 128 * ld;
        for (int pass = 0; pass < 2; ++pass) {
            __syncthreads();
            const unsigned long long t0 = wall_clock64();
            if (V == 2) {
                v4f64 acc[4][4];
                for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = v4f64{0, 0, 0, 0};
                gemm_nt_mainloop<2, 2, 4, 0, 8>(acc, Ag, ld, Bg, ld, 16, smem, threadIdx.x);
                { v4f64 sm = v4f64{0, 0, 0, 0}; for (auto& row : acc) for (auto& v : row) sm += v; if (sm[0] + sm[1] + sm[2] + sm[3] == 12345.678) C[threadIdx.x] = sm[0]; }
            } else if (V == 3) {
                v4f64 acc[4][4];
                for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = v4f64{0, 0, 0, 0};
                gemm_nt_mainloop3<4, 4>(acc, Ag, ld, Bg, ld, 16, smem, threadIdx.x);
                { v4f64 sm = v4f64{0, 0, 0, 0}; for (auto& row : acc) for (auto& v : row) sm += v; if (sm[0] + sm[1] + sm[2] + sm[3] == 12345.678) C[threadIdx.x] = sm[0]; }
            } else {
                v4f64 acc[8][2];
                for (int i = 0; i < 8; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = v4f64{0, 0, 0, 0};
                gemm_nt_mainloop3<2, 8, true>(acc, Ag, ld, Bg, ld, 16, smem, threadIdx.x);
                { v4f64 sm = v4f64{0, 0, 0, 0}; for (auto& row : acc) for (auto& v : row) sm += v; if (sm[0] + sm[1] + sm[2] + sm[3] == 12345.678) C[threadIdx.x] = sm[0]; }
            }
            const unsigned long long t1 = wall_clock64();
            if (threadIdx.x == 0) out[((size_t)blockIdx.x * reps + rep) * 2 + pass] = t1 - t0;
        }
    }
}
template <int V>
void run(const char* name, int nwg, const double* A, const double* B, double* C, int ld, unsigned long long* out, int reps, int share = 0) {
    const size_t smem = V == 2 ? 2 * 8 * 288 * 8 : TILE3_LDS_BYTES;
    hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(k<V>, dim3(nwg), dim3(256), smem, 0, A, B, C, ld, out, reps, share);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)nwg * reps * 2);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    double cold = 0, warm = 0;
    for (size_t i = 0; i < h.size(); i += 2) { cold += h[i]; warm += h[i + 1]; }
    printf("%-34s %4d workgroups: cold %.2f us, warm %.2f us per K-step (16 k-tiles)\n", name, nwg, cold / (h.size() / 2) / 100.0, warm / (h.size() / 2) / 100.0);
}
int main() {
    const int ld = 128 * 768, reps = 8;
    double *A, *B, *C; unsigned long long* out;
    hipMalloc(&A, (size_t)ld * 128 * reps * 8); hipMalloc(&B, (size_t)ld * 128 * reps * 8); hipMalloc(&C, 4096); hipMalloc(&out, 768 * reps * 2 * 8);
    for (int nwg : {1, 8, 64, 256, 768}) {
        hipMemset(A, 0, (size_t)ld * 128 * reps * 8); hipMemset(B, 0, (size_t)ld * 128 * reps * 8);
        run<2>("two buffers, 64x64 wave tiles", nwg, A, B, C, ld, out, reps);
        hipMemset(A, 0, (size_t)ld * 128 * reps * 8); hipMemset(B, 0, (size_t)ld * 128 * reps * 8);
        run<3>("three buffers, 64x64 wave tiles", nwg, A, B, C, ld, out, reps);
        hipMemset(A, 0, (size_t)ld * 128 * reps * 8); hipMemset(B, 0, (size_t)ld * 128 * reps * 8);
        run<4>("three buffers, 32x128 wave tiles", nwg, A, B, C, ld, out, reps);
    }
    for (int share : {1, 8, 64}) {
        printf("operands shared by the workgroups (%d distinct row blocks):\n", share);
        run<2>("two buffers, 64x64 wave tiles", 768, A, B, C, ld, out, reps, share);
        run<3>("three buffers, 64x64 wave tiles", 768, A, B, C, ld, out, reps, share);
    }
    return 0;
}
