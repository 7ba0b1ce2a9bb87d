// Latency / issue laboratory for the pivot leaf's instruction kinds (one wave): cycles per instruction in a DEPENDENT chain and
// in an independent stream, fp64 VALU, v_rcp_f64, v_readlane round trips, v_cndmask_b32 and the two fp64 MFMA shapes.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/hip/valu_lat.hip -o tools/hip/valu_lat
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
#define REP 512
template <int KIND>
__global__ __launch_bounds__(64) void k(double* out, unsigned long long* cyc, double a0, double b0) {
    double a = a0 + threadIdx.x * 1e-9, b = b0, c = 1e-3, d = 2e-3, e = 3e-3, f = 4e-3, g = 5e-3, h = 6e-3;
    v4d acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
    double s44 = 0.0;
    int i1 = threadIdx.x, i2 = threadIdx.x + 1;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < REP; ++i) {
        if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a) : "v"(b));                       // dependent fma
        if (KIND == 1) { asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a) : "v"(b)); asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(c) : "v"(b));
                         asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d) : "v"(b)); asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(e) : "v"(b)); }   // 4 independent
        if (KIND == 2) asm volatile("v_rcp_f64 %0, %0" : "+v"(a));                                           // dependent rcp
        if (KIND == 3) { asm volatile("v_rcp_f64 %0, %0" : "+v"(a)); asm volatile("v_rcp_f64 %0, %0" : "+v"(c)); asm volatile("v_rcp_f64 %0, %0" : "+v"(d)); asm volatile("v_rcp_f64 %0, %0" : "+v"(e)); }
        if (KIND == 4) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));                              // dependent mul
        if (KIND == 5) { int lo = __builtin_amdgcn_readlane(__double2loint(a), 5), hi = __builtin_amdgcn_readlane(__double2hiint(a), 5);
                         double u = __hiloint2double(hi, lo); asm volatile("v_fma_f64 %0, %1, %2, %2" : "=v"(a) : "s"(u), "v"(b)); }   // readlane -> fma -> readlane
        if (KIND == 6) { int x = __double2loint(a); int seven = 7; asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x) : "v"(seven) : "vcc"); a = __hiloint2double(__double2hiint(a), x); }  // dependent cndmask
        if (KIND == 7) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);                        // dependent accumulate (same acc)
        if (KIND == 8) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0); a = acc[0]; asm volatile("" : "+v"(a)); }   // mfma -> its result is the next A operand
        if (KIND == 9) { s44 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s44, 0, 0, 0); }                       // 4x4x4 dependent accumulate
        if (KIND == 10) { a = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0); asm volatile("" : "+v"(a)); }   // 4x4x4 -> next A operand
        if (KIND == 11) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0); }  // 2 independent
        if (KIND == 12) { asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a) : "v"(b)); asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(c) : "v"(b)); }   // 2 independent chains
        if (KIND == 13) { asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a) : "v"(b)); asm volatile("v_mov_b32 %0, %0" : "+v"(i1)); asm volatile("v_mov_b32 %0, %0" : "+v"(i2)); }   // dependent fma + 2 unrelated b32 ops
        if (KIND == 14) { s44 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0); asm volatile("v_fma_f64 %0, %1, %2, %2" : "=v"(a) : "v"(s44), "v"(b)); }   // 4x4x4 -> fma -> 4x4x4
        if (KIND == 15) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0); double u = acc[1]; int lo = __builtin_amdgcn_readlane(__double2loint(u), 5), hi = __builtin_amdgcn_readlane(__double2hiint(u), 5);
                          a = __hiloint2double(hi, lo); asm volatile("" : "+v"(a)); }   // 16x16x4 -> readlane of its result -> next operand
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = a + c + d + e + f + g + h + acc[0] + acc[1] + acc2[0] + s44 + i1 + i2;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int KIND>
static void run(const char* name, int per, double* out, unsigned long long* cyc) {
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<KIND>), dim3(1), dim3(64), 0, 0, out, cyc, 0.9999, 1e-4);
    hipDeviceSynchronize();
    unsigned long long h = 0;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-78s: %7.1f cycles per iteration (%d instr) %s\n", name, (double)h / REP, per, hipGetErrorString(hipGetLastError()));
}
int main() {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 64);
    run<0>("v_fma_f64, dependent chain", 1, out, cyc);
    run<1>("v_fma_f64, 4 independent chains", 4, out, cyc);
    run<12>("v_fma_f64, 2 independent chains", 2, out, cyc);
    run<13>("v_fma_f64 dependent + 2 unrelated v_mov_b32", 3, out, cyc);
    run<4>("v_mul_f64, dependent chain", 1, out, cyc);
    run<2>("v_rcp_f64, dependent chain", 1, out, cyc);
    run<3>("v_rcp_f64, 4 independent chains", 4, out, cyc);
    run<5>("v_readlane x2 -> v_fma_f64 (SGPR operand) -> v_readlane ...", 3, out, cyc);
    run<6>("v_cndmask_b32, dependent chain", 1, out, cyc);
    run<7>("v_mfma_f64_16x16x4, same accumulator", 1, out, cyc);
    run<11>("v_mfma_f64_16x16x4, two accumulators", 2, out, cyc);
    run<8>("v_mfma_f64_16x16x4 -> result is the next A operand", 1, out, cyc);
    run<15>("v_mfma_f64_16x16x4 -> v_readlane x2 of the result -> next A operand", 3, out, cyc);
    run<9>("v_mfma_f64_4x4x4_4b, same accumulator", 1, out, cyc);
    run<10>("v_mfma_f64_4x4x4_4b -> result is the next A operand", 1, out, cyc);
    run<14>("v_mfma_f64_4x4x4_4b -> v_fma_f64 -> v_mfma_f64_4x4x4_4b", 2, out, cyc);
    return 0;
}
