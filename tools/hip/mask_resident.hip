// Can W workgroups that need a CU each (96 KB of LDS) all be RESIDENT at once on a stream whose CU mask has the first n
// bits set?  Each workgroup announces itself and waits (bounded) until all W have: the persistent pivot chains of the
// factorization need exactly this.  Prints, per (n, W): how many arrived, and where the workgroups ran (per XCC / SE).
// build: hipcc --offload-arch=gfx950 -O2 tools/hip/mask_resident.hip -o tools/hip/mask_resident
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ void meet(int* ctr, int W, long limit, unsigned* where, int* arrived) {
    extern __shared__ char lds[];
    if (threadIdx.x == 0) {
        lds[0] = 1;
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        where[2 * blockIdx.x] = xcc & 0xf;
        where[2 * blockIdx.x + 1] = hwid;
        atomicAdd(ctr, 1);
        long spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < W && ++spins < limit) __builtin_amdgcn_s_sleep(8);
        arrived[blockIdx.x] = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main(int argc, char** argv) {
    (void)hipFuncSetAttribute((const void*)meet, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    const int cases[][2] = {{16, 16}, {32, 32}, {40, 34}, {40, 40}, {48, 34}, {48, 48}, {56, 56}, {64, 34}, {64, 64}, {72, 68}, {72, 72},
                            {96, 96}, {104, 102}, {176, 170}, {192, 192}, {256, 256}};
    for (auto& c : cases) {
        const int n = c[0], W = c[1];
        uint32_t mask[8] = {0};
        for (int b = 0; b < n; ++b) mask[b / 32] |= 1u << (b % 32);
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) { printf("n=%d: mask rejected\n", n); continue; }
        int *ctr, *arr;
        unsigned* where;
        (void)hipMalloc(&ctr, 4); (void)hipMalloc(&arr, 4 * W); (void)hipMalloc(&where, 8 * W);
        (void)hipMemset(ctr, 0, 4);
        hipLaunchKernelGGL(meet, dim3(W), dim3(256), 96 * 1024, s, ctr, W, 2000000L, where, arr);
        (void)hipStreamSynchronize(s);
        std::vector<int> ha(W);
        std::vector<unsigned> hw(2 * W);
        (void)hipMemcpy(ha.data(), arr, 4 * W, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hw.data(), where, 8 * W, hipMemcpyDeviceToHost);
        int mn = W;
        for (int v : ha) mn = v < mn ? v : mn;
        std::map<int, std::map<int, int>> per;   // xcc -> se -> count
        for (int i = 0; i < W; ++i) per[hw[2 * i]][(hw[2 * i + 1] >> 13) & 7]++;
        printf("mask first %3d bits, %3d workgroups: %s (min seen %d) |", n, W, mn >= W ? "ALL RESIDENT" : "NOT all resident", mn);
        for (auto& x : per) { printf(" x%d:", x.first); for (auto& se : x.second) printf("%d", se.second); }
        printf("\n");
        (void)hipFree(ctr); (void)hipFree(arr); (void)hipFree(where);
        (void)hipStreamDestroy(s);
    }
    return 0;
}
