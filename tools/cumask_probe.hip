// Which compute units does a CU-mask bit select?  For a few masks, launch many small workgroups on a masked
// stream and histogram (XCC_ID, SE_ID, CU_ID) of where they ran.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <tuple>
#include <vector>

__global__ void where(unsigned* out) {
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    // burn a little time so workgroups spread over every enabled CU
    double x = threadIdx.x;
    for (int i = 0; i < 2000; ++i) x = x * 1.0000001 + 1e-9;
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = xcc & 0xf;
        out[2 * blockIdx.x + 1] = hwid;
    }
    if (x == 1.234) out[0] = 0;
}

static void run(const char* name, const std::vector<int>& bits) {
    uint32_t mask[8] = {0};
    for (int b : bits) mask[b / 32] |= 1u << (b % 32);
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) { printf("%s: mask rejected\n", name); return; }
    const int n = 8192;
    unsigned* d;
    (void)hipMalloc(&d, n * 8);
    hipLaunchKernelGGL(where, dim3(n), dim3(64), 0, s, d);
    (void)hipStreamSynchronize(s);
    std::vector<unsigned> h(2 * n);
    (void)hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
    std::map<int, std::set<std::pair<int, int>>> per_xcc;  // xcc -> {(se, cu)}
    for (int i = 0; i < n; ++i) {
        const unsigned hw = h[2 * i + 1];
        const int cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;  // gfx9 HW_ID layout
        per_xcc[h[2 * i]].insert({se * 2 + sh, cu});
    }
    printf("%-34s %3zu bits ->", name, bits.size());
    int total = 0;
    for (auto& kv : per_xcc) { printf(" xcc%d:%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
    printf("  = %d CUs\n", total);
    (void)hipFree(d);
    (void)hipStreamDestroy(s);
}

int main() {
    auto range = [](int a, int b) { std::vector<int> v; for (int i = a; i < b; ++i) v.push_back(i); return v; };
    run("bits 0..31", range(0, 32));
    run("bits 0..63", range(0, 64));
    run("bits 64..127", range(64, 128));
    run("bits 0..127", range(0, 128));
    run("bits 128..255", range(128, 256));
    run("bits 0..255", range(0, 256));
    { std::vector<int> v; for (int j = 0; j < 32; ++j) v.push_back(8 * j); run("bits 0,8,16,..,248", v); }
    { std::vector<int> v; for (int j = 0; j < 32; ++j) { v.push_back(8 * j); v.push_back(8 * j + 1); } run("bits {0,1}+8j", v); }
    { std::vector<int> v; for (int j = 0; j < 8; ++j) v.push_back(j); run("bits 0..7", v); }
    { std::vector<int> v; for (int j = 0; j < 16; ++j) v.push_back(j); run("bits 0..15", v); }
    return 0;
}
