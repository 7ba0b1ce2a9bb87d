// Hand-over latency between two workgroups on two given CUs of an MI355X: a 40 KB payload (the 64 x 64 diagonal block + its inverses of
// the pivot chain) and a flag, ping-pong.  What does the chain's 8 us hand-over consist of, and would a chain whose CUs share
// ONE XCD's L2 hand over faster?  Variants of the producer's stores / the consumer's loads:
//   A  stores: relaxed agent-scope atomics (the chain's put<WT>); consumer: acquire fence (agent) + plain loads        [shipped]
//   B  stores: plain + release fence (agent); consumer as A
//   C  stores as A; consumer: no fence, loads with sc1 (agent-scope atomics)
//   D  stores: plain; consumer: no fence, loads with sc0 only (by-pass the CU's L1: correct only inside one XCD)
//   E  stores as A; consumer: no fence, loads with sc0 only
// build: hipcc --offload-arch=gfx950 -O3 -o tools/hip/handover_lab tools/hip/handover_lab.hip ; run: ./handover_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int PAY = 5120;   // doubles: 64 x 64 block + 4 x 16 x 16 inverses
constexpr int NT = 256;

__device__ __forceinline__ double load_sc(const double* p, int kind) {
    double v;
    if (kind == 0) asm volatile("global_load_dwordx2 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (kind == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx2 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int VAR>
__global__ __launch_bounds__(NT) void pingpong(int role, double* buf0, double* buf1, int* flag0, int* flag1, int iters,
                                                unsigned long long* out, int* errs, int pay, int pthreads) {
    // role 0 sends on (buf0, flag0) and receives on (buf1, flag1); role 1 the other way round
    double* sbuf = role == 0 ? buf0 : buf1;
    double* rbuf = role == 0 ? buf1 : buf0;
    int* sflag = role == 0 ? flag0 : flag1;
    int* rflag = role == 0 ? flag1 : flag0;
    const int tid = threadIdx.x;
    __shared__ int s_dummy;
    int bad = 0;
    const unsigned long long t0 = wall_clock64();
    for (int it = 1; it <= iters; ++it) {
        if (role == 0 || it > 1 || true) {
            if (role == 1) {
                // receive first
            }
        }
        auto send = [&](int val) {
            if (tid < pthreads) for (int i = tid; i < pay; i += pthreads) {
                const double v = (double)val;
                if (VAR == 0 || VAR == 2 || VAR == 4) __hip_atomic_store(sbuf + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else sbuf[i] = v;
            }
            if (VAR == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(sflag, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        auto recv = [&](int val) {
            if (tid == 0) {
                long spins = 0;
                while (__hip_atomic_load(rflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < val) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > 200000000L) break;
                }
                s_dummy = 1;
            }
            __syncthreads();
            if (VAR == 0 || VAR == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            double s = 0.0;
            for (int i = tid; i < pay; i += NT) {
                if (VAR == 0 || VAR == 1) s += rbuf[i];
                else if (VAR == 2) s += load_sc(rbuf + i, 2);
                else s += load_sc(rbuf + i, 1);
            }
            if (s != (double)val * (pay / NT)) ++bad;
        };
        if (role == 0) { send(it); recv(it); }
        else { recv(it); send(it); }
    }
    const unsigned long long t1 = wall_clock64();
    if (tid == 0) out[role] = t1 - t0;
    if (bad) atomicAdd(errs, 1);
}

static bool masked_stream(int bit, hipStream_t* s) {
    uint32_t m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    m[bit / 32] = 1u << (bit % 32);
    return hipExtStreamCreateWithCUMask(s, 8, m) == hipSuccess;
}

template <int VAR>
static void run(const char* name, int bitA, int bitB, const char* where, int pay = PAY, int pthreads = NT) {
    double *b0, *b1; int *f; unsigned long long* out; int* errs;
    CK(hipMalloc(&b0, PAY * 8)); CK(hipMalloc(&b1, PAY * 8)); CK(hipMalloc(&f, 256)); CK(hipMalloc(&out, 16)); CK(hipMalloc(&errs, 4));
    CK(hipMemset(b0, 0, PAY * 8)); CK(hipMemset(b1, 0, PAY * 8)); CK(hipMemset(f, 0, 256)); CK(hipMemset(errs, 0, 4));
    hipStream_t sa, sb;
    if (!masked_stream(bitA, &sa) || !masked_stream(bitB, &sb)) { printf("no masked streams\n"); exit(1); }
    CK(hipDeviceSynchronize());
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(f, 0, 256));
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(pingpong<VAR>, dim3(1), dim3(NT), 0, sa, 0, b0, b1, f, f + 32, iters, out, errs, pay, pthreads);
        hipLaunchKernelGGL(pingpong<VAR>, dim3(1), dim3(NT), 0, sb, 1, b0, b1, f, f + 32, iters, out, errs, pay, pthreads);
        CK(hipDeviceSynchronize());
    }
    unsigned long long h[2]; int he;
    CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(&he, errs, 4, hipMemcpyDeviceToHost));
    printf("%-78s %-16s CUs %3d,%3d: %6.2f us per hand-over (%5.1f KB by %3d threads + flag)%s\n", name, where, bitA, bitB, h[0] * 0.01 / (2.0 * iters),
           pay * 8 / 1024.0, pthreads, he ? "   ** WRONG DATA **" : "");
    fflush(stdout);
    CK(hipStreamDestroy(sa)); CK(hipStreamDestroy(sb));
    CK(hipFree(b0)); CK(hipFree(b1)); CK(hipFree(f)); CK(hipFree(out)); CK(hipFree(errs));
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    // mask bit b is CU b / 8 of XCD b % 8: bits 0 and 8 share XCD 0; bits 0 and 1 are in XCDs 0 and 1; 0 and 4: XCDs 0 and 4
    const int pairs[3][2] = {{0, 8}, {0, 1}, {0, 4}};
    const char* where[3] = {"same XCD", "XCD 0 -> XCD 1", "XCD 0 -> XCD 4"};
    for (int p = 0; p < 3; ++p) {
        run<0>("A  agent-scope atomic stores | acquire fence + plain loads   [shipped]", pairs[p][0], pairs[p][1], where[p]);
        run<1>("B  plain stores + release fence | acquire fence + plain loads", pairs[p][0], pairs[p][1], where[p]);
        run<2>("C  agent-scope atomic stores | no fence, sc1 loads", pairs[p][0], pairs[p][1], where[p]);
        run<3>("D  plain stores | no fence, sc0 loads", pairs[p][0], pairs[p][1], where[p]);
        run<4>("E  agent-scope atomic stores | no fence, sc0 loads", pairs[p][0], pairs[p][1], where[p]);
    }
    for (int pay : {0, 256, 1024, 2560, 5120})
        for (int pt : {256, 64})
            run<0>("A  agent-scope atomic stores | acquire fence + plain loads   [shipped]", 0, 1, "XCD 0 -> XCD 1", pay, pt);
    for (int pay : {0, 256, 1024, 2560, 5120}) run<1>("B  plain stores + release fence | acquire fence + plain loads", 0, 1, "XCD 0 -> XCD 1", pay, 64);
    printf("LAB_EXIT 0\n");
    return 0;
}
