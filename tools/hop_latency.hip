// Cross-workgroup signalling latency on the device: two workgroups ping-pong a flag in global memory
// (release store / acquire load at agent scope).  Decides how a persistent triangular-solve kernel can
// hand x_k from the diagonal solver to the panel updaters.  Spins are bounded (no hang on a bug).
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void pingpong(int* flag, int rounds, int partner_block, unsigned long long* out, int* fail) {
    const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == partner_block ? 1 : -1);
    if (me < 0 || threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    for (int r = 0; r < rounds; ++r) {
        const int want = 2 * r + me;  // flag value that hands the turn to me
        long spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != want) {
            if (++spins > 20000000) { *fail = 1; return; }
        }
        __hip_atomic_store(flag, want + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (me == 0) out[0] = wall_clock64() - t0;
}

// payload variant: the sender also writes 256 doubles before the flag, the receiver reads them after
__global__ void pingpong_payload(int* flag, double* buf, int rounds, int partner_block, unsigned long long* out,
                                 int* fail) {
    const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == partner_block ? 1 : -1);
    if (me < 0) return;
    __shared__ int ok;
    const unsigned long long t0 = wall_clock64();
    double acc = 0.0;
    for (int r = 0; r < rounds; ++r) {
        const int want = 2 * r + me;
        if (threadIdx.x == 0) {
            long spins = 0;
            ok = 1;
            while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != want) {
                if (++spins > 20000000) { *fail = 1; ok = 0; break; }
            }
        }
        __syncthreads();
        if (!ok) return;
        acc += __builtin_nontemporal_load(buf + threadIdx.x);  // the partner's payload
        __syncthreads();
        buf[threadIdx.x] = acc + 1.0;
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(flag, want + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (me == 0 && threadIdx.x == 0) out[0] = wall_clock64() - t0;
    if (acc == 1.2345) out[1] = 1;
}

int main() {
    int *flag, *fail;
    unsigned long long* out;
    double* buf;
    (void)hipMalloc(&flag, 4);
    (void)hipMalloc(&fail, 4);
    (void)hipMalloc(&out, 16);
    (void)hipMalloc(&buf, 256 * 8);
    const int rounds = 2000;
    for (int partner : {1, 8, 4, 16}) {
        for (int payload = 0; payload < 2; ++payload) {
            (void)hipMemset(flag, 0, 4);
            (void)hipMemset(fail, 0, 4);
            (void)hipMemset(buf, 0, 256 * 8);
            if (payload)
                hipLaunchKernelGGL(pingpong_payload, dim3(partner + 1), dim3(256), 0, 0, flag, buf, rounds, partner, out, fail);
            else
                hipLaunchKernelGGL(pingpong, dim3(partner + 1), dim3(64), 0, 0, flag, rounds, partner, out, fail);
            (void)hipDeviceSynchronize();
            unsigned long long h = 0;
            int f = 0;
            (void)hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
            (void)hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
            printf("partner block %2d (%s XCD) %s: %.3f us per one-way hop%s\n", partner, partner % 8 ? "other" : "same",
                   payload ? "flag + 2 KB payload" : "flag only         ", h * 10.0 / 1e3 / (2.0 * rounds), f ? "  [SPIN LIMIT HIT]" : "");
        }
    }
    return 0;
}
