// What happens to workgroups of a persistent grid that the hardware places LATE, beside workgroups that spin?
// (in the task-DAG schedule such workgroups took a task and then made no progress until the
// spinning ones went home.)  A grid of G workgroups, three per CU by their LDS (50 KB + 2 KB static), on a stream whose CU mask leaves out
// the first 16 bits -- the bulk kernel's mask, which holds 672 at launch.  Workgroups with a block id below `nspin` poll a
// counter (s_sleep 4 between polls, as the schedule's waits do) until the others -- the workers -- have all added to it, or a
// bound expires; a worker reads through a buffer for ~100-200 us and adds one.  Printed per worker: when it started relative
// to the first workgroup of the grid, how long its work took, where it ran; for the spinners: when they left.
// usage: late_wg_probe [G nspin iters]   (default: the three cases of main)
// build: hipcc --offload-arch=gfx950 -O2 tools/hip/late_wg_probe.hip -o tools/hip/late_wg_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Rec { unsigned long long t0, t1; unsigned hwid, xcc; long polls; double sum; };

__global__ __launch_bounds__(256) void probe(int* flag, int target, int nspin, long limit, const double* buf, int nbuf_mask, int iters, Rec* recs) {
    extern __shared__ char lds[];
    __shared__ double red[256];
    lds[threadIdx.x] = 1;
    Rec r{};
    if (threadIdx.x == 0) {
        r.t0 = wall_clock64();
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(r.xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(r.hwid));
    }
    if ((int)blockIdx.x < nspin) {
        if (threadIdx.x == 0) {
            long polls = 0;
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++polls < limit) __builtin_amdgcn_s_sleep(4);
            r.polls = polls;
        }
        __syncthreads();
    } else {
        double acc = 0.0;
        int idx = (int)(threadIdx.x + 977u * blockIdx.x) & nbuf_mask;
        for (int i = 0; i < iters; ++i) {
            const double v = buf[idx];
            acc += v;
            idx = (idx + 4099 + (int)(v * 0.0)) & nbuf_mask;   // (dependent loads: one round trip to memory per iteration)
        }
        red[threadIdx.x] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int i = 1; i < 256; ++i) acc += red[i];
            r.sum = acc;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            atomicAdd(flag, 1);
        }
    }
    if (threadIdx.x == 0) { r.t1 = wall_clock64(); recs[blockIdx.x] = r; }
}

static void run_case(hipStream_t s, int G, int nspin, int iters, const double* buf, int nbuf_mask) {
    int* flag;
    Rec* recs;
    (void)hipMalloc(&flag, 4); (void)hipMalloc(&recs, sizeof(Rec) * G);
    (void)hipMemset(flag, 0, 4); (void)hipMemset(recs, 0, sizeof(Rec) * G);
    const long limit = 400000;   // ~80 ms of polls: then the spinners leave and whatever was not placed runs
    hipLaunchKernelGGL(probe, dim3(G), dim3(256), 50 * 1024, s, flag, G - nspin, nspin, limit, buf, nbuf_mask, iters, recs);
    (void)hipStreamSynchronize(s);
    std::vector<Rec> h(G);
    (void)hipMemcpy(h.data(), recs, sizeof(Rec) * G, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull;
    for (auto& r : h) if (r.t0 && r.t0 < t0) t0 = r.t0;
    long maxpolls = 0;
    double spin_end_min = 1e30, spin_end_max = 0, spin_start_max = 0;
    for (int i = 0; i < nspin; ++i) {
        maxpolls = std::max(maxpolls, h[i].polls);
        spin_end_min = std::min(spin_end_min, (h[i].t1 - t0) / 100.0);
        spin_end_max = std::max(spin_end_max, (h[i].t1 - t0) / 100.0);
        spin_start_max = std::max(spin_start_max, (h[i].t0 - t0) / 100.0);
    }
    printf("G = %d, spinners %d (last one started at %.0f us; they left between %.0f and %.0f us, %s), workers %d, %d dependent loads each:\n", G, nspin,
           spin_start_max, spin_end_min, spin_end_max, maxpolls >= limit - 1 ? "SOME BY THE BOUND" : "all released by the workers", G - nspin, iters);
    std::vector<double> dur;
    for (int i = nspin; i < G; ++i) {
        const Rec& r = h[i];
        dur.push_back((r.t1 - r.t0) / 100.0);
        printf("  worker %3d: started %9.0f us, work took %9.0f us   xcc %u se %u cu %u\n", i, (r.t0 - t0) / 100.0, (r.t1 - r.t0) / 100.0, r.xcc & 15, (r.hwid >> 13) & 7,
               (r.hwid >> 8) & 15);
    }
    std::sort(dur.begin(), dur.end());
    if (!dur.empty()) printf("  work: min %.0f, median %.0f, max %.0f us\n", dur.front(), dur[dur.size() / 2], dur.back());
    (void)hipFree(flag); (void)hipFree(recs);
}

int main(int argc, char** argv) {
    (void)hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 50 * 1024);
    uint32_t mask[8];
    for (int w = 0; w < 8; ++w) mask[w] = 0xffffffffu;
    mask[0] = 0xffff0000u;   // (without the 16 CUs of the pivot chain)
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) { printf("CU mask rejected\n"); return 1; }
    const int nbuf = 1 << 22;   // 32 MB
    double* buf;
    (void)hipMalloc(&buf, sizeof(double) * nbuf);
    (void)hipMemset(buf, 0, sizeof(double) * nbuf);
    if (argc > 3) {
        run_case(s, atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), buf, nbuf - 1);
    } else {
        run_case(s, 672, 624, 200, buf, nbuf - 1);   // control: everything placed at launch
        run_case(s, 720, 672, 200, buf, nbuf - 1);   // the 48 workgroups beyond what is placed at launch are the workers
        run_case(s, 720, 640, 200, buf, nbuf - 1);   // workers: 32 placed at launch + 48 late
    }
    return 0;
}
