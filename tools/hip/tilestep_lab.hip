// Tile-step laboratory: the left-looking prologue of a pivot-chain strip (factor.hip, pp_strip) on its own -- one workgroup of four
// waves applies K = 256 columns to its 64 x 256 strip, T[t, c] -= V[t, k-chunk] L[c, k-chunk]^T for c = 0..3: 16 tile steps of
// 64 fp64 MFMAs per wave (1.7 us at the matrix pipe's rate), the L tiles read from global memory that ANOTHER kernel has just
// written with write-through stores (as the chain's other strips do).  Variants: prefetch distance 1 (shipped) / 2, loads issued
// before / behind the barrier.  Prints the time per tile step.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/hip/tilestep_lab.hip -o tools/hip/tilestep_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned long long lab_clock() { return __builtin_readcyclecounter(); }

__global__ void writer(double* F, int64_t n, double v) {   // the producer: write-through stores from many CUs
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        __hip_atomic_store(F + i, v + 1e-9 * (double)(i & 1023), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// VAR 0: shipped loop (one register set, load of the next tile issued behind the barrier)
// VAR 1: two register sets (prefetch distance 2)
// VAR 2: no global loads at all (tiles constant in LDS): the loop's own floor
// VAR 3: eight 16-byte loads per lane (two consecutive rows of one column) instead of sixteen 8-byte loads, two 8-byte LDS writes each
// VAR 4: the sixteen loads of the next tile issued ONE PER BLOCK of four products (the texture addresser works under the matrix pipe)
// VAR 5: the next tile goes from global memory straight into LDS (global_load_lds_dwordx4: 8 wave instructions per wave and step, one =
//        two k-columns x 64 rows = 1 KB, no registers, no LDS writes); the tile lies k-column-major in LDS, column pairs (4g, 4g + 2) and
//        (4g + 1, 4g + 3) 1152 bytes apart so that the two k-columns a ds_read_b64 lane group reads sit on different bank halves;
//        same products in the same order as VAR 0
// VAR 6: as VAR 5, the eight requests of the next tile spread over the step (one per two blocks of four products)
template <int VAR>
__global__ __launch_bounds__(256) void pro(const double* __restrict__ F, const double* __restrict__ Vp, int64_t ld, int64_t ldv, int64_t p0,
                                            int t, double* out, unsigned long long* cyc, int reps) {
    extern __shared__ __attribute__((aligned(128))) char smem[];
    v4d* stage = reinterpret_cast<v4d*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t r0 = p0 + 64 * (int64_t)t + 16 * w;
    const int Kp = 256, nch = 4, ncb = 4;
    v4d X[16];
    for (int g = 0; g < 16; ++g) X[g] = v4d{0, 0, 0, 0};
    unsigned long long best = ~0ull;
    for (int rep = 0; rep < reps; ++rep) {
        __syncthreads();
        const unsigned long long t0 = lab_clock();
        v4d pre[4], pre2[4], Bv[4], Bn[4];
        typedef double v2d __attribute__((ext_vector_type(2)));
        v2d pq[8];
        auto tile_load4 = [&](int kc, int c) {
            const double* src = F + (p0 + 64 * (int64_t)c + 2 * (lane & 31)) + (p0 - Kp + 64 * (int64_t)kc + 2 * w + (lane >> 5)) * ld;
#pragma unroll
            for (int k = 0; k < 8; ++k) pq[k] = *reinterpret_cast<const v2d*>(src + (8 * k) * ld);
        };
        auto tile_store4 = [&](v4d* tile) {
            double* td = reinterpret_cast<double*>(tile);
            const int r = 2 * (lane & 31);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int cc = 8 * k + 2 * w + (lane >> 5);
                const int idx = (((r >> 4) * 4 + (cc >> 4)) * 64 + (r & 15) + 16 * (cc & 3)) * 4 + ((cc >> 2) & 3);
                td[idx] = pq[k][0];
                td[idx + 4] = pq[k][1];
            }
        };
        auto tile_load = [&](int kc, int c, v4d (&P)[4]) {
            const double* src = F + (p0 + 64 * (int64_t)c + lane) + (p0 - Kp + 64 * (int64_t)kc + w) * ld;
#pragma unroll
            for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                for (int s = 0; s < 4; ++s) P[ib][s] = src[(16 * ib + 4 * s) * ld];
        };
        auto b_load = [&](int kc, v4d (&B)[4]) {
            const double* src = Vp + (r0 + l15) + (64 * (int64_t)kc + l4) * ldv;
#pragma unroll
            for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                for (int s = 0; s < 4; ++s) B[ib][s] = src[(16 * ib + 4 * s) * ldv];
        };
        auto tile_dma = [&](int kc, int c, char* tile, int q0, int q1) {   // (VAR 5 / 6) requests q0 .. q1 - 1 of this wave's eight
            const int half = lane >> 5, r = 2 * (lane & 31);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (q < q0 || q >= q1) continue;
                const int j = 8 * w + q, g = j >> 1, odd = j & 1;
                const double* src = F + (p0 + 64 * (int64_t)c + r) + (p0 - Kp + 64 * (int64_t)kc + 4 * g + odd + 2 * half) * ld;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(tile + 2176 * g + 1152 * odd), 16, 0, 0);
            }
        };
        auto nxt = [&](int& kc, int& c) { if (++c >= ncb) { c = 0; ++kc; } };
        int it = 0;
        if (VAR == 3) tile_load4(0, 0);
        else if (VAR >= 5) { }
        else if (VAR != 2) tile_load(0, 0, pre);   // (VAR 4: the first tile as usual)
        int k2 = 0, c2 = 0;
        nxt(k2, c2);
        if (VAR == 1) tile_load(k2, c2, pre2);
        b_load(0, Bn);
        if (VAR >= 5) tile_dma(0, 0, smem, 0, 8);
        for (int kc = 0; kc < nch; ++kc) {
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) Bv[ib] = VAR >= 5 ? -Bn[ib] : Bn[ib];
            if (kc + 1 < nch) b_load(kc + 1, Bn);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (VAR >= 5) {
                    char* tile5 = smem + (it & 1) * 34816;
                    char* next5 = smem + ((it + 1) & 1) * 34816;
                    ++it;
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    int c3 = c + 1, k3 = kc;
                    if (c3 >= ncb) { c3 = 0; k3 = kc + 1; }
                    const bool more = k3 < nch;
                    if (VAR == 5 && more) tile_dma(k3, c3, next5, 0, 8);
                    const char* tb = tile5 + 8 * l15 + 1152 * (l4 & 1) + 512 * (l4 >> 1);
#pragma unroll
                    for (int qq = 0; qq < 16; ++qq) {
                        const int cb2 = qq >> 2, ib = qq & 3;
                        if (VAR == 6 && more && (qq & 1) == 0) tile_dma(k3, c3, next5, qq >> 1, (qq >> 1) + 1);
                        double a[4];
#pragma unroll
                        for (int s2 = 0; s2 < 4; ++s2) a[s2] = *reinterpret_cast<const double*>(tb + 2176 * (4 * ib + s2) + 128 * cb2);
#pragma unroll
                        for (int s2 = 0; s2 < 4; ++s2)
                            X[4 * c + cb2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s2], Bv[ib][s2], X[4 * c + cb2], 0, 0, 0);
                    }
                    continue;
                }
                v4d* tile = stage + (it & 1) * 1024;
                if (VAR == 3) tile_store4(tile);
                else if (VAR != 2) {
#pragma unroll
                    for (int ib = 0; ib < 4; ++ib) tile[((lane >> 4) * 4 + ib) * 64 + (lane & 15) + 16 * w] = (VAR == 1 && (it & 1)) ? pre2[ib] : pre[ib];
                }
                ++it;
                __syncthreads();
                if (VAR == 0 || VAR == 3) {
                    int c3 = c + 1, k3 = kc;
                    if (c3 >= ncb) { c3 = 0; k3 = kc + 1; }
                    if (k3 < nch) { if (VAR == 3) tile_load4(k3, c3); else tile_load(k3, c3, pre); }
                }
                if (VAR == 1) {   // the tile two steps ahead, into the register set this step has just emptied
                    int c3 = c, k3 = kc;
                    nxt(k3, c3); nxt(k3, c3);
                    if (k3 < nch) { if (it & 1) tile_load(k3, c3, pre); else tile_load(k3, c3, pre2); }
                }
                if (VAR == 4) {
                    int c3 = c + 1, k3 = kc;
                    if (c3 >= ncb) { c3 = 0; k3 = kc + 1; }
                    const bool more = k3 < nch;
                    const double* src = F + (p0 + 64 * (int64_t)c3 + lane) + (p0 - Kp + 64 * (int64_t)k3 + w) * ld;
                    v4d a = tile[lane];
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int cb2 = q >> 2, ib = q & 3;
                        v4d an = a;
                        if (q < 15) an = tile[(q + 1) * 64 + lane];
                        if (more) pre[q >> 2][q & 3] = src[(16 * (q >> 2) + 4 * (q & 3)) * ld];
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                            X[4 * c + cb2] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[s], Bv[ib][s], X[4 * c + cb2], 0, 0, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                        a = an;
                    }
                } else {
#pragma unroll
                for (int cb2 = 0; cb2 < 4; ++cb2) {
#pragma unroll
                    for (int ib = 0; ib < 4; ++ib) {
                        const v4d a = tile[(cb2 * 4 + ib) * 64 + lane];
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                            X[4 * c + cb2] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[s], Bv[ib][s], X[4 * c + cb2], 0, 0, 0);
                    }
                }
                }
            }
        }
        __syncthreads();
        const unsigned long long t1 = lab_clock();
        if (t1 - t0 < best) best = t1 - t0;
    }
    double sm = 0;
    for (int g = 0; g < 16; ++g) sm += X[g][0] + X[g][1] + X[g][2] + X[g][3];
    out[tid] = sm;
    if (tid == 0) cyc[0] = best;
}

template <int VAR>
static void run(const char* name, double* F, double* V, int64_t ld, double* out, unsigned long long* cyc, bool cold) {
    hipFuncSetAttribute((const void*)pro<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 34816);
    unsigned long long best = ~0ull;
    for (int i = 0; i < 8; ++i) {
        if (cold) { hipLaunchKernelGGL(writer, dim3(512), dim3(256), 0, 0, F, ld * 2048, 0.25 + i); hipLaunchKernelGGL(writer, dim3(512), dim3(256), 0, 0, V, ld * 2048, 0.5 + i); }
        hipLaunchKernelGGL((pro<VAR>), dim3(1), dim3(256), 2 * 34816, 0, F, V, ld, ld, (int64_t)1024, 5, out, cyc, cold ? 1 : 20);
        hipDeviceSynchronize();
        unsigned long long h = 0;
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        if (h < best) best = h;
    }
    std::vector<double> ho(256);
    hipMemcpy(ho.data(), out, 256 * 8, hipMemcpyDeviceToHost);
    double cs = 0;
    for (int i = 0; i < 256; ++i) cs += ho[i] * (1.0 + 0.001 * i);
    printf("[checksum %.17g] ", cs);
    printf("%-66s %s: %7llu cycles = %5.2f us per tile step (16 steps)  %s\n", name, cold ? "operands just written by other CUs" : "operands warm in this CU's L2      ", best,
           best / 2400.0 / 16.0, hipGetErrorString(hipGetLastError()));
}
int main() {
    const int64_t ld = 2048 + 64;
    double *F, *V, *out; unsigned long long* cyc;
    hipMalloc(&F, ld * 2048 * 8); hipMalloc(&V, ld * 2048 * 8); hipMalloc(&out, 4096); hipMalloc(&cyc, 64);
    hipMemset(F, 0, ld * 2048 * 8); hipMemset(V, 0, ld * 2048 * 8);
    for (int cold = 0; cold < 2; ++cold) {
        run<2>("no global loads (LDS tiles as they are): the loop's floor", F, V, ld, out, cyc, cold);
        run<0>("shipped: one register set, next tile requested behind the barrier", F, V, ld, out, cyc, cold);
        run<1>("two register sets: the tile two steps ahead", F, V, ld, out, cyc, cold);
        run<3>("eight 16-byte loads (row pairs) + 8-byte LDS writes", F, V, ld, out, cyc, cold);
        run<4>("one load of the next tile per block of four products", F, V, ld, out, cyc, cold);
        run<5>("next tile straight into LDS (global_load_lds_dwordx4), all 8 first", F, V, ld, out, cyc, cold);
        run<6>("next tile straight into LDS, one request per two product blocks", F, V, ld, out, cyc, cold);
    }
    return 0;
}
