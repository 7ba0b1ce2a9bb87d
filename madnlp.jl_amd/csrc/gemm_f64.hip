// fp64 MFMA tile kernel:  C (MxN) op= A (MxK) * B (NxK)^T, everything column-major.
//
// This is the trailing-update / panel-solve workhorse of the blocked right-looking
// Cholesky / LDL^T factorization that stands in for LAPACK dpotrf / dsytrf
// (reference src/LinearSolvers/lapack.jl:145-148,164-167) and of the J_i' D J_i
// product of build_kkt!(::DenseCondensedKKTSystem) (reference
// src/KKT/Dense/condensed.jl:178).
//
// gfx950 mapping
//   * v_mfma_f64_16x16x4_f64: one wave owns a 64x64 block of C as 4x4 MFMA tiles
//     (16 accumulators x 4 f64 = 128 VGPRs); a workgroup is WM x WN waves.
//   * We compute C^T tiles (MFMA "A" operand = B fragment, "B" operand = A fragment)
//     so that a lane's 16 lanes-in-a-row hold 16 consecutive rows of one column
//     of C: the epilogue touches C in 128-byte column segments.
//   * Both operands are "row index contiguous", so an LDS tile is [BK][rows+16]:
//     a fragment read (ds_read_b64) has lanes 0-15 on 16 consecutive rows of
//     k, lanes 16-31 on k+1, ...; the +16 pad makes the row pitch = 16 mod 32
//     doubles so the two 16-lane halves of a 32-lane LDS group hit disjoint banks.
//   * Global -> LDS staging goes through registers in 16-byte pieces, double
//     buffered: the loads of k-tile t+1 are issued before the MFMAs of k-tile t.
#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "gemm_tile.h"

namespace mnk {

template <int WM, int WN, int WT, int MODE, bool LDL_EPI>
__global__ __launch_bounds__(64 * WM * WN, tile_occ(WM, WN, WT)) void gemm_nt_kernel(
    int64_t M, int64_t N, int64_t K, const double* __restrict__ A, int64_t lda,
    const double* __restrict__ B, int64_t ldb, double* C, int64_t ldc,
    const double* __restrict__ colscale, double* C2, int64_t ldc2, int ntm,
    const int* __restrict__ info_flag, int tile_off, int nc, int sw) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    if (info_flag != nullptr && *info_flag != 0) return;

    // XCD-aware remap: hardware places consecutive workgroup ids round-robin on the
    // 8 XCDs; give each XCD a contiguous run of logical tiles so that tiles sharing
    // an A row-block / B column-block share one L2.
    int nblk = gridDim.x;
    int bid = blockIdx.x;
    int per = nblk >> 3;
    int logical = tile_off + ((per > 0 && bid < per * 8) ? (bid & 7) * per + (bid >> 3) : bid);
    int tm, tn;
    decode_tile<MODE>(logical, ntm, nc, sw, tm, tn);
    gemm_nt_tile<WM, WN, WT, MODE, LDL_EPI, 0, tile_bk(WM, WN, WT)>(tm, tn, M, N, K, A, lda, B, ldb, C, ldc, colscale,
                                                                       C2, ldc2, smem_raw);
}

template <int WM, int WN, int WT, int MODE, int DBG>
__global__ __launch_bounds__(64 * WM * WN, DBG == 6 ? 2 : tile_occ(WM, WN, WT)) void gemm_nt_dbg_kernel(
    int64_t M, int64_t N, int64_t K, const double* __restrict__ A, int64_t lda,
    const double* __restrict__ B, int64_t ldb, double* C, int64_t ldc, int ntm) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    int nblk = gridDim.x, bid = blockIdx.x, per = nblk >> 3;
    int logical = (per > 0 && bid < per * 8) ? (bid & 7) * per + (bid >> 3) : bid;
    int tm, tn;
    decode_tile<MODE>(logical, ntm, ntm, 1, tm, tn);
    gemm_nt_tile<WM, WN, WT, MODE, false, DBG == 6 ? 0 : DBG, DBG == 6 ? 16 : tile_bk(WM, WN, WT)>(tm, tn, M, N, K, A, lda, B, ldb, C, ldc, nullptr, nullptr, 0, smem_raw);
}

// Work-queue variant: the workgroups of one or more launches (possibly on different streams with
// different CU masks) pull logical tile indices from device counters until `ntiles` are taken.
// Used to let the panel stream's CUs join the trailing update once the next panel is factored.
// The tile range is cut into 8 contiguous chunks, one per XCD, each with its own counter: a
// workgroup drains the chunk of the XCD it runs on first (tiles of one chunk share A/B blocks in
// that XCD's L2, like the static remap of gemm_nt_kernel) and then steals from the other chunks.
template <int WM, int WN, int WT, int MODE>
__global__ __launch_bounds__(64 * WM * WN, tile_occ(WM, WN, WT)) void gemm_nt_queue_kernel(
    int64_t M, int64_t N, int64_t K, const double* __restrict__ A, int64_t lda,
    const double* __restrict__ B, int64_t ldb, double* C, int64_t ldc, int ntm, int ntiles,
    int* __restrict__ counters /* [8] */, const int* __restrict__ info_flag, int nc, int sw) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ int s_tile;
    if (info_flag != nullptr && *info_flag != 0) return;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int home = (int)(xcc & 7u);
    int probe = 0;
    while (probe < 8) {
        const int q = (home + probe) & 7;
        const int lo = (int)(((int64_t)ntiles * q) >> 3), hi = (int)(((int64_t)ntiles * (q + 1)) >> 3);
        if (threadIdx.x == 0) s_tile = atomicAdd(counters + q, 1);
        __syncthreads();
        const int logical = lo + s_tile;
        __syncthreads();  // everyone has read s_tile before the next round overwrites it
        if (logical >= hi) { ++probe; continue; }
        int tm, tn;
        decode_tile<MODE>(logical, ntm, nc, sw, tm, tn);
        gemm_nt_tile<WM, WN, WT, MODE, false, 0, tile_bk(WM, WN, WT)>(tm, tn, M, N, K, A, lda, B, ldb, C, ldc, nullptr,
                                                                          nullptr, 0, smem_raw);
        __syncthreads();  // LDS tiles are reused by the next round
    }
}

// The same product for a BATCH of independent triples (A, B, C) of one shape in one launch: blockIdx.y is the batch index,
// the pointers come from a record array in device memory (the Schur stage's per-scenario sweeps: schur.hip).
template <int WM, int WN, int WT, int MODE>
__global__ __launch_bounds__(64 * WM * WN, tile_occ(WM, WN, WT)) void gemm_nt_batch_kernel(
    int64_t M, int64_t N, int64_t K, const GemmBatchRec* __restrict__ recs, int64_t lda, int64_t ldb, int64_t ldc, int ntm,
    int nc, int sw) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const GemmBatchRec rec = recs[blockIdx.y];
    if (rec.info != nullptr && *rec.info != 0) return;
    int tm, tn;
    decode_tile<MODE>((int)blockIdx.x, ntm, nc, sw, tm, tn);
    gemm_nt_tile<WM, WN, WT, MODE, false, 0, tile_bk(WM, WN, WT)>(tm, tn, M, N, K, rec.A, lda, rec.B, ldb, rec.C, ldc, nullptr,
                                                                      nullptr, 0, smem_raw);
}

template <int WM, int WN, int WT>
static int launch_batch_t(hipStream_t s, int64_t M, int64_t N, int64_t K, const GemmBatchRec* recs, int nbatch, int64_t lda,
                          int64_t ldb, int64_t ldc) {
    constexpr int BM = 16 * WT * WM, BN = 16 * WT * WN;
    const int ntm = (int)((M + BM - 1) / BM), ntn = (int)((N + BN - 1) / BN);
    const size_t smem = 2 * tile_bk(WM, WN, WT) * ((BM + 16) + (BN + 16)) * sizeof(double);
    auto kern = gemm_nt_batch_kernel<WM, WN, WT, 0>;
    static std::atomic<uint64_t> attr_devs{0};
    int dev = 0;
    MNK_HIP(hipGetDevice(&dev));
    if (!(attr_devs.load(std::memory_order_relaxed) >> (dev & 63) & 1)) {
        MNK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_devs.fetch_or(1ull << (dev & 63), std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(kern, dim3(ntm * ntn, nbatch), dim3(64 * WM * WN), smem, s, M, N, K, recs, lda, ldb, ldc, ntm,
                       ntn < ntm ? ntn : ntm, 1);
    MNK_HIP(hipGetLastError());
    return 0;
}

// C_i -= A_i B_i^T for i < nbatch (mode 0 of launch_gemm_nt; the same tile shapes)
int launch_gemm_nt_batch(hipStream_t s, int64_t M, int64_t N, int64_t K, const GemmBatchRec* recs, int nbatch, int64_t lda,
                         int64_t ldb, int64_t ldc) {
    if (M <= 0 || N <= 0 || nbatch <= 0) return 0;
    MNK_REQUIRE(M % 64 == 0 && N % 64 == 0 && K % BK == 0 && K > 0, "gemm_nt_batch: M,N must be multiples of 64, K of 16");
    if (N <= 64) return launch_batch_t<4, 1, 4>(s, M, N, K, recs, nbatch, lda, ldb, ldc);
    return launch_batch_t<2, 2, 4>(s, M, N, K, recs, nbatch, lda, ldb, ldc);
}

// width of the super-columns of the lower-tile enumeration (MNK_SUPER_W overrides; 1 = column-by-column)
static int tile_super_width() {
    static const int sw = getenv("MNK_SUPER_W") ? std::max(1, atoi(getenv("MNK_SUPER_W"))) : 8;
    return sw;
}

template <int WM, int WN, int WT, int MODE, bool LDL_EPI>
static int launch_t(hipStream_t s, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda,
                    const double* B, int64_t ldb, double* C, int64_t ldc, const double* colscale,
                    double* C2, int64_t ldc2, const int* info_flag, int tile_begin = 0, int tile_count = -1) {
    constexpr int BM = 16 * WT * WM, BN = 16 * WT * WN;
    const int ntm = (int)((M + BM - 1) / BM), ntn = (int)((N + BN - 1) / BN);
    int ntiles = ntm * ntn;
    const int nc = ntn < ntm ? ntn : ntm;  // tile columns that contain a lower tile
    if (MODE == 2 || MODE == 4) {
        static_assert(MODE == 0 || MODE == 1 || WM == WN, "lower-only modes need square tiles");
        ntiles = nc * ntm - nc * (nc - 1) / 2;
    }
    const size_t smem = 2 * tile_bk(WM, WN, WT) * ((BM + 16) + (BN + 16)) * sizeof(double);
    auto kern = gemm_nt_kernel<WM, WN, WT, MODE, LDL_EPI>;
    // the attribute belongs to the (kernel, device) pair: set it once per device of this process
    static std::atomic<uint64_t> attr_devs{0};
    int dev = 0;
    MNK_HIP(hipGetDevice(&dev));
    if (!(attr_devs.load(std::memory_order_relaxed) >> (dev & 63) & 1)) {
        MNK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_devs.fetch_or(1ull << (dev & 63), std::memory_order_relaxed);
    }
    if (tile_count >= 0) ntiles = std::min(ntiles - tile_begin, tile_count);
    if (ntiles <= 0) return 0;
    hipLaunchKernelGGL(kern, dim3(ntiles), dim3(64 * WM * WN), smem, s, M, N, K, A, lda, B, ldb, C,
                       ldc, colscale, C2, ldc2, ntm, info_flag, tile_begin, nc, tile_super_width());
    MNK_HIP(hipGetLastError());
    return 0;
}

int launch_gemm_nt(hipStream_t s, int mode, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda,
                   const double* B, int64_t ldb, double* C, int64_t ldc, const double* colscale,
                   double* C2, int64_t ldc2, const int* info_flag) {
    if (M <= 0 || N <= 0) return 0;
    MNK_REQUIRE(M % 64 == 0 && N % 64 == 0 && K % BK == 0 && K > 0, "gemm_nt: M,N must be multiples of 64, K of 16");
    if (mode == 1) {
        // panel solve shape: N is one or two inner blocks wide; use a tall 256 x 64 tile
        if (N <= 64) {
            if (colscale)
                return launch_t<4, 1, 4, 1, true>(s, M, N, K, A, lda, B, ldb, C, ldc, colscale, C2, ldc2, info_flag);
            return launch_t<4, 1, 4, 1, false>(s, M, N, K, A, lda, B, ldb, C, ldc, nullptr, nullptr, 0, info_flag);
        }
        if (colscale)
            return launch_t<2, 2, 4, 1, true>(s, M, N, K, A, lda, B, ldb, C, ldc, colscale, C2, ldc2, info_flag);
        return launch_t<2, 2, 4, 1, false>(s, M, N, K, A, lda, B, ldb, C, ldc, nullptr, nullptr, 0, info_flag);
    }
    if (mode == 0) {
        // skinny left-looking updates: 64x64 workgroup tiles of 4 waves x (32x32) keep the per-wave
        // MFMA chain short and put >= M/64 workgroups on the chip (latency-bound shape)
        if (N <= 64 && M >= 1024)
            return launch_t<2, 2, 2, 0, false>(s, M, N, K, A, lda, B, ldb, C, ldc, nullptr, nullptr, 0, info_flag);
        if (N <= 64) return launch_t<4, 1, 4, 0, false>(s, M, N, K, A, lda, B, ldb, C, ldc, nullptr, nullptr, 0, info_flag);
        return launch_t<2, 2, 4, 0, false>(s, M, N, K, A, lda, B, ldb, C, ldc, nullptr, nullptr, 0, info_flag);
    }
    if (mode == 2) {
        return launch_t<2, 2, 4, 2, false>(s, M, N, K, A, lda, B, ldb, C, ldc, nullptr, nullptr, 0, info_flag);
    }
    if (mode == 4) {
        // (the Gram product of the dense condensed KKT system: at n = 2048 the lower triangle is 136 tiles of 128 x 128 -- half
        // the CUs idle, 74 us for 2.1 GFLOP; 528 tiles of 64 x 64 fill the chip.  Same bits: the order of a C entry's sum is the K-loop's.)
        const int64_t mt = (std::min(M, N) + 127) / 128;
        if (mt * (mt + 1) / 2 < 512 && M % 64 == 0 && N % 64 == 0)
            return launch_t<2, 2, 2, 4, false>(s, M, N, K, A, lda, B, ldb, C, ldc, nullptr, nullptr, 0, info_flag);
        return launch_t<2, 2, 4, 4, false>(s, M, N, K, A, lda, B, ldb, C, ldc, nullptr, nullptr, 0, info_flag);
    }
    set_error("gemm_nt: bad mode %d", mode);
    return -1;
}

// C(lower tiles) -= A * B^T with 64x64 workgroup tiles: four times the workgroups of the default
// 128x128 tiling, for updates with too few tiles to fill the chip (latency-bound tail of the
// factorization).
int launch_gemm_nt_lower_small(hipStream_t s, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda,
                               const double* B, int64_t ldb, double* C, int64_t ldc, const int* info_flag) {
    if (M <= 0 || N <= 0) return 0;
    MNK_REQUIRE(M % 64 == 0 && N % 64 == 0 && K % BK == 0 && K > 0, "gemm_nt: M,N must be multiples of 64, K of 16");
    return launch_t<2, 2, 2, 2, false>(s, M, N, K, A, lda, B, ldb, C, ldc, nullptr, nullptr, 0, info_flag);
}

// a contiguous range of the lower tiles of the 128x128 tiling (mode 2), for chunked launches
int launch_gemm_nt_lower_range(hipStream_t s, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda,
                               const double* B, int64_t ldb, double* C, int64_t ldc, const int* info_flag,
                               int tile_begin, int tile_count) {
    if (M <= 0 || N <= 0) return 0;
    MNK_REQUIRE(M % 64 == 0 && N % 64 == 0 && K % BK == 0 && K > 0, "gemm_nt: M,N must be multiples of 64, K of 16");
    return launch_t<2, 2, 4, 2, false>(s, M, N, K, A, lda, B, ldb, C, ldc, nullptr, nullptr, 0, info_flag, tile_begin,
                                       tile_count);
}

int gemm_nt_lower_tiles(int64_t M, int64_t N);
int launch_gemm_nt_dbg(hipStream_t s, int shared_ab, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda,
                       const double* B, int64_t ldb, double* C, int64_t ldc) {
    const int ntm = (int)((M + 127) / 128);
    const int ntiles = gemm_nt_lower_tiles(M, N);
    const size_t smem = 2 * tile_bk(2, 2, 4) * ((128 + 16) + (128 + 16)) * sizeof(double);
#define MNK_DBG_LAUNCH(D)                                                                                        \
    do {                                                                                                         \
        auto kern = gemm_nt_dbg_kernel<2, 2, 4, 2, D>;                                                           \
        const size_t smem_d = D == 6 ? smem * 2 : smem; /* variant 6 runs BK = 16 */                             \
        MNK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_d)); \
        hipLaunchKernelGGL(kern, dim3(ntiles), dim3(256), smem_d, s, M, N, K, A, lda, B, ldb, C, ldc, ntm);      \
    } while (0)
    if (shared_ab == 1) MNK_DBG_LAUNCH(1);
    else if (shared_ab == 2) MNK_DBG_LAUNCH(2);
    else if (shared_ab == 3) MNK_DBG_LAUNCH(3);
    else if (shared_ab == 4) MNK_DBG_LAUNCH(4);
    else if (shared_ab == 5) MNK_DBG_LAUNCH(5);
    else if (shared_ab == 6) MNK_DBG_LAUNCH(6);  // BK = 16, two workgroups per CU (the earlier configuration)
    else MNK_DBG_LAUNCH(0);
#undef MNK_DBG_LAUNCH
    MNK_HIP(hipGetLastError());
    return 0;
}

int gemm_nt_lower_tiles(int64_t M, int64_t N) {
    const int ntm = (int)((M + 127) / 128), ntn = (int)((N + 127) / 128);
    const int nc = ntn < ntm ? ntn : ntm;
    return nc * ntm - nc * (nc - 1) / 2;
}

// Work-queue launch of the lower-tile update (mode 2, 128x128 tiles): as many workgroups as fit on `cus`
// CUs pull tiles from counter[0..7] (zeroed by the caller before the first launch that shares them).
int launch_gemm_nt_queue(hipStream_t s, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda,
                         const double* B, int64_t ldb, double* C, int64_t ldc, int* counter, int cus,
                         const int* info_flag) {
    int nwg = cus * tile_occ(2, 2, 4);
    if (M <= 0 || N <= 0) return 0;
    MNK_REQUIRE(M % 64 == 0 && N % 64 == 0 && K % BK == 0 && K > 0, "gemm_nt: M,N must be multiples of 64, K of 16");
    constexpr int BM = 128;
    const int ntm = (int)((M + BM - 1) / BM);
    const int ntiles = gemm_nt_lower_tiles(M, N);
    if (nwg > ntiles) nwg = ntiles;
    const size_t smem = 2 * tile_bk(2, 2, 4) * ((BM + 16) + (BM + 16)) * sizeof(double);
    auto kern = gemm_nt_queue_kernel<2, 2, 4, 2>;
    // the attribute belongs to the (kernel, device) pair: set it once per device of this process
    static std::atomic<uint64_t> attr_devs{0};
    int dev = 0;
    MNK_HIP(hipGetDevice(&dev));
    if (!(attr_devs.load(std::memory_order_relaxed) >> (dev & 63) & 1)) {
        MNK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_devs.fetch_or(1ull << (dev & 63), std::memory_order_relaxed);
    }
    const int ntn_q = (int)((N + BM - 1) / BM);
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), smem, s, M, N, K, A, lda, B, ldb, C, ldc, ntm, ntiles, counter,
                       info_flag, ntn_q < ntm ? ntn_q : ntm, tile_super_width());
    MNK_HIP(hipGetLastError());
    return 0;
}

// Shader clock under fp64 MFMA load: every wave of a whole-chip launch issues v_mfma_f64_16x16x4 from registers and compares the
// shader-cycle counter (s_memtime) with the 100 MHz wall counter (s_memrealtime) over the same stretch.
__global__ __launch_bounds__(256) void shader_clock_kernel(unsigned long long* __restrict__ out, int iters, double seed) {
    v4f64 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = v4f64{0, 0, 0, 0};
    double a = seed * (1.0 + threadIdx.x * 0.37), b = seed * (0.7 - threadIdx.x * 0.011);
    const unsigned long long w0 = wall_clock64(), c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        a = -a;
    }
    const unsigned long long w1 = wall_clock64(), c1 = __builtin_readcyclecounter();
    double sm = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) sm += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * 4 + threadIdx.x / 64;
        out[2 * w] = w1 - w0;
        out[2 * w + 1] = (c1 - c0) + (sm == 12345.678 ? 1 : 0);
    }
}

}  // namespace mnk

// Diagnostics (bench.py `clocks`): the shader clock the chip sustains under fp64 MFMA load right now, in MHz -- ~5 ms of every CU
// on the context's stream (the sysfs / rocm-smi readings show the idle state between two steps: 95 MHz on a box that runs 2.4 GHz).
extern "C" int mnk_debug_shader_clock(mnk_ctx* ctx, double* mhz) {
    MNK_REQUIRE(ctx && mhz, "mnk_debug_shader_clock: NULL argument");
    MNK_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int wgs = 3 * ctx->num_cu, iters = 6000;   // (~2.5 ms per launch; the first launch takes the clock's ramp, the second is read)
    mnk::DevBuf<unsigned long long> out;
    if (out.alloc((size_t)wgs * 8)) return -2;
    for (int rep = 0; rep < 2; ++rep)
        hipLaunchKernelGGL(mnk::shader_clock_kernel, dim3(wgs), dim3(256), 0, s, out.p, iters, 1.2345678901234567);
    MNK_HIP(hipGetLastError());
    std::vector<unsigned long long> h((size_t)wgs * 8);
    MNK_HIP(mnk::d2h_copy(h.data(), out.p, h.size() * sizeof(unsigned long long), s));
    MNK_HIP(mnk::stream_wait(s));
    double wsum = 0.0, csum = 0.0;
    for (size_t i = 0; i < h.size(); i += 2) { wsum += (double)h[i]; csum += (double)h[i + 1]; }
    *mhz = wsum > 0.0 ? csum / wsum * 100.0 : 0.0;
    return 0;
}
