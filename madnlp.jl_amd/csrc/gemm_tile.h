// fp64 MFMA tile of C (MxN) op= A (MxK) * B (NxK)^T, column-major: the device-side pieces shared by the tile kernels of
// gemm_f64.hip and the persistent left-looking tile kernel of dag.hip.  See gemm_f64.hip for the gfx950 mapping.
#pragma once
#include "common.h"

namespace mnk {

typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef double v2f64 __attribute__((ext_vector_type(2)));

constexpr int BK = 16;
// k-tile depth and workgroups per CU of a tile shape.  The 128x128 tile (4 waves x 64x64) runs BK = 8:
// 36.9 KB of LDS and <= 168 VGPRs let three workgroups share a CU (three waves per SIMD hide the
// barrier / LDS-DMA stalls of one another): measured +4 % over BK = 16 with two workgroups, same sums.
constexpr bool tile_big(int wm, int wn, int wt) { return wm == 2 && wn == 2 && wt == 4; }
constexpr int tile_bk(int wm, int wn, int wt) { return tile_big(wm, wn, wt) ? 8 : BK; }
constexpr int tile_occ(int wm, int wn, int wt) { return tile_big(wm, wn, wt) ? 3 : 2; }

// Decode a logical tile index into (tm, tn).  Lower-only modes enumerate the tiles on/below the diagonal in
// SUPER-COLUMNS of `sw` tile columns, row by row inside a super-column: a window of consecutive logical tiles
// (what one XCD's workgroups hold at a time: ~72 tiles of the 128x128 kernel) then spans ~72/sw tile rows x sw tile
// columns and shares ~72/sw + sw operand blocks through that XCD's L2, instead of 73 with a column-by-column order
// (sw = 1, the round-1 order).  Tiles before tile column c: first(c) = c*ntm - c*(c-1)/2.
template <int MODE>
__device__ __forceinline__ void decode_tile(int logical, int ntm, int nc, int sw, int& tm, int& tn) {
    if (MODE == 2 || MODE == 4) {
        const double bq = 2.0 * ntm + 1.0;
        int c = (int)((bq - sqrt(bq * bq - 8.0 * (double)logical)) * 0.5);
        if (c < 0) c = 0;
        while (c > 0 && c * ntm - c * (c - 1) / 2 > logical) --c;
        while ((c + 1) * ntm - (c + 1) * c / 2 <= logical) ++c;
        if (sw <= 1) {
            tn = c;
            tm = tn + (logical - (c * ntm - c * (c - 1) / 2));
            return;
        }
        const int r0 = (c / sw) * sw;                       // first tile row/column of the super-column
        const int W = nc - r0 < sw ? nc - r0 : sw;           // its width
        int rem = logical - (r0 * ntm - r0 * (r0 - 1) / 2);  // index inside the super-column, row-major
        const int tri = W * (W + 1) / 2;                     // its triangular top: row r0 + i holds i + 1 tiles
        if (rem < tri) {
            int i = (int)((sqrt(8.0 * (double)rem + 1.0) - 1.0) * 0.5);
            while (i > 0 && i * (i + 1) / 2 > rem) --i;
            while ((i + 1) * (i + 2) / 2 <= rem) ++i;
            tm = r0 + i;
            tn = r0 + (rem - i * (i + 1) / 2);
        } else {
            rem -= tri;
            tm = r0 + W + rem / W;
            tn = r0 + rem % W;
        }
    } else {
        tm = logical % ntm;
        tn = logical / ntm;
    }
}

// K-loop of one workgroup tile: acc += (rows of A) x (rows of B)^T over `nk` k-tiles of depth BKT, starting at the
// columns Ag / Bg point to (Ag, Bg already offset to the tile's first row).  acc[ni][mi] register r of lane (l15, l4) of
// wave (wm, wn) is C^T: row wm*WS + mi*16 + l15, column wn*WS + ni*16 + l4 + 4r.  Every thread of the workgroup must
// call it with the same arguments; it ends with a workgroup barrier (the LDS tiles are free on return).
// GATE: `gate(kt)` is called (by every thread, uniformly) before the loads of k-tile kt are issued and may block until
// the operands of that k-tile exist (dag.hip: the factor's columns become final while the loop runs); false = give up.
struct GemmNoGate {
    __device__ __forceinline__ bool operator()(int) const { return true; }
};
template <int WM, int WN, int WT, int DBG, int BKT, class GATE = GemmNoGate>
__device__ __forceinline__ bool gemm_nt_mainloop(v4f64 (&acc)[WT][WT], const double* __restrict__ Ag, int64_t lda,
                                                 const double* __restrict__ Bg, int64_t ldb, int nk, char* smem_raw,
                                                 int tid, GATE gate = GATE()) {
    constexpr int NT = 64 * WM * WN;
    constexpr int WS = 16 * WT;  // wave tile edge (WT x WT MFMA 16x16 tiles per wave)
    constexpr int BM = WS * WM, BN = WS * WN;
    constexpr int LDA_S = BM + 16, LDB_S = BN + 16;
    constexpr int APIECES = (BM / 2) * BKT / NT;  // 16-byte pieces per thread per k-tile
    constexpr int BPIECES = (BN / 2) * BKT / NT;
    static_assert(APIECES >= 1 && BPIECES >= 1, "tile too small for the thread count");

    double* As = reinterpret_cast<double*>(smem_raw);           // [2][BKT][LDA_S]
    double* Bs = As + 2 * BKT * LDA_S;                           // [2][BKT][LDB_S]

    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int l15 = lane & 15, l4 = lane >> 4;

    v2f64 ra[APIECES], rb[BPIECES];

    auto gload = [&](int kt) {
        const int64_t k0 = (int64_t)kt * BKT;
#pragma unroll
        for (int q = 0; q < APIECES; ++q) {
            int p = tid + NT * q;
            int k = p / (BM / 2), r = (p % (BM / 2)) * 2;
            ra[q] = *reinterpret_cast<const v2f64*>(Ag + r + (k0 + k) * lda);
        }
#pragma unroll
        for (int q = 0; q < BPIECES; ++q) {
            int p = tid + NT * q;
            int k = p / (BN / 2), r = (p % (BN / 2)) * 2;
            rb[q] = *reinterpret_cast<const v2f64*>(Bg + r + (k0 + k) * ldb);
        }
    };
    auto sstore = [&](int buf) {
        double* as = As + buf * BKT * LDA_S;
        double* bs = Bs + buf * BKT * LDB_S;
#pragma unroll
        for (int q = 0; q < APIECES; ++q) {
            int p = tid + NT * q;
            int k = p / (BM / 2), r = (p % (BM / 2)) * 2;
            *reinterpret_cast<v2f64*>(as + k * LDA_S + r) = ra[q];
        }
#pragma unroll
        for (int q = 0; q < BPIECES; ++q) {
            int p = tid + NT * q;
            int k = p / (BN / 2), r = (p % (BN / 2)) * 2;
            *reinterpret_cast<v2f64*>(bs + k * LDB_S + r) = rb[q];
        }
    };

    // 128x128 tiles stage through the LDS-DMA path (global_load_lds_dwordx4: one wave instruction moves one
    // 128-row k-column, 1 KiB, straight into its LDS row; no staging registers, no ds_write pass):
    // measured +3 % (192 CUs) / +5 % (256 CUs) on the trailing update, bit-identical results.
    // DBG == 5 (diagnostics) forces the register-staged path for A/B runs.
    constexpr bool DMA = BM == 128 && BN == 128 && (DBG == 0 || DBG == 1 || DBG == 4);
    constexpr int NW = NT / 64;
    auto gl_lds = [&](int kt, int buf) {
        const int64_t k0 = (int64_t)kt * BKT;
        double* as = As + buf * BKT * LDA_S;
        double* bs = Bs + buf * BKT * LDB_S;
#pragma unroll
        for (int i = 0; i < BKT / NW; ++i) {
            const int k = wave * (BKT / NW) + i;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Ag + (k0 + k) * lda + lane * 2),
                                             (__attribute__((address_space(3))) void*)(as + k * LDA_S), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Bg + (k0 + k) * ldb + lane * 2),
                                             (__attribute__((address_space(3))) void*)(bs + k * LDB_S), 16, 0, 0);
        }
    };
    if (nk <= 0) return true;
    if (!gate(0)) return false;
    if (DMA) {
        gl_lds(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        gload(0);
        sstore(0);
    }
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk && !gate(kt + 1)) return false;
        if (DMA) {
            if (kt + 1 < nk) gl_lds(kt + 1, cur ^ 1);
        } else if (DBG != 3 && kt + 1 < nk) {
            gload(kt + 1);
        }
        const double* as = As + cur * BKT * LDA_S + wm * WS + l15;
        const double* bs = Bs + cur * BKT * LDB_S + wn * WS + l15;
#pragma unroll
        for (int kk = 0; kk < BKT / 4; ++kk) {
            double af[WT], bf[WT];
#pragma unroll
            for (int i = 0; i < WT; ++i) {
                af[i] = as[(kk * 4 + l4) * LDA_S + i * 16];
                bf[i] = bs[(kk * 4 + l4) * LDB_S + i * 16];
            }
#pragma unroll
            for (int ni = 0; ni < WT; ++ni)
#pragma unroll
                for (int mi = 0; mi < WT; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[ni], af[mi], acc[ni][mi], 0, 0, 0);
        }
        if (DMA) {
            // the DMA writes are ordered for the readers by this wave's vmcnt followed by the barrier
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        } else {
            if (DBG != 3 && kt + 1 < nk) sstore(cur ^ 1);
            if (DBG != 2 && DBG != 3) __syncthreads();
        }
    }
    return true;
}

// The same K-loop for the 128x128 tile (4 waves x 64x64, k-tiles of 8) with THREE LDS buffers: the LDS-DMA loads of k-tile
// kt + 2 are issued before k-tile kt is multiplied, so two k-tiles are in flight instead of one.  A workgroup that has its
// CU for itself (the tile-closing tasks of dag.hip at the end of a factorization: the rows' chains are serial) is bound by
// the load latency of one k-tile per iteration with two buffers: 16 k-tiles x 2.6 us = 41.7 us per 128-column step
// (measured, r03 traces) against 0.85 us of MFMA work per k-tile.
// LDS layout: row k of a buffer sits at k * 128 + 16 * ((k + 1) / 2) doubles -- a pad of 16 in front of every ODD row only:
// the rows (2j, 2j + 1) a half-wave reads together are 144 doubles apart (bank-conflict free, like the layout of the
// two-buffer loop that pads every row), 3 x 2 x 1088 doubles = 52 224 B: three workgroups per CU still fit the 160 KB.
constexpr int TILE3_ROWS = 8 * 128 + 16 * 4;             // doubles per operand per buffer
constexpr int TILE3_LDS_BYTES = 3 * 2 * TILE3_ROWS * 8;  // 52 224
__device__ __forceinline__ constexpr int tile3_row(int k) { return k * 128 + 16 * ((k + 1) >> 1); }
// Wave tile: MI x NI blocks of 16x16 (MI * NI = 16): 4 x 4 = the 64x64 wave tiles of the two-buffer loop (2 x 2 waves);
// 2 x 8 = every wave owns 32 full rows of the tile (4 x 1 waves) -- the register layout of dag.hip's finalization, so a
// tile-closing task runs its substitutions on the accumulators without a round trip through memory.
// NEG: acc -= A B^T (the A fragments are negated on their way to the MFMA).
template <int MI = 4, int NI = 4, bool NEG = false, class GATE = GemmNoGate>
__device__ __forceinline__ bool gemm_nt_mainloop3(v4f64 (&acc)[NI][MI], const double* __restrict__ Ag, int64_t lda,
                                                  const double* __restrict__ Bg, int64_t ldb, int nk, char* smem_raw, int tid,
                                                  GATE gate = GATE()) {
    static_assert(MI * NI == 16 && 128 % (16 * MI) == 0, "four waves cover the 128x128 tile");
    constexpr int BKT = 8, WSM = 16 * MI, WSN = 16 * NI, WAVES_M = 128 / WSM;
    double* Ls = reinterpret_cast<double*>(smem_raw);  // [3][2][TILE3_ROWS]
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
    const int l15 = lane & 15, l4 = lane >> 4;
    auto gl_lds = [&](int kt, int buf) {
        const int64_t k0 = (int64_t)kt * BKT;
        double* as = Ls + buf * 2 * TILE3_ROWS;
        double* bs = as + TILE3_ROWS;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = wave * 2 + i;
            const int off = tile3_row(k);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Ag + (k0 + k) * lda + lane * 2),
                                             (__attribute__((address_space(3))) void*)(as + off), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Bg + (k0 + k) * ldb + lane * 2),
                                             (__attribute__((address_space(3))) void*)(bs + off), 16, 0, 0);
        }
    };
    if (nk <= 0) return true;
    if (!gate(0)) return false;
    gl_lds(0, 0);
    if (nk > 1) {
        if (!gate(1)) return false;
        gl_lds(1, 1);
        asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const int nxt2 = cur == 0 ? 2 : cur - 1;  // (kt + 2) % 3: the buffer the barrier of iteration kt - 1 released
        const bool more = kt + 2 < nk;
        if (more) {
            if (!gate(kt + 2)) return false;
            gl_lds(kt + 2, nxt2);
        }
        const double* as = Ls + cur * 2 * TILE3_ROWS + wm * WSM + l15 + tile3_row(l4);
        const double* bs = Ls + cur * 2 * TILE3_ROWS + TILE3_ROWS + wn * WSN + l15 + tile3_row(l4);
#pragma unroll
        for (int kk = 0; kk < BKT / 4; ++kk) {
            double af[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {   // row kk * 4 + l4: tile3_row(kk * 4 + l4) = kk * 544 + tile3_row(l4)
                af[i] = as[kk * (4 * 128 + 32) + i * 16];
                if (NEG) af[i] = -af[i];
            }
#pragma unroll
            for (int n0 = 0; n0 < NI; n0 += 4) {   // (four B fragments at a time: the 2 x 8 shape has no registers to spare)
                double bf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) bf[i] = bs[kk * (4 * 128 + 32) + (n0 + i) * 16];
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        acc[n0 + ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[ni], af[mi], acc[n0 + ni][mi], 0, 0, 0);
                if (NI > 4) __builtin_amdgcn_sched_barrier(0);  // (keeps the scheduler from hoisting the next fragments: registers)
            }
        }
        // k-tile kt + 1 must have landed (this wave's share: its vmcnt; the others': the barrier); kt + 2 stays in flight.
        // A bare s_barrier: __syncthreads() carries a workgroup fence that the compiler implements as vmcnt(0), which would
        // drain the loads of k-tile kt + 2 here.  The LDS reads of this iteration were consumed by its MFMAs (lgkmcnt waits
        // in front of them), so the buffer may be overwritten once every wave has passed the barrier.
        if (more)
            asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        cur = cur == 2 ? 0 : cur + 1;
    }
    __syncthreads();  // (callers reuse the LDS right away)
    return true;
}

// Epilogue of one workgroup tile: lane (l15, l4), reg r of acc[ni][mi] is C[row0+wm*64+mi*16+l15, col0+wn*64+ni*16+l4+4r]
// BATCH (MODE 2 / 4): all 16 loads of a column block are issued before its 16 stores -- four memory round trips per tile
// instead of sixteen (the tasks of dag.hip are 1 .. 12 k-steps long and end with this read-modify-write: 21 us of ~230).
template <int WM, int WN, int WT, int MODE, bool LDL_EPI, bool BATCH = false>
__device__ __forceinline__ void gemm_nt_epilogue(const v4f64 (&acc)[WT][WT], int64_t row0, int64_t col0, int64_t M, int64_t N,
                                                 double* C, int64_t ldc, const double* __restrict__ colscale, double* C2,
                                                 int64_t ldc2, int tid) {
    constexpr int WS = 16 * WT;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t wrow = row0 + wm * WS, wcol = col0 + wn * WS;
    if (wrow >= M || wcol >= N) return;
    if ((MODE == 2 || MODE == 4) && wrow + WS <= wcol) return;  // wave tile entirely above the diagonal
    // (Measured on gfx950: replacing this load/subtract/store sequence by no-return L2 atomics or by
    // batching all loads ahead of the stores does not change the kernel time -- with two workgroups
    // per CU the epilogue of one hides behind the k-loop of the other.)
    if (BATCH && (MODE == 2 || MODE == 4)) {
#pragma unroll
        for (int ni = 0; ni < WT; ++ni) {
            double cv[4][WT];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double* cp = C + (wcol + ni * 16 + l4 + 4 * r) * ldc + wrow + l15;
#pragma unroll
                for (int mi = 0; mi < WT; ++mi) cv[r][mi] = cp[mi * 16];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double* cp = C + (wcol + ni * 16 + l4 + 4 * r) * ldc + wrow + l15;
#pragma unroll
                for (int mi = 0; mi < WT; ++mi) cp[mi * 16] = MODE == 4 ? cv[r][mi] + acc[ni][mi][r] : cv[r][mi] - acc[ni][mi][r];
            }
        }
        return;
    }
#pragma unroll
    for (int ni = 0; ni < WT; ++ni) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t n = wcol + ni * 16 + l4 + 4 * r;
            double* cp = C + n * ldc + wrow + l15;
            if (MODE == 1) {
                if (LDL_EPI) {
                    const double sc = colscale[n];
                    double* c2p = C2 + n * ldc2 + wrow + l15;
#pragma unroll
                    for (int mi = 0; mi < WT; ++mi) {
                        c2p[mi * 16] = acc[ni][mi][r];
                        cp[mi * 16] = acc[ni][mi][r] * sc;
                    }
                } else {
#pragma unroll
                    for (int mi = 0; mi < WT; ++mi) cp[mi * 16] = acc[ni][mi][r];
                }
            } else {
                double cv[WT];
#pragma unroll
                for (int mi = 0; mi < WT; ++mi) cv[mi] = cp[mi * 16];
#pragma unroll
                for (int mi = 0; mi < WT; ++mi) cp[mi * 16] = MODE == 4 ? cv[mi] + acc[ni][mi][r] : cv[mi] - acc[ni][mi][r];
            }
        }
    }
}

// A diagonal tile accumulated in SUBTRACT order: acc = -C before the K-loop, C = -acc after it (the lower wave tiles; the
// register layout of gemm_nt_epilogue).  The partial results are then C - (the products so far), which shrink as the
// cancellation proceeds, instead of a sum that grows from zero to |C| and is subtracted at the end: a pivot that is the
// difference of two numbers of size 1e13 agreeing to 15 digits keeps its sign (dag.hip).
template <int WM, int WN, int WT>
__device__ __forceinline__ void gemm_nt_load_neg_lower(v4f64 (&acc)[WT][WT], int64_t row0, int64_t col0, const double* C, int64_t ldc, int tid) {
    constexpr int WS = 16 * WT;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t wrow = row0 + wm * WS, wcol = col0 + wn * WS;
    if (wrow + WS <= wcol) return;
#pragma unroll
    for (int ni = 0; ni < WT; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double* cp = C + (wcol + ni * 16 + l4 + 4 * r) * ldc + wrow + l15;
#pragma unroll
            for (int mi = 0; mi < WT; ++mi) acc[ni][mi][r] = -cp[mi * 16];
        }
}
template <int WM, int WN, int WT>
__device__ __forceinline__ void gemm_nt_store_neg_lower(const v4f64 (&acc)[WT][WT], int64_t row0, int64_t col0, double* C, int64_t ldc, int tid) {
    constexpr int WS = 16 * WT;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t wrow = row0 + wm * WS, wcol = col0 + wn * WS;
    if (wrow + WS <= wcol) return;
#pragma unroll
    for (int ni = 0; ni < WT; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double* cp = C + (wcol + ni * 16 + l4 + 4 * r) * ldc + wrow + l15;
#pragma unroll
            for (int mi = 0; mi < WT; ++mi) cp[mi * 16] = -acc[ni][mi][r];
        }
}

// One workgroup tile of C.  Every thread of the workgroup must call it with the same (tm, tn).
template <int WM, int WN, int WT, int MODE, bool LDL_EPI, int DBG = 0, int BKT = BK>
__device__ __forceinline__ void gemm_nt_tile(
    int tm, int tn, int64_t M, int64_t N, int64_t K, const double* __restrict__ A, int64_t lda,
    const double* __restrict__ B, int64_t ldb, double* C, int64_t ldc,
    const double* __restrict__ colscale, double* C2, int64_t ldc2, char* smem_raw) {
    constexpr int WS = 16 * WT;
    constexpr int BM = WS * WM, BN = WS * WN;
    const int64_t row0 = (int64_t)tm * BM;
    const int64_t col0 = (int64_t)tn * BN;
    int tid = threadIdx.x;
    // Opaque to the optimizer: inside the work-queue loop everything derived from the thread id would be
    // hoisted out of the loop and then spilled (168-VGPR budget); recomputing it per tile costs a few VALU ops.
    asm volatile("" : "+v"(tid));
    // DBG (diagnostics only, results are wrong for DBG >= 2): 1 = every tile reads the same two operand
    // blocks (perfect cache reuse: separates the memory-side cost from the MFMA/LDS cost); 2 = no barrier in
    // the k-loop; 3 = no global loads / LDS stores in the k-loop (MFMA + LDS-read loop alone)
    const double* Ag = A + (DBG == 1 ? (int64_t)(tm & 1) * BM : row0);
    const double* Bg = B + (DBG == 1 ? (int64_t)(tn & 1) * BN : col0);
    v4f64 acc[WT][WT];  // [ni][mi]
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j) acc[i][j] = v4f64{0.0, 0.0, 0.0, 0.0};
    (void)gemm_nt_mainloop<WM, WN, WT, DBG, BKT>(acc, Ag, lda, Bg, ldb, (int)(K / BKT), smem_raw, tid);
    gemm_nt_epilogue<WM, WN, WT, MODE, LDL_EPI>(acc, row0, col0, M, N, C, ldc, colscale, C2, ldc2, tid);
}

}  // namespace mnk
