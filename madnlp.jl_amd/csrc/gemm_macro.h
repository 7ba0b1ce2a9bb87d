// K-loop of a MACRO tile: RT vertically adjacent 128 x 128 tiles of one tile column, C(I0 .. I0 + RT - 1, J) += A B^T over the
// same k-range, computed by ONE workgroup of 4 RT waves (wave tile 64 x 64, as in gemm_tile.h) that stages the B rows once for
// all RT tiles: (RT + 1) operand streams for RT tiles instead of 2 RT -- 3/4 (RT = 2) or 2/3 (RT = 3) of the operand bytes
// per flop of the 128 x 128 kernel.  Why: the left-looking bulk kernel of dag.hip streams its operands from beyond the L2s
// (profiles/r04_pmc_dag_C3.md: 27.6 GB per factorization at N = 11 192, 31.7 x the algorithmic traffic); at the MFMA-bound
// rate the 128 x 128 tiles of the whole chip ask for ~4.4 TB/s, which is what HBM delivers to that access pattern -- the
// K-loop ran at 0.88-0.92 of its MFMA bound.  Fewer bytes per flop is the lever; an L2 cannot help (4 MB per XCD turn over in
// ~8 us against tasks of hundreds of microseconds that start at unrelated times).
//
// One workgroup per CU (LDS: NS stages of 8 k-columns, 26 / 34 KB each), so nothing else hides this workgroup's stalls: the
// LDS-DMA loads (global_load_lds_dwordx4, one wave instruction = one 128-row k-column = 1 KB) run NS - 1 stages ahead of the
// MFMAs and every iteration waits only for the OLDEST stage in flight (s_waitcnt vmcnt(n), n = this wave's loads of the
// younger stages) in front of ONE bare s_barrier, which also releases the stage buffer that is refilled next.
#pragma once
#include "gemm_tile.h"

namespace mnk {

template <int RT>
struct MacroCfg {
    static constexpr int NW = 4 * RT;                    // waves per workgroup
    static constexpr int NT = 64 * NW;                   // threads
    static constexpr int LDA_S = RT * 128 + 16;          // doubles per k-column of the A stage (pad: rows 4 apart hit different banks)
    static constexpr int LDB_S = 128 + 16;
    static constexpr int STAGE = 8 * (LDA_S + LDB_S);    // doubles per stage (8 k-columns of A, then 8 of B)
    static constexpr int JOBS = 8 * (RT + 1);            // wave-wide DMA instructions per stage
    static constexpr int MAXJ = (JOBS + NW - 1) / NW;    // ... per wave, at most
};
template <int RT, int NS>
constexpr int macro_lds_bytes() { return NS * MacroCfg<RT>::STAGE * 8; }

template <int N>
__device__ __forceinline__ void macro_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// acc[ni][mi] of wave (rt, wm, wn): register r of lane (l15, l4) is C^T of tile rt: row wm*64 + mi*16 + l15, column
// wn*64 + ni*16 + l4 + 4r -- the layout of gemm_nt_mainloop<2, 2, 4>, so gemm_nt_epilogue<2, 2, 4, ...> serves every tile
// of the macro tile with (tid & 255, row0 + 128 rt).
// Ag: first row of tile 0 at the first k-column; tile b's rows start `ablk` doubles further (128 in a column-major factor).
// rtn <= RT: number of tiles that exist (a ragged last macro row): the loads of the others are redirected to the last tile
// that exists and their waves multiply what they find (nobody reads their accumulators) -- no branch in the loop.
// Every wave issues exactly MAXJ wave-wide DMA instructions per stage (RT = 3: 36 for 32 k-columns -- four B columns are
// loaded twice, same bytes to the same place), so one immediate serves all waves in `s_waitcnt vmcnt(n)`.
// gate(kt): as in gemm_nt_mainloop (called uniformly before the loads of k-tile kt are issued; false = give up -- the caller
// must then drain vmcnt and pass a barrier before it reuses the LDS).
template <int RT, int NS, class GATE = GemmNoGate>
__device__ __forceinline__ bool macro_mainloop(v4f64 (&acc)[4][4], const double* __restrict__ Ag, int64_t lda, int64_t ablk,
                                               const double* __restrict__ Bg, int64_t ldb, int nk, int rtn, char* smem_raw,
                                               int tid, GATE gate = GATE()) {
    using C = MacroCfg<RT>;
    static_assert(NS >= 2 && NS <= 4, "stages");
    constexpr int MJ = C::MAXJ;
    double* L = reinterpret_cast<double*>(smem_raw);
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rt = wave >> 2, wq = wave & 3, wm = wq & 1, wn = wq >> 1;
    const int l15 = lane & 15, l4 = lane >> 4;
    // DMA job j of a stage: operand block b = j % (RT + 1) (b < RT: the rows of tile b; b == RT: the B rows), k-column
    // j / (RT + 1); wave w issues the jobs w, w + NW, w + 2 NW (beyond the last job: the B columns again)
    const double* src[MJ];
    int64_t step[MJ];     // doubles per stage (8 k-columns)
    int dst[MJ];          // doubles from the start of a stage
#pragma unroll
    for (int q = 0; q < MJ; ++q) {
        int j = wave + C::NW * q;
        int b, k;
        if (j < C::JOBS) { b = j % (RT + 1); k = j / (RT + 1); }
        else { b = RT; k = j - C::JOBS; }
        if (b == RT) {
            src[q] = Bg + k * ldb + lane * 2;
            step[q] = 8 * ldb;
            dst[q] = 8 * C::LDA_S + k * C::LDB_S;
        } else {
            const int bb = b < rtn ? b : rtn - 1;
            src[q] = Ag + bb * ablk + k * lda + lane * 2;
            step[q] = 8 * lda;
            dst[q] = k * C::LDA_S + b * 128;
        }
    }
    auto issue = [&](int buf) {
        double* st = L + buf * C::STAGE;
#pragma unroll
        for (int q = 0; q < MJ; ++q) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src[q],
                                             (__attribute__((address_space(3))) void*)(st + dst[q]), 16, 0, 0);
            src[q] += step[q];
        }
    };
    auto compute = [&](int cur) {
        const double* as = L + cur * C::STAGE + rt * 128 + wm * 64 + l15 + l4 * C::LDA_S;
        const double* bs = L + cur * C::STAGE + 8 * C::LDA_S + wn * 64 + l15 + l4 * C::LDB_S;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            double af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = as[kk * 4 * C::LDA_S + i * 16];
                bf[i] = bs[kk * 4 * C::LDB_S + i * 16];
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[ni], af[mi], acc[ni][mi], 0, 0, 0);
        }
    };
    if (nk <= 0) return true;
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nk) {
            if (!gate(s)) return false;
            issue(s);
        }
    int cur = 0, kt = 0;
    // steady state: stages kt + 1 .. kt + NS - 2 stay in flight, stage kt + NS - 1 is issued into the buffer that the barrier frees
    for (; kt + NS - 1 < nk; ++kt) {
        macro_wait_vm<MJ * (NS - 2)>();
        __builtin_amdgcn_s_barrier();   // stage kt has landed for everybody; the buffer of stage kt - 1 is free
        if (!gate(kt + NS - 1)) return false;
        issue(cur == 0 ? NS - 1 : cur - 1);
        compute(cur);
        cur = cur == NS - 1 ? 0 : cur + 1;
    }
    // drain: r = nk - 1 - kt younger stages in flight
    for (; kt < nk; ++kt) {
        const int r = nk - 1 - kt;
        if (NS >= 4 && r >= 2) macro_wait_vm<MJ * 2>();
        else if (NS >= 3 && r == 1) macro_wait_vm<MJ>();
        else macro_wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        compute(cur);
        cur = cur == NS - 1 ? 0 : cur + 1;
    }
    __builtin_amdgcn_s_barrier();   // (callers reuse the LDS)
    return true;
}

}  // namespace mnk
