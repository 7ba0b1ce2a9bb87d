// Shared declarations for libmadnlp_hip.so (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/madnlp_hip.h"

namespace mnk {

void set_error(const char* fmt, ...);

#define MNK_HIP(call)                                                                   \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            mnk::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return -2;                                                                  \
        }                                                                               \
    } while (0)

#define MNK_REQUIRE(cond, msg)                                                          \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            mnk::set_error("%s:%d: %s", __FILE__, __LINE__, msg);                       \
            return -1;                                                                  \
        }                                                                               \
    } while (0)

// Factorization granularity: every dense buffer is padded to a multiple of PAD
// rows/columns (unit diagonal in the padding) so that tile kernels never see a
// ragged edge; NBI is the width of one inner panel (one diagonal-block kernel).
constexpr int NBI = 64;
constexpr int PAD = 128;
// Rows of slack behind every dense operand buffer: tile loads may run up to one
// wave tile past the last row (values are discarded, memory must be mapped).
constexpr int SLACK = 256;
// max|a_ij| of a transfer is folded into AMAX_SLOTS words 128 bytes apart behind the four result words (ls.h: amax_dev)
constexpr int AMAX_SLOT0 = 16, AMAX_STRIDE = 16, AMAX_SLOTS = 64, AMAX_WORDS = AMAX_SLOT0 + AMAX_STRIDE * AMAX_SLOTS;

// Host-side wait for a stream: hipStreamSynchronize (short spin, then an interrupt-driven sleep) by default; with
// MNK_SPIN_WAIT=1 the stream is polled instead (one host core busy for the duration of every wait).  Round 2 chased
// 40-80 ms stalls of the inertia fetch down to the HOST being frozen by its cgroup CPU quota (BLAS / OpenMP pools sized
// after 256 visible cpus on a 16-CPU quota), not to this wait: both forms measure the same once the pools are bounded
// (bench.py: _cpu_quota).
inline hipError_t stream_wait(hipStream_t s) {
    static const bool spin = []() {
        const char* e = getenv("MNK_SPIN_WAIT");
        return e != nullptr && atoi(e) != 0;
    }();
    if (!spin) return hipStreamSynchronize(s);
    for (;;) {
        const hipError_t e = hipStreamQuery(s);
        if (e != hipErrorNotReady) return e;
        __builtin_ia32_pause();
    }
}

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// One process-wide lock around (a) the launch sequence of every group of PERSISTENT kernels (pivot chain + bulk kernel,
// persistent panel launches, the one-launch solve: mnk_persist_begin / _end) and (b) every runtime call of this library
// that may synchronize the whole device (hipFree, hipMalloc, stream / event destruction).  A persistent group is only
// complete when all of its kernels are in their queues: a device-wide synchronization issued by ANOTHER host thread between
// two of its launches waits for the resident half of the group, which waits for the half that cannot be launched while the
// runtime is synchronizing -- seen with four host threads driving four solvers (a solver freed on one thread, a chain
// waiting 2.5 s for its bulk kernel on another: info = -7, schedule 1).  Frees issued by code outside this library
// (another allocator in the process) are outside this lock: INTEGRATION.md section 0.
std::recursive_mutex& launch_mutex();   // ls.hip
// Called with the lock held, in front of a runtime call that waits for the device to go idle (hipFree, hipHostFree, stream
// destruction): waits until the last persistent operation of this process has completed.  The lock only keeps such a call
// from falling BETWEEN the launches of a group; a group that is launched but not yet fully RUNNING is just as exposed --
// stress runs with six host threads still lost a factorization to a 2.6 s stall (a chain strip waiting for a bulk kernel
// that had been launched, at the moment another thread destroyed its solver) until frees also waited for the groups in flight.
void quiesce_persistent();              // ls.hip
struct LaunchLock {
    LaunchLock() { launch_mutex().lock(); }
    ~LaunchLock() { launch_mutex().unlock(); }
    LaunchLock(const LaunchLock&) = delete;
    LaunchLock& operator=(const LaunchLock&) = delete;
};

// A copy between PAGEABLE host memory and the device (either direction) that the runtime carries out while a persistent group
// of this process is in flight on another stream stops that group for good: both of its kernels are running according to the kernel trace, neither
// makes progress, the bounded waits expire (tools/first_group_probe.py: ONE such copy by a second host thread is enough;
// copies from / to PINNED host memory, allocations, kernels and synchronizations of the other thread are harmless -- the first
// round of a multi-threaded run met it through the other threads' task-list uploads, tools/thread_stress.py; a solve with a
// host right-hand side through its copy back).  Every copy of this library between the device and host memory it does not
// know to be pinned therefore holds the launch mutex (no group is launched meanwhile) and first waits for
// the last persistent operation in flight; it must be complete (stream_wait) before the guard goes.  Copies issued by OTHER
// code of the process are outside this protection: INTEGRATION.md section 0.
struct H2DGuard {
    bool held;
    explicit H2DGuard(bool needed = true) : held(needed) {
        if (held) { launch_mutex().lock(); quiesce_persistent(); }
    }
    ~H2DGuard() { if (held) launch_mutex().unlock(); }
    H2DGuard(const H2DGuard&) = delete;
    H2DGuard& operator=(const H2DGuard&) = delete;
};

// host (pageable or not) -> device, complete when it returns
inline hipError_t h2d_copy(void* dst, const void* src, size_t bytes, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    H2DGuard guard;
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s);
    return e != hipSuccess ? e : stream_wait(s);
}
inline hipError_t h2d_copy_2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipStream_t s) {
    if (width == 0 || height == 0) return hipSuccess;
    H2DGuard guard;
    const hipError_t e = hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyHostToDevice, s);
    return e != hipSuccess ? e : stream_wait(s);
}

// device -> host (pageable or not), complete when it returns
inline hipError_t d2h_copy(void* dst, const void* src, size_t bytes, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    H2DGuard guard;
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s);
    return e != hipSuccess ? e : stream_wait(s);
}
inline hipError_t d2h_copy_2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipStream_t s) {
    if (width == 0 || height == 0) return hipSuccess;
    H2DGuard guard;
    const hipError_t e = hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyDeviceToHost, s);
    return e != hipSuccess ? e : stream_wait(s);
}

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    int alloc(size_t count) {
        release();
        if (count == 0) count = 1;
        LaunchLock lock;   // (never between two launches of a persistent group of another thread)
        hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
        if (e != hipSuccess) {
            p = nullptr;
            set_error("hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
            return -2;
        }
        n = count;
        return 0;
    }
    int upload(const std::vector<T>& h, hipStream_t s) {
        int rc = alloc(h.size());
        if (rc) return rc;
        if (!h.empty()) {
            H2DGuard h2d;
            hipError_t e = hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s);
            if (e != hipSuccess) { set_error("upload failed: %s", hipGetErrorString(e)); return -2; }
            e = stream_wait(s);
            if (e != hipSuccess) { set_error("upload sync failed: %s", hipGetErrorString(e)); return -2; }
        }
        return 0;
    }
    void release() {
        if (p) {
            LaunchLock lock;   // (hipFree synchronizes the device)
            quiesce_persistent();
            (void)hipFree(p);
        }
        p = nullptr;
        n = 0;
    }
    ~DevBuf() { release(); }
};

}  // namespace mnk

struct mnk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // look-ahead of the factorization: panel stream (high priority), update stream, fork/join events
    hipStream_t sp = nullptr, su = nullptr;
    int panel_cus = 0;  // > 0: sp is restricted to this many CUs and su to the others (CU masks)
    // task-DAG schedule (dag.hip): the pivot chain needs only a few CUs, the persistent bulk kernel gets all the others
    hipStream_t sp_dag = nullptr, su_dag = nullptr;
    hipStream_t sp_dagB = nullptr, su_dagB = nullptr;   // batches (dag.hip): a second chain partition (the next dag_cus mask bits) and a bulk stream on the CUs outside both
    bool shared_dag_streams = false;   // the four task-DAG streams belong to the device's shared set (ls.hip), not to this context
    int dag_cus = 0;    // > 0: sp_dag is restricted to this many CUs and su_dag to the others
    // ... and a third pair for its second phase, where every remaining row is in the chain's band (one CU per 64 rows)
    hipStream_t sp_dag2 = nullptr, su_dag2 = nullptr;
    int dag_cus2 = 0;
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    hipStream_t s_fill = nullptr;   // background zero-fill of the spare factor buffer (mnk_ls::fact_spare), created on first use
    std::vector<hipEvent_t> ev_panel, ev_next, ev_next2, ev_bdone;
    int num_cu = 256;   // CUs this context may use (the whole device, or its partition)
    int cu_first = 0;   // first CU-mask bit of the partition
    int total_cu = 256; // CUs of the device
    bool partitioned = false;
    void* persist_slot = nullptr;   // partitioned contexts: their own arbiter slot (ls.hip: mnk_persist_begin)
    // Handles created on this context (solvers, KKT systems).  Garbage-collected hosts (Julia) run finalizers
    // in no particular order: destroying a context that still has children only marks it released, and the
    // last child to go frees it (mnk_ctx_child_gone).
    int children = 0;
    bool released = false;
};
int mnk_masked_stream_pair(mnk_ctx* ctx, int chain_cus, hipStream_t* sp, hipStream_t* su);   // ls.hip
// CU-masked streams that are made when a caller first needs them (every one is a hardware queue of the device): ls.hip
int mnk_ctx_ensure_panel_streams(mnk_ctx* ctx);   // ctx->sp / su: look-ahead streams of the launch-per-panel schedules
int mnk_ctx_ensure_dag(mnk_ctx* ctx);             // ctx->sp_dag / su_dag again after mnk_release_idle_streams took them
int mnk_ctx_ensure_dag2(mnk_ctx* ctx);            // ctx->sp_dag2 / su_dag2: deep-band pair of the task-DAG schedule
int mnk_ctx_ensure_batch_streams(mnk_ctx* ctx);   // ctx->sp_dagB / su_dagB: second chain partition + bulk stream of batches
int mnk_solve_warmup(hipStream_t s);               // solve.hip: first (no-op) launch of the inverse kernel that needs scratch
int mnk_ctx_bulk_wgs(const mnk_ctx* ctx, int first, int per_cu);   // ls.hip: workgroups the mask {CUs first.. of the context} holds AT ONCE
int mnk_dag_warmup(hipStream_t* streams, int n, int nwg);   // dag.hip: first (empty) launch of the bulk kernels on these streams
int mnk_live_contexts(int device);  // contexts alive on this device in this process
// Device arbiter of the PERSISTENT kernels (task-DAG schedule, persistent panel launches, one-launch solve): their waiting
// workgroups stay resident, so two of them from different contexts must never share the chip (each could hold CUs the
// other's producers need).  Every such operation is bracketed by begin / end: `begin` makes the caller's stream wait for
// the previous persistent operation of ANY context of this process on the device, `end` records its own completion --
// the operations run back to back at full speed, in the order they were enqueued, whatever stream they come from
// (reference behaviour: distinct solver instances are driven concurrently, src/KKT/Schur/schur.jl:953).  `begin` returns
// with the arbiter's mutex held; `end` releases it (enqueue order = execution order).  Contexts confined to a CU
// partition arbitrate among themselves only.
int mnk_persist_begin(mnk_ctx* ctx, hipStream_t s);
int mnk_persist_end(mnk_ctx* ctx, hipStream_t s, int rc);   // returns rc (or the error of its own event record)
void mnk_ctx_child_added(mnk_ctx* ctx);
void mnk_ctx_child_gone(mnk_ctx* ctx);

namespace mnk {
// Task-DAG schedule (dag.hip): the launch covers only the band of its strip-column and synchronizes with the persistent
// bulk kernel through progress counters instead of stream events.  front == nullptr: off.
struct PpDag {
    int* front;          // front[t]: leading 128-column tile columns for which the 64-row strip t of L is final
    const int* af;       // "band tile (I, Jt) accumulated" flags written by the bulk kernel, [I * ntile + Jt]
    int ntile;
    int need_front;      // > 0: strips t >= front_from wait for front[t] >= need_front (their rows of the older columns)
    int front_from;      // 4: one launch per strip-column (stream order covers the strips above); 0: persistent chain
    int af_tilecol;      // >= 0: first tile column of this launch; its band tiles were pre-accumulated by the bulk kernel
    long spin_limit;
    unsigned long long* trace;  // diagnostics: 8 time stamps per strip of this launch
    unsigned long long* vmax;   // growth monitor (LDL^T with the BUNCHKAUFMAN guard on): receives max|V|, see growth_fold
    int* dbg;            // diagnostics (option dag_debug): 8 words per strip of the chain -- what a strip that has been waiting for
                         // a long time waits for: {site, strip-column, strip, word index (front: block; af: -1 - tile index),
                         // target, value seen, spins >> 20, 0}
};
}  // namespace mnk

// ---- kernels / launchers shared between translation units -------------------
namespace mnk {

// C (MxN) op= A (MxK) * B (NxK)^T on fp64 MFMA tiles.  mode: 0 sub, 1 set, 2 sub-lower, 4 add-lower.
// `colscale`/`C2`: optional LDL epilogue (C2 = acc, C = acc * colscale[n]); mode 1 only.
int launch_gemm_nt(hipStream_t s, int mode, int64_t M, int64_t N, int64_t K,
                   const double* A, int64_t lda, const double* B, int64_t ldb,
                   double* C, int64_t ldc, const double* colscale, double* C2, int64_t ldc2,
                   const int* info_flag);
// one right-hand-side block of mnk_launch_trsm64_batch (factor.hip): its rows X (and V for LDL^T, else NULL) and the factor's
// diagonal blocks / their 16 x 16 inverses / D^-1 / info word
struct TrsmBatchRec { double* X; double* V; const double* dblk; const double* inv16; const double* dinv; const int* info; };
struct GemmBatchRec { const double* A; const double* B; double* C; const int* info; };   // (info != 0: the triple is skipped)
int launch_gemm_nt_batch(hipStream_t s, int64_t M, int64_t N, int64_t K, const GemmBatchRec* recs_dev, int nbatch, int64_t lda,
                         int64_t ldb, int64_t ldc);
int launch_gemm_nt_lower_small(hipStream_t s, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda,
                               const double* B, int64_t ldb, double* C, int64_t ldc, const int* info_flag);
int gemm_nt_lower_tiles(int64_t M, int64_t N);
int launch_gemm_nt_dbg(hipStream_t s, int shared_ab, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda,
                       const double* B, int64_t ldb, double* C, int64_t ldc);
int launch_gemm_nt_lower_range(hipStream_t s, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda,
                               const double* B, int64_t ldb, double* C, int64_t ldc, const int* info_flag,
                               int tile_begin, int tile_count);
int launch_gemm_nt_queue(hipStream_t s, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda,
                         const double* B, int64_t ldb, double* C, int64_t ldc, int* counter, int cus,
                         const int* info_flag);

// ---- task-DAG schedule (dag.hip): persistent left-looking tile kernel beside the pivot chain -------------------------
// Task list (4 ints per task, see dag.hip).  Strip-columns Js < js2 have a band of `band_tiles` tile rows, the others a band
// that covers every remaining row; returns the number of tasks of the first phase (ready before the chain enters js2).
int dag_build_tasks(int ntile, int chunk, int band_tiles, int js2, std::vector<int>& out, int taper0, std::vector<int>* ready);
int dag_add_fill_tasks(int ntile, std::vector<int>& tasks, std::vector<int>& ready, int n1);
void dag_merge_tasks(const std::vector<int>& tasks, const std::vector<int>& ready, int ninst, int period, std::vector<int>& out);

}  // namespace mnk
