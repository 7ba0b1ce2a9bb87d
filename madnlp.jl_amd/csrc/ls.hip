// C-ABI of the linear solver handle (mnk_ls_*): allocation, `transfer_matrix!`
// (reference src/LinearSolvers/lapack_common.jl:28 and its device twin
// lib/MadNLPGPU/src/utils.jl:12-23 / kernels_sparse.jl:27-33), factorize / inertia /
// solve entry points.  See include/madnlp_hip.h for the per-function citations.
#include <cstdarg>
#include <cstdlib>
#include <atomic>
#include <map>
#include <mutex>

#include <cfloat>
#include "ls.h"

namespace mnk {

static thread_local std::string g_err;
void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

// Zero the (padded) factor buffer and put a unit diagonal on the padding rows.
// Only column blocks on/below the diagonal tile row are touched: nothing in the
// factorization or the solves ever reads above the 128-row tile containing the diagonal.
__global__ __launch_bounds__(256) void fill_lower_kernel(double* __restrict__ F, int64_t ld, int64_t N, int64_t Np) {
    const int64_t col = blockIdx.x;
    const int64_t first = (col / PAD) * PAD;  // first row of the diagonal tile of this column
    const int64_t r = first + ((int64_t)blockIdx.y * 256 + threadIdx.x) * 2;
    if (r >= Np) return;
    double2 v = make_double2(0.0, 0.0);
    if (col >= N) {
        if (r == col) v.x = 1.0;
        if (r + 1 == col) v.y = 1.0;
    }
    *reinterpret_cast<double2*>(F + r + col * ld) = v;
}

// aug_com (lower CSC) -> dense: one thread per stored entry (coordinates precomputed).
// max |v| over the wave -> one atomicMax on the bit pattern (non-negative doubles order like unsigned integers; NaN -> +Inf)
__device__ __forceinline__ void wave_absmax_to(unsigned long long* word, double v) {
    double a = fabs(v);
    if (!(a <= DBL_MAX)) a = __longlong_as_double(0x7ff0000000000000LL);
    for (int off = 32; off > 0; off >>= 1) a = fmax(a, __shfl_xor(a, off));
    // (a relaxed look first: once the word holds a large value almost every wave skips the atomic -- 80 000 atomics on
    // one address took 390 us of a 2112-row transfer)
    if ((threadIdx.x & 63) == 0 && a > 0.0) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(a);
        if (bits > __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(word, bits);
    }
}

// (`amax`: optional, max|a_ij| of what is transferred -- the growth guard of BUNCHKAUFMAN's static-pivot tier)
// (`lim`: only the leading principal block of that order is transferred -- a solver of order lim on a larger KKT handle, the
// leading-block probe of mnk_ls_factorize_sc_async)
__global__ void scatter_csc_kernel(double* __restrict__ F, int64_t ld, const int32_t* __restrict__ row,
                                   const int32_t* __restrict__ col, const double* __restrict__ nz, int64_t nnz,
                                   unsigned long long* __restrict__ amax, int32_t lim = INT32_MAX) {
    const int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    double v = 0.0;
    if (k < nnz && row[k] < lim && col[k] < lim) {
        v = nz[k];
        F[row[k] + (int64_t)col[k] * ld] = v;
    }
    if (amax != nullptr) wave_absmax_to(amax + AMAX_SLOT0 + AMAX_STRIDE * (blockIdx.x & (AMAX_SLOTS - 1)), v);
}

// dense source -> factor buffer, rows >= first row of the diagonal tile of each column.
__global__ __launch_bounds__(256) void copy_lower_kernel(double* __restrict__ F, int64_t ld,
                                                         const double* __restrict__ A, int64_t lda, int64_t N,
                                                         int64_t Np, unsigned long long* __restrict__ amax) {
    const int64_t col = blockIdx.x;
    const int64_t first = (col / PAD) * PAD;
    // four rows per thread, 256 apart (coalesced, four loads in flight; a quarter of the workgroups and of the wave
    // reductions of one row per thread: 50-70 -> ~25 us for the 2112 x 2112 matrix of config C2)
    const int64_t r0 = first + (int64_t)blockIdx.y * 1024 + threadIdx.x;
    if (r0 - threadIdx.x >= Np) return;   // (uniform)
    double v[4], vl = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t r = r0 + 256 * i;
        v[i] = 0.0;
        if (r < Np) {
            if (col < N && r < N) {
                v[i] = A[r + col * lda];
                if (r >= col) vl = fmax(vl, fabs(v[i]));  // the lower triangle is the matrix ('L' storage: the rest may hold anything)
                if (r >= col && !(fabs(v[i]) <= DBL_MAX)) vl = __longlong_as_double(0x7ff0000000000000LL);
            } else if (r == col) {
                v[i] = 1.0;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t r = r0 + 256 * i;
        if (r < Np) F[r + col * ld] = v[i];
    }
    if (amax != nullptr) {
        // one fold per workgroup, spread over AMAX_SLOTS words (publish_info_kernel takes their maximum): 25 000 waves
        // looking at ONE word cost 40 us of the 2112-row transfer
        __shared__ double wmax[4];
        for (int off = 32; off > 0; off >>= 1) vl = fmax(vl, __shfl_xor(vl, off));
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = vl;
        __syncthreads();
        if (threadIdx.x < 64)
            wave_absmax_to(amax + AMAX_SLOT0 + AMAX_STRIDE * ((blockIdx.x + 5 * blockIdx.y) & (AMAX_SLOTS - 1)),
                           fmax(fmax(wmax[0], wmax[1]), fmax(wmax[2], wmax[3])));
    }
}

}  // namespace mnk

using namespace mnk;

static std::mutex g_ctx_mutex;
static std::atomic<int> g_live_ctx[64];
static std::vector<mnk_ctx*> g_ctx_list[64];   // live whole-device contexts per device (g_ctx_mutex): mnk_release_idle_streams

int mnk_live_contexts(int device) { return g_live_ctx[device & 63].load(std::memory_order_relaxed); }

std::recursive_mutex& mnk::launch_mutex() {
    static std::recursive_mutex m;
    return m;
}

// ---- device arbiter of the persistent kernels (common.h) ----
namespace {
struct PersistSlot {   // (guarded by mnk::launch_mutex(): recursive -- a factorization that is redone inside mnk_ls_fetch_info re-enters on the same thread)
    hipEvent_t last = nullptr;
    bool recorded = false;
};
PersistSlot g_persist[64];
PersistSlot& persist_slot_of(mnk_ctx* ctx) {
    if (!ctx->partitioned) return g_persist[ctx->device & 63];
    if (ctx->persist_slot == nullptr) ctx->persist_slot = new PersistSlot();   // (created before the context is shared between threads: mnk_ctx_create_partition)
    return *static_cast<PersistSlot*>(ctx->persist_slot);
}
}  // namespace

void mnk::quiesce_persistent() {
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    for (int d = 0; d < 64; ++d) {
        PersistSlot& ps = g_persist[d];
        if (ps.last == nullptr || !ps.recorded) continue;
        (void)hipSetDevice(d);
        (void)hipEventSynchronize(ps.last);
    }
    if (have_cur) (void)hipSetDevice(cur);
    (void)hipGetLastError();
}

int mnk_persist_begin(mnk_ctx* ctx, hipStream_t s) {
    PersistSlot& ps = persist_slot_of(ctx);
    mnk::launch_mutex().lock();
    if (ps.last == nullptr && hipEventCreateWithFlags(&ps.last, hipEventDisableTiming) != hipSuccess) {
        ps.last = nullptr;
        mnk::launch_mutex().unlock();
        mnk::set_error("mnk_persist_begin: hipEventCreate failed");
        return -2;
    }
    if (ps.recorded && hipStreamWaitEvent(s, ps.last, 0) != hipSuccess) {
        mnk::launch_mutex().unlock();
        mnk::set_error("mnk_persist_begin: hipStreamWaitEvent failed");
        return -2;
    }
    return 0;
}

int mnk_persist_end(mnk_ctx* ctx, hipStream_t s, int rc) {
    PersistSlot& ps = persist_slot_of(ctx);
    if (hipEventRecord(ps.last, s) == hipSuccess) ps.recorded = true;
    else if (rc == 0) { mnk::set_error("mnk_persist_end: hipEventRecord failed"); rc = -2; }
    mnk::launch_mutex().unlock();
    return rc;
}

// ---- CU-masked streams the whole-device contexts of a process share per device ----
namespace {
struct DagStreams { hipStream_t sp = nullptr, su = nullptr, sp2 = nullptr, su2 = nullptr, spB = nullptr, suB = nullptr; int cus = 0; bool made = false; };
DagStreams g_dag_streams[64];
struct MaskedPair { hipStream_t sp = nullptr, su = nullptr; };
std::map<std::pair<int, int>, MaskedPair> g_pair_cache;   // (device, chain CUs) -> pair (mnk_masked_stream_pair)
}  // namespace

// Destroys the masked streams of `device` that are shared between contexts -- the deep-band, batch and small-batch pairs, the
// look-ahead pairs of the live contexts, and with `primary` the task-DAG schedule's first pair as well -- and forgets them in
// every live context: each is made again by the first operation that needs it (mnk_ctx_ensure_*).  Why: every CU-masked stream
// is a hardware queue, the device runs only so many of them side by side -- ALL PROCESSES TOGETHER -- and a process that
// merely holds a dozen idle ones was seen to keep ANOTHER process' pivot chain and bulk kernel from being scheduled together
// (tools/parent_child_probe.py).  Caller holds the launch mutex; nothing persistent is in flight.
static void release_shared_streams_locked(int device, bool primary) {
    mnk::quiesce_persistent();
    (void)hipSetDevice(device);
    std::lock_guard<std::mutex> lk(g_ctx_mutex);
    DagStreams& d = g_dag_streams[device & 63];
    auto kill = [](hipStream_t& st) { if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); st = nullptr; } };
    kill(d.sp2); kill(d.su2); kill(d.spB); kill(d.suB);
    if (primary) { kill(d.sp); kill(d.su); d.made = false; d.cus = 0; }
    for (auto it = g_pair_cache.begin(); it != g_pair_cache.end();) {
        if (it->first.first == device) { kill(it->second.sp); kill(it->second.su); it = g_pair_cache.erase(it); }
        else ++it;
    }
    for (mnk_ctx* c : g_ctx_list[device & 63]) {
        if (!c->shared_dag_streams) continue;
        c->sp_dag2 = c->su_dag2 = c->sp_dagB = c->su_dagB = nullptr;
        if (primary) c->sp_dag = c->su_dag = nullptr;
        kill(c->sp); kill(c->su);
    }
    (void)hipGetLastError();
}

static void ctx_free(mnk_ctx* c) {
    mnk::LaunchLock lock;   // (stream / event destruction may synchronize)
    mnk::quiesce_persistent();
    (void)hipSetDevice(c->device);
    const int live_left = g_live_ctx[c->device & 63].fetch_sub(1, std::memory_order_relaxed) - 1;
    {
        std::lock_guard<std::mutex> lk(g_ctx_mutex);
        auto& v = g_ctx_list[c->device & 63];
        v.erase(std::remove(v.begin(), v.end(), c), v.end());
    }
    // the last context of a device: the process keeps no CU-masked stream (= hardware queue) of that device behind
    if (live_left <= 0 && !c->partitioned) release_shared_streams_locked(c->device, true);
    if (c->ev_a) (void)hipEventDestroy(c->ev_a);
    if (c->ev_b) (void)hipEventDestroy(c->ev_b);
    for (hipEvent_t e : c->ev_panel) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ev_next) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ev_next2) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ev_bdone) (void)hipEventDestroy(e);
    if (c->sp) (void)hipStreamDestroy(c->sp);
    if (c->su) (void)hipStreamDestroy(c->su);
    if (!c->shared_dag_streams) {   // (the per-device set of the whole-device contexts lives as long as the process)
        if (c->sp_dag) (void)hipStreamDestroy(c->sp_dag);
        if (c->su_dag) (void)hipStreamDestroy(c->su_dag);
        if (c->sp_dag2) (void)hipStreamDestroy(c->sp_dag2);
        if (c->su_dag2) (void)hipStreamDestroy(c->su_dag2);
        if (c->sp_dagB) (void)hipStreamDestroy(c->sp_dagB);
        if (c->su_dagB) (void)hipStreamDestroy(c->su_dagB);
    }
    if (c->s_fill) (void)hipStreamDestroy(c->s_fill);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    if (c->persist_slot) {
        PersistSlot* ps = static_cast<PersistSlot*>(c->persist_slot);
        if (ps->last) (void)hipEventDestroy(ps->last);
        delete ps;
    }
    delete c;
}

void mnk_ctx_child_added(mnk_ctx* ctx) {
    if (!ctx) return;
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    ++ctx->children;
}

void mnk_ctx_child_gone(mnk_ctx* ctx) {
    if (!ctx) return;
    bool free_now = false;
    {
        std::lock_guard<std::mutex> lock(g_ctx_mutex);
        --ctx->children;
        free_now = ctx->released && ctx->children <= 0;
    }
    if (free_now) ctx_free(ctx);
}

bool mnk_ls_take_solve_abort(mnk_ls* ls) {
    if (!ls->solve_abort || *ls->solve_abort == 0) return false;
    *ls->solve_abort = 0;
    ls->persistent_solve = 0;
    ls->pub_clean[0] = ls->pub_clean[1] = false;
    return true;
}

extern "C" {

int mnk_version(void) { return MNK_VERSION; }
const char* mnk_last_error_string(void) { return g_err.c_str(); }

// A stream restricted to the listed CU-mask bits (bit b is CU b/8 of XCD b%8 on this part).
static bool make_masked_stream(int num_cu_total, const int* bits, int count, hipStream_t& out) {
    const int words = (num_cu_total + 31) / 32;
    std::vector<uint32_t> m(words, 0u);
    for (int i = 0; i < count; ++i)
        if (bits[i] >= 0 && bits[i] < num_cu_total) m[bits[i] / 32] |= 1u << (bits[i] % 32);
    if (hipExtStreamCreateWithCUMask(&out, (uint32_t)words, m.data()) == hipSuccess) return true;
    (void)hipGetLastError();
    out = nullptr;
    return false;
}

}  // extern "C"
// The CU-masked stream pairs of the task-DAG schedule that whole-device contexts share per device (process lifetime).  Only
// the first pair (pivot chain | bulk kernel) is made with the first context; the others when a caller first needs them --
// every masked stream is a hardware queue (see ctx_create_common).
namespace {
std::vector<int> ctx_bits(const mnk_ctx* c) {
    std::vector<int> bits;
    for (int b = 0; b < c->num_cu; ++b) bits.push_back(c->cu_first + b);
    return bits;
}
}  // namespace

// Size of a persistent grid on the CUs [first, num_cu) of the context (`per_cu` workgroups fit on a CU): the number of
// workgroups the hardware places AT LAUNCH.  The dispatcher deals the workgroups of a grid out evenly -- XCD by XCD, and
// inside an XCD shader engine by shader engine (mask bit b is CU b / 8 of XCD b % 8, shader engine (b / 8) % 4) -- whatever
// the mask left of an engine, so the engine with the fewest CUs bounds what is resident at once: 16 chain CUs take one CU
// from engines 0 and 1 of every XCD, which then hold 7 x 3 = 21 workgroups each, and of a grid of 3 x 240 = 720 exactly
// 32 x 21 = 672 start (tools/stall_hunt.py on a diagnostic build: 21 / 21 / 22 / 21-22 workgroups per engine, 35 CUs of
// engines 2 and 3 holding two).  The others are placed LATER, in mid-kernel, when the hardware finds room -- and a workgroup
// placed in mid-kernel is what the one-in-~3000 time-out of the task-DAG schedule was: in both captured
// events 6 resp. 12 workgroups with block ids 672..686 had all just been started, within 0.2 ms of each other, took a task
// each, and then did nothing for the ~965 ms until the time-out sent the others home -- at which point each ran its whole
// task at the usual 13 us per tile column.  Tasks they held were ones the pivot chain needed.  (They were never in a wait of
// their own, and polling with a back-off changed nothing.  Why the hardware stalls them is not known; a persistent grid
// simply must not exceed what is co-resident at launch.)  Those workgroups did 0.9 % of the tasks.
extern "C" int mnk_debug_grid_at_launch(int cu_first, int num_cu, int first, int per_cu) {
    int per_engine[32] = {0};
    for (int b = std::max(first, 0); b < num_cu; ++b) {
        const int bit = cu_first + b;
        ++per_engine[(bit % 8) * 4 + (bit / 8) % 4];
    }
    int engines = 0, least = INT_MAX;
    for (int e = 0; e < 32; ++e)
        if (per_engine[e] > 0) { ++engines; least = std::min(least, per_engine[e]); }
    return engines > 0 ? per_cu * least * engines : per_cu;
}
int mnk_ctx_bulk_wgs(const mnk_ctx* c, int first, int per_cu) { return mnk_debug_grid_at_launch(c->cu_first, c->num_cu, first, per_cu); }

// the deep-band pair (every row of a small system in the chain's band: dag_cus2 chain CUs | the others)
int mnk_ctx_ensure_dag2(mnk_ctx* c) {
    if (c->sp_dag2 != nullptr || c->dag_cus2 <= 0 || !c->shared_dag_streams) return 0;
    mnk::LaunchLock lock;   // (stream creation may synchronize the device)
    std::lock_guard<std::mutex> lk(g_ctx_mutex);
    DagStreams& d = g_dag_streams[c->device & 63];
    if (d.sp2 == nullptr) {
        mnk::quiesce_persistent();
        MNK_HIP(hipSetDevice(c->device));
        const std::vector<int> bits = ctx_bits(c);
        if (!(make_masked_stream(c->total_cu, bits.data(), c->dag_cus2, d.sp2) &&
              make_masked_stream(c->total_cu, bits.data() + c->dag_cus2, c->num_cu - c->dag_cus2, d.su2))) {
            if (d.sp2) (void)hipStreamDestroy(d.sp2);
            d.sp2 = d.su2 = nullptr;
            return -2;
        }
        hipStream_t warm[1] = {d.su2};
        if (mnk_dag_warmup(warm, 1, 3 * c->num_cu) != 0) (void)hipGetLastError();
        if (mnk_solve_warmup(d.su2) != 0) (void)hipGetLastError();
    }
    c->sp_dag2 = d.sp2; c->su_dag2 = d.su2;
    return 0;
}

// batches of independent factorizations: a second chain partition and a bulk stream on the CUs outside both
int mnk_ctx_ensure_batch_streams(mnk_ctx* c) {
    if (c->sp_dagB != nullptr || c->dag_cus <= 0 || !c->shared_dag_streams || 2 * c->dag_cus + 32 > c->num_cu) return 0;
    mnk::LaunchLock lock;
    std::lock_guard<std::mutex> lk(g_ctx_mutex);
    DagStreams& d = g_dag_streams[c->device & 63];
    if (d.spB == nullptr) {
        mnk::quiesce_persistent();
        MNK_HIP(hipSetDevice(c->device));
        const std::vector<int> bits = ctx_bits(c);
        if (!(make_masked_stream(c->total_cu, bits.data() + c->dag_cus, c->dag_cus, d.spB) &&
              make_masked_stream(c->total_cu, bits.data() + 2 * c->dag_cus, c->num_cu - 2 * c->dag_cus, d.suB))) {
            if (d.spB) (void)hipStreamDestroy(d.spB);
            d.spB = d.suB = nullptr;
            return -2;
        }
        hipStream_t warm[1] = {d.suB};
        if (mnk_dag_warmup(warm, 1, 3 * c->num_cu) != 0) (void)hipGetLastError();
        for (hipStream_t w : {d.suB, d.spB})
            if (mnk_solve_warmup(w) != 0) (void)hipGetLastError();
    }
    c->sp_dagB = d.spB; c->su_dagB = d.suB;
    return 0;
}

// the task-DAG schedule's first pair (pivot chain | bulk kernel) after mnk_release_idle_streams has taken it away
int mnk_ctx_ensure_dag(mnk_ctx* c) {
    if (c->sp_dag != nullptr || c->dag_cus <= 0 || !c->shared_dag_streams) return 0;
    mnk::LaunchLock lock;
    std::lock_guard<std::mutex> lk(g_ctx_mutex);
    DagStreams& d = g_dag_streams[c->device & 63];
    if (d.sp == nullptr) {
        mnk::quiesce_persistent();
        MNK_HIP(hipSetDevice(c->device));
        const std::vector<int> bits = ctx_bits(c);
        if (!(make_masked_stream(c->total_cu, bits.data(), c->dag_cus, d.sp) &&
              make_masked_stream(c->total_cu, bits.data() + c->dag_cus, c->num_cu - c->dag_cus, d.su))) {
            if (d.sp) (void)hipStreamDestroy(d.sp);
            d.sp = d.su = nullptr;
            return -2;
        }
        d.cus = c->dag_cus; d.made = true;
        hipStream_t warm[1] = {d.su};
        if (mnk_dag_warmup(warm, 1, 3 * c->num_cu) != 0) (void)hipGetLastError();
        for (hipStream_t w : {d.su, d.sp})
            if (mnk_solve_warmup(w) != 0) (void)hipGetLastError();
    }
    c->sp_dag = d.sp; c->su_dag = d.su;
    return 0;
}

extern "C" int mnk_release_idle_streams(int device) {
    MNK_REQUIRE(device >= 0 && device < 64, "mnk_release_idle_streams: bad device");
    mnk::LaunchLock lock;   // (no persistent group is being launched; the ones in flight are waited for)
    release_shared_streams_locked(device, true);
    return 0;
}

// the look-ahead streams of the launch-per-panel schedules (1, 4): a quarter of the CUs for the panel stream, the rest for the
// update stream (measured, profiles/; MNK_PANEL_CUS overrides); stream priorities where masks are not to be had
int mnk_ctx_ensure_panel_streams(mnk_ctx* c) {
    if (c->sp != nullptr) return 0;
    mnk::LaunchLock lock;
    mnk::quiesce_persistent();
    MNK_HIP(hipSetDevice(c->device));
    const std::vector<int> bits = ctx_bits(c);
    auto make_pair = [&](int want) -> bool {
        if (want <= 0 || want >= c->num_cu) return false;
        if (make_masked_stream(c->total_cu, bits.data(), want, c->sp) &&
            make_masked_stream(c->total_cu, bits.data() + want, c->num_cu - want, c->su))
            return true;
        if (c->sp) { (void)hipStreamDestroy(c->sp); c->sp = nullptr; }
        c->su = nullptr;
        return false;
    };
    if (const char* e = getenv("MNK_PANEL_CUS")) {
        const int want = atoi(e);
        if (make_pair(want)) c->panel_cus = want;
    } else if (c->num_cu >= 16) {
        const int q4 = std::max(4, c->num_cu / 4);
        if (make_pair(q4)) c->panel_cus = q4;
    }
    if (!c->sp) {
        int prio_lo = 0, prio_hi = 0;  // numerically lower = higher priority
        MNK_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        MNK_HIP(hipStreamCreateWithPriority(&c->sp, hipStreamNonBlocking, prio_hi));
        MNK_HIP(hipStreamCreateWithPriority(&c->su, hipStreamNonBlocking, prio_lo));
    }
    return 0;
}

extern "C" {
// `part_first/part_count` confine the context to a contiguous range of CU-mask bits.  Kept for experiments
// only (no public entry point): partitions run MFMA-bound kernels side by side at full rate, but the
// latency-bound panel chain of one instance slows ~2x next to the updates of the others -- every partition
// has CUs on every XCD and shares all L2s, and masks that differ between XCDs are not honoured
// (tools/cumask_probe.hip, tools/partition_probe.py) -- so 16 instances per GPU run faster time-sharing the
// whole chip (bench.py --batch) than on 2/4/8 partitions (72 vs 58/45/26 it/s).
static int ctx_create_common(int device, void* stream, int part_first, int part_count, mnk_ctx** out) {
    MNK_REQUIRE(out != nullptr, "mnk_ctx_create: out is NULL");
    mnk::LaunchLock lock;   // (stream / event creation: not between two launches of another thread's persistent group)
    int ndev = 0;
    MNK_HIP(hipGetDeviceCount(&ndev));
    MNK_REQUIRE(device >= 0 && device < ndev, "mnk_ctx_create: no such device");
    MNK_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    MNK_HIP(hipGetDeviceProperties(&prop, device));
    const int total = prop.multiProcessorCount;
    const bool part = part_count > 0;
    if (part)
        MNK_REQUIRE(part_first >= 0 && part_count >= 16 && part_first + part_count <= total && stream == nullptr,
                    "mnk_ctx_create_partition: need 16 <= cu_count, the range inside the device, and a library-owned stream");
    mnk_ctx* c = new mnk_ctx();
    c->device = device;
    c->cu_first = part ? part_first : 0;
    c->partitioned = part;
    c->total_cu = total;
    c->num_cu = part ? part_count : total;
    // The CU-mask bits this context owns: a contiguous range (bits are dealt round-robin over the XCDs, so every
    // XCD contributes the same number of CUs).  Masks that differ between XCDs are not honoured by the runtime
    // (tools/cumask_probe.hip: a strided mask runs on all 256 CUs), so partitions cannot be whole XCDs.
    std::vector<int> bits;
    for (int b = 0; b < c->num_cu; ++b) bits.push_back(c->cu_first + b);
    if (stream) {
        c->stream = (hipStream_t)stream;
    } else if (part) {
        if (!make_masked_stream(total, bits.data(), part_count, c->stream)) {
            delete c;
            set_error("mnk_ctx_create_partition: CU-masked streams are not available on this device");
            return -2;
        }
        c->own_stream = true;
    } else {
        MNK_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    // (The look-ahead streams of the launch-per-panel schedules are created when such a schedule first runs:
    // mnk_ctx_ensure_panel_streams.  Every CU-masked stream is a hardware queue of its own, and the hardware runs only so
    // many queues of a device side by side -- all processes together: with four idle contexts in a PARENT process, each
    // holding its masked pair, a child's chain and bulk kernels were no longer scheduled together and every run of
    // tools/parent_child_probe.py lost factorizations to their bounded waits; without the pairs, none.)
    auto make_pair = [&](int want, hipStream_t& sp, hipStream_t& su) -> bool {
        if (want <= 0 || want >= c->num_cu) return false;
        if (make_masked_stream(total, bits.data(), want, sp) &&
            make_masked_stream(total, bits.data() + want, c->num_cu - want, su))
            return true;
        if (sp) { (void)hipStreamDestroy(sp); sp = nullptr; }
        su = nullptr;
        return false;
    };
    // second pair for the task-DAG schedule: a handful of CUs for the pivot chain (two per XCD), the rest for the bulk kernel.
    // Whole-device contexts SHARE these four streams per device: the persistent operations of a process take turns on the
    // device anyway (mnk_persist_begin), and every stream more is one more client of the runtime's few hardware queues -- two
    // streams whose kernels must run side by side (chain and bulk) must not end up multiplexed behind a third one.
    if (c->num_cu >= 64) {
        DagStreams own;
        std::lock_guard<std::mutex> lock(g_ctx_mutex);
        DagStreams& d = part ? own : g_dag_streams[device & 63];
        if (!d.made) {
            const int want = getenv("MNK_DAG_CUS") ? atoi(getenv("MNK_DAG_CUS")) : 16;
            if (make_pair(want, d.sp, d.su)) d.cus = want;
            d.made = true;
            hipStream_t warm[1] = {d.su};
            if (mnk_dag_warmup(warm, 1, 3 * c->num_cu) != 0) (void)hipGetLastError();   // (best effort)
            for (hipStream_t w : {d.su, d.sp})              // (where the inverses for the solves are launched)
                if (w != nullptr && mnk_solve_warmup(w) != 0) (void)hipGetLastError();
        }
        if (!part) g_ctx_list[device & 63].push_back(c);
        c->sp_dag = d.sp; c->su_dag = d.su; c->dag_cus = d.cus;
        c->dag_cus2 = d.cus > 0 ? (getenv("MNK_DAG_CUS2") ? atoi(getenv("MNK_DAG_CUS2")) : 96) : 0;   // (its streams: mnk_ctx_ensure_dag2)
        if (c->dag_cus2 <= 0 || c->dag_cus2 >= c->num_cu) c->dag_cus2 = 0;
        c->shared_dag_streams = !part;
        if (part) {   // (a partition's streams are its own: made now, destroyed with it)
            if (c->dag_cus2 > 0 && !make_pair(c->dag_cus2, c->sp_dag2, c->su_dag2)) c->dag_cus2 = 0;
        }
    }
    MNK_HIP(hipEventCreateWithFlags(&c->ev_a, hipEventDisableTiming));
    MNK_HIP(hipEventCreateWithFlags(&c->ev_b, hipEventDisableTiming));
    if (mnk_solve_warmup(c->stream) != 0) (void)hipGetLastError();   // (the context's own stream runs the last inverses of every factorization)
    g_live_ctx[device & 63].fetch_add(1, std::memory_order_relaxed);
    *out = c;
    return 0;
}

int mnk_ctx_create(int device, void* stream, mnk_ctx** out) { return ctx_create_common(device, stream, 0, 0, out); }

}  // extern "C"
// A pair of CU-masked streams of the whole device: `sp` on the first `chain_cus` mask bits, `su` on all the others (nullptr
// when none are left); cached per device and size until mnk_release_idle_streams / the device's last context goes (masked
// streams are hardware queues).
int mnk_masked_stream_pair(mnk_ctx* ctx, int chain_cus, hipStream_t* sp, hipStream_t* su) {
    mnk::LaunchLock lock;
    MNK_REQUIRE(!ctx->partitioned && chain_cus > 0 && chain_cus <= ctx->total_cu, "mnk_masked_stream_pair: bad size");
    MaskedPair& p = g_pair_cache[{ctx->device, chain_cus}];
    if (p.sp == nullptr) {
        std::vector<int> bits;
        for (int b = 0; b < ctx->total_cu; ++b) bits.push_back(b);
        if (!make_masked_stream(ctx->total_cu, bits.data(), chain_cus, p.sp)) { set_error("mnk_masked_stream_pair: CU-masked streams are not available"); return -2; }
        if (chain_cus < ctx->total_cu && !make_masked_stream(ctx->total_cu, bits.data() + chain_cus, ctx->total_cu - chain_cus, p.su)) {
            (void)hipStreamDestroy(p.sp);
            p.sp = nullptr;
            set_error("mnk_masked_stream_pair: CU-masked streams are not available");
            return -2;
        }
        if (p.su != nullptr) {   // (first launches of the kernels that need scratch: see mnk_dag_warmup)
            hipStream_t w[1] = {p.su};
            if (mnk_dag_warmup(w, 1, 3 * ctx->num_cu) != 0) (void)hipGetLastError();
        }
    }
    *sp = p.sp;
    *su = p.su;
    return 0;
}
extern "C" {

int mnk_ctx_destroy(mnk_ctx* c) {
    if (!c) return 0;
    {
        std::lock_guard<std::mutex> lock(g_ctx_mutex);
        if (c->children > 0) {  // a solver / KKT handle still points here: the last of them frees the context
            c->released = true;
            return 0;
        }
    }
    ctx_free(c);
    return 0;
}


int mnk_ctx_synchronize(mnk_ctx* c) {
    MNK_REQUIRE(c != nullptr, "ctx is NULL");
    MNK_HIP(mnk::stream_wait(c->stream));
    return 0;
}

void* mnk_ctx_stream(mnk_ctx* c) { return c ? (void*)c->stream : nullptr; }

int mnk_ls_create(mnk_ctx* ctx, int64_t N, int algo, mnk_ls** out) {
    MNK_REQUIRE(ctx && out, "mnk_ls_create: NULL argument");
    mnk::LaunchLock lock;   // (pinned host allocations, buffers)
    MNK_REQUIRE(N > 0, "mnk_ls_create: N must be positive");
    const bool bk_requested = algo == MNK_BUNCHKAUFMAN;
    if (bk_requested) algo = MNK_LDL;  // tier 1: static-pivot blocked LDL^T; tier 2 on breakdown: bk.hip
    MNK_REQUIRE(algo == MNK_CHOLESKY || algo == MNK_LDL,
                "mnk_ls_create: CHOLESKY, LDL and BUNCHKAUFMAN are implemented on device");
    MNK_HIP(hipSetDevice(ctx->device));
    mnk_ls* ls = new mnk_ls();
    ls->ctx = ctx;
    ls->N = N;
    ls->algo = algo;
    ls->bk_requested = bk_requested;
    // Tuning overrides from the environment, for runs of unmodified callers: MNK_OPTIONS="key=value,key=value" with the keys
    // of mnk_ls_set_option (plus MNK_PANEL_ALGO / MNK_PERSISTENT_SOLVE, the two a deployment may need: INTEGRATION.md).  An
    // option set this way is not changed by later mnk_ls_set_option calls.
    {
        auto from_env = [&](const std::string& key, double v) {
            if (mnk_ls_set_option(ls, key.c_str(), v) == 0) ls->env_keys.push_back(key);
            else fprintf(stderr, "madnlp_hip: ignoring environment override '%s': %s\n", key.c_str(), mnk_last_error_string());
        };
        if (const char* e = getenv("MNK_PANEL_ALGO")) from_env("panel_algo", atof(e));
        if (const char* e = getenv("MNK_PERSISTENT_SOLVE")) from_env("persistent_solve", atof(e));
        if (const char* e = getenv("MNK_OPTIONS")) {
            std::string all(e);
            size_t pos = 0;
            while (pos < all.size()) {
                size_t end = all.find(',', pos);
                if (end == std::string::npos) end = all.size();
                const std::string item = all.substr(pos, end - pos);
                const size_t eq = item.find('=');
                if (eq != std::string::npos && eq > 0) from_env(item.substr(0, eq), atof(item.c_str() + eq + 1));
                pos = end + 1;
            }
        }
    }
    ls->Np = round_up(N, PAD);
    ls->ld = ls->Np;
    ls->ldw = ls->Np;
    int rc = 0;
    rc |= ls->fact.alloc((size_t)ls->ld * ls->Np + SLACK);
    rc |= ls->linv.alloc((size_t)(ls->Np / NBI) * NBI * NBI);
    rc |= ls->dblk.alloc((size_t)(ls->Np / NBI) * NBI * NBI);
    rc |= ls->inv16.alloc((size_t)(ls->Np / NBI) * 1024);
    rc |= ls->linv256.alloc((size_t)((ls->Np + 255) / 256) * 65536);
    rc |= ls->linv256t.alloc((size_t)((ls->Np + 255) / 256) * 65536);
    rc |= ls->dvec.alloc(ls->Np);
    rc |= ls->dinv.alloc(ls->Np);
    rc |= ls->xwork.alloc(10 * ls->Np);
    if (hipHostMalloc((void**)&ls->solve_abort, sizeof(int), hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        ls->solve_abort = nullptr;
        rc |= -2;
    } else {
        *ls->solve_abort = 0;
    }
    if (hipHostMalloc((void**)&ls->pin, 8 * sizeof(unsigned long long), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&ls->pin_dev, ls->pin, 0) != hipSuccess) {
        (void)hipGetLastError();
        rc |= -2;
    }
    rc |= ls->info_dev.alloc(4);   // info | which bounded device-side wait expired (info = -7) / last valid pivot (info = -9) | early-rejection switch | -
    rc |= ls->inertia_dev.alloc(3);
    if (rc) { delete ls; return -2; }
    MNK_HIP(hipMemsetAsync(ls->fact.p, 0, ((size_t)ls->ld * ls->Np + SLACK) * sizeof(double), ctx->stream));
    mnk_ctx_child_added(ctx);
    *out = ls;
    return 0;
}

int mnk_ls_destroy(mnk_ls* ls) {
    if (!ls) return 0;
    (void)hipSetDevice(ls->ctx->device);
    if (mnk_ls_pending_elsewhere(ls)) {
        // the other thread's batch still holds this pointer and will launch on it at its end: deleting now would leave it dangling
        set_error("mnk_ls_destroy: a factorize! call of this solver is pending in a batch that another thread opened "
                  "(that thread's mnk_factorize_batch_end must come first); the solver was NOT destroyed");
        return -1;
    }
    (void)mnk_ls_sync_deferred(ls);
    (void)mnk::stream_wait(ls->ctx->stream);
    if (ls->probe_ls) { (void)mnk_ls_destroy(ls->probe_ls); ls->probe_ls = nullptr; }
    mnk::LaunchLock lock;   // (host-memory frees and event destruction below; the device buffers lock for themselves)
    mnk::quiesce_persistent();
    if (ls->solve_abort) (void)hipHostFree(ls->solve_abort);
    if (ls->pin) (void)hipHostFree(ls->pin);
    if (ls->ctx->s_fill) (void)mnk::stream_wait(ls->ctx->s_fill);   // a background fill of this solver's spare buffer
    if (ls->ev_spare) (void)hipEventDestroy(ls->ev_spare);
    if (ls->ev_free) (void)hipEventDestroy(ls->ev_free);
    if (ls->ev_defer) (void)hipEventDestroy(ls->ev_defer);
    if (ls->ev_info) (void)hipEventDestroy(ls->ev_info);
    mnk_ctx* ctx = ls->ctx;
    delete ls;
    mnk_ctx_child_gone(ctx);
    return 0;
}

int mnk_ls_set_option(mnk_ls* ls, const char* key, double value) {
    MNK_REQUIRE(ls && key, "mnk_ls_set_option: NULL argument");
    { int rc_d = mnk_ls_sync_deferred(ls); if (rc_d) return rc_d; }
    for (const std::string& k : ls->env_keys)
        if (k == key) {   // fixed by the environment for this process (MNK_OPTIONS): the call is ignored, once per key out loud
            static std::mutex warn_mutex;
            static std::vector<std::string> warned;
            std::lock_guard<std::mutex> lock(warn_mutex);
            if (std::find(warned.begin(), warned.end(), k) == warned.end()) {
                warned.push_back(k);
                fprintf(stderr, "madnlp_hip: set_option('%s', %g) ignored: the option is pinned by MNK_OPTIONS / MNK_PANEL_ALGO / "
                                "MNK_PERSISTENT_SOLVE for this process (query: get_stat(\"pinned:%s\"))\n", key, value, key);
            }
            return 0;
        }
    if (!strcmp(key, "pivot_tol")) { ls->pivot_tol = value; return 0; }
    if (!strcmp(key, "split_a")) { ls->split_a = (int)value; return 0; }
    if (!strcmp(key, "tail_rows")) { ls->tail_rows = (int64_t)value; return 0; }
    if (!strcmp(key, "tail_nbo")) { ls->tail_nbo = (int64_t)value; return 0; }
    if (!strcmp(key, "own_cols")) { ls->own_cols = std::max<long>(64, (long)value / 64 * 64); return 0; }
    if (!strcmp(key, "panel0_whole")) { ls->panel0_whole = (int)value; return 0; }
    if (!strcmp(key, "dag_js2")) { ls->dag_js2_override = (int)value; return 0; }   // experiments: strip-column at which every row joins the band (-1: by size)
    if (!strcmp(key, "outer_block")) {
        int64_t v = (int64_t)value;
        MNK_REQUIRE(v == 0 || (v >= NBI && v % NBI == 0), "outer_block must be 0 (by size) or a positive multiple of 64");
        ls->nbo_auto = v == 0;
        if (v > 0) ls->nbo = v;
        ls->wbuf[0].release();
        ls->wbuf[1].release();
        return 0;
    }
    if (!strcmp(key, "solve512")) { ls->solve512 = value != 0; return 0; }   // 512-column steps of the one-launch solve
    if (!strcmp(key, "solve512_min_rows")) { ls->solve512_min_rows = (int64_t)value; return 0; }
    if (!strcmp(key, "linv_mfma")) { ls->linv_mfma = value != 0.0; return 0; }
    if (!strcmp(key, "dag_chain_inline")) { ls->dag_chain_inline = value != 0.0; return 0; }
    if (!strcmp(key, "prefill")) { ls->prefill = value != 0; return 0; }   // background zero-fill into a second factor buffer
    if (!strcmp(key, "prefill_max_rows")) { ls->prefill_max_rows = (int64_t)value; return 0; }
    if (!strcmp(key, "single_rows")) {  // systems up to this order: one outer panel, no look-ahead (0: never)
        ls->single_rows = (int64_t)value;
        ls->wbuf[0].release();
        ls->wbuf[1].release();
        return 0;
    }
    if (!strcmp(key, "lookahead")) { ls->lookahead = value != 0.0; return 0; }
    // 0: no work sharing, 1: the panel stream joins the trailing update when that is the longer leg,
    // 2: always (used by the schedule tests)
    if (!strcmp(key, "share")) { ls->share = (int)value; return 0; }
    if (!strcmp(key, "small_tiles")) { ls->small_tiles = (int)value; return 0; }
    if (!strcmp(key, "small_tiles_mid")) { ls->small_tiles_mid = (int)value; return 0; }
    // 5: task-DAG schedule (persistent left-looking bulk kernel beside the pivot chain, dag.hip); 4: persistent panel
    // kernel per 256 columns + one trailing update per outer panel; 1: one launch per piece (the fallback of 4 and 5)
    if (!strcmp(key, "panel_algo")) {
        MNK_REQUIRE((int)value == 1 || (int)value == 4 || (int)value == 5, "panel_algo must be 1, 4 or 5");
        ls->panel_algo = (int)value;
        return 0;
    }
    if (!strcmp(key, "pp_fuse_rows")) { ls->pp_fuse_rows = (int64_t)value; return 0; }
    if (!strcmp(key, "dag_min_rows")) { ls->dag_min_rows = (int64_t)value; return 0; }
    if (!strcmp(key, "dag_taper0")) {
        MNK_REQUIRE(value >= 1.0 && value <= 16.0, "dag_taper0 must be in 1..16");
        ls->dag_taper0 = (int)value;
        ls->dag_tasks.release();
        return 0;
    }
    if (!strcmp(key, "dag_chunk")) {   // tile columns per bulk task of the task-DAG schedule (the task list is rebuilt)
        MNK_REQUIRE(value >= 1.0 && value <= 64.0, "dag_chunk must be in 1..64");
        ls->dag_chunk = (int)value;
        ls->dag_tasks.release();
        return 0;
    }
    if (!strcmp(key, "dag_band")) {    // 64-row strips of the pivot chain's band
        MNK_REQUIRE(value >= 8.0 && value <= 64.0 && (int)value % 4 == 0 && (double)(int)value == value, "dag_band must be a multiple of 4 in 8..64");
        ls->dag_band = (int)value;
        ls->dag_tasks.release();
        return 0;
    }
    if (!strcmp(key, "dag_spin_limit")) {  // polls a wait of the task-DAG schedule's kernels may take before it gives up
        MNK_REQUIRE(value >= 1024.0, "dag_spin_limit must be at least 1024");
        ls->dag_spin_limit = (long)value;
        return 0;
    }
    if (!strcmp(key, "dag_fill")) { ls->dag_fill = value != 0.0; ls->dag_tasks.release(); return 0; }   // zero-fill of the spare buffer as tasks of the bulk queue
    if (!strcmp(key, "batch_period")) { ls->batch_period = (int)value; return 0; }
    if (!strcmp(key, "dag_trace")) { ls->dag_trace_on = value != 0.0; return 0; }  // diagnostics: tools/dag_timeline.py
    if (!strcmp(key, "dag_max_rows")) { ls->dag_max_rows = (int64_t)value; return 0; }
    if (!strcmp(key, "dag_deep_rows")) { ls->dag_deep_rows = (int64_t)value; ls->dag_tasks.release(); return 0; }   // largest order with every row in the chain's band
    // BUNCHKAUFMAN only: 1 (default) = refactor with the pivoted Bunch-Kaufman tier when the static-pivot
    // factorization breaks down; 0 = report the breakdown as num_zero and let the IPM regularize
    if (!strcmp(key, "bk_fallback")) { ls->bk_fallback = (int)value; return 0; }
    if (!strcmp(key, "bk_panel_wgs")) {   // pivoted tier: 0 = a workgroup per 256 rows of a panel, 1 = one workgroup per panel
        MNK_REQUIRE(value == 0.0 || value == 1.0, "bk_panel_wgs must be 0 or 1");
        ls->bk_panel_wgs = (int)value;
        return 0;
    }
    if (!strcmp(key, "accept_only_pd")) { ls->accept_only_pd = value != 0; return 0; }  // see mnk_ls_fetch_info
    if (!strcmp(key, "probe")) { ls->probe = value != 0; return 0; }   // leading-block probe in front of a factorization that is likely to be rejected early (ls.h)
    if (!strcmp(key, "early_reject")) { ls->early_reject = value != 0; return 0; }      // with accept_only_pd: stop at the first non-positive pivot (leaf64.h)
    // BUNCHKAUFMAN only: element growth max|d_k| / max|a_ij| of the static-pivot tier above which the pivoted tier takes over
    if (!strcmp(key, "dag_debug")) {
        ls->dag_debug = value != 0.0;
        if (ls->dag_debug && !ls->dag_dbg.p) {
            // (8 words per chain strip, then -- diagnostic builds with MNK_DIAG_BULK_DBG -- 16 words per bulk workgroup)
            if (ls->dag_dbg.alloc(8 * 128 + 16 * 1024)) return -2;
            MNK_HIP(hipMemset(ls->dag_dbg.p, 0, (8 * 128 + 16 * 1024) * sizeof(int)));
        }
        return 0;
    }
    if (!strcmp(key, "bk_max_wgs")) {
        MNK_REQUIRE(value >= 0.0 && value <= 256.0, "bk_max_wgs must be in 0..256");
        ls->bk_max_wgs = (int)value;
        return 0;
    }
    if (!strcmp(key, "bk_spin_limit")) {
        MNK_REQUIRE(value >= 1024.0, "bk_spin_limit must be at least 1024");
        ls->bk_spin_limit = (long)value;
        return 0;
    }
    if (!strcmp(key, "debug_bk_missing")) { ls->debug_bk_missing = (int)value; return 0; }
    if (!strcmp(key, "bk_growth_tol")) {
        MNK_REQUIRE(value > 1.0, "bk_growth_tol must be > 1");
        ls->bk_growth_tol = value;
        return 0;
    }
    if (!strcmp(key, "bk_growth_tol_qd")) {  // the same for pivot sequences "all positive, then all negative"
        MNK_REQUIRE(value > 1.0, "bk_growth_tol_qd must be > 1");
        ls->bk_growth_tol_qd = value;
        return 0;
    }
    if (!strcmp(key, "persistent_solve")) { ls->persistent_solve = value != 0.0; return 0; }
    if (!strcmp(key, "ps_spin_limit")) {  // polls a persistent-solve wait may take before it gives up
        MNK_REQUIRE(value >= 1024.0, "ps_spin_limit must be at least 1024");
        ls->ps_spin_limit = (long)value;
        return 0;
    }
    // tests only: workgroup `value` of every persistent solve leaves at once, as a peer that never became
    // resident would (< 0: off); the others must give up after ps_spin_limit polls instead of hanging
    if (!strcmp(key, "debug_pp_missing")) { ls->debug_pp_missing = (int)value; return 0; }
    if (!strcmp(key, "debug_ps_missing")) { ls->debug_ps_missing = (int)value; return 0; }
    if (!strcmp(key, "solve_trace")) {  // diagnostics: time stamps of the forward sweep's critical path
        if (value != 0.0) {
            int rc = ls->solve_trace.alloc((size_t)(ls->Np / 64) * 8);
            if (rc) return rc;
            MNK_HIP(hipMemset(ls->solve_trace.p, 0, (size_t)(ls->Np / 64) * 8 * sizeof(unsigned long long)));
        } else {
            ls->solve_trace.release();
        }
        return 0;
    }
    set_error("mnk_ls_set_option: unknown option '%s'", key);
    return -1;
}

static int ensure_wbuf(mnk_ls* ls) {
    if (ls->algo != MNK_LDL || ls->wbuf[0].p) return 0;
    const size_t cnt = (size_t)ls->ldw * std::min<int64_t>(mnk_ls_effective_nbo(ls), ls->Np) + SLACK;
    int rc = ls->wbuf[0].alloc(cnt);
    // double buffered for the look-ahead: panel k+1 is factored while panel k is applied
    if (!rc && mnk_ls_effective_nbo(ls) < ls->Np) rc = ls->wbuf[1].alloc(cnt);
    return rc;
}

// The word that receives max|a_ij| of the matrix being transferred (zeroed here), or NULL when nobody will ask for it.
static unsigned long long* amax_word(mnk_ls* ls) {
    if (!(ls->bk_requested && ls->bk_fallback && ls->algo == MNK_LDL)) return nullptr;
    if (!ls->amax_dev.p && ls->amax_dev.alloc(AMAX_WORDS)) return nullptr;
    (void)hipMemsetAsync(ls->amax_dev.p, 0, AMAX_WORDS * sizeof(unsigned long long), ls->ctx->stream);
    return ls->amax_dev.p;
}

static int prepare_fill(mnk_ls* ls) {
    mnk_ctx* ctx = ls->ctx;
    hipStream_t s = ctx->stream;
    dim3 grid((unsigned)ls->Np, (unsigned)((ls->Np / 2 + 255) / 256));
    bool pre = ls->prefill && ls->Np <= ls->prefill_max_rows;
    if (pre && !ls->fact_spare.p) {
        // second factor buffer, events, side stream; if any of it cannot be had the fill stays in line
        if (ls->fact_spare.alloc(ls->fact.n) != 0) { (void)hipGetLastError(); ls->prefill = 0; pre = false; }
        if (pre && !ctx->s_fill && hipStreamCreateWithFlags(&ctx->s_fill, hipStreamNonBlocking) != hipSuccess) { ls->prefill = 0; pre = false; }
        if (pre && (hipEventCreateWithFlags(&ls->ev_spare, hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&ls->ev_free, hipEventDisableTiming) != hipSuccess)) { ls->prefill = 0; pre = false; }
        if (pre) MNK_HIP(hipMemsetAsync(ls->fact_spare.p, 0, ls->fact_spare.n * sizeof(double), s));  // (slack behind the matrix)
    }
    if (pre && ls->spare_zeroed && ls->spare_by_dag && !ls->info_valid) {
        // The zero-fill was left to the task queue of the previous factorization, and nobody has looked at its `info` yet: a
        // factorization that fails (a time-out, a Cholesky breakdown) drops the rest of its queue, fill tasks included.
        ls->spare_zeroed = false;
        ls->spare_by_dag = false;
    }
    if (pre && ls->spare_zeroed) {
        std::swap(ls->fact.p, ls->fact_spare.p);          // the buffer zeroed in the background becomes the factor buffer
        MNK_HIP(hipStreamWaitEvent(s, ls->ev_spare, 0));
        // the other buffer now holds the previous factor: it counts as zeroed again only after a COMPLETED
        // mnk_ls_prefill_spare() (a second transfer for the same factorize! -- the pivoted tier, a redo after a time-out --
        // would otherwise swap the old factor back in and scatter the sparse entries over it)
        ls->spare_zeroed = false;
    } else {
        hipLaunchKernelGGL(fill_lower_kernel, grid, dim3(256), 0, s, ls->fact.p, ls->ld, ls->N, ls->Np);
    }
    ls->spare_pending = pre;   // mnk_ls_prefill_spare() zeroes the other buffer once this factorization is queued
    MNK_HIP(hipGetLastError());
    return 0;
}

// The other factor buffer (the previous factor, or fresh memory) is free once everything queued so far has run: from then
// on it is zeroed on the side stream, behind the factorization that was just queued (whose persistent kernels want the
// whole chip from their first microsecond) and concurrently with the inertia fetch and the solves that follow.
}  // extern "C"
int mnk_ls_prefill_spare(mnk_ls* ls) {
    if (!ls->spare_pending) return 0;
    ls->spare_pending = false;
    if (ls->dag_filled) {   // the bulk queue of the factorization that was just queued zeroes the spare buffer (DAG_FILL tasks): stream order does the rest
        ls->dag_filled = false;
        MNK_HIP(hipEventRecord(ls->ev_spare, ls->ctx->stream));
        ls->spare_zeroed = true;
        ls->spare_by_dag = true;
        return 0;
    }
    ls->spare_by_dag = false;
    mnk_ctx* ctx = ls->ctx;
    dim3 grid((unsigned)ls->Np, (unsigned)((ls->Np / 2 + 255) / 256));
    MNK_HIP(hipEventRecord(ls->ev_free, ctx->stream));
    MNK_HIP(hipStreamWaitEvent(ctx->s_fill, ls->ev_free, 0));
    hipLaunchKernelGGL(fill_lower_kernel, grid, dim3(256), 0, ctx->s_fill, ls->fact_spare.p, ls->ld, ls->N, ls->Np);
    MNK_HIP(hipGetLastError());
    MNK_HIP(hipEventRecord(ls->ev_spare, ctx->s_fill));
    ls->spare_zeroed = true;
    return 0;
}
extern "C" {

static int transfer_sc(mnk_ls* ls, mnk_sc* sc) {
    int rc = prepare_fill(ls);
    if (rc) return rc;
    const int64_t nnz = sc->nnz_aug;
    hipLaunchKernelGGL(scatter_csc_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, ls->ctx->stream,
                       ls->fact.p, ls->ld, sc->aug_row.p, sc->aug_col.p, sc->aug_nz.p, nnz, amax_word(ls),
                       ls->N < sc->n ? (int32_t)ls->N : INT32_MAX);
    MNK_HIP(hipGetLastError());
    return 0;
}

// The leading-block probe (ls.h).  On the AC-OPF run of the bench line 13 of 17 rejected trials stop in the same 64 columns
// (3072-3135 of 11 192): the static-pivot elimination has done 1 - (1 - 0.28)^3 = 63 % of a factorization's flops by then (4.4 ms),
// the leading principal block of order 3328 alone is 2.6 % of them (1.8 ms as a factorization of its own).  A matrix whose leading
// block is not positive definite is not positive definite, and the pivots of the block do not depend on the rest of the matrix: a
// probe that fails gives the verdict of the full factorization.  When: early rejection armed, the previous verdict of this solver
// was an acceptance (in inertia_correction! the matrix after a rejection is the regularized one), and a rejection that stopped in
// the first half of the columns is at most three verdicts old -- solvers that never reject never probe.
// Returns 0 and *rejected; on a rejection the solver's state is that of an early-rejected factorization (inertia: the block's
// positive pivots, everything else negative; a solve that comes all the same completes the factorization from the handle).
static int probe_leading_block(mnk_ls* ls, mnk_sc* sc, bool* rejected) {
    *rejected = false;
    if (!(ls->probe && ls->early_reject && ls->accept_only_pd && ls->algo == MNK_LDL && ls->N == sc->n && !mnk_batch_active())) return 0;
    if (ls->probe_last_rejected || ls->probe_hint_col < 0 || ls->probe_since_hint > 3) return 0;
    const int64_t m = (ls->probe_hint_col + 256) / 256 * 256;
    if (m < 512 || 2 * m > ls->N) return 0;
    if (ls->probe_ls == nullptr || ls->probe_order != m) {
        if (ls->probe_ls) { (void)mnk_ls_destroy(ls->probe_ls); ls->probe_ls = nullptr; }
        mnk_ls* child = nullptr;
        if (mnk_ls_create(ls->ctx, m, ls->algo, &child) != 0) { (void)hipGetLastError(); return 0; }   // (no memory for it: no probe)
        child->probe = 0;
        child->accept_only_pd = 1;
        child->early_reject = 1;
        child->pivot_tol = ls->pivot_tol;
        ls->probe_ls = child;
        ls->probe_order = m;
    }
    mnk_ls* c = ls->probe_ls;
    int rc = mnk_ls_factorize_sc_async(c, sc);
    if (!rc) rc = mnk_ls_fetch_info(c);
    if (rc) return rc;
    if (c->npos == m && c->nzero == 0 && c->nneg == 0) { ++ls->probe_misses; return 0; }
    // not positive definite: the verdict, without the full factorization
    ++ls->probe_hits;
    ++ls->early_rejects;
    ls->early_reject_col = c->factor_invalid ? c->early_reject_col : m - 1;
    ls->info = 1;
    ls->npos = c->npos;
    ls->nzero = c->nzero;
    ls->nneg = ls->N - ls->npos - ls->nzero;
    ls->factor_invalid = true;
    ls->info_valid = true;
    ls->factorized = true;
    ls->bk_active = false;
    ls->probe_last_rejected = true;
    if (2 * ls->early_reject_col < ls->N) { ls->probe_hint_col = ls->early_reject_col; ls->probe_since_hint = 0; }
    *rejected = true;
    return 0;
}

int mnk_ls_factorize_sc_async(mnk_ls* ls, mnk_sc* sc) {
    MNK_REQUIRE(ls && sc && sc->ctx, "mnk_ls_factorize_sc: NULL argument or host-only handle");
    // A solver of SMALLER order factors the leading principal block of the handle's matrix (round 6: the probe of
    // madnlp_jl_amd.ipm_dev -- a matrix whose leading block is not positive definite is not positive definite, and a static-pivot
    // elimination that stops at column c has done 1 - (1 - c/N)^3 of the full factorization's work where the block alone costs (c/N)^3)
    MNK_REQUIRE(sc->n >= ls->N, "mnk_ls_factorize_sc: the solver's order exceeds the KKT system's");
    MNK_HIP(hipSetDevice(ls->ctx->device));
    { int rc_d = mnk_ls_sync_deferred(ls); if (rc_d) return rc_d; }   // (a factorize! of this solver pending in an open batch runs first)
    int rc = ensure_wbuf(ls);
    if (rc) return rc;
    bool probe_rejected = false;
    rc = probe_leading_block(ls, sc, &probe_rejected);
    if (rc) return rc;
    if (!probe_rejected) {
        rc = transfer_sc(ls, sc);
        if (rc) return rc;
    }
    // (the KKT handle keeps aug_com until the next build_kkt!, so the pivoted tier can fetch the matrix again when
    // the inertia is asked for)
    ls->src_persistent = true;   // (the handle keeps aug_com until the next build_kkt!)
    ls->retransfer = [ls, sc, w = std::weak_ptr<int>(sc->alive)]() {
        if (w.expired()) {
            set_error("factorize!: the KKT handle of the last factorize! call was destroyed before its inertia was fetched; the "
                      "matrix cannot be transferred again for the fall-back tier");
            return -5;
        }
        return transfer_sc(ls, sc);
    };
    if (probe_rejected) return 0;   // (the verdict is in; ensure_complete_factor knows the way back to the matrix)
    return mnk_ls_run_factorization(ls);
}

static int transfer_dense(mnk_ls* ls, const double* Adev, int64_t lda) {
    dim3 grid((unsigned)ls->Np, (unsigned)((ls->Np + 1023) / 1024));
    hipLaunchKernelGGL(copy_lower_kernel, grid, dim3(256), 0, ls->ctx->stream, ls->fact.p, ls->ld, Adev, lda,
                       ls->N, ls->Np, amax_word(ls));
    MNK_HIP(hipGetLastError());
    return 0;
}

static int factorize_dense_dev(mnk_ls* ls, const double* Adev, int64_t lda, bool src_persistent = false) {
    int rc = ensure_wbuf(ls);
    if (rc) return rc;
    rc = transfer_dense(ls, Adev, lda);
    if (rc) return rc;
    ls->src_persistent = src_persistent;   // (early rejection needs a source that outlives the call: ensure_complete_factor)
    ls->retransfer = [ls, Adev, lda]() { return transfer_dense(ls, Adev, lda); };  // (callers clear it when Adev's life ends)
    return mnk_ls_run_factorization(ls);
}

}  // extern "C"
// (internal: the dense device source of the Schur stage's scenario blocks; asynchronous -- the caller keeps `Adev` alive and
// fetches the info itself)
int mnk_ls_factorize_dense_dev_async(mnk_ls* ls, const double* Adev, int64_t lda) {
    MNK_REQUIRE(ls && Adev && lda >= ls->N, "mnk_ls_factorize_dense_dev_async: bad argument");
    MNK_HIP(hipSetDevice(ls->ctx->device));
    { int rc_d = mnk_ls_sync_deferred(ls); if (rc_d) return rc_d; }
    return factorize_dense_dev(ls, Adev, lda);
}
extern "C" {

int mnk_ls_factorize_dc_async(mnk_ls* ls, mnk_dc* dc) {
    MNK_REQUIRE(ls && dc, "mnk_ls_factorize_dc: NULL argument");
    MNK_REQUIRE(dc->order == ls->N, "mnk_ls_factorize_dc: order mismatch");
    MNK_HIP(hipSetDevice(ls->ctx->device));
    { int rc_d = mnk_ls_sync_deferred(ls); if (rc_d) return rc_d; }   // (a factorize! of this solver pending in an open batch runs first)
    int rc = factorize_dense_dev(ls, dc->aug.p, round_up(dc->order, PAD), true);
    // the way back goes through the handle (its buffer may be reallocated, the handle may be destroyed before the inertia
    // of this asynchronous call is fetched)
    ls->retransfer = [ls, dc, w = std::weak_ptr<int>(dc->alive)]() {
        if (w.expired()) {
            set_error("factorize!: the KKT handle of the last factorize! call was destroyed before its inertia was fetched; the "
                      "matrix cannot be transferred again for the fall-back tier");
            return -5;
        }
        return transfer_dense(ls, dc->aug.p, round_up(dc->order, PAD));
    };
    return rc;
}

static int finish_info(mnk_ls* ls, int* info) {
    int rc = mnk_ls_fetch_info(ls);
    if (rc) return rc;
    if (info) *info = ls->info;
    return 0;
}

int mnk_ls_factorize_sc(mnk_ls* ls, mnk_sc* sc, int* info) {
    int rc = mnk_ls_factorize_sc_async(ls, sc);
    return rc ? rc : finish_info(ls, info);
}

int mnk_ls_factorize_dc(mnk_ls* ls, mnk_dc* dc, int* info) {
    int rc = mnk_ls_factorize_dc_async(ls, dc);
    return rc ? rc : finish_info(ls, info);
}

int mnk_ls_factorize_dense(mnk_ls* ls, const double* A, int64_t lda, int loc, int* info) {
    MNK_REQUIRE(ls && A, "mnk_ls_factorize_dense: NULL argument");
    MNK_REQUIRE(lda >= ls->N, "mnk_ls_factorize_dense: lda < N");
    MNK_HIP(hipSetDevice(ls->ctx->device));
    { int rc_d = mnk_ls_sync_deferred(ls); if (rc_d) return rc_d; }   // (a factorize! of this solver pending in an open batch runs first)
    int rc;
    if (loc == MNK_DEVICE) {
        rc = factorize_dense_dev(ls, A, lda);
    } else {
        DevBuf<double> tmp;
        rc = tmp.alloc((size_t)ls->N * ls->N);
        if (rc) return rc;
        MNK_HIP(mnk::h2d_copy_2d(tmp.p, ls->N * sizeof(double), A, lda * sizeof(double), ls->N * sizeof(double), ls->N,
                                 ls->ctx->stream));
        rc = factorize_dense_dev(ls, tmp.p, ls->N);
        MNK_HIP(mnk::stream_wait(ls->ctx->stream));
        if (!rc) rc = finish_info(ls, info);  // (a breakdown is handled while the staging buffer is alive)
        ls->retransfer = nullptr;
        return rc;
    }
    rc = rc ? rc : finish_info(ls, info);
    ls->retransfer = nullptr;  // the caller's device buffer is only guaranteed for the duration of the call
    return rc;
}

int mnk_ls_factorize_csc(mnk_ls* ls, const int32_t* colptr, const int32_t* rowval, const double* nzval,
                         int index_base, int* info) {
    MNK_REQUIRE(ls && colptr && rowval && nzval, "mnk_ls_factorize_csc: NULL argument");
    MNK_HIP(hipSetDevice(ls->ctx->device));
    { int rc_d = mnk_ls_sync_deferred(ls); if (rc_d) return rc_d; }   // (a factorize! of this solver pending in an open batch runs first)
    const int64_t N = ls->N;
    const int64_t nnz = colptr[N] - index_base;
    std::vector<int32_t> row(nnz), col(nnz);
    for (int64_t c = 0; c < N; ++c)
        for (int64_t k = colptr[c] - index_base; k < colptr[c + 1] - index_base; ++k) {
            row[k] = rowval[k] - index_base;
            col[k] = (int32_t)c;
            MNK_REQUIRE(row[k] >= c && row[k] < N, "mnk_ls_factorize_csc: matrix must be lower triangular");
        }
    DevBuf<int32_t> drow, dcol;
    DevBuf<double> dnz;
    std::vector<double> nzv(nzval, nzval + nnz);
    int rc = drow.upload(row, ls->ctx->stream);
    rc |= dcol.upload(col, ls->ctx->stream);
    rc |= dnz.upload(nzv, ls->ctx->stream);
    if (rc) return -2;
    rc = ensure_wbuf(ls);
    if (rc) return rc;
    auto transfer = [ls, nnz, &drow, &dcol, &dnz]() -> int {
        int r = prepare_fill(ls);
        if (r) return r;
        if (nnz > 0)
            hipLaunchKernelGGL(scatter_csc_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, ls->ctx->stream,
                               ls->fact.p, ls->ld, drow.p, dcol.p, dnz.p, nnz, amax_word(ls));
        MNK_HIP(hipGetLastError());
        return 0;
    };
    rc = transfer();
    if (rc) return rc;
    ls->src_persistent = false;
    ls->retransfer = transfer;
    rc = mnk_ls_run_factorization(ls);
    if (!rc) {
        MNK_HIP(mnk::stream_wait(ls->ctx->stream));
        rc = finish_info(ls, info);
    }
    ls->retransfer = nullptr;  // the staging buffers die with this call
    return rc;
}

int mnk_ls_inertia(mnk_ls* ls, int64_t* num_pos, int64_t* num_zero, int64_t* num_neg) {
    MNK_REQUIRE(ls, "mnk_ls_inertia: NULL argument");
    { int rc_d = mnk_ls_sync_deferred(ls); if (rc_d) return rc_d; }
    MNK_REQUIRE(ls->factorized, "mnk_ls_inertia: factorize first");
    MNK_HIP(hipSetDevice(ls->ctx->device));
    int rc = mnk_ls_fetch_info(ls);
    if (rc) return rc;
    if (num_pos) *num_pos = ls->npos;
    if (num_zero) *num_zero = ls->nzero;
    if (num_neg) *num_neg = ls->nneg;
    return 0;
}

int mnk_ls_check_solve(mnk_ls* ls) {
    MNK_REQUIRE(ls, "mnk_ls_check_solve: NULL argument");
    MNK_HIP(hipSetDevice(ls->ctx->device));
    { int rc_q = mnk_solve_sync_deferred(ls); if (rc_q) return rc_q; }
    MNK_HIP(mnk::stream_wait(ls->ctx->stream));
    if (mnk_ls_take_solve_abort(ls)) {
        set_error("mnk_ls_check_solve: a persistent solve on device-resident data gave up waiting for a peer workgroup "
                  "(device oversubscribed by another process?); its result is invalid -- solve again "
                  "(persistent_solve is now off for this solver)");
        return -3;
    }
    return 0;
}

// A factorization that was stopped at its first non-positive pivot (early_reject) left no factor.  A caller that solves all the
// same did not ask for the inertia, or overrules its own test: MadNLP does where it initializes the multipliers by least
// squares and in the restoration phases' first steps (reference src/IPM/solver.jl:147-190, src/IPM/restoration.jl: factorize_wrapper!
// followed by solve_refine_wrapper! without inertia_correction!).  The matrix is then factored again, to the end: early
// rejection is an optimization of the rejected trials, never a change of what the solver can do.
static int ensure_complete_factor(mnk_ls* ls, const char* who) {
    if (ls->reject_on_device == 1) { int rc_i = mnk_ls_fetch_info(ls); if (rc_i) return rc_i; }   // (known when the pivots are)
    if (!ls->factor_invalid) return 0;
    { static const bool peek = getenv("MNK_DBG_NO_REDO") != nullptr; if (peek) return 0; }   // (diagnostics: look at what a rejected factorization left)
    if (!ls->retransfer) {
        set_error("%s: the last factorization was stopped at its first non-positive pivot (early_reject) and its source is gone: "
                  "there is no factor", who);
        return -1;
    }
    const int keep = ls->early_reject;
    ls->early_reject = 0;
    int rc = ls->retransfer();
    if (!rc) rc = mnk_ls_run_factorization(ls);
    if (!rc) rc = mnk_ls_fetch_info(ls);
    ls->early_reject = keep;
    ++ls->early_reject_redone;
    return rc;
}

int mnk_ls_solve(mnk_ls* ls, double* x, int64_t nrhs, int64_t ldx, int loc) {
    MNK_REQUIRE(ls && x, "mnk_ls_solve: NULL argument");
    { int rc_d = mnk_ls_sync_deferred_fact(ls); if (rc_d) return rc_d; }
    MNK_REQUIRE(ls->factorized, "mnk_ls_solve: factorize first");
    MNK_HIP(hipSetDevice(ls->ctx->device));
    { int rc_f = ensure_complete_factor(ls, "mnk_ls_solve"); if (rc_f) return rc_f; }
    MNK_REQUIRE(nrhs >= 1 && ldx >= ls->N, "mnk_ls_solve: bad nrhs/ldx");
    MNK_HIP(hipSetDevice(ls->ctx->device));
    if (ls->solve_abort && *ls->solve_abort != 0) {
        // a previous solve on device-resident vectors gave up (its result is invalid): fail loudly now and use
        // the stepwise solve from here on
        *ls->solve_abort = 0;
        ls->persistent_solve = 0;
        ls->pub_clean[0] = ls->pub_clean[1] = false;
        set_error("mnk_ls_solve: an earlier persistent solve on device-resident data gave up waiting for a peer "
                  "workgroup (device oversubscribed by another process?); its result is invalid -- refactorize/solve "
                  "again (persistent_solve is now off for this solver)");
        return -3;
    }
    hipStream_t s = ls->ctx->stream;
    const int64_t N = ls->N, Np = ls->Np;
    double* w = ls->xwork.p;
    for (int64_t k = 0; k < nrhs; ++k) {
        double* xk = x + k * ldx;
        if (loc == MNK_DEVICE) {   // (the one-launch solve works on the caller's vector itself)
            if (nrhs == 1 && mnk_solve_defer(ls, xk)) continue;   // an open solve batch of this thread took it
            int rc = mnk_solve_sync_deferred(ls);                   // (earlier queued solves of this solver keep their place)
            if (rc) return rc;
            rc = mnk_ls_run_solve(ls, w, xk);
            if (rc) return rc;
            continue;
        }
        { int rc_q = mnk_solve_sync_deferred(ls); if (rc_q) return rc_q; }
        MNK_HIP(hipMemsetAsync(w, 0, Np * sizeof(double), s));
        MNK_HIP(mnk::h2d_copy(w, xk, N * sizeof(double), s));
        int rc = mnk_ls_run_solve(ls, w);
        if (rc) return rc;
        if (loc != MNK_DEVICE && ls->persistent_solve) {
            // The one-launch solve gives up (instead of hanging the device) if its workgroups cannot all become
            // resident, e.g. another process saturates the GPU with its own persistent kernels.  The host still
            // owns the right-hand side here: redo this and all later solves with one launch per step.
            MNK_HIP(mnk::stream_wait(s));
            if (mnk_ls_take_solve_abort(ls)) {
                MNK_HIP(hipMemsetAsync(w, 0, Np * sizeof(double), s));
                MNK_HIP(mnk::h2d_copy(w, xk, N * sizeof(double), s));
                rc = mnk_ls_run_solve(ls, w);
                if (rc) return rc;
            }
        }
        if (loc == MNK_DEVICE) MNK_HIP(hipMemcpyAsync(xk, w, N * sizeof(double), hipMemcpyDeviceToDevice, s));
        else MNK_HIP(mnk::d2h_copy(xk, w, N * sizeof(double), s));
    }
    return 0;
}

// ---- array entry points for batches of independent instances (one call per phase of an iteration instead of four to six
// calls per instance: from an interpreted host the per-call overhead of 16 instances was 4.6 ms of a 150 ms step) ----
int mnk_sc_step_batch(int n, mnk_sc* const* sc, mnk_ls* const* ls, const double* const* jac_coo, const double* const* hess_coo,
                      const double* const* pr_diag, const double* const* du_diag, int loc) {
    MNK_REQUIRE(n >= 0 && (n == 0 || (sc && ls && jac_coo && hess_coo && pr_diag && du_diag)), "mnk_sc_step_batch: NULL argument");
    int rc = mnk_factorize_batch_begin();
    if (rc) return rc;
    for (int i = 0; i < n && !rc; ++i) {
        rc = mnk_sc_compress_jacobian(sc[i], jac_coo[i], loc);
        if (!rc) rc = mnk_sc_compress_hessian(sc[i], hess_coo[i], loc);
        if (!rc) rc = mnk_sc_build(sc[i], pr_diag[i], du_diag[i], loc);
        if (!rc) rc = mnk_ls_factorize_sc_async(ls[i], sc[i]);
    }
    const int rc_end = mnk_factorize_batch_end();
    return rc ? rc : rc_end;
}

int mnk_ls_inertia_batch(int n, mnk_ls* const* ls, int64_t* num_pos, int64_t* num_zero, int64_t* num_neg) {
    MNK_REQUIRE(n >= 0 && (n == 0 || (ls && num_pos && num_zero && num_neg)), "mnk_ls_inertia_batch: NULL argument");
    for (int i = 0; i < n; ++i) {
        int rc = mnk_ls_inertia(ls[i], num_pos + i, num_zero + i, num_neg + i);
        if (rc) return rc;
    }
    return 0;
}

int mnk_ls_solve_batch(int n, mnk_ls* const* ls, double* const* x, int loc) {
    MNK_REQUIRE(n >= 0 && (n == 0 || (ls && x)), "mnk_ls_solve_batch: NULL argument");
    int rc = mnk_solve_batch_begin();
    if (rc) return rc;
    for (int i = 0; i < n && !rc; ++i) rc = mnk_ls_solve(ls[i], x[i], 1, ls[i]->N, loc);
    const int rc_end = mnk_solve_batch_end();
    return rc ? rc : rc_end;
}

int mnk_ls_debug_dag_state(mnk_ls* ls, int* flags, int64_t nflags, int* chain, int64_t nchain, int* have) {
    MNK_REQUIRE(ls && have, "mnk_ls_debug_dag_state: NULL argument");
    *have = ls->dbg_flags.empty() ? 0 : 1;
    if (flags) memcpy(flags, ls->dbg_flags.data(), sizeof(int) * (size_t)std::min<int64_t>(nflags, (int64_t)ls->dbg_flags.size()));
    if (chain) memcpy(chain, ls->dbg_chain.data(), sizeof(int) * (size_t)std::min<int64_t>(nchain, (int64_t)ls->dbg_chain.size()));
    return 0;
}

int mnk_ls_debug_solve_trace(mnk_ls* ls, unsigned long long* out, int64_t n) {
    if (ls && out && ls->dag_trace.p && ls->dag_trace_on) {  // the task-DAG schedule's trace (option dag_trace) shares this read-out
        MNK_HIP(hipSetDevice(ls->ctx->device));
        MNK_HIP(mnk::stream_wait(ls->ctx->stream));
        const int64_t cnt = std::min<int64_t>(n, (int64_t)ls->dag_trace.n);
        MNK_HIP(mnk::d2h_copy(out, ls->dag_trace.p, cnt * sizeof(unsigned long long), ls->ctx->stream));
        return 0;
    }
    MNK_REQUIRE(ls && out && ls->solve_trace.p, "mnk_ls_debug_solve_trace: tracing is off (option solve_trace)");
    MNK_HIP(hipSetDevice(ls->ctx->device));
    MNK_HIP(mnk::stream_wait(ls->ctx->stream));
    const int64_t cnt = std::min<int64_t>(n, (ls->Np / 64) * 8);
    MNK_HIP(mnk::d2h_copy(out, ls->solve_trace.p, cnt * sizeof(unsigned long long), ls->ctx->stream));
    return 0;
}

int mnk_ls_get_factor(mnk_ls* ls, double* L, double* D, int loc) {
    MNK_REQUIRE(ls && L, "mnk_ls_get_factor: NULL argument");
    { int rc_d = mnk_ls_sync_deferred(ls); if (rc_d) return rc_d; }
    MNK_REQUIRE(ls->factorized, "mnk_ls_get_factor: factorize first");
    { int rc_f = ensure_complete_factor(ls, "mnk_ls_get_factor"); if (rc_f) return rc_f; }
    MNK_HIP(hipSetDevice(ls->ctx->device));
    hipStream_t s = ls->ctx->stream;
    if (loc == MNK_DEVICE) {
        MNK_HIP(hipMemcpy2DAsync(L, ls->N * sizeof(double), ls->fact.p, ls->ld * sizeof(double), ls->N * sizeof(double), ls->N,
                                 hipMemcpyDeviceToDevice, s));
        if (D) MNK_HIP(hipMemcpyAsync(D, ls->dvec.p, ls->N * sizeof(double), hipMemcpyDeviceToDevice, s));
        MNK_HIP(mnk::stream_wait(s));
    } else {
        MNK_HIP(mnk::d2h_copy_2d(L, ls->N * sizeof(double), ls->fact.p, ls->ld * sizeof(double), ls->N * sizeof(double), ls->N, s));
        if (D) MNK_HIP(mnk::d2h_copy(D, ls->dvec.p, ls->N * sizeof(double), s));
    }
    return 0;
}

int mnk_ls_get_stat(mnk_ls* ls, const char* key, double* value) {
    MNK_REQUIRE(ls && key && value, "mnk_ls_get_stat: NULL argument");
    MNK_HIP(hipSetDevice(ls->ctx->device));
    { int rc_d = mnk_ls_sync_deferred(ls); if (rc_d) return rc_d; }
    if (ls->factorized) {
        int rc = mnk_ls_fetch_info(ls);
        if (rc) return rc;
    }
    if (!strncmp(key, "pinned:", 7)) {   // 1 if MNK_OPTIONS & co. fixed this option for the process (set_option calls on it are ignored)
        *value = std::find(ls->env_keys.begin(), ls->env_keys.end(), std::string(key + 7)) != ls->env_keys.end() ? 1.0 : 0.0;
        return 0;
    }
    if (!strcmp(key, "panel_algo")) { *value = ls->algo_now; return 0; }
    if (!strcmp(key, "pp_fallbacks")) { *value = ls->pp_fallbacks; return 0; }
    // which bounded device-side wait expired last (info = -7): 1 bulk task / operand rows, 2 bulk task / chunk order, 3 gate on the bulk stream, 4 chain strip / diagonal block, 5 chain strip / bulk kernel's rows or band tiles; 0: none
    if (!strcmp(key, "timeout_site")) { *value = ls->last_timeout_site; return 0; }
    if (!strcmp(key, "probe_hits")) { *value = (double)ls->probe_hits; return 0; }       // matrices rejected by the leading-block probe alone
    if (!strcmp(key, "probe_misses")) { *value = (double)ls->probe_misses; return 0; }   // probes that passed (the full factorization followed)
    if (!strcmp(key, "early_rejects")) { *value = (double)ls->early_rejects; return 0; }       // factorizations stopped at their first non-positive pivot
    if (!strcmp(key, "early_reject_redone")) { *value = (double)ls->early_reject_redone; return 0; }   // ... rejected factorizations completed after all because the caller solved
    if (!strcmp(key, "early_reject_col")) { *value = (double)ls->early_reject_col; return 0; } // ... the last valid pivot of the latest one
    if (!strcmp(key, "stall_ms_total")) { *value = ls->stall_ms_total; return 0; }      // what this solver's expired waits (fall-backs) have cost, host ms
    if (!strcmp(key, "stall_ms_process")) { *value = mnk_process_stall_ms(); return 0; }  // ... all solvers of the process
    if (!strcmp(key, "growth")) { *value = ls->last_growth; return 0; }  // BUNCHKAUFMAN: max|d_k| / max|a_ij| of the static-pivot tier
    if (!strcmp(key, "sign_changes")) { *value = (double)ls->last_sign_changes; return 0; }  // ... and sign changes along its pivots
    if (!strcmp(key, "bk_count")) { *value = ls->bk_count; return 0; }
    if (!strcmp(key, "bk_panel_multi")) { *value = ls->bk_multi_last ? 1.0 : 0.0; return 0; }
    if (!strcmp(key, "bk_mw_fallbacks")) { *value = ls->bk_mw_fallbacks; return 0; }
    if (!strcmp(key, "dag_bulk_wgs")) { *value = ls->ctx->dag_cus > 0 ? mnk_ctx_bulk_wgs(ls->ctx, ls->ctx->dag_cus, 3) : 0; return 0; }   // grid of the bulk kernel beside the 16-CU chain
    if (!strcmp(key, "dag_ntasks")) { *value = ls->dag_ntasks; return 0; }    // task-DAG schedule: bulk tasks, ...
    if (!strcmp(key, "dag_ntasks1")) { *value = ls->dag_ntasks1; return 0; }  // ... of them in the first phase, ...
    if (!strcmp(key, "dag_js2")) { *value = ls->dag_js2; return 0; }          // ... first strip-column of the second phase
    set_error("mnk_ls_get_stat: unknown key '%s'", key);
    return -1;
}

int mnk_ls_bk_info(mnk_ls* ls, int* active, int* count, int32_t* perm, double* doff) {
    MNK_REQUIRE(ls, "mnk_ls_bk_info: NULL argument");
    MNK_HIP(hipSetDevice(ls->ctx->device));
    { int rc_d = mnk_ls_sync_deferred(ls); if (rc_d) return rc_d; }
    if (ls->factorized) {
        int rc = mnk_ls_fetch_info(ls);  // the tier is decided when the static factorization's pivots are known
        if (rc) return rc;
    }
    if (active) *active = ls->bk_active ? 1 : 0;
    if (count) *count = ls->bk_count;
    if (ls->bk_active && (perm || doff)) {
        hipStream_t s = ls->ctx->stream;
        if (perm) MNK_HIP(mnk::d2h_copy(perm, ls->bk_perm.p, ls->N * sizeof(int32_t), s));
        if (doff) MNK_HIP(mnk::d2h_copy(doff, ls->bk_doff.p, ls->N * sizeof(double), s));
    }
    return 0;
}

int mnk_gemm_nt(mnk_ctx* ctx, int mode, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda,
                const double* B, int64_t ldb, double* C, int64_t ldc) {
    MNK_REQUIRE(ctx && A && B && C, "mnk_gemm_nt: NULL argument");
    MNK_HIP(hipSetDevice(ctx->device));
    return launch_gemm_nt(ctx->stream, mode, M, N, K, A, lda, B, ldb, C, ldc, nullptr, nullptr, 0, nullptr);
}

// Clock probe: one wave runs a fixed dependent chain of fp64 FMAs and reports the elapsed constant-rate
// (100 MHz) timer ticks; run beside a load, the ratio to the idle value is the shader-clock ratio.
__global__ void clock_probe_kernel(unsigned long long* out, int iters, double a, double b) {
    double x = (double)threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) x = __builtin_fma(x, a, b);
    }
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (x == 1.2345) out[1] = 1;
}

// Diagnostics: time `reps` lower-tile updates C -= A*A^T (M x M, depth K) under one of the
// schedules the factorization uses.  variant 0: static tiling on the context stream (all CUs);
// 1: tile queue on the context stream; 2: static on the update stream (its CU mask); 3: queue on the
// update stream; 4: queue drained by update + panel streams together; 5: static on the update
// stream while the panel stream runs the same product on a second matrix C2 (interference probe);
// 6-8: static in chunks of 512/1024/256 tiles; 9/10: every tile reads the same operand blocks
// (context / update stream); 20: clock probe alone, 21: clock probe beside an update (ms <- ticks).
int mnk_debug_update(mnk_ctx* ctx, int variant, int64_t M, int64_t K, const double* A, int64_t lda, double* C,
                     double* C2, int64_t ldc, int reps, double* ms) {
    MNK_REQUIRE(ctx && A && C && ms, "mnk_debug_update: NULL argument");
    MNK_HIP(hipSetDevice(ctx->device));
    { int rc_s = mnk_ctx_ensure_panel_streams(ctx); if (rc_s) return rc_s; }
    hipStream_t s = ctx->stream, su = ctx->su, sp = ctx->sp;
    MNK_REQUIRE(su && sp, "mnk_debug_update: no look-ahead streams");
    DevBuf<int> ctr;
    int rc = ctr.alloc(8 * (size_t)reps);
    if (rc) return rc;
    MNK_HIP(hipMemsetAsync(ctr.p, 0, 8 * (size_t)reps * sizeof(int), s));
    hipEvent_t e0, e1;
    MNK_HIP(hipEventCreate(&e0));
    MNK_HIP(hipEventCreate(&e1));
    const int pcus = ctx->panel_cus, ucus = ctx->num_cu - pcus;
    MNK_HIP(hipEventRecord(e0, s));
    MNK_HIP(hipEventRecord(ctx->ev_a, s));
    MNK_HIP(hipStreamWaitEvent(su, ctx->ev_a, 0));
    MNK_HIP(hipStreamWaitEvent(sp, ctx->ev_a, 0));
    for (int r = 0; r < reps && !rc; ++r) {
        switch (variant) {
        case 0: rc = launch_gemm_nt(s, 2, M, M, K, A, lda, A, lda, C, ldc, nullptr, nullptr, 0, nullptr); break;
        case 1: rc = launch_gemm_nt_queue(s, M, M, K, A, lda, A, lda, C, ldc, ctr.p + 8 * r, ctx->num_cu, nullptr); break;
        case 2: rc = launch_gemm_nt(su, 2, M, M, K, A, lda, A, lda, C, ldc, nullptr, nullptr, 0, nullptr); break;
        case 3: rc = launch_gemm_nt_queue(su, M, M, K, A, lda, A, lda, C, ldc, ctr.p + 8 * r, ucus, nullptr); break;
        case 4:
            rc = launch_gemm_nt_queue(su, M, M, K, A, lda, A, lda, C, ldc, ctr.p + 8 * r, ucus, nullptr);
            if (!rc) rc = launch_gemm_nt_queue(sp, M, M, K, A, lda, A, lda, C, ldc, ctr.p + 8 * r, pcus, nullptr);
            break;
        case 20: break;  // probe only
        case 21: rc = launch_gemm_nt(su, 2, M, M, K, A, lda, A, lda, C, ldc, nullptr, nullptr, 0, nullptr); break;
        case 9: rc = launch_gemm_nt_dbg(s, 1, M, M, K, A, lda, A, lda, C, ldc); break;
        case 10: rc = launch_gemm_nt_dbg(su, 1, M, M, K, A, lda, A, lda, C, ldc); break;
        case 11: rc = launch_gemm_nt_dbg(su, 2, M, M, K, A, lda, A, lda, C, ldc); break;  // no k-loop barrier
        case 12: rc = launch_gemm_nt_dbg(su, 3, M, M, K, A, lda, A, lda, C, ldc); break;  // MFMA + LDS reads only
        case 13: rc = launch_gemm_nt_dbg(su, 5, M, M, K, A, lda, A, lda, C, ldc); break;  // register staging
        case 14: rc = launch_gemm_nt_dbg(s, 5, M, M, K, A, lda, A, lda, C, ldc); break;
        case 15: rc = launch_gemm_nt_dbg(su, 6, M, M, K, A, lda, A, lda, C, ldc); break;  // BK = 16, 2 workgroups/CU
        case 16: rc = launch_gemm_nt_dbg(s, 6, M, M, K, A, lda, A, lda, C, ldc); break;
        case 6: case 7: case 8: {  // static tiling on the context stream, one launch per `chunk` tiles
            const int chunk = variant == 6 ? 2 * ctx->num_cu : variant == 7 ? 4 * ctx->num_cu : ctx->num_cu;
            const int nt = gemm_nt_lower_tiles(M, M);
            for (int t = 0; t < nt && !rc; t += chunk)
                rc = launch_gemm_nt_lower_range(s, M, M, K, A, lda, A, lda, C, ldc, nullptr, t, chunk);
            break;
        }
        default:
            rc = launch_gemm_nt(su, 2, M, M, K, A, lda, A, lda, C, ldc, nullptr, nullptr, 0, nullptr);
            if (!rc && C2) rc = launch_gemm_nt(sp, 2, M / 2, M / 2, K, A, lda, A, lda, C2, ldc, nullptr, nullptr, 0, nullptr);
            break;
        }
    }
    DevBuf<unsigned long long> probe;
    if (variant == 20 || variant == 21) {
        rc |= probe.alloc(2);
        if (!rc) hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, sp, probe.p, 12000, 0.999999, 1e-9);
    }
    MNK_HIP(hipEventRecord(ctx->ev_a, sp));
    MNK_HIP(hipEventRecord(ctx->ev_b, su));
    MNK_HIP(hipStreamWaitEvent(s, ctx->ev_a, 0));
    MNK_HIP(hipStreamWaitEvent(s, ctx->ev_b, 0));
    MNK_HIP(hipEventRecord(e1, s));
    MNK_HIP(mnk::stream_wait(s));
    float t = 0.f;
    MNK_HIP(hipEventElapsedTime(&t, e0, e1));
    *ms = (double)t;
    if (variant == 20 || variant == 21) {  // report the probe's ticks (10 ns each) instead
        unsigned long long h = 0;
        { mnk::H2DGuard guard; MNK_HIP(hipMemcpy(&h, probe.p, sizeof(h), hipMemcpyDeviceToHost)); }
        *ms = (double)h;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

}  // extern "C"
