// Device-side IPM scalar reductions over device-resident iterates (second slice): the regular-phase
// functions of reference src/IPM/kernels.jl:263-388,675-695 (GPU twins: lib/MadNLPGPU/src/IPM/kernels.jl:4-116, which
// are mapreduce calls over the same expressions).  HBM-bound one-pass reductions: algorithmic bytes = 8 x (vectors read)
// per element; two launches (grid-stride partials in a fixed order -> one block) + one 8/16-byte D2H per call, so results
// are deterministic run to run.  Expressions are evaluated exactly as the reference writes them (no FMA contraction):
// max/min-type results are bit-identical to the host restatement, sum-type results agree to summation-order rounding.
#pragma clang fp contract(off)
#include <cfloat>
#include <cmath>
#include <functional>
#include <utility>

#include "common.h"

using namespace mnk;

struct mnk_ipm {
    mnk_ctx* ctx = nullptr;
    int64_t ntot = 0, nlb = 0, nub = 0;
    int64_t nllb = 0, nuub = 0;
    DevBuf<int64_t> ind_lb, ind_ub, ind_llb, ind_uub;
    DevBuf<double> gemv_part;   // column-slab partial sums of mnk_ipm_gemv (grown on demand)
    DevBuf<double> part;   // IPM_SLOTS x IPM_BLOCKS partials
    double* pin = nullptr;     // IPM_SLOTS pinned, device-mapped host words: the final reduction stores its result here
    double* pin_dev = nullptr;
    // batch mode (mnk_ipm_batch_begin / _end): the get_* calls only enqueue their reductions into successive slots and
    // leave a finalizer behind; ONE synchronization at batch_end, then every deferred `out` receives its value
    bool batching = false;
    int next_slot = 0;
    std::vector<std::pair<int, std::function<void(const double*)>>> pending;  // (first slot, finalizer)
};

namespace {

constexpr int IPM_BLOCKS = 256;
constexpr int IPM_SLOTS = 32;  // reductions in flight: up to four per call, several calls per batch
constexpr int IPM_THREADS = 256;
enum { R_SUM = 0, R_MAX = 1, R_MIN = 2 };

template <int RED>
__device__ __forceinline__ double red_id() { return RED == R_SUM ? 0.0 : (RED == R_MAX ? -INFINITY : INFINITY); }
template <int RED>
__device__ __forceinline__ double red_op(double a, double b) {
    return RED == R_SUM ? a + b : (RED == R_MAX ? fmax(a, b) : fmin(a, b));
}
// NaN must propagate like the reference's max(a, b) / min(a, b) (Julia: NaN if either is NaN)
template <int RED>
__device__ __forceinline__ double red_op_nan(double a, double b) {
    if (RED != R_SUM && (a != a || b != b)) return NAN;
    return red_op<RED>(a, b);
}

template <int RED>
__device__ __forceinline__ void block_reduce_store(double v, double* __restrict__ out) {
    __shared__ double sh[IPM_THREADS];
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int s = IPM_THREADS / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] = red_op_nan<RED>(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = sh[0];
}

template <int RED>
__global__ __launch_bounds__(IPM_THREADS) void final_reduce_kernel(const double* __restrict__ part, int nparts,
                                                                     double* __restrict__ res) {
    double v = red_id<RED>();
    for (int i = threadIdx.x; i < nparts; i += IPM_THREADS) v = red_op_nan<RED>(v, part[i]);
    __shared__ double sh[IPM_THREADS];
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int s = IPM_THREADS / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] = red_op_nan<RED>(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0)
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(res), (unsigned long long)__double_as_longlong(sh[0]),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- element functors.  side 0: lower-bounded entries (index set ind_lb), side 1: upper-bounded (ind_ub) ----------
// get_varphi (kernels.jl:263-283): -mu log(x_lr - xl_r) resp. -mu log(xu_r - x_ur), Inf when the slack is negative
struct FVarphi {
    const double *x, *xb; const int64_t* ind; double mu; int upper;
    __device__ double operator()(int64_t i) const {
        const int64_t p = ind[i];
        const double d = upper ? xb[p] - x[p] : x[p] - xb[p];
        return d < 0 ? INFINITY : -mu * log(d);
    }
};
// get_inf_du (:285-291): |f - zl + zu + jacl|
struct FInfDu {
    const double *f, *zl, *zu, *jacl;
    __device__ double operator()(int64_t i) const { return fabs(f[i] - zl[i] + zu[i] + jacl[i]); }
};
// get_inf_compl (:293-303): |(x_lr - xl_r) zl_r - mu| resp. |(xu_r - x_ur) zu_r - mu|
struct FCompl {
    const double *x, *xb, *z; const int64_t* ind; double mu; int upper; int absolute;
    __device__ double operator()(int64_t i) const {
        const int64_t p = ind[i];
        const double d = upper ? xb[p] - x[p] : x[p] - xb[p];
        const double c = d * z[p];
        return absolute ? fabs(c - mu) : c;   // absolute = 0: the complementarity product itself (min / average)
    }
};
// get_varphi_d (:341-354): (f - mu/(x - xl) + mu/(xu - x)) dx
struct FVarphiD {
    const double *f, *x, *xl, *xu, *dx; double mu;
    __device__ double operator()(int64_t i) const { return (f[i] - mu / (x[i] - xl[i]) + mu / (xu[i] - x[i])) * dx[i]; }
};
// get_alpha_max (:356-371)
struct FAlphaMax {
    const double *x, *xl, *xu, *dx; double tau;
    __device__ double operator()(int64_t i) const {
        const double a = dx[i] < 0 ? (-x[i] + xl[i]) * tau / dx[i] : INFINITY;
        const double b = dx[i] > 0 ? (-x[i] + xu[i]) * tau / dx[i] : INFINITY;
        return fmin(a, b);
    }
};
// get_alpha_z (:373-388): dz is the bound-length step, z the full-length multiplier
struct FAlphaZ {
    const double *z, *dz; const int64_t* ind; double tau;
    __device__ double operator()(int64_t i) const { return dz[i] < 0 ? (-z[ind[i]]) * tau / dz[i] : INFINITY; }
};
// get_rel_search_norm (:675-682)
struct FRelNorm {
    const double *x, *dx;
    __device__ double operator()(int64_t i) const { return fabs(dx[i]) / (1.0 + fabs(x[i])); }
};
// norm(v, 1) / norm(v, Inf) pieces, optionally gathered
struct FAbs {
    const double* v; const int64_t* ind;
    __device__ double operator()(int64_t i) const { return fabs(ind ? v[ind[i]] : v[i]); }
};

template <int RED, class F>
__global__ __launch_bounds__(IPM_THREADS) void map_reduce_kernel(F f, int64_t n, double* __restrict__ part) {
    double v = red_id<RED>();
    for (int64_t i = blockIdx.x * (int64_t)IPM_THREADS + threadIdx.x; i < n; i += (int64_t)IPM_BLOCKS * IPM_THREADS)
        v = red_op_nan<RED>(v, f(i));
    block_reduce_store<RED>(v, part);
}

// enqueue "res[slot] = reduce_{i < n} f(i)" (identity when n == 0)
template <int RED, class F>
int enqueue(mnk_ipm* h, F f, int64_t n, int slot) {
    hipStream_t s = h->ctx->stream;
    double* part = h->part.p + slot * IPM_BLOCKS;
    hipLaunchKernelGGL((map_reduce_kernel<RED, F>), dim3(IPM_BLOCKS), dim3(IPM_THREADS), 0, s, f, n, part);
    hipLaunchKernelGGL((final_reduce_kernel<RED>), dim3(1), dim3(IPM_THREADS), 0, s, part, IPM_BLOCKS, h->pin_dev + slot);
    MNK_HIP(hipGetLastError());
    return 0;
}

inline double max0(double r) { return r != r ? r : fmax(0.0, r); }  // reference: max(zero, ...), NaN propagates

// first slot of a call that needs `count` reductions (-1: the batch is full)
int slot_base(mnk_ipm* h, int count) {
    if (!h->batching) return 0;
    if (h->next_slot + count > IPM_SLOTS) {
        set_error("mnk_ipm: more than %d reductions in one batch", IPM_SLOTS);
        return -1;
    }
    const int b = h->next_slot;
    h->next_slot += count;
    return b;
}

// Outside a batch: wait for the stream, read the `count` results that start at slot `base` and run the finalizer (which
// writes the caller's `out`).  Inside a batch: keep the finalizer for mnk_ipm_batch_end.
template <class Fin>
int finish(mnk_ipm* h, int base, int count, Fin fin) {
    if (h->batching) {
        h->pending.emplace_back(base, std::function<void(const double*)>(fin));
        return 0;
    }
    // the final reductions stored their results straight into pinned, device-mapped host memory
    MNK_HIP(mnk::stream_wait(h->ctx->stream));
    volatile double* pw = h->pin + base;
    double r[4];
    for (int i = 0; i < count; ++i) r[i] = pw[i];
    fin(r);
    return 0;
}

}  // namespace

extern "C" {

int mnk_ipm_create(mnk_ctx* ctx, int64_t ntot, int64_t nlb, const int64_t* ind_lb, int64_t nub, const int64_t* ind_ub,
                   int index_base, mnk_ipm** out) {
    MNK_REQUIRE(ctx && out && ntot > 0 && nlb >= 0 && nub >= 0, "mnk_ipm_create: bad argument");
    MNK_REQUIRE((nlb == 0 || ind_lb) && (nub == 0 || ind_ub) && (index_base == 0 || index_base == 1),
                "mnk_ipm_create: bad index sets");
    MNK_HIP(hipSetDevice(ctx->device));
    std::vector<int64_t> lb(nlb), ub(nub);
    for (int64_t i = 0; i < nlb; ++i) {
        lb[i] = ind_lb[i] - index_base;
        MNK_REQUIRE(lb[i] >= 0 && lb[i] < ntot, "mnk_ipm_create: ind_lb out of range");
    }
    for (int64_t i = 0; i < nub; ++i) {
        ub[i] = ind_ub[i] - index_base;
        MNK_REQUIRE(ub[i] >= 0 && ub[i] < ntot, "mnk_ipm_create: ind_ub out of range");
    }
    mnk_ipm* h = new mnk_ipm();
    h->ctx = ctx;
    h->ntot = ntot; h->nlb = nlb; h->nub = nub;
    int rc = h->ind_lb.upload(lb, ctx->stream) | h->ind_ub.upload(ub, ctx->stream) | h->part.alloc(IPM_SLOTS * IPM_BLOCKS);
    if (!rc && (hipHostMalloc((void**)&h->pin, IPM_SLOTS * sizeof(double), hipHostMallocMapped) != hipSuccess ||
                hipHostGetDevicePointer((void**)&h->pin_dev, h->pin, 0) != hipSuccess)) {
        (void)hipGetLastError();
        rc = -2;
    }
    if (rc) { delete h; return rc; }
    mnk_ctx_child_added(ctx);
    *out = h;
    return 0;
}

int mnk_ipm_destroy(mnk_ipm* h) {
    if (!h) return 0;
    (void)hipSetDevice(h->ctx->device);
    (void)mnk::stream_wait(h->ctx->stream);
    mnk_ctx* ctx = h->ctx;
    if (h->pin) (void)hipHostFree(h->pin);
    delete h;
    mnk_ctx_child_gone(ctx);
    return 0;
}

#define IPM_ENTER(h, who)                                              \
    MNK_REQUIRE((h) != nullptr && out != nullptr, who ": NULL argument"); \
    MNK_HIP(hipSetDevice((h)->ctx->device))

int mnk_ipm_batch_begin(mnk_ipm* h) {
    MNK_REQUIRE(h != nullptr && !h->batching, "mnk_ipm_batch_begin: NULL handle or a batch is already open");
    h->batching = true;
    h->next_slot = 0;
    h->pending.clear();
    return 0;
}

int mnk_ipm_batch_end(mnk_ipm* h) {
    MNK_REQUIRE(h != nullptr && h->batching, "mnk_ipm_batch_end: no open batch");
    h->batching = false;
    MNK_HIP(hipSetDevice(h->ctx->device));
    MNK_HIP(mnk::stream_wait(h->ctx->stream));
    double r[IPM_SLOTS];
    volatile double* pw = h->pin;
    for (int i = 0; i < h->next_slot; ++i) r[i] = pw[i];
    for (auto& pf : h->pending) pf.second(r + pf.first);
    h->pending.clear();
    h->next_slot = 0;
    return 0;
}

#define IPM_BASE(cnt)                 \
    const int b = slot_base(h, cnt);  \
    if (b < 0) return -1

int mnk_ipm_get_varphi(mnk_ipm* h, double obj_val, const double* x, const double* xl, const double* xu, double mu,
                       double* out) {
    IPM_ENTER(h, "mnk_ipm_get_varphi");
    IPM_BASE(2);
    int rc = enqueue<R_SUM>(h, FVarphi{x, xl, h->ind_lb.p, mu, 0}, h->nlb, b) |
             enqueue<R_SUM>(h, FVarphi{x, xu, h->ind_ub.p, mu, 1}, h->nub, b + 1);
    if (rc) return rc;
    return finish(h, b, 2, [=](const double* r) { *out = obj_val + r[0] + r[1]; });
}

int mnk_ipm_get_inf_du(mnk_ipm* h, const double* f, const double* zl, const double* zu, const double* jacl, double sd,
                       double* out) {
    IPM_ENTER(h, "mnk_ipm_get_inf_du");
    IPM_BASE(1);
    int rc = enqueue<R_MAX>(h, FInfDu{f, zl, zu, jacl}, h->ntot, b);
    if (rc) return rc;
    return finish(h, b, 1, [=](const double* r) { *out = max0(r[0]) / sd; });
}

int mnk_ipm_get_inf_compl(mnk_ipm* h, const double* x, const double* xl, const double* xu, const double* zl,
                          const double* zu, double mu, double sc, double* out) {
    IPM_ENTER(h, "mnk_ipm_get_inf_compl");
    IPM_BASE(2);
    int rc = enqueue<R_MAX>(h, FCompl{x, xl, zl, h->ind_lb.p, mu, 0, 1}, h->nlb, b) |
             enqueue<R_MAX>(h, FCompl{x, xu, zu, h->ind_ub.p, mu, 1, 1}, h->nub, b + 1);
    if (rc) return rc;
    return finish(h, b, 2, [=](const double* r) { *out = ((r[0] != r[0] || r[1] != r[1]) ? NAN : max0(fmax(r[0], r[1]))) / sc; });
}

int mnk_ipm_get_min_complementarity(mnk_ipm* h, const double* x, const double* xl, const double* xu, const double* zl,
                                    const double* zu, double* out) {
    IPM_ENTER(h, "mnk_ipm_get_min_complementarity");
    IPM_BASE(2);
    int rc = enqueue<R_MIN>(h, FCompl{x, xl, zl, h->ind_lb.p, 0.0, 0, 0}, h->nlb, b) |
             enqueue<R_MIN>(h, FCompl{x, xu, zu, h->ind_ub.p, 0.0, 1, 0}, h->nub, b + 1);
    if (rc) return rc;
    return finish(h, b, 2, [=](const double* r) { *out = (r[0] != r[0] || r[1] != r[1]) ? NAN : fmin(r[0], r[1]); });
}

int mnk_ipm_get_average_complementarity(mnk_ipm* h, const double* x, const double* xl, const double* xu, const double* zl,
                                        const double* zu, double* out) {
    IPM_ENTER(h, "mnk_ipm_get_average_complementarity");
    if (h->nlb + h->nub == 0) { *out = 0.0; return 0; }
    IPM_BASE(2);
    int rc = enqueue<R_SUM>(h, FCompl{x, xl, zl, h->ind_lb.p, 0.0, 0, 0}, h->nlb, b) |
             enqueue<R_SUM>(h, FCompl{x, xu, zu, h->ind_ub.p, 0.0, 1, 0}, h->nub, b + 1);
    if (rc) return rc;
    const double cnt = (double)(h->nlb + h->nub);
    return finish(h, b, 2, [=](const double* r) { *out = (r[0] + r[1]) / cnt; });
}

int mnk_ipm_get_varphi_d(mnk_ipm* h, const double* f, const double* x, const double* xl, const double* xu,
                         const double* dx, double mu, double* out) {
    IPM_ENTER(h, "mnk_ipm_get_varphi_d");
    IPM_BASE(1);
    int rc = enqueue<R_SUM>(h, FVarphiD{f, x, xl, xu, dx, mu}, h->ntot, b);
    if (rc) return rc;
    return finish(h, b, 1, [=](const double* r) { *out = r[0]; });
}

int mnk_ipm_get_alpha_max(mnk_ipm* h, const double* x, const double* xl, const double* xu, const double* dx, double tau,
                          double* out) {
    IPM_ENTER(h, "mnk_ipm_get_alpha_max");
    IPM_BASE(1);
    int rc = enqueue<R_MIN>(h, FAlphaMax{x, xl, xu, dx, tau}, h->ntot, b);
    if (rc) return rc;
    return finish(h, b, 1, [=](const double* r) { *out = r[0] != r[0] ? r[0] : fmin(1.0, r[0]); });
}

int mnk_ipm_get_alpha_z(mnk_ipm* h, const double* zl, const double* zu, const double* dzl, const double* dzu, double tau,
                        double* out) {
    IPM_ENTER(h, "mnk_ipm_get_alpha_z");
    IPM_BASE(2);
    int rc = enqueue<R_MIN>(h, FAlphaZ{zl, dzl, h->ind_lb.p, tau}, h->nlb, b) |
             enqueue<R_MIN>(h, FAlphaZ{zu, dzu, h->ind_ub.p, tau}, h->nub, b + 1);
    if (rc) return rc;
    return finish(h, b, 2, [=](const double* r) {
        const double m = fmin(r[0], r[1]);
        *out = (r[0] != r[0] || r[1] != r[1]) ? NAN : fmin(1.0, m);
    });
}

int mnk_ipm_get_rel_search_norm(mnk_ipm* h, const double* x, const double* dx, double* out) {
    IPM_ENTER(h, "mnk_ipm_get_rel_search_norm");
    IPM_BASE(1);
    int rc = enqueue<R_MAX>(h, FRelNorm{x, dx}, h->ntot, b);
    if (rc) return rc;
    return finish(h, b, 1, [=](const double* r) { *out = max0(r[0]); });
}

// get_sd / get_sc (kernels.jl:684-695): l = the constraint multipliers (m entries), zl / zu full-length
int mnk_ipm_get_sd_sc(mnk_ipm* h, const double* l, int64_t m, const double* zl, const double* zu, double s_max,
                      double* out /* [sd, sc] */) {
    IPM_ENTER(h, "mnk_ipm_get_sd_sc");
    MNK_REQUIRE(m >= 0, "mnk_ipm_get_sd_sc: bad size");
    IPM_BASE(3);
    int rc = enqueue<R_SUM>(h, FAbs{zl, h->ind_lb.p}, h->nlb, b) | enqueue<R_SUM>(h, FAbs{zu, h->ind_ub.p}, h->nub, b + 1) |
             enqueue<R_SUM>(h, FAbs{l, nullptr}, m, b + 2);
    if (rc) return rc;
    const double cnt_d = (double)std::max<int64_t>(1, m + h->nlb + h->nub), cnt_c = (double)std::max<int64_t>(1, h->nlb + h->nub);
    return finish(h, b, 3, [=](const double* r) {
        const double nz = r[0] + r[1];
        out[0] = fmax(s_max, (r[2] + r[0] + r[1]) / cnt_d) / s_max;
        out[1] = fmax(s_max, nz / cnt_c) / s_max;
    });
}

// get_inf_pr = norm(c, Inf) (kernels.jl:284) and theta = norm(c, 1) (solver.jl get_theta)
int mnk_ipm_get_norms(mnk_ipm* h, const double* c, int64_t m, double* out /* [inf, one] */) {
    IPM_ENTER(h, "mnk_ipm_get_norms");
    MNK_REQUIRE(m >= 0, "mnk_ipm_get_norms: bad size");
    IPM_BASE(2);
    int rc = enqueue<R_MAX>(h, FAbs{c, nullptr}, m, b) | enqueue<R_SUM>(h, FAbs{c, nullptr}, m, b + 1);
    if (rc) return rc;
    return finish(h, b, 2, [=](const double* r) {
        out[0] = max0(r[0]);
        out[1] = r[1];
    });
}

}  // extern "C"

// ---- elementwise pieces of the regular phase (reference src/IPM/kernels.jl:113-131, 656-673, 775-801, 818-823) ------
namespace {
__device__ __forceinline__ double jl_min(double a, double b) { return (a != a || b != b) ? NAN : fmin(a, b); }
__device__ __forceinline__ double jl_max(double a, double b) { return (a != a || b != b) ? NAN : fmax(a, b); }

__global__ void aug_rhs_primal_kernel(double* __restrict__ px, const double* __restrict__ f, const double* __restrict__ zl,
                                      const double* __restrict__ zu, const double* __restrict__ jacl, int64_t n) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) px[i] = -f[i] + zl[i] - zu[i] - jacl[i];
}
__global__ void negate_kernel(double* __restrict__ py, const double* __restrict__ c, int64_t m) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < m) py[i] = -c[i];
}
// pzl = (xl_r - x_lr) zl_r + mu (upper = 0) ; pzu = (xu_r - x_ur) zu_r - mu (upper = 1)
__global__ void aug_rhs_bound_kernel(double* __restrict__ pz, const double* __restrict__ x, const double* __restrict__ xb,
                                     const double* __restrict__ z, const int64_t* __restrict__ ind, double mu, int64_t nb,
                                     int upper) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= nb) return;
    const int64_t p = ind[i];
    const double t = (xb[p] - x[p]) * z[p];
    pz[i] = upper ? t - mu : t + mu;
}
__global__ void shift_gather_kernel(double* __restrict__ v, const int64_t* __restrict__ ind, double delta, int64_t nb) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < nb) v[ind[i]] += delta;
}
// adjust_boundary!: xl_r = (x_lr - xl_r < c1) ? xl_r - c2 max(1, |x_lr|) : xl_r ; xu_r = (xu_r - x_ur < c1) ? xu_r + c2 max(1, |x_ur|) : xu_r
__global__ void adjust_boundary_kernel(double* __restrict__ xb, const double* __restrict__ x, const int64_t* __restrict__ ind,
                                       double c1, double c2, int64_t nb, int upper) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= nb) return;
    const int64_t p = ind[i];
    const double xv = x[p], b = xb[p];
    const double slack = upper ? b - xv : xv - b;
    if (slack < c1) xb[p] = upper ? b + c2 * fmax(1.0, fabs(xv)) : b - c2 * fmax(1.0, fabs(xv));
}
// reset_bound_dual!(z, x1, x2, mu, kappa_sigma) on full-length vectors: z = max(min(z, (ks mu)/(x1-x2)), (mu/ks)/(x1-x2))
__global__ void reset_bound_dual_kernel(double* __restrict__ z, const double* __restrict__ x1, const double* __restrict__ x2,
                                        double ksmu, double muks, int64_t n) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double d = x1[i] - x2[i];
    z[i] = jl_max(jl_min(z[i], ksmu / d), muks / d);
}
}  // namespace

extern "C" {

#define IPM_G(cnt) dim3((unsigned)(((cnt) + 255) / 256)), dim3(256), 0, h->ctx->stream

int mnk_ipm_set_perturbation_sets(mnk_ipm* h, int64_t nllb, const int64_t* ind_llb, int64_t nuub, const int64_t* ind_uub,
                                  int index_base) {
    MNK_REQUIRE(h && nllb >= 0 && nuub >= 0 && (nllb == 0 || ind_llb) && (nuub == 0 || ind_uub) &&
                    (index_base == 0 || index_base == 1), "mnk_ipm_set_perturbation_sets: bad argument");
    MNK_HIP(hipSetDevice(h->ctx->device));
    std::vector<int64_t> a(nllb), b(nuub);
    for (int64_t i = 0; i < nllb; ++i) {
        a[i] = ind_llb[i] - index_base;
        MNK_REQUIRE(a[i] >= 0 && a[i] < h->ntot, "mnk_ipm_set_perturbation_sets: ind_llb out of range");
    }
    for (int64_t i = 0; i < nuub; ++i) {
        b[i] = ind_uub[i] - index_base;
        MNK_REQUIRE(b[i] >= 0 && b[i] < h->ntot, "mnk_ipm_set_perturbation_sets: ind_uub out of range");
    }
    h->nllb = nllb; h->nuub = nuub;
    return h->ind_llb.upload(a, h->ctx->stream) | h->ind_uub.upload(b, h->ctx->stream);
}

// set_aug_rhs!(solver, kkt, c, mu) (kernels.jl:113-131): the four blocks of the right-hand side p
int mnk_ipm_set_aug_rhs(mnk_ipm* h, const double* f, const double* zl, const double* zu, const double* jacl,
                        const double* c, int64_t m, const double* x, const double* xl, const double* xu, double mu,
                        double* px, double* py, double* pzl, double* pzu) {
    MNK_REQUIRE(h && f && zl && zu && jacl && x && xl && xu && px && (m == 0 || (c && py)) && (h->nlb == 0 || pzl) &&
                    (h->nub == 0 || pzu), "mnk_ipm_set_aug_rhs: NULL argument");
    MNK_HIP(hipSetDevice(h->ctx->device));
    hipLaunchKernelGGL(aug_rhs_primal_kernel, IPM_G(h->ntot), px, f, zl, zu, jacl, h->ntot);
    if (m > 0) hipLaunchKernelGGL(negate_kernel, IPM_G(m), py, c, m);
    if (h->nlb > 0) hipLaunchKernelGGL(aug_rhs_bound_kernel, IPM_G(h->nlb), pzl, x, xl, zl, h->ind_lb.p, mu, h->nlb, 0);
    if (h->nub > 0) hipLaunchKernelGGL(aug_rhs_bound_kernel, IPM_G(h->nub), pzu, x, xu, zu, h->ind_ub.p, mu, h->nub, 1);
    MNK_HIP(hipGetLastError());
    return 0;
}

// dual_inf_perturbation!(px, ind_llb, ind_uub, mu, kappa_d) (kernels.jl:818-823)
int mnk_ipm_dual_inf_perturbation(mnk_ipm* h, double* px, double mu, double kappa_d) {
    MNK_REQUIRE(h && px, "mnk_ipm_dual_inf_perturbation: NULL argument");
    MNK_HIP(hipSetDevice(h->ctx->device));
    if (h->nllb > 0) hipLaunchKernelGGL(shift_gather_kernel, IPM_G(h->nllb), px, h->ind_llb.p, -(mu * kappa_d), h->nllb);
    if (h->nuub > 0) hipLaunchKernelGGL(shift_gather_kernel, IPM_G(h->nuub), px, h->ind_uub.p, mu * kappa_d, h->nuub);
    MNK_HIP(hipGetLastError());
    return 0;
}

// adjust_boundary!(x_lr, xl_r, x_ur, xu_r, mu) (kernels.jl:656-673), xl / xu full-length, updated in place
int mnk_ipm_adjust_boundary(mnk_ipm* h, const double* x, double* xl, double* xu, double mu) {
    MNK_REQUIRE(h && x && xl && xu, "mnk_ipm_adjust_boundary: NULL argument");
    MNK_HIP(hipSetDevice(h->ctx->device));
    const double c1 = DBL_EPSILON * mu, c2 = pow(DBL_EPSILON, 0.75);
    if (h->nlb > 0) hipLaunchKernelGGL(adjust_boundary_kernel, IPM_G(h->nlb), xl, x, h->ind_lb.p, c1, c2, h->nlb, 0);
    if (h->nub > 0) hipLaunchKernelGGL(adjust_boundary_kernel, IPM_G(h->nub), xu, x, h->ind_ub.p, c1, c2, h->nub, 1);
    MNK_HIP(hipGetLastError());
    return 0;
}

// the two reset_bound_dual! calls of the accepted step (solver.jl:280-291), full primal-length vectors (unbounded
// entries carry -Inf / +Inf bounds and come out as exactly 0, as in the reference)
int mnk_ipm_reset_bound_dual(mnk_ipm* h, double* zl, double* zu, const double* x, const double* xl, const double* xu,
                             double mu, double kappa_sigma) {
    MNK_REQUIRE(h && zl && zu && x && xl && xu, "mnk_ipm_reset_bound_dual: NULL argument");
    MNK_HIP(hipSetDevice(h->ctx->device));
    hipLaunchKernelGGL(reset_bound_dual_kernel, IPM_G(h->ntot), zl, x, xl, kappa_sigma * mu, mu / kappa_sigma, h->ntot);
    hipLaunchKernelGGL(reset_bound_dual_kernel, IPM_G(h->ntot), zu, xu, x, kappa_sigma * mu, mu / kappa_sigma, h->ntot);
    MNK_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"

// =====================================================================================================================
// Restoration phase (robust restorer), SURVEY 8(f).4 third slice: reference src/IPM/kernels.jl:390-636 (GPU twins
// lib/MadNLPGPU/src/IPM/kernels.jl:117-462), the elementwise pieces :72-110,133-158,206-257,638-654,775-786,825-829 and the
// vector part of initialize_robust_restorer! (src/IPM/restoration.jl:39-76).  pp / nn / zp / zn / dpp / ... are the
// m-vectors of the RobustRestorer, everything else is full primal length (ntot) unless noted.
// =====================================================================================================================
namespace {

struct FObjPN {  // rho (p + n)
    const double *p, *n; double rho;
    __device__ double operator()(int64_t i) const { return rho * (p[i] + n[i]); }
};
struct FObjProx {  // zeta/2 D_R^2 (x - x_ref)^2
    const double *D, *x, *xr; double zeta;
    __device__ double operator()(int64_t i) const {
        const double d = x[i] - xr[i];
        return zeta / 2 * (D[i] * D[i]) * (d * d);
    }
};
struct FCPN {  // |c - p + n|
    const double *c, *p, *n;
    __device__ double operator()(int64_t i) const { return fabs(c[i] - p[i] + n[i]); }
};
struct FRhoL {  // |rho - l - zp| (plus = 0) / |rho + l - zn| (plus = 1)
    const double *l, *z; double rho; int plus;
    __device__ double operator()(int64_t i) const { return plus ? fabs(rho + l[i] - z[i]) : fabs(rho - l[i] - z[i]); }
};
struct FProdMu {  // |a z - mu|
    const double *a, *z; double mu;
    __device__ double operator()(int64_t i) const { return fabs(a[i] * z[i] - mu); }
};
struct FStepRatio {  // dv < 0 ? -v tau / dv : Inf
    const double *v, *dv; double tau;
    __device__ double operator()(int64_t i) const { return dv[i] < 0 ? (-v[i]) * tau / dv[i] : INFINITY; }
};
struct FLogBar {  // d = x_lr - xl_r (upper = 0) / xu_r - x_ur (upper = 1): d < 0 ? Inf : mu log(d)
    const double *x, *xb; const int64_t* ind; double mu; int upper;
    __device__ double operator()(int64_t i) const {
        const int64_t p = ind[i];
        const double d = upper ? xb[p] - x[p] : x[p] - xb[p];
        return d < 0 ? INFINITY : mu * log(d);
    }
};
struct FLog1 {  // v < 0 ? Inf : mu log(v)
    const double* v; double mu;
    __device__ double operator()(int64_t i) const { return v[i] < 0 ? INFINITY : mu * log(v[i]); }
};
struct FSumDu {  // |f - zl + zu + jacl| (summed by get_F)
    const double *f, *zl, *zu, *jacl;
    __device__ double operator()(int64_t i) const { return fabs(f[i] - zl[i] + zu[i] + jacl[i]); }
};
// get_F (:572-610): lower side (x_lr >= xl_r && zl_r >= 0) ? |(x_lr - xl_r) zl_r - mu| : Inf; the upper side is restated
// as the reference computes it, |(xu_r - xu_r) zu_r - mu| under the guard (xu_r >= x_ur && zu_r >= 0) (:606)
struct FFBound {
    const double *x, *xb, *z; const int64_t* ind; double mu; int upper;
    __device__ double operator()(int64_t i) const {
        const int64_t p = ind[i];
        if (upper) return (xb[p] >= x[p] && z[p] >= 0) ? fabs((xb[p] - xb[p]) * z[p] - mu) : INFINITY;
        return (x[p] >= xb[p] && z[p] >= 0) ? fabs((x[p] - xb[p]) * z[p] - mu) : INFINITY;
    }
};
struct FRhoMu {  // (rho - mu / v) dv
    const double *v, *dv; double mu, rho;
    __device__ double operator()(int64_t i) const { return (rho - mu / v[i]) * dv[i]; }
};

inline double nanmax(double a, double b) { return (a != a || b != b) ? NAN : fmax(a, b); }
inline double nanmin(double a, double b) { return (a != a || b != b) ? NAN : fmin(a, b); }

}  // namespace

extern "C" {

#define IPM_M(who) MNK_REQUIRE(m >= 0, who ": bad size")

int mnk_ipm_get_obj_val_R(mnk_ipm* h, const double* p, const double* n, int64_t m, const double* D_R, const double* x,
                          const double* x_ref, double rho, double zeta, double* out) {
    IPM_ENTER(h, "mnk_ipm_get_obj_val_R");
    IPM_M("mnk_ipm_get_obj_val_R");
    IPM_BASE(2);
    int rc = enqueue<R_SUM>(h, FObjPN{p, n, rho}, m, b) | enqueue<R_SUM>(h, FObjProx{D_R, x, x_ref, zeta}, h->ntot, b + 1);
    if (rc) return rc;
    return finish(h, b, 2, [=](const double* r) { *out = r[0] + r[1]; });
}

int mnk_ipm_get_theta_R(mnk_ipm* h, const double* c, const double* p, const double* n, int64_t m, double* out) {
    IPM_ENTER(h, "mnk_ipm_get_theta_R");
    IPM_M("mnk_ipm_get_theta_R");
    IPM_BASE(1);
    int rc = enqueue<R_SUM>(h, FCPN{c, p, n}, m, b);
    if (rc) return rc;
    return finish(h, b, 1, [=](const double* r) { *out = r[0]; });
}

int mnk_ipm_get_inf_pr_R(mnk_ipm* h, const double* c, const double* p, const double* n, int64_t m, double* out) {
    IPM_ENTER(h, "mnk_ipm_get_inf_pr_R");
    IPM_M("mnk_ipm_get_inf_pr_R");
    IPM_BASE(1);
    int rc = enqueue<R_MAX>(h, FCPN{c, p, n}, m, b);
    if (rc) return rc;
    return finish(h, b, 1, [=](const double* r) { *out = max0(r[0]); });
}

int mnk_ipm_get_inf_du_R(mnk_ipm* h, const double* f_R, const double* l, const double* zl, const double* zu,
                         const double* jacl, const double* zp, const double* zn, int64_t m, double rho, double sd,
                         double* out) {
    IPM_ENTER(h, "mnk_ipm_get_inf_du_R");
    IPM_M("mnk_ipm_get_inf_du_R");
    IPM_BASE(3);
    int rc = enqueue<R_MAX>(h, FInfDu{f_R, zl, zu, jacl}, h->ntot, b) | enqueue<R_MAX>(h, FRhoL{l, zp, rho, 0}, m, b + 1) |
             enqueue<R_MAX>(h, FRhoL{l, zn, rho, 1}, m, b + 2);
    if (rc) return rc;
    return finish(h, b, 3, [=](const double* r) { *out = nanmax(0.0, nanmax(r[0], nanmax(r[1], r[2]))) / sd; });
}

int mnk_ipm_get_inf_compl_R(mnk_ipm* h, const double* x, const double* xl, const double* xu, const double* zl,
                            const double* zu, const double* pp, const double* zp, const double* nn, const double* zn,
                            int64_t m, double mu_R, double sc, double* out) {
    IPM_ENTER(h, "mnk_ipm_get_inf_compl_R");
    IPM_M("mnk_ipm_get_inf_compl_R");
    IPM_BASE(4);
    int rc = enqueue<R_MAX>(h, FCompl{x, xl, zl, h->ind_lb.p, mu_R, 0, 1}, h->nlb, b) |
             enqueue<R_MAX>(h, FCompl{x, xu, zu, h->ind_ub.p, mu_R, 1, 1}, h->nub, b + 1) |
             enqueue<R_MAX>(h, FProdMu{pp, zp, mu_R}, m, b + 2) | enqueue<R_MAX>(h, FProdMu{nn, zn, mu_R}, m, b + 3);
    if (rc) return rc;
    return finish(h, b, 4, [=](const double* r) {
        *out = nanmax(0.0, nanmax(nanmax(r[0], r[1]), nanmax(r[2], r[3]))) / sc;
    });
}

int mnk_ipm_get_alpha_max_R(mnk_ipm* h, const double* x, const double* xl, const double* xu, const double* dx,
                            const double* pp, const double* dpp, const double* nn, const double* dnn, int64_t m,
                            double tau_R, double* out) {
    IPM_ENTER(h, "mnk_ipm_get_alpha_max_R");
    IPM_M("mnk_ipm_get_alpha_max_R");
    IPM_BASE(3);
    int rc = enqueue<R_MIN>(h, FAlphaMax{x, xl, xu, dx, tau_R}, h->ntot, b) |
             enqueue<R_MIN>(h, FStepRatio{pp, dpp, tau_R}, m, b + 1) | enqueue<R_MIN>(h, FStepRatio{nn, dnn, tau_R}, m, b + 2);
    if (rc) return rc;
    return finish(h, b, 3, [=](const double* r) { *out = nanmin(1.0, nanmin(r[0], nanmin(r[1], r[2]))); });
}

int mnk_ipm_get_alpha_z_R(mnk_ipm* h, const double* zl, const double* zu, const double* dzl, const double* dzu,
                          const double* zp, const double* dzp, const double* zn, const double* dzn, int64_t m,
                          double tau_R, double* out) {
    IPM_ENTER(h, "mnk_ipm_get_alpha_z_R");
    IPM_M("mnk_ipm_get_alpha_z_R");
    IPM_BASE(4);
    int rc = enqueue<R_MIN>(h, FAlphaZ{zl, dzl, h->ind_lb.p, tau_R}, h->nlb, b) |
             enqueue<R_MIN>(h, FAlphaZ{zu, dzu, h->ind_ub.p, tau_R}, h->nub, b + 1) |
             enqueue<R_MIN>(h, FStepRatio{zp, dzp, tau_R}, m, b + 2) | enqueue<R_MIN>(h, FStepRatio{zn, dzn, tau_R}, m, b + 3);
    if (rc) return rc;
    return finish(h, b, 4, [=](const double* r) {
        *out = nanmin(1.0, nanmin(nanmin(r[0], r[1]), nanmin(r[2], r[3])));
    });
}

int mnk_ipm_get_varphi_R(mnk_ipm* h, double obj_val, const double* x, const double* xl, const double* xu, const double* pp,
                         const double* nn, int64_t m, double mu_R, double* out) {
    IPM_ENTER(h, "mnk_ipm_get_varphi_R");
    IPM_M("mnk_ipm_get_varphi_R");
    IPM_BASE(4);
    int rc = enqueue<R_SUM>(h, FLogBar{x, xl, h->ind_lb.p, mu_R, 0}, h->nlb, b) |
             enqueue<R_SUM>(h, FLogBar{x, xu, h->ind_ub.p, mu_R, 1}, h->nub, b + 1) |
             enqueue<R_SUM>(h, FLog1{pp, mu_R}, m, b + 2) | enqueue<R_SUM>(h, FLog1{nn, mu_R}, m, b + 3);
    if (rc) return rc;
    return finish(h, b, 4, [=](const double* r) { *out = obj_val - (r[0] + r[1] + r[2] + r[3]); });
}

int mnk_ipm_get_F(mnk_ipm* h, const double* c, int64_t m, const double* f, const double* zl, const double* zu,
                  const double* jacl, const double* x, const double* xl, const double* xu, double mu, double* out) {
    IPM_ENTER(h, "mnk_ipm_get_F");
    IPM_M("mnk_ipm_get_F");
    IPM_BASE(4);
    int rc = enqueue<R_SUM>(h, FAbs{c, nullptr}, m, b) | enqueue<R_SUM>(h, FSumDu{f, zl, zu, jacl}, h->ntot, b + 1) |
             enqueue<R_SUM>(h, FFBound{x, xl, zl, h->ind_lb.p, mu, 0}, h->nlb, b + 2) |
             enqueue<R_SUM>(h, FFBound{x, xu, zu, h->ind_ub.p, mu, 1}, h->nub, b + 3);
    if (rc) return rc;
    return finish(h, b, 4, [=](const double* r) { *out = r[0] + r[1] + r[2] + r[3]; });
}

int mnk_ipm_get_varphi_d_R(mnk_ipm* h, const double* f_R, const double* x, const double* xl, const double* xu,
                           const double* dx, const double* pp, const double* nn, const double* dpp, const double* dnn,
                           int64_t m, double mu_R, double rho, double* out) {
    IPM_ENTER(h, "mnk_ipm_get_varphi_d_R");
    IPM_M("mnk_ipm_get_varphi_d_R");
    IPM_BASE(3);
    int rc = enqueue<R_SUM>(h, FVarphiD{f_R, x, xl, xu, dx, mu_R}, h->ntot, b) |
             enqueue<R_SUM>(h, FRhoMu{pp, dpp, mu_R, rho}, m, b + 1) | enqueue<R_SUM>(h, FRhoMu{nn, dnn, mu_R, rho}, m, b + 2);
    if (rc) return rc;
    return finish(h, b, 3, [=](const double* r) { *out = r[0] + r[1] + r[2]; });
}

}  // extern "C"

// ---- elementwise pieces of the restoration phase ----------------------------------------------------------------------
namespace {
#define IPM_IDX(cnt) const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; if (i >= (cnt)) return

// populate_RR_nn! (:825-829): nn = t + sqrt(t^2 + mu c / (2 rho)), t = (mu - rho c) / (2 rho)
__global__ void populate_nn_kernel(double* __restrict__ nn, const double* __restrict__ c, double mu, double rho, int64_t m) {
    IPM_IDX(m);
    const double t = (mu - rho * c[i]) / (2 * rho);
    nn[i] = t + sqrt(t * t + mu * c[i] / (2 * rho));
}
// initialize_robust_restorer! (restoration.jl:46-53): x_ref = x, D_R = min(1, 1 / |x_ref|)
__global__ void rr_ref_kernel(double* __restrict__ x_ref, double* __restrict__ D_R, const double* __restrict__ x, int64_t n) {
    IPM_IDX(n);
    const double v = x[i];
    x_ref[i] = v;
    D_R[i] = jl_min(1.0, 1.0 / fabs(v));
}
// (:59-62): populate_RR_nn!, pp = c + nn, zp = mu_R / pp, zn = mu_R / nn
__global__ void rr_slack_kernel(double* __restrict__ nn, double* __restrict__ pp, double* __restrict__ zp,
                                double* __restrict__ zn, const double* __restrict__ c, double mu, double rho, int64_t m) {
    IPM_IDX(m);
    const double t = (mu - rho * c[i]) / (2 * rho);
    const double nv = t + sqrt(t * t + mu * c[i] / (2 * rho));
    const double pv = c[i] + nv;
    nn[i] = nv;
    pp[i] = pv;
    zp[i] = mu / pv;
    zn[i] = mu / nv;
}
// (:69-70): z_r = min(rho, z_r) through the index set
__global__ void cap_gather_kernel(double* __restrict__ z, const int64_t* __restrict__ ind, double rho, int64_t nb) {
    IPM_IDX(nb);
    const int64_t p = ind[i];
    z[p] = jl_min(rho, z[p]);
}
// set_f_RR! (:106-110): f_R = zeta D_R^2 (x - x_ref)
__global__ void f_rr_kernel(double* __restrict__ f_R, const double* __restrict__ D, const double* __restrict__ x,
                            const double* __restrict__ xr, double zeta, int64_t n) {
    IPM_IDX(n);
    f_R[i] = zeta * (D[i] * D[i]) * (x[i] - xr[i]);
}
// set_aug_rhs_RR! (:133-158), py = -c + pp - nn + (mu - (rho - y) pp) / zp - (mu - (rho + y) nn) / zn
__global__ void aug_rhs_rr_dual_kernel(double* __restrict__ py, const double* __restrict__ c, const double* __restrict__ y,
                                       const double* __restrict__ pp, const double* __restrict__ nn,
                                       const double* __restrict__ zp, const double* __restrict__ zn, double mu, double rho,
                                       int64_t m) {
    IPM_IDX(m);
    py[i] = -c[i] + pp[i] - nn[i] + (mu - (rho - y[i]) * pp[i]) / zp[i] - (mu - (rho + y[i]) * nn[i]) / zn[i];
}
// finish_aug_solve_RR! (:251-257)
__global__ void finish_rr_kernel(double* __restrict__ dpp, double* __restrict__ dnn, double* __restrict__ dzp,
                                 double* __restrict__ dzn, const double* __restrict__ l, const double* __restrict__ dl,
                                 const double* __restrict__ pp, const double* __restrict__ nn, const double* __restrict__ zp,
                                 const double* __restrict__ zn, double mu, double rho, int64_t m) {
    IPM_IDX(m);
    const double a = rho - l[i] - dl[i] - zp[i];
    const double b = rho + l[i] + dl[i] - zn[i];
    dzp[i] = a;
    dzn[i] = b;
    dpp[i] = -pp[i] + mu / zp[i] - (pp[i] / zp[i]) * a;
    dnn[i] = -nn[i] + mu / zn[i] - (nn[i] / zn[i]) * b;
}
// reset_bound_dual!(z, x, mu, kappa_sigma) (:775-786)
__global__ void reset_bound_dual1_kernel(double* __restrict__ z, const double* __restrict__ x, double ksmu, double muks,
                                         int64_t n) {
    IPM_IDX(n);
    z[i] = jl_max(jl_min(z[i], ksmu / x[i]), muks / x[i]);
}
// set_initial_bounds! (:206-218)
__global__ void initial_bounds_kernel(double* __restrict__ xl, double* __restrict__ xu, double tol, int64_t n) {
    IPM_IDX(n);
    const double l = xl[i], u = xu[i];
    xl[i] = l - jl_max(1.0, fabs(l)) * tol;
    xu[i] = u + jl_max(1.0, fabs(u)) * tol;
}
// set_initial_rhs! (:220-230): px = -f + zl - zu
__global__ void initial_rhs_kernel(double* __restrict__ px, const double* __restrict__ f, const double* __restrict__ zl,
                                   const double* __restrict__ zu, int64_t n) {
    IPM_IDX(n);
    px[i] = -f[i] + zl[i] - zu[i];
}
__global__ void zero_kernel(double* __restrict__ v, int64_t n) {
    IPM_IDX(n);
    v[i] = 0.0;
}
// set_g_ifr! (:242-248): g = f - mu / (x - xl) + mu / (xu - x) + jacl
__global__ void g_ifr_kernel(double* __restrict__ g, const double* __restrict__ f, const double* __restrict__ x,
                             const double* __restrict__ xl, const double* __restrict__ xu, const double* __restrict__ jacl,
                             double mu, int64_t n) {
    IPM_IDX(n);
    g[i] = f[i] - mu / (x[i] - xl[i]) + mu / (xu[i] - x[i]) + jacl[i];
}
// _initialize_variables! (:638-650)
__global__ void initialize_variables_kernel(double* __restrict__ x, const double* __restrict__ xl,
                                            const double* __restrict__ xu, double bound_push, double bound_fac, int64_t n) {
    IPM_IDX(n);
    const double l = xl[i], u = xu[i], v = x[i];
    const bool hl = l != -INFINITY, hu = u != INFINITY;
    if (hl && hu)
        x[i] = jl_min(u - jl_min(bound_push * jl_max(1.0, fabs(u)), bound_fac * (u - l)),
                      jl_max(l + jl_min(bound_push * jl_max(1.0, fabs(l)), bound_fac * (u - l)), v));
    else if (hl && !hu)
        x[i] = jl_max(l + bound_push * jl_max(1.0, fabs(l)), v);
    else if (!hl && hu)
        x[i] = jl_min(u - bound_push * jl_max(1.0, fabs(u)), v);
}
}  // namespace

extern "C" {

#define IPM_VOID_ENTER(cond, who)                   \
    MNK_REQUIRE(h != nullptr && (cond), who ": bad argument"); \
    MNK_HIP(hipSetDevice(h->ctx->device))
#define IPM_DONE() MNK_HIP(hipGetLastError()); return 0

int mnk_ipm_populate_RR_nn(mnk_ipm* h, double* nn, const double* c, int64_t m, double mu, double rho) {
    IPM_VOID_ENTER(m >= 0 && (m == 0 || (nn && c)), "mnk_ipm_populate_RR_nn");
    if (m > 0) hipLaunchKernelGGL(populate_nn_kernel, IPM_G(m), nn, c, mu, rho, m);
    IPM_DONE();
}

/* vector part of initialize_robust_restorer!: x_ref, D_R from x; nn, pp, zp, zn from c (mu_R, rho given by the caller,
 * mu_R = max(mu, norm(c, Inf)) through mnk_ipm_get_norms); zl_r / zu_r capped at rho in the full-length zl / zu */
int mnk_ipm_initialize_robust_restorer(mnk_ipm* h, const double* x, const double* c, int64_t m, double mu_R, double rho,
                                       double* x_ref, double* D_R, double* nn, double* pp, double* zp, double* zn,
                                       double* zl, double* zu) {
    IPM_VOID_ENTER(m >= 0 && x && x_ref && D_R && zl && zu && (m == 0 || (c && nn && pp && zp && zn)),
                   "mnk_ipm_initialize_robust_restorer");
    hipLaunchKernelGGL(rr_ref_kernel, IPM_G(h->ntot), x_ref, D_R, x, h->ntot);
    if (m > 0) hipLaunchKernelGGL(rr_slack_kernel, IPM_G(m), nn, pp, zp, zn, c, mu_R, rho, m);
    if (h->nlb > 0) hipLaunchKernelGGL(cap_gather_kernel, IPM_G(h->nlb), zl, h->ind_lb.p, rho, h->nlb);
    if (h->nub > 0) hipLaunchKernelGGL(cap_gather_kernel, IPM_G(h->nub), zu, h->ind_ub.p, rho, h->nub);
    IPM_DONE();
}

int mnk_ipm_set_f_RR(mnk_ipm* h, double* f_R, const double* D_R, const double* x, const double* x_ref, double zeta) {
    IPM_VOID_ENTER(f_R && D_R && x && x_ref, "mnk_ipm_set_f_RR");
    hipLaunchKernelGGL(f_rr_kernel, IPM_G(h->ntot), f_R, D_R, x, x_ref, zeta, h->ntot);
    IPM_DONE();
}

int mnk_ipm_set_aug_rhs_RR(mnk_ipm* h, const double* f_R, const double* zl, const double* zu, const double* jacl,
                           const double* c, const double* y, const double* pp, const double* nn, const double* zp,
                           const double* zn, int64_t m, const double* x, const double* xl, const double* xu, double mu_R,
                           double rho, double* px, double* py, double* pzl, double* pzu) {
    IPM_VOID_ENTER(m >= 0 && f_R && zl && zu && jacl && x && xl && xu && px &&
                       (m == 0 || (c && y && pp && nn && zp && zn && py)) && (h->nlb == 0 || pzl) && (h->nub == 0 || pzu),
                   "mnk_ipm_set_aug_rhs_RR");
    hipLaunchKernelGGL(aug_rhs_primal_kernel, IPM_G(h->ntot), px, f_R, zl, zu, jacl, h->ntot);
    if (m > 0) hipLaunchKernelGGL(aug_rhs_rr_dual_kernel, IPM_G(m), py, c, y, pp, nn, zp, zn, mu_R, rho, m);
    if (h->nlb > 0) hipLaunchKernelGGL(aug_rhs_bound_kernel, IPM_G(h->nlb), pzl, x, xl, zl, h->ind_lb.p, mu_R, h->nlb, 0);
    if (h->nub > 0) hipLaunchKernelGGL(aug_rhs_bound_kernel, IPM_G(h->nub), pzu, x, xu, zu, h->ind_ub.p, mu_R, h->nub, 1);
    IPM_DONE();
}

int mnk_ipm_finish_aug_solve_RR(mnk_ipm* h, double* dpp, double* dnn, double* dzp, double* dzn, const double* l,
                                const double* dl, const double* pp, const double* nn, const double* zp, const double* zn,
                                int64_t m, double mu_R, double rho) {
    IPM_VOID_ENTER(m >= 0 && (m == 0 || (dpp && dnn && dzp && dzn && l && dl && pp && nn && zp && zn)),
                   "mnk_ipm_finish_aug_solve_RR");
    if (m > 0) hipLaunchKernelGGL(finish_rr_kernel, IPM_G(m), dpp, dnn, dzp, dzn, l, dl, pp, nn, zp, zn, mu_R, rho, m);
    IPM_DONE();
}

int mnk_ipm_reset_bound_dual_1(mnk_ipm* h, double* z, const double* x, int64_t n, double mu, double kappa_sigma) {
    IPM_VOID_ENTER(n >= 0 && (n == 0 || (z && x)), "mnk_ipm_reset_bound_dual_1");
    if (n > 0) hipLaunchKernelGGL(reset_bound_dual1_kernel, IPM_G(n), z, x, kappa_sigma * mu, mu / kappa_sigma, n);
    IPM_DONE();
}

int mnk_ipm_set_initial_bounds(mnk_ipm* h, double* xl, double* xu, int64_t n, double tol) {
    IPM_VOID_ENTER(n >= 0 && (n == 0 || (xl && xu)), "mnk_ipm_set_initial_bounds");
    if (n > 0 && tol > 0) hipLaunchKernelGGL(initial_bounds_kernel, IPM_G(n), xl, xu, tol, n);
    IPM_DONE();
}

int mnk_ipm_set_initial_rhs(mnk_ipm* h, const double* f, const double* zl, const double* zu, double* px, double* py,
                            int64_t m, double* pzl, double* pzu) {
    IPM_VOID_ENTER(m >= 0 && f && zl && zu && px && (m == 0 || py) && (h->nlb == 0 || pzl) && (h->nub == 0 || pzu),
                   "mnk_ipm_set_initial_rhs");
    hipLaunchKernelGGL(initial_rhs_kernel, IPM_G(h->ntot), px, f, zl, zu, h->ntot);
    if (m > 0) hipLaunchKernelGGL(zero_kernel, IPM_G(m), py, m);
    if (h->nlb > 0) hipLaunchKernelGGL(zero_kernel, IPM_G(h->nlb), pzl, h->nlb);
    if (h->nub > 0) hipLaunchKernelGGL(zero_kernel, IPM_G(h->nub), pzu, h->nub);
    IPM_DONE();
}

int mnk_ipm_set_aug_rhs_ifr(mnk_ipm* h, const double* c, int64_t m, double* px, double* py, double* pzl, double* pzu) {
    IPM_VOID_ENTER(m >= 0 && px && (m == 0 || (c && py)) && (h->nlb == 0 || pzl) && (h->nub == 0 || pzu),
                   "mnk_ipm_set_aug_rhs_ifr");
    hipLaunchKernelGGL(zero_kernel, IPM_G(h->ntot), px, h->ntot);
    if (m > 0) hipLaunchKernelGGL(negate_kernel, IPM_G(m), py, c, m);
    if (h->nlb > 0) hipLaunchKernelGGL(zero_kernel, IPM_G(h->nlb), pzl, h->nlb);
    if (h->nub > 0) hipLaunchKernelGGL(zero_kernel, IPM_G(h->nub), pzu, h->nub);
    IPM_DONE();
}

int mnk_ipm_set_g_ifr(mnk_ipm* h, double* g, const double* f, const double* x, const double* xl, const double* xu,
                      const double* jacl, double mu) {
    IPM_VOID_ENTER(g && f && x && xl && xu && jacl, "mnk_ipm_set_g_ifr");
    hipLaunchKernelGGL(g_ifr_kernel, IPM_G(h->ntot), g, f, x, xl, xu, jacl, mu, h->ntot);
    IPM_DONE();
}

int mnk_ipm_initialize_variables(mnk_ipm* h, double* x, const double* xl, const double* xu, int64_t n, double bound_push,
                                 double bound_fac) {
    IPM_VOID_ENTER(n >= 0 && (n == 0 || (x && xl && xu)), "mnk_ipm_initialize_variables");
    if (n > 0) hipLaunchKernelGGL(initialize_variables_kernel, IPM_G(n), x, xl, xu, bound_push, bound_fac, n);
    IPM_DONE();
}

}  // extern "C"

// ---- plain vector work of the solver loop: the copyto! / fill! / axpy! / dot / norm / mul! calls that reference
// src/IPM/solver.jl (:236-291 regular!, :300-411 restore!, :413-545 robust!), src/IPM/line_search.jl and
// src/LinearSolvers/backsolve.jl:27-76 make on the iterate, and the dense products of a QP model's callbacks.  They keep
// the whole loop on the context's stream (no second runtime in the data path).  No FMA contraction: axpby evaluates
// a x + b y as numpy / Julia broadcast do.  HBM-bound: 8 bytes per element read / written, gemv 8 m n.
namespace {
struct FDot {
    const double *x, *y;
    __device__ double operator()(int64_t i) const { return x[i] * y[i]; }
};
struct FId {
    const double* v;
    __device__ double operator()(int64_t i) const { return v[i]; }
};
__global__ void vec_copy_kernel(double* __restrict__ d, const double* __restrict__ s, int64_t n) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) d[i] = s[i];
}
__global__ void vec_fill_kernel(double* __restrict__ d, double v, int64_t n) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) d[i] = v;
}
// out = a x + b y; a == 1 / b == 1 / b == 0 keep the operand exactly (x + b y, a x + y, a x); out may alias x or y
__global__ void vec_axpby_kernel(double* out, double a, const double* x, double b, const double* y, int64_t n) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double ax = a == 1.0 ? x[i] : a * x[i];
    out[i] = y == nullptr ? ax : (b == 1.0 ? ax + y[i] : ax + b * y[i]);
}
__global__ void vec_scatter_axpy_kernel(double* __restrict__ y, const int64_t* __restrict__ idx, double a,
                                        const double* __restrict__ x, int64_t n) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) y[idx[i]] += a * x[i];
}
__global__ void vec_scatter_fill_kernel(double* __restrict__ y, const int64_t* __restrict__ idx, double v, int64_t n) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) y[idx[i]] = v;
}
__global__ void vec_gather_kernel(double* __restrict__ out, double a, const double* __restrict__ x,
                                  const int64_t* __restrict__ idx, int64_t n) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = a * x[idx[i]];
}
// y = alpha op(A) x + beta y, A column-major m x n.  trans = 0: the columns are cut into slabs, workgroup (rb, s) sums slab
// s for the 256 rows of row block rb (one thread per row, coalesced over rows) into part[s][row]; a second kernel adds the
// slabs in a fixed order (deterministic, and 256 workgroups instead of m / 256 for a 2048 x 2048 matrix: 100 -> ~12 us).
// trans = 1: one wavefront per column of A, lanes stride the rows, butterfly sum.
constexpr int GEMV_SLAB = 64;
__global__ __launch_bounds__(256) void gemv_n_slab_kernel(int64_t m, int64_t n, const double* __restrict__ A, int64_t lda,
                                                           const double* __restrict__ x, double* __restrict__ part) {
    const int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.y * GEMV_SLAB, c1 = c0 + GEMV_SLAB < n ? c0 + GEMV_SLAB : n;
    if (i >= m) return;
    double s = 0.0;
    for (int64_t j = c0; j < c1; ++j) s += A[i + j * lda] * x[j];
    part[(int64_t)blockIdx.y * m + i] = s;
}
__global__ void gemv_n_sum_kernel(int64_t m, int nslab, double alpha, const double* __restrict__ part, double beta,
                                  double* __restrict__ y) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= m) return;
    double s = 0.0;
    for (int k = 0; k < nslab; ++k) s += part[(int64_t)k * m + i];
    y[i] = beta == 0.0 ? alpha * s : alpha * s + beta * y[i];
}
__global__ __launch_bounds__(256) void gemv_t_kernel(int64_t m, int64_t n, double alpha, const double* __restrict__ A,
                                                      int64_t lda, const double* __restrict__ x, double beta,
                                                      double* __restrict__ y) {
    const int64_t j = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (j >= n) return;
    double s = 0.0;
    for (int64_t i = lane; i < m; i += 64) s += A[i + j * lda] * x[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) y[j] = beta == 0.0 ? alpha * s : alpha * s + beta * y[j];
}
}  // namespace

extern "C" {

int mnk_ipm_get_dot(mnk_ipm* h, const double* x, const double* y, int64_t n, double* out) {
    IPM_ENTER(h, "mnk_ipm_get_dot");
    MNK_REQUIRE(n >= 0 && (n == 0 || (x && y)), "mnk_ipm_get_dot: bad argument");
    IPM_BASE(1);
    int rc = enqueue<R_SUM>(h, FDot{x, y}, n, b);
    if (rc) return rc;
    return finish(h, b, 1, [=](const double* r) { *out = r[0]; });
}

int mnk_ipm_get_sum(mnk_ipm* h, const double* v, int64_t n, double* out) {
    IPM_ENTER(h, "mnk_ipm_get_sum");
    MNK_REQUIRE(n >= 0 && (n == 0 || v), "mnk_ipm_get_sum: bad argument");
    IPM_BASE(1);
    int rc = enqueue<R_SUM>(h, FId{v}, n, b);
    if (rc) return rc;
    return finish(h, b, 1, [=](const double* r) { *out = r[0]; });
}

int mnk_ipm_get_norm2(mnk_ipm* h, const double* v, int64_t n, double* out) {
    IPM_ENTER(h, "mnk_ipm_get_norm2");
    MNK_REQUIRE(n >= 0 && (n == 0 || v), "mnk_ipm_get_norm2: bad argument");
    IPM_BASE(1);
    int rc = enqueue<R_SUM>(h, FDot{v, v}, n, b);
    if (rc) return rc;
    return finish(h, b, 1, [=](const double* r) { *out = sqrt(r[0]); });
}

#define VEC_ENTER(cond, who)                                  \
    MNK_REQUIRE(h != nullptr && n >= 0 && (n == 0 || (cond)), who ": bad argument"); \
    MNK_HIP(hipSetDevice(h->ctx->device));                    \
    if (n == 0) return 0

int mnk_ipm_vec_copy(mnk_ipm* h, double* dst, const double* src, int64_t n) {
    VEC_ENTER(dst && src, "mnk_ipm_vec_copy");
    hipLaunchKernelGGL(vec_copy_kernel, IPM_G(n), dst, src, n);
    MNK_HIP(hipGetLastError());
    return 0;
}

int mnk_ipm_vec_fill(mnk_ipm* h, double* v, int64_t n, double value) {
    VEC_ENTER(v, "mnk_ipm_vec_fill");
    hipLaunchKernelGGL(vec_fill_kernel, IPM_G(n), v, value, n);
    MNK_HIP(hipGetLastError());
    return 0;
}

// out = a x + b y (y may be NULL: out = a x)
int mnk_ipm_vec_axpby(mnk_ipm* h, double* out, double a, const double* x, double b, const double* y, int64_t n) {
    VEC_ENTER(out && x, "mnk_ipm_vec_axpby");
    hipLaunchKernelGGL(vec_axpby_kernel, IPM_G(n), out, a, x, b, y, n);
    MNK_HIP(hipGetLastError());
    return 0;
}

// y[idx[i]] += a x[i], i < n (idx: device, 0-based, distinct)
int mnk_ipm_vec_scatter_axpy(mnk_ipm* h, double* y, const int64_t* idx, double a, const double* x, int64_t n) {
    VEC_ENTER(y && idx && x, "mnk_ipm_vec_scatter_axpy");
    hipLaunchKernelGGL(vec_scatter_axpy_kernel, IPM_G(n), y, idx, a, x, n);
    MNK_HIP(hipGetLastError());
    return 0;
}

// out[i] = a x[idx[i]], i < n
int mnk_ipm_vec_gather(mnk_ipm* h, double* out, double a, const double* x, const int64_t* idx, int64_t n) {
    VEC_ENTER(out && idx && x, "mnk_ipm_vec_gather");
    hipLaunchKernelGGL(vec_gather_kernel, IPM_G(n), out, a, x, idx, n);
    MNK_HIP(hipGetLastError());
    return 0;
}

// zl_r += a dzl, zu_r += a dzu (solver.jl:282-283 axpy!(alpha_z, dual_lb(d), zl_r) ...), zl / zu full-length
int mnk_ipm_bound_dual_axpy(mnk_ipm* h, double* zl, double* zu, double a, const double* dzl, const double* dzu) {
    MNK_REQUIRE(h && zl && zu && (h->nlb == 0 || dzl) && (h->nub == 0 || dzu), "mnk_ipm_bound_dual_axpy: NULL argument");
    MNK_HIP(hipSetDevice(h->ctx->device));
    if (h->nlb > 0) hipLaunchKernelGGL(vec_scatter_axpy_kernel, IPM_G(h->nlb), zl, h->ind_lb.p, a, dzl, h->nlb);
    if (h->nub > 0) hipLaunchKernelGGL(vec_scatter_axpy_kernel, IPM_G(h->nub), zu, h->ind_ub.p, a, dzu, h->nub);
    MNK_HIP(hipGetLastError());
    return 0;
}

// zl_r .= v, zu_r .= v (the "second chance" of filter_line_search_RR!, line_search.jl:193-196)
int mnk_ipm_bound_dual_fill(mnk_ipm* h, double* zl, double* zu, double v) {
    MNK_REQUIRE(h && zl && zu, "mnk_ipm_bound_dual_fill: NULL argument");
    MNK_HIP(hipSetDevice(h->ctx->device));
    if (h->nlb > 0) hipLaunchKernelGGL(vec_scatter_fill_kernel, IPM_G(h->nlb), zl, h->ind_lb.p, v, h->nlb);
    if (h->nub > 0) hipLaunchKernelGGL(vec_scatter_fill_kernel, IPM_G(h->nub), zu, h->ind_ub.p, v, h->nub);
    MNK_HIP(hipGetLastError());
    return 0;
}

// y = alpha op(A) x + beta y; A: m x n column-major (lda >= m), op = A (trans = 0, y has m entries) or A' (trans = 1, n entries)
int mnk_ipm_gemv(mnk_ipm* h, int trans, int64_t m, int64_t n, double alpha, const double* A, int64_t lda, const double* x,
                 double beta, double* y) {
    MNK_REQUIRE(h && m >= 0 && n >= 0 && (trans == 0 || trans == 1) && lda >= std::max<int64_t>(1, m) && y &&
                    (m == 0 || n == 0 || (A && x)), "mnk_ipm_gemv: bad argument");
    MNK_HIP(hipSetDevice(h->ctx->device));
    if ((trans ? n : m) == 0) return 0;
    if (trans == 0) {
        const int nslab = (int)((n + GEMV_SLAB - 1) / GEMV_SLAB);
        if (n == 0) {   // y = beta y
            hipLaunchKernelGGL(gemv_n_sum_kernel, IPM_G(m), m, 0, alpha, (const double*)nullptr, beta, y);
        } else {
            if (h->gemv_part.n < (size_t)nslab * (size_t)m && h->gemv_part.alloc((size_t)nslab * (size_t)m)) return -2;
            hipLaunchKernelGGL(gemv_n_slab_kernel, dim3((unsigned)((m + 255) / 256), (unsigned)nslab), dim3(256), 0,
                               h->ctx->stream, m, n, A, lda, x, h->gemv_part.p);
            hipLaunchKernelGGL(gemv_n_sum_kernel, IPM_G(m), m, nslab, alpha, h->gemv_part.p, beta, y);
        }
    } else
        hipLaunchKernelGGL(gemv_t_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, h->ctx->stream, m, n, alpha, A, lda, x,
                           beta, y);
    MNK_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
