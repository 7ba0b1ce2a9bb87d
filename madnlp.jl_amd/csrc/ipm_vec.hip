// Device-side IPM scalar reductions over device-resident iterates (second slice): the regular-phase
// functions of reference src/IPM/kernels.jl:263-388,675-695 (GPU twins: lib/MadNLPGPU/src/IPM/kernels.jl:4-116, which
// are mapreduce calls over the same expressions).  HBM-bound one-pass reductions: algorithmic bytes = 8 x (vectors read)
// per element; two launches (grid-stride partials in a fixed order -> one block) + one 8/16-byte D2H per call, so results
// are deterministic run to run.  Expressions are evaluated exactly as the reference writes them (no FMA contraction):
// max/min-type results are bit-identical to the host restatement, sum-type results agree to summation-order rounding.
#pragma clang fp contract(off)
#include <cfloat>
#include <cmath>

#include "common.h"

using namespace mnk;

struct mnk_ipm {
    mnk_ctx* ctx = nullptr;
    int64_t ntot = 0, nlb = 0, nub = 0;
    int64_t nllb = 0, nuub = 0;
    DevBuf<int64_t> ind_lb, ind_ub, ind_llb, ind_uub;
    DevBuf<double> part;   // 2 x IPM_BLOCKS partials
    DevBuf<double> res;    // 2 results
    double* pin = nullptr;     // 2 pinned, device-mapped host words: the final reduction stores its result here
    double* pin_dev = nullptr;
};

namespace {

constexpr int IPM_BLOCKS = 256;
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

int fetch(mnk_ipm* h, int count, double* out) {
    // the final reductions stored their results straight into pinned, device-mapped host memory: poll the stream, read
    MNK_HIP(mnk::stream_wait(h->ctx->stream));
    volatile double* pw = h->pin;
    for (int i = 0; i < count; ++i) out[i] = pw[i];
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
    int rc = h->ind_lb.upload(lb, ctx->stream) | h->ind_ub.upload(ub, ctx->stream) | h->part.alloc(2 * IPM_BLOCKS) |
             h->res.alloc(2);
    if (!rc && (hipHostMalloc((void**)&h->pin, 2 * sizeof(double), hipHostMallocMapped) != hipSuccess ||
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

int mnk_ipm_get_varphi(mnk_ipm* h, double obj_val, const double* x, const double* xl, const double* xu, double mu,
                       double* out) {
    IPM_ENTER(h, "mnk_ipm_get_varphi");
    int rc = enqueue<R_SUM>(h, FVarphi{x, xl, h->ind_lb.p, mu, 0}, h->nlb, 0) |
             enqueue<R_SUM>(h, FVarphi{x, xu, h->ind_ub.p, mu, 1}, h->nub, 1);
    if (rc) return rc;
    double r[2];
    rc = fetch(h, 2, r);
    if (rc) return rc;
    *out = obj_val + r[0] + r[1];
    return 0;
}

int mnk_ipm_get_inf_du(mnk_ipm* h, const double* f, const double* zl, const double* zu, const double* jacl, double sd,
                       double* out) {
    IPM_ENTER(h, "mnk_ipm_get_inf_du");
    int rc = enqueue<R_MAX>(h, FInfDu{f, zl, zu, jacl}, h->ntot, 0);
    if (rc) return rc;
    double r;
    rc = fetch(h, 1, &r);
    if (rc) return rc;
    *out = max0(r) / sd;
    return 0;
}

int mnk_ipm_get_inf_compl(mnk_ipm* h, const double* x, const double* xl, const double* xu, const double* zl,
                          const double* zu, double mu, double sc, double* out) {
    IPM_ENTER(h, "mnk_ipm_get_inf_compl");
    int rc = enqueue<R_MAX>(h, FCompl{x, xl, zl, h->ind_lb.p, mu, 0, 1}, h->nlb, 0) |
             enqueue<R_MAX>(h, FCompl{x, xu, zu, h->ind_ub.p, mu, 1, 1}, h->nub, 1);
    if (rc) return rc;
    double r[2];
    rc = fetch(h, 2, r);
    if (rc) return rc;
    *out = ((r[0] != r[0] || r[1] != r[1]) ? NAN : max0(fmax(r[0], r[1]))) / sc;
    return 0;
}

int mnk_ipm_get_min_complementarity(mnk_ipm* h, const double* x, const double* xl, const double* xu, const double* zl,
                                    const double* zu, double* out) {
    IPM_ENTER(h, "mnk_ipm_get_min_complementarity");
    int rc = enqueue<R_MIN>(h, FCompl{x, xl, zl, h->ind_lb.p, 0.0, 0, 0}, h->nlb, 0) |
             enqueue<R_MIN>(h, FCompl{x, xu, zu, h->ind_ub.p, 0.0, 1, 0}, h->nub, 1);
    if (rc) return rc;
    double r[2];
    rc = fetch(h, 2, r);
    if (rc) return rc;
    *out = (r[0] != r[0] || r[1] != r[1]) ? NAN : fmin(r[0], r[1]);
    return 0;
}

int mnk_ipm_get_average_complementarity(mnk_ipm* h, const double* x, const double* xl, const double* xu, const double* zl,
                                        const double* zu, double* out) {
    IPM_ENTER(h, "mnk_ipm_get_average_complementarity");
    if (h->nlb + h->nub == 0) { *out = 0.0; return 0; }
    int rc = enqueue<R_SUM>(h, FCompl{x, xl, zl, h->ind_lb.p, 0.0, 0, 0}, h->nlb, 0) |
             enqueue<R_SUM>(h, FCompl{x, xu, zu, h->ind_ub.p, 0.0, 1, 0}, h->nub, 1);
    if (rc) return rc;
    double r[2];
    rc = fetch(h, 2, r);
    if (rc) return rc;
    *out = (r[0] + r[1]) / (double)(h->nlb + h->nub);
    return 0;
}

int mnk_ipm_get_varphi_d(mnk_ipm* h, const double* f, const double* x, const double* xl, const double* xu,
                         const double* dx, double mu, double* out) {
    IPM_ENTER(h, "mnk_ipm_get_varphi_d");
    int rc = enqueue<R_SUM>(h, FVarphiD{f, x, xl, xu, dx, mu}, h->ntot, 0);
    if (rc) return rc;
    return fetch(h, 1, out);
}

int mnk_ipm_get_alpha_max(mnk_ipm* h, const double* x, const double* xl, const double* xu, const double* dx, double tau,
                          double* out) {
    IPM_ENTER(h, "mnk_ipm_get_alpha_max");
    int rc = enqueue<R_MIN>(h, FAlphaMax{x, xl, xu, dx, tau}, h->ntot, 0);
    if (rc) return rc;
    double r;
    rc = fetch(h, 1, &r);
    if (rc) return rc;
    *out = r != r ? r : fmin(1.0, r);
    return 0;
}

int mnk_ipm_get_alpha_z(mnk_ipm* h, const double* zl, const double* zu, const double* dzl, const double* dzu, double tau,
                        double* out) {
    IPM_ENTER(h, "mnk_ipm_get_alpha_z");
    int rc = enqueue<R_MIN>(h, FAlphaZ{zl, dzl, h->ind_lb.p, tau}, h->nlb, 0) |
             enqueue<R_MIN>(h, FAlphaZ{zu, dzu, h->ind_ub.p, tau}, h->nub, 1);
    if (rc) return rc;
    double r[2];
    rc = fetch(h, 2, r);
    if (rc) return rc;
    const double m = fmin(r[0], r[1]);
    *out = (r[0] != r[0] || r[1] != r[1]) ? NAN : fmin(1.0, m);
    return 0;
}

int mnk_ipm_get_rel_search_norm(mnk_ipm* h, const double* x, const double* dx, double* out) {
    IPM_ENTER(h, "mnk_ipm_get_rel_search_norm");
    int rc = enqueue<R_MAX>(h, FRelNorm{x, dx}, h->ntot, 0);
    if (rc) return rc;
    double r;
    rc = fetch(h, 1, &r);
    if (rc) return rc;
    *out = max0(r);
    return 0;
}

// get_sd / get_sc (kernels.jl:684-695): l = the constraint multipliers (m entries), zl / zu full-length
int mnk_ipm_get_sd_sc(mnk_ipm* h, const double* l, int64_t m, const double* zl, const double* zu, double s_max,
                      double* out /* [sd, sc] */) {
    IPM_ENTER(h, "mnk_ipm_get_sd_sc");
    MNK_REQUIRE(m >= 0, "mnk_ipm_get_sd_sc: bad size");
    double r[2], nl = 0.0;
    int rc = enqueue<R_SUM>(h, FAbs{zl, h->ind_lb.p}, h->nlb, 0) | enqueue<R_SUM>(h, FAbs{zu, h->ind_ub.p}, h->nub, 1);
    if (rc) return rc;
    rc = fetch(h, 2, r);
    if (rc) return rc;
    rc = enqueue<R_SUM>(h, FAbs{l, nullptr}, m, 0);
    if (rc) return rc;
    rc = fetch(h, 1, &nl);
    if (rc) return rc;
    const double nz = r[0] + r[1];
    const double cnt_d = (double)std::max<int64_t>(1, m + h->nlb + h->nub), cnt_c = (double)std::max<int64_t>(1, h->nlb + h->nub);
    out[0] = fmax(s_max, (nl + r[0] + r[1]) / cnt_d) / s_max;
    out[1] = fmax(s_max, nz / cnt_c) / s_max;
    return 0;
}

// get_inf_pr = norm(c, Inf) (kernels.jl:284) and theta = norm(c, 1) (solver.jl get_theta)
int mnk_ipm_get_norms(mnk_ipm* h, const double* c, int64_t m, double* out /* [inf, one] */) {
    IPM_ENTER(h, "mnk_ipm_get_norms");
    MNK_REQUIRE(m >= 0, "mnk_ipm_get_norms: bad size");
    int rc = enqueue<R_MAX>(h, FAbs{c, nullptr}, m, 0) | enqueue<R_SUM>(h, FAbs{c, nullptr}, m, 1);
    if (rc) return rc;
    double r[2];
    rc = fetch(h, 2, r);
    if (rc) return rc;
    out[0] = max0(r[0]);
    out[1] = r[1];
    return 0;
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
