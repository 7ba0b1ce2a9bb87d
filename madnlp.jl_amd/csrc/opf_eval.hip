// Device evaluation of the NLP callbacks of a polar AC optimal power flow model (callback half): what the
// reference's eval_f_wrapper / eval_grad_f_wrapper! / eval_cons_wrapper! / eval_jac_wrapper! / eval_lag_hess_wrapper!
// (src/IPM/callbacks.jl:1-96) obtain from the model through NLPModels.obj / grad! / cons! / jac_coord! / hess_coord!
// (in the reference's GPU benchmarks the model is ExaModels' AC-OPF, whose kernels run on the device).  Here the model is
// `madnlp_jl_amd.problems.ACOPFModel`; this file evaluates exactly its expressions, in its constraint and COO order:
//
//   x = [va (nbus) | vm (nbus) | pg (ngen) | qg (ngen) | p (narc) | q (narc)],  narc = 2 nbranch; arc a < nbranch is the
//       from side of branch a, arc a + nbranch its to side; own end f(a), far end t(a)
//   c = [va_0 | p_a - P_a | q_a - Q_a | va_fr - va_to | p_a^2 + q_a^2 | active balance | reactive balance]
//   T(u, w, d) = k0 u^2 + u w (k1 cos d + k2 sin d),  u = vm_f, w = vm_t, d = va_f - va_t;  (k0,k1,k2) = coef[a][0:3] for P,
//       coef[a][3:6] for Q
//
// HBM-bound gather kernels, one thread per output group, no atomics (bus sums walk a CSR incidence in arc order, like the
// host model's CSR product), no FMA contraction: everything except sin / cos (ocml vs libm, <= 2 ulp) is bit-identical to
// the numpy model.  Algorithmic bytes per evaluation: cons 8 (n + m) + 48 narc, jac_coord 8 (n + nnzj), hess_coord
// 8 (n + m + nnzh).
#pragma clang fp contract(off)
#include <cmath>

#include "common.h"

using namespace mnk;

struct mnk_opf {
    mnk_ctx* ctx = nullptr;
    int64_t nbus = 0, ngen = 0, nbr = 0, narc = 0, n = 0, m = 0, nnzj = 0, nnzh = 0;
    DevBuf<int32_t> arc_f, arc_t, gen_bus, bus_arc_ptr, bus_arc, bus_gen_ptr, bus_gen;
    DevBuf<double> coef, bus, cost;
};

namespace {

struct OpfDims {
    int nbus, ngen, nbr, narc;
    __host__ __device__ int va() const { return 0; }
    __host__ __device__ int vm() const { return nbus; }
    __host__ __device__ int pg() const { return 2 * nbus; }
    __host__ __device__ int qg() const { return 2 * nbus + ngen; }
    __host__ __device__ int p() const { return 2 * nbus + 2 * ngen; }
    __host__ __device__ int q() const { return 2 * nbus + 2 * ngen + narc; }
};

struct ArcPoint {
    double u, w, cs, sn;
};
__device__ __forceinline__ ArcPoint arc_point(const OpfDims d, const double* __restrict__ x, const int32_t* __restrict__ arc_f,
                                              const int32_t* __restrict__ arc_t, int a) {
    const int f = arc_f[a], t = arc_t[a];
    ArcPoint r;
    r.u = x[d.vm() + f];
    r.w = x[d.vm() + t];
    const double dl = x[d.va() + f] - x[d.va() + t];
    r.cs = cos(dl);
    r.sn = sin(dl);
    return r;
}

__global__ void opf_obj_terms_kernel(OpfDims d, const double* __restrict__ x, const double* __restrict__ cost,
                                     double* __restrict__ terms) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= d.ngen) return;
    const double pg = x[d.pg() + g];
    terms[g] = cost[3 * g] * pg * pg + cost[3 * g + 1] * pg + cost[3 * g + 2];
}

__global__ void opf_grad_kernel(OpfDims d, int n, const double* __restrict__ x, const double* __restrict__ cost,
                                double* __restrict__ g) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int k = i - d.pg();
    g[i] = (k >= 0 && k < d.ngen) ? 2.0 * cost[3 * k] * x[i] + cost[3 * k + 1] : 0.0;
}

// one thread per constraint row
__global__ void opf_cons_kernel(OpfDims d, int m, const double* __restrict__ x, const int32_t* __restrict__ arc_f,
                                const int32_t* __restrict__ arc_t, const double* __restrict__ coef,
                                const double* __restrict__ bus, const int32_t* __restrict__ bus_arc_ptr,
                                const int32_t* __restrict__ bus_arc, const int32_t* __restrict__ bus_gen_ptr,
                                const int32_t* __restrict__ bus_gen, double* __restrict__ c) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m) return;
    if (r == 0) {
        c[0] = x[d.va()];
        return;
    }
    const int row = r;
    r -= 1;
    if (r < 2 * d.narc) {
        const int blk = r >= d.narc, a = r - blk * d.narc;
        const ArcPoint P = arc_point(d, x, arc_f, arc_t, a);
        const double* k = coef + 6 * a + 3 * blk;
        const double own = x[(blk ? d.q() : d.p()) + a];
        c[row] = own - (k[0] * P.u * P.u + P.u * P.w * (k[1] * P.cs + k[2] * P.sn));
        return;
    }
    r -= 2 * d.narc;
    if (r < d.nbr) {
        c[row] = x[d.va() + arc_f[r]] - x[d.va() + arc_t[r]];   // arc r < nbr: (fr, to) of branch r
        return;
    }
    r -= d.nbr;
    if (r < d.narc) {
        const double p = x[d.p() + r], q = x[d.q() + r];
        c[row] = p * p + q * q;
        return;
    }
    r -= d.narc;
    const int blk = r >= d.nbus, i = r - blk * d.nbus;
    const double vm = x[d.vm() + i];
    double sa = 0.0, sg = 0.0;
    const int flow0 = blk ? d.q() : d.p(), gen0 = blk ? d.qg() : d.pg();
    for (int k = bus_arc_ptr[i]; k < bus_arc_ptr[i + 1]; ++k) sa += x[flow0 + bus_arc[k]];
    for (int k = bus_gen_ptr[i]; k < bus_gen_ptr[i + 1]; ++k) sg += x[gen0 + bus_gen[k]];
    const double* b = bus + 4 * i;   // pd, qd, gs, bs
    c[row] = blk ? b[1] - b[3] * vm * vm + sa - sg : b[0] + b[2] * vm * vm + sa - sg;
}

// COO values in the order of the pattern: [1 | 5 per p row | 5 per q row | 2 per branch | 2 per arc | vm, arcs, gens | vm, arcs, gens]
__global__ void opf_jac_kernel(OpfDims d, int nmax, const double* __restrict__ x, const int32_t* __restrict__ arc_f,
                               const int32_t* __restrict__ arc_t, const double* __restrict__ coef,
                               const double* __restrict__ bus, double* __restrict__ J) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nmax) return;
    if (i == 0) J[0] = 1.0;
    const int o_q = 1 + 5 * d.narc, o_ang = o_q + 5 * d.narc, o_th = o_ang + 2 * d.nbr, o_pb = o_th + 2 * d.narc,
              o_qb = o_pb + d.nbus + d.narc + d.ngen;
    if (i < d.narc) {
        const ArcPoint P = arc_point(d, x, arc_f, arc_t, i);
        for (int blk = 0; blk < 2; ++blk) {
            const double* k = coef + 6 * i + 3 * blk;
            const double K = k[1] * P.cs + k[2] * P.sn, Kp = -k[1] * P.sn + k[2] * P.cs;
            const double Tu = 2.0 * k[0] * P.u + P.w * K, Tw = P.u * K, Td = P.u * P.w * Kp;
            double* o = J + (blk ? o_q : 1) + 5 * i;
            o[0] = 1.0; o[1] = -Tu; o[2] = -Tw; o[3] = -Td; o[4] = Td;
        }
        J[o_th + 2 * i] = 2.0 * x[d.p() + i];
        J[o_th + 2 * i + 1] = 2.0 * x[d.q() + i];
        J[o_pb + d.nbus + i] = 1.0;
        J[o_qb + d.nbus + i] = 1.0;
    }
    if (i < d.nbr) {
        J[o_ang + 2 * i] = 1.0;
        J[o_ang + 2 * i + 1] = -1.0;
    }
    if (i < d.nbus) {
        const double vm = x[d.vm() + i];
        J[o_pb + i] = 2.0 * bus[4 * i + 2] * vm;
        J[o_qb + i] = -2.0 * bus[4 * i + 3] * vm;
    }
    if (i < d.ngen) {
        J[o_pb + d.nbus + d.narc + i] = -1.0;
        J[o_qb + d.nbus + d.narc + i] = -1.0;
    }
}

// Hessian of the Lagrangian sigma f + y' c, COO values: [10 per p row | 10 per q row | p diag | q diag | vm diag | pg diag];
// the 10 = lower triangle of the clique (vm_f, vm_t, va_f, va_t) in numpy's tril_indices order
__global__ void opf_hess_kernel(OpfDims d, int nmax, const double* __restrict__ x, const double* __restrict__ y, double sigma,
                                const int32_t* __restrict__ arc_f, const int32_t* __restrict__ arc_t,
                                const double* __restrict__ coef, const double* __restrict__ bus,
                                const double* __restrict__ cost, double* __restrict__ H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nmax) return;
    const int o_pd = 20 * d.narc, o_qd = o_pd + d.narc, o_vm = o_qd + d.narc, o_pg = o_vm + d.nbus;
    const int r_th = 1 + 2 * d.narc + d.nbr, r_pb = r_th + d.narc, r_qb = r_pb + d.nbus;
    if (i < d.narc) {
        const ArcPoint P = arc_point(d, x, arc_f, arc_t, i);
        for (int blk = 0; blk < 2; ++blk) {
            const double* k = coef + 6 * i + 3 * blk;
            const double yy = -y[1 + blk * d.narc + i];
            const double K = k[1] * P.cs + k[2] * P.sn, Kp = -k[1] * P.sn + k[2] * P.cs;
            const double uwK = P.u * P.w * K, wKp = P.w * Kp, uKp = P.u * Kp;
            double* o = H + 10 * (blk * d.narc + i);
            o[0] = yy * (2.0 * k[0]); o[1] = yy * K; o[2] = yy * 0.0; o[3] = yy * wKp; o[4] = yy * uKp;
            o[5] = yy * -uwK; o[6] = yy * -wKp; o[7] = yy * -uKp; o[8] = yy * uwK; o[9] = yy * -uwK;
        }
        const double yth = y[r_th + i];
        H[o_pd + i] = 2.0 * yth;
        H[o_qd + i] = 2.0 * yth;
    }
    if (i < d.nbus) H[o_vm + i] = 2.0 * bus[4 * i + 2] * y[r_pb + i] - 2.0 * bus[4 * i + 3] * y[r_qb + i];
    if (i < d.ngen) H[o_pg + i] = sigma * 2.0 * cost[3 * i];
}

// CSR incidence bus -> items, items of a bus in increasing item order
void incidence(int64_t nbus, const std::vector<int32_t>& bus_of, std::vector<int32_t>& ptr, std::vector<int32_t>& idx) {
    ptr.assign(nbus + 1, 0);
    for (int32_t b : bus_of) ptr[b + 1]++;
    for (int64_t i = 0; i < nbus; ++i) ptr[i + 1] += ptr[i];
    idx.resize(bus_of.size());
    std::vector<int32_t> fill(ptr.begin(), ptr.end() - 1);
    for (size_t k = 0; k < bus_of.size(); ++k) idx[fill[bus_of[k]]++] = (int32_t)k;
}

inline OpfDims dims(const mnk_opf* h) { return OpfDims{(int)h->nbus, (int)h->ngen, (int)h->nbr, (int)h->narc}; }

}  // namespace

extern "C" {

#define OPF_G(cnt) dim3((unsigned)(((cnt) + 255) / 256)), dim3(256), 0, h->ctx->stream
#define OPF_ENTER(h, cond, who)                              \
    MNK_REQUIRE((h) != nullptr && (cond), who ": NULL argument"); \
    MNK_HIP(hipSetDevice((h)->ctx->device))

int mnk_opf_create(mnk_ctx* ctx, int64_t nbus, int64_t ngen, int64_t nbranch, const int32_t* fr, const int32_t* to,
                   const int32_t* gen_bus, const double* arc_coef, const double* bus_data, const double* gen_cost,
                   mnk_opf** out) {
    MNK_REQUIRE(ctx && out && fr && to && gen_bus && arc_coef && bus_data && gen_cost, "mnk_opf_create: NULL argument");
    MNK_REQUIRE(nbus > 0 && ngen > 0 && nbranch > 0 && 2 * nbus + 2 * ngen + 4 * nbranch < (1LL << 30),
                "mnk_opf_create: bad sizes");
    MNK_HIP(hipSetDevice(ctx->device));
    auto* h = new mnk_opf();
    h->ctx = ctx;
    h->nbus = nbus; h->ngen = ngen; h->nbr = nbranch; h->narc = 2 * nbranch;
    h->n = 2 * nbus + 2 * ngen + 2 * h->narc;
    h->m = 1 + 2 * h->narc + nbranch + h->narc + 2 * nbus;
    h->nnzj = 1 + 10 * h->narc + 2 * nbranch + 2 * h->narc + 2 * (nbus + h->narc + ngen);
    h->nnzh = 20 * h->narc + 2 * h->narc + nbus + ngen;
    std::vector<int32_t> af(h->narc), at(h->narc), gb(ngen);
    for (int64_t l = 0; l < nbranch; ++l) {
        if (fr[l] < 0 || fr[l] >= nbus || to[l] < 0 || to[l] >= nbus) {
            delete h;
            set_error("mnk_opf_create: branch %lld has an end outside [0, nbus)", (long long)l);
            return -1;
        }
        af[l] = fr[l]; at[l] = to[l];
        af[l + nbranch] = to[l]; at[l + nbranch] = fr[l];
    }
    for (int64_t g = 0; g < ngen; ++g) {
        if (gen_bus[g] < 0 || gen_bus[g] >= nbus) {
            delete h;
            set_error("mnk_opf_create: generator %lld sits on a bus outside [0, nbus)", (long long)g);
            return -1;
        }
        gb[g] = gen_bus[g];
    }
    std::vector<int32_t> aptr, aidx, gptr, gidx;
    incidence(nbus, af, aptr, aidx);
    incidence(nbus, gb, gptr, gidx);
    hipStream_t s = ctx->stream;
    int rc = h->arc_f.upload(af, s) | h->arc_t.upload(at, s) | h->gen_bus.upload(gb, s) | h->bus_arc_ptr.upload(aptr, s) |
             h->bus_arc.upload(aidx, s) | h->bus_gen_ptr.upload(gptr, s) | h->bus_gen.upload(gidx, s) |
             h->coef.upload(std::vector<double>(arc_coef, arc_coef + 6 * h->narc), s) |
             h->bus.upload(std::vector<double>(bus_data, bus_data + 4 * nbus), s) |
             h->cost.upload(std::vector<double>(gen_cost, gen_cost + 3 * ngen), s);
    if (rc) {
        delete h;
        return rc;
    }
    mnk_ctx_child_added(ctx);
    *out = h;
    return 0;
}

int mnk_opf_destroy(mnk_opf* h) {
    if (!h) return 0;
    mnk_ctx* ctx = h->ctx;
    (void)hipSetDevice(ctx->device);
    (void)mnk::stream_wait(ctx->stream);
    delete h;
    mnk_ctx_child_gone(ctx);
    return 0;
}

int mnk_opf_sizes(mnk_opf* h, int64_t* n, int64_t* m, int64_t* nnzj, int64_t* nnzh) {
    MNK_REQUIRE(h != nullptr, "mnk_opf_sizes: NULL handle");
    if (n) *n = h->n;
    if (m) *m = h->m;
    if (nnzj) *nnzj = h->nnzj;
    if (nnzh) *nnzh = h->nnzh;
    return 0;
}

// per-generator terms of the objective; the caller sums them (mnk_ipm_get_sum) -- NLPModels.obj
int mnk_opf_obj_terms(mnk_opf* h, const double* x, double* terms) {
    OPF_ENTER(h, x && terms, "mnk_opf_obj_terms");
    hipLaunchKernelGGL(opf_obj_terms_kernel, OPF_G(h->ngen), dims(h), x, h->cost.p, terms);
    MNK_HIP(hipGetLastError());
    return 0;
}

int mnk_opf_grad(mnk_opf* h, const double* x, double* g) {   // NLPModels.grad!
    OPF_ENTER(h, x && g, "mnk_opf_grad");
    hipLaunchKernelGGL(opf_grad_kernel, OPF_G(h->n), dims(h), (int)h->n, x, h->cost.p, g);
    MNK_HIP(hipGetLastError());
    return 0;
}

int mnk_opf_cons(mnk_opf* h, const double* x, double* c) {   // NLPModels.cons!
    OPF_ENTER(h, x && c, "mnk_opf_cons");
    hipLaunchKernelGGL(opf_cons_kernel, OPF_G(h->m), dims(h), (int)h->m, x, h->arc_f.p, h->arc_t.p, h->coef.p, h->bus.p,
                       h->bus_arc_ptr.p, h->bus_arc.p, h->bus_gen_ptr.p, h->bus_gen.p, c);
    MNK_HIP(hipGetLastError());
    return 0;
}

int mnk_opf_jac_coord(mnk_opf* h, const double* x, double* jac) {   // NLPModels.jac_coord!
    OPF_ENTER(h, x && jac, "mnk_opf_jac_coord");
    const int nmax = (int)std::max(std::max(h->narc, h->nbus), h->ngen);
    hipLaunchKernelGGL(opf_jac_kernel, OPF_G(nmax), dims(h), nmax, x, h->arc_f.p, h->arc_t.p, h->coef.p, h->bus.p, jac);
    MNK_HIP(hipGetLastError());
    return 0;
}

int mnk_opf_hess_coord(mnk_opf* h, const double* x, const double* y, double obj_weight, double* hess) {   // NLPModels.hess_coord!
    OPF_ENTER(h, x && y && hess, "mnk_opf_hess_coord");
    const int nmax = (int)std::max(std::max(h->narc, h->nbus), h->ngen);
    hipLaunchKernelGGL(opf_hess_kernel, OPF_G(nmax), dims(h), nmax, x, y, obj_weight, h->arc_f.p, h->arc_t.p, h->coef.p,
                       h->bus.p, h->cost.p, hess);
    MNK_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
