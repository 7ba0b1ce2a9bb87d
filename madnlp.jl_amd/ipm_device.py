"""Mirror of the reference's device IPM reductions (`lib/MadNLPGPU/src/IPM/kernels.jl:4-116`, host definitions
`src/IPM/kernels.jl:263-388,675-695`) over the C ABI (`mnk_ipm_*`): every vector is a device tensor, every result a
Python float.  Same names as the reference; the `_r` views are taken inside the library through `ind_lb` / `ind_ub`."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib as L
from .linear_solver import HipContext, _ptr


def _dev(v):
    p, loc = _ptr(v)
    assert loc == L.MNK_DEVICE, "mnk_ipm_* take device-resident vectors"
    return p


class _Batch:
    """MNK_IPM_NO_BATCH=1 (A/B): the calls of the block synchronize one by one, as outside a batch."""
    _off = os.environ.get("MNK_IPM_NO_BATCH", "0") not in ("", "0")

    def __init__(self, K):
        self.K = K

    def __enter__(self):
        if not self._off:
            L.check(L.lib().mnk_ipm_batch_begin(self.K._h), "mnk_ipm_batch_begin")
        self.K._batching, self.K._keep = True, []
        return self.K

    def __exit__(self, *exc):
        self.K._batching = False
        if not self._off:
            L.check(L.lib().mnk_ipm_batch_end(self.K._h), "mnk_ipm_batch_end")
        self.K._keep = []
        return False


class IPMDeviceKernels:
    def __init__(self, ntot, ind_lb, ind_ub, ctx: HipContext | None = None):
        self.ctx = ctx or HipContext()
        self.ntot = int(ntot)
        lb = np.ascontiguousarray(ind_lb, dtype=np.int64)
        ub = np.ascontiguousarray(ind_ub, dtype=np.int64)
        self._h = C.c_void_p()
        L.check(L.lib().mnk_ipm_create(self.ctx.handle, self.ntot, len(lb), lb.ctypes.data, len(ub), ub.ctypes.data, 0,
                                       C.byref(self._h)), "mnk_ipm_create")

    _batching = False

    def _call(self, name, *args, n=1):
        out = (C.c_double * n)()
        L.check(getattr(L.lib(), name)(self._h, *args, out), name)
        if self._batching:          # filled by batch_end: the caller reads out[i] after the `with` block
            self._keep.append(out)
            return out
        return out[0] if n == 1 else tuple(out)

    def batch(self):
        """`with K.batch(): a = K.get_...(); b = K.get_...()` -- the calls only enqueue their reductions and return their
        result buffers; ONE synchronization when the block ends, after which `a[0]`, `b[0]`, ... hold the values."""
        return _Batch(self)

    def get_varphi(self, obj_val, x, xl, xu, mu):
        return self._call("mnk_ipm_get_varphi", float(obj_val), _dev(x), _dev(xl), _dev(xu), float(mu))

    def get_inf_du(self, f, zl, zu, jacl, sd):
        return self._call("mnk_ipm_get_inf_du", _dev(f), _dev(zl), _dev(zu), _dev(jacl), float(sd))

    def get_inf_compl(self, x, xl, xu, zl, zu, mu, sc):
        return self._call("mnk_ipm_get_inf_compl", _dev(x), _dev(xl), _dev(xu), _dev(zl), _dev(zu), float(mu), float(sc))

    def get_min_complementarity(self, x, xl, xu, zl, zu):
        return self._call("mnk_ipm_get_min_complementarity", _dev(x), _dev(xl), _dev(xu), _dev(zl), _dev(zu))

    def get_average_complementarity(self, x, xl, xu, zl, zu):
        return self._call("mnk_ipm_get_average_complementarity", _dev(x), _dev(xl), _dev(xu), _dev(zl), _dev(zu))

    def get_varphi_d(self, f, x, xl, xu, dx, mu):
        return self._call("mnk_ipm_get_varphi_d", _dev(f), _dev(x), _dev(xl), _dev(xu), _dev(dx), float(mu))

    def get_alpha_max(self, x, xl, xu, dx, tau):
        return self._call("mnk_ipm_get_alpha_max", _dev(x), _dev(xl), _dev(xu), _dev(dx), float(tau))

    def get_alpha_z(self, zl, zu, dzl, dzu, tau):
        return self._call("mnk_ipm_get_alpha_z", _dev(zl), _dev(zu), _dev(dzl), _dev(dzu), float(tau))

    def get_rel_search_norm(self, x, dx):
        return self._call("mnk_ipm_get_rel_search_norm", _dev(x), _dev(dx))

    def get_sd_sc(self, l, zl, zu, s_max):
        return self._call("mnk_ipm_get_sd_sc", _dev(l), int(l.numel()), _dev(zl), _dev(zu), float(s_max), n=2)

    def get_norms(self, c):
        """(norm(c, Inf), norm(c, 1)): `get_inf_pr` and `get_theta`."""
        return self._call("mnk_ipm_get_norms", _dev(c), int(c.numel()), n=2)

    # ---- elementwise pieces (asynchronous on the context's stream; outputs are device tensors of the caller)
    def set_perturbation_sets(self, ind_llb, ind_uub):
        a = np.ascontiguousarray(ind_llb, dtype=np.int64)
        b = np.ascontiguousarray(ind_uub, dtype=np.int64)
        L.check(L.lib().mnk_ipm_set_perturbation_sets(self._h, len(a), a.ctypes.data, len(b), b.ctypes.data, 0),
                "mnk_ipm_set_perturbation_sets")

    def set_aug_rhs(self, f, zl, zu, jacl, c, x, xl, xu, mu, px, py, pzl, pzu):
        L.check(L.lib().mnk_ipm_set_aug_rhs(self._h, _dev(f), _dev(zl), _dev(zu), _dev(jacl), _dev(c), int(c.numel()),
                                            _dev(x), _dev(xl), _dev(xu), float(mu), _dev(px), _dev(py), _dev(pzl),
                                            _dev(pzu)), "mnk_ipm_set_aug_rhs")

    def dual_inf_perturbation(self, px, mu, kappa_d):
        L.check(L.lib().mnk_ipm_dual_inf_perturbation(self._h, _dev(px), float(mu), float(kappa_d)),
                "mnk_ipm_dual_inf_perturbation")

    def adjust_boundary(self, x, xl, xu, mu):
        L.check(L.lib().mnk_ipm_adjust_boundary(self._h, _dev(x), _dev(xl), _dev(xu), float(mu)), "mnk_ipm_adjust_boundary")

    def reset_bound_dual(self, zl, zu, x, xl, xu, mu, kappa_sigma):
        L.check(L.lib().mnk_ipm_reset_bound_dual(self._h, _dev(zl), _dev(zu), _dev(x), _dev(xl), _dev(xu), float(mu),
                                                 float(kappa_sigma)), "mnk_ipm_reset_bound_dual")

    # ---- restoration phase (robust restorer): reference src/IPM/kernels.jl:390-636 and the elementwise pieces
    def get_obj_val_R(self, p, n, D_R, x, x_ref, rho, zeta):
        return self._call("mnk_ipm_get_obj_val_R", _dev(p), _dev(n), int(p.numel()), _dev(D_R), _dev(x), _dev(x_ref),
                          float(rho), float(zeta))

    def get_theta_R(self, c, p, n):
        return self._call("mnk_ipm_get_theta_R", _dev(c), _dev(p), _dev(n), int(c.numel()))

    def get_inf_pr_R(self, c, p, n):
        return self._call("mnk_ipm_get_inf_pr_R", _dev(c), _dev(p), _dev(n), int(c.numel()))

    def get_inf_du_R(self, f_R, l, zl, zu, jacl, zp, zn, rho, sd):
        return self._call("mnk_ipm_get_inf_du_R", _dev(f_R), _dev(l), _dev(zl), _dev(zu), _dev(jacl), _dev(zp), _dev(zn),
                          int(l.numel()), float(rho), float(sd))

    def get_inf_compl_R(self, x, xl, xu, zl, zu, pp, zp, nn, zn, mu_R, sc):
        return self._call("mnk_ipm_get_inf_compl_R", _dev(x), _dev(xl), _dev(xu), _dev(zl), _dev(zu), _dev(pp), _dev(zp),
                          _dev(nn), _dev(zn), int(pp.numel()), float(mu_R), float(sc))

    def get_alpha_max_R(self, x, xl, xu, dx, pp, dpp, nn, dnn, tau_R):
        return self._call("mnk_ipm_get_alpha_max_R", _dev(x), _dev(xl), _dev(xu), _dev(dx), _dev(pp), _dev(dpp), _dev(nn),
                          _dev(dnn), int(pp.numel()), float(tau_R))

    def get_alpha_z_R(self, zl, zu, dzl, dzu, zp, dzp, zn, dzn, tau_R):
        return self._call("mnk_ipm_get_alpha_z_R", _dev(zl), _dev(zu), _dev(dzl), _dev(dzu), _dev(zp), _dev(dzp), _dev(zn),
                          _dev(dzn), int(zp.numel()), float(tau_R))

    def get_varphi_R(self, obj_val, x, xl, xu, pp, nn, mu_R):
        return self._call("mnk_ipm_get_varphi_R", float(obj_val), _dev(x), _dev(xl), _dev(xu), _dev(pp), _dev(nn),
                          int(pp.numel()), float(mu_R))

    def get_F(self, c, f, zl, zu, jacl, x, xl, xu, mu):
        return self._call("mnk_ipm_get_F", _dev(c), int(c.numel()), _dev(f), _dev(zl), _dev(zu), _dev(jacl), _dev(x),
                          _dev(xl), _dev(xu), float(mu))

    def get_varphi_d_R(self, f_R, x, xl, xu, dx, pp, nn, dpp, dnn, mu_R, rho):
        return self._call("mnk_ipm_get_varphi_d_R", _dev(f_R), _dev(x), _dev(xl), _dev(xu), _dev(dx), _dev(pp), _dev(nn),
                          _dev(dpp), _dev(dnn), int(pp.numel()), float(mu_R), float(rho))

    def _void(self, name, *args):
        L.check(getattr(L.lib(), name)(self._h, *args), name)

    def populate_RR_nn(self, nn, c, mu, rho):
        self._void("mnk_ipm_populate_RR_nn", _dev(nn), _dev(c), int(c.numel()), float(mu), float(rho))

    def initialize_robust_restorer(self, x, c, mu_R, rho, x_ref, D_R, nn, pp, zp, zn, zl, zu):
        self._void("mnk_ipm_initialize_robust_restorer", _dev(x), _dev(c), int(c.numel()), float(mu_R), float(rho),
                   _dev(x_ref), _dev(D_R), _dev(nn), _dev(pp), _dev(zp), _dev(zn), _dev(zl), _dev(zu))

    def set_f_RR(self, f_R, D_R, x, x_ref, zeta):
        self._void("mnk_ipm_set_f_RR", _dev(f_R), _dev(D_R), _dev(x), _dev(x_ref), float(zeta))

    def set_aug_rhs_RR(self, f_R, zl, zu, jacl, c, y, pp, nn, zp, zn, x, xl, xu, mu_R, rho, px, py, pzl, pzu):
        self._void("mnk_ipm_set_aug_rhs_RR", _dev(f_R), _dev(zl), _dev(zu), _dev(jacl), _dev(c), _dev(y), _dev(pp), _dev(nn),
                   _dev(zp), _dev(zn), int(c.numel()), _dev(x), _dev(xl), _dev(xu), float(mu_R), float(rho), _dev(px),
                   _dev(py), _dev(pzl), _dev(pzu))

    def finish_aug_solve_RR(self, dpp, dnn, dzp, dzn, l, dl, pp, nn, zp, zn, mu_R, rho):
        self._void("mnk_ipm_finish_aug_solve_RR", _dev(dpp), _dev(dnn), _dev(dzp), _dev(dzn), _dev(l), _dev(dl), _dev(pp),
                   _dev(nn), _dev(zp), _dev(zn), int(pp.numel()), float(mu_R), float(rho))

    def reset_bound_dual_1(self, z, x, mu, kappa_sigma):
        self._void("mnk_ipm_reset_bound_dual_1", _dev(z), _dev(x), int(z.numel()), float(mu), float(kappa_sigma))

    def set_initial_bounds(self, xl, xu, tol):
        self._void("mnk_ipm_set_initial_bounds", _dev(xl), _dev(xu), int(xl.numel()), float(tol))

    def set_initial_rhs(self, f, zl, zu, px, py, pzl, pzu):
        self._void("mnk_ipm_set_initial_rhs", _dev(f), _dev(zl), _dev(zu), _dev(px), _dev(py), int(py.numel()), _dev(pzl),
                   _dev(pzu))

    def set_aug_rhs_ifr(self, c, px, py, pzl, pzu):
        self._void("mnk_ipm_set_aug_rhs_ifr", _dev(c), int(c.numel()), _dev(px), _dev(py), _dev(pzl), _dev(pzu))

    def set_g_ifr(self, g, f, x, xl, xu, jacl, mu):
        self._void("mnk_ipm_set_g_ifr", _dev(g), _dev(f), _dev(x), _dev(xl), _dev(xu), _dev(jacl), float(mu))

    def initialize_variables(self, x, xl, xu, bound_push, bound_fac):
        self._void("mnk_ipm_initialize_variables", _dev(x), _dev(xl), _dev(xu), int(x.numel()), float(bound_push),
                   float(bound_fac))

    # ---- plain vector work of the loop (copyto! / fill! / axpy! / dot / norm / mul! of the reference's solver.jl,
    # line_search.jl, backsolve.jl) on the context's stream
    def get_dot(self, x, y):
        return self._call("mnk_ipm_get_dot", _dev(x), _dev(y), int(x.numel()))

    def get_sum(self, v):
        return self._call("mnk_ipm_get_sum", _dev(v), int(v.numel()))

    def get_norm2(self, v):
        return self._call("mnk_ipm_get_norm2", _dev(v), int(v.numel()))

    def vec_copy(self, dst, src):
        assert dst.numel() == src.numel()
        self._void("mnk_ipm_vec_copy", _dev(dst), _dev(src), int(dst.numel()))

    def vec_fill(self, v, value):
        if v.numel():
            self._void("mnk_ipm_vec_fill", _dev(v), int(v.numel()), float(value))

    def vec_axpby(self, out, a, x, b=0.0, y=None):
        """out = a x + b y (y None: out = a x); out may alias x or y."""
        assert out.numel() == x.numel() and (y is None or y.numel() == x.numel())
        if out.numel():
            self._void("mnk_ipm_vec_axpby", _dev(out), float(a), _dev(x), float(b), None if y is None else _dev(y),
                       int(out.numel()))

    def vec_scatter_axpy(self, y, idx, a, x):
        if x.numel():
            self._void("mnk_ipm_vec_scatter_axpy", _dev(y), idx.data_ptr(), float(a), _dev(x), int(x.numel()))

    def vec_gather(self, out, a, x, idx):
        if out.numel():
            self._void("mnk_ipm_vec_gather", _dev(out), float(a), _dev(x), idx.data_ptr(), int(out.numel()))

    def bound_dual_axpy(self, zl, zu, a, dzl, dzu):
        self._void("mnk_ipm_bound_dual_axpy", _dev(zl), _dev(zu), float(a), _dev(dzl), _dev(dzu))

    def bound_dual_fill(self, zl, zu, v):
        self._void("mnk_ipm_bound_dual_fill", _dev(zl), _dev(zu), float(v))

    def gemv(self, trans, m, n, alpha, A, lda, x, beta, y):
        """y = alpha op(A) x + beta y, A column-major m x n on the device."""
        self._void("mnk_ipm_gemv", int(trans), int(m), int(n), float(alpha), A.data_ptr(), int(lda), _dev(x), float(beta),
                   _dev(y))

    def close(self):
        if self._h:
            L.lib().mnk_ipm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
