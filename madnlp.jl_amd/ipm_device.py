"""Mirror of the reference's device IPM reductions (`lib/MadNLPGPU/src/IPM/kernels.jl:4-116`, host definitions
`src/IPM/kernels.jl:263-388,675-695`) over the C ABI (`mnk_ipm_*`): every vector is a device tensor, every result a
Python float.  Same names as the reference; the `_r` views are taken inside the library through `ind_lb` / `ind_ub`."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .linear_solver import HipContext, _ptr


def _dev(v):
    p, loc = _ptr(v)
    assert loc == L.MNK_DEVICE, "mnk_ipm_* take device-resident vectors"
    return p


class IPMDeviceKernels:
    def __init__(self, ntot, ind_lb, ind_ub, ctx: HipContext | None = None):
        self.ctx = ctx or HipContext()
        self.ntot = int(ntot)
        lb = np.ascontiguousarray(ind_lb, dtype=np.int64)
        ub = np.ascontiguousarray(ind_ub, dtype=np.int64)
        self._h = C.c_void_p()
        L.check(L.lib().mnk_ipm_create(self.ctx.handle, self.ntot, len(lb), lb.ctypes.data, len(ub), ub.ctypes.data, 0,
                                       C.byref(self._h)), "mnk_ipm_create")

    def _call(self, name, *args, n=1):
        out = (C.c_double * n)()
        L.check(getattr(L.lib(), name)(self._h, *args, out), name)
        return out[0] if n == 1 else tuple(out)

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

    def close(self):
        if self._h:
            L.lib().mnk_ipm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
