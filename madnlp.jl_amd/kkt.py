"""Host-side mirror of MadNLP's KKT-system interface over the HIP library.

Mirrors `AbstractKKTSystem` (reference `src/KKT/KKTsystem.jl:104-256`) for the three
formulations on the hot path:

  * `SparseCondensedKKTSystem`  reference `src/KKT/Sparse/condensed.jl`
  * `DenseCondensedKKTSystem`   reference `src/KKT/Dense/condensed.jl`
  * `DenseKKTSystem`            reference `src/KKT/Dense/augmented.jl`

Same field names (`hess, jac, reg, pr_diag, du_diag, l_diag, u_diag, l_lower, u_lower,
aug_com, linear_solver, ind_ineq, ind_lb, ind_ub`) and method names minus the `!`.
Design stance: IPM vectors and callbacks stay on the host, the KKT
matrices, factors and work buffers are device-resident behind the C handles; per
iteration only O(nnz) values and O(n) vectors cross PCIe.  All indices are 0-based.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.sparse as sp

from . import _lib as L
from .linear_solver import (DeviceCSC, DeviceDense, HipContext, HipLinearSolver, HipSolverOptions, _ptr,
                            _LIVE_OBJECTS)


# ------------------------------------------------------------------ KKT vectors
class UnreducedKKTVector:
    """reference `src/KKT/rhs.jl:90-150`: [x, s ; y ; zl ; zu] in one array with views."""

    def __init__(self, n, m, nlb, nub, ind_lb, ind_ub):
        self.n, self.m, self.nlb, self.nub = n, m, nlb, nub
        self.values = np.zeros(n + m + nlb + nub)
        self.ind_lb = np.asarray(ind_lb, dtype=np.int64)
        self.ind_ub = np.asarray(ind_ub, dtype=np.int64)

    @classmethod
    def from_kkt(cls, kkt):
        return cls(len(kkt.pr_diag), len(kkt.du_diag), len(kkt.l_diag), len(kkt.u_diag), kkt.ind_lb, kkt.ind_ub)

    def copy(self):
        c = UnreducedKKTVector(self.n, self.m, self.nlb, self.nub, self.ind_lb, self.ind_ub)
        c.values[:] = self.values
        return c

    def full(self):
        return self.values

    def primal(self):
        return self.values[:self.n]

    def dual(self):
        return self.values[self.n:self.n + self.m]

    def primal_dual(self):
        return self.values[:self.n + self.m]

    def dual_lb(self):
        return self.values[self.n + self.m:self.n + self.m + self.nlb]

    def dual_ub(self):
        return self.values[self.n + self.m + self.nlb:]


# ------------------------------------------------ generic pieces (KKTsystem.jl, IPM/kernels.jl)
class _KKTCommon:
    def initialize(self):
        """`initialize!` reference `src/KKT/KKTsystem.jl:210-216`."""
        self.reg[:] = 1.0
        self.pr_diag[:] = 1.0
        self.du_diag[:] = 0.0
        self.hess[...] = 0.0

    def regularize_diagonal(self, primal, dual):
        """`regularize_diagonal!` reference `src/KKT/KKTsystem.jl:222-226`."""
        self.reg += primal
        self.pr_diag += primal
        self.du_diag -= dual

    def set_aug_diagonal(self):
        """`_set_aug_diagonal!` reference `src/IPM/kernels.jl:22-27`."""
        self.pr_diag[:] = self.reg
        self.pr_diag[self.ind_lb] -= self.l_lower / self.l_diag
        self.pr_diag[self.ind_ub] -= self.u_lower / self.u_diag

    # ---- device-side feeders (SURVEY 8(a)11 on the device; `_PFX` = "mnk_sc" / "mnk_dc")
    def set_aug_diagonal_device(self, x, xl, xu, zl, zu, primal_reg=0.0, dual_reg=0.0):
        """`set_aug_diagonal!(kkt, solver)` (reference `src/IPM/kernels.jl:4-27`) inside the handle: `x, xl, xu, zl, zu`
        are the full primal-length vectors of the iterate (host arrays or device tensors); reg, du_diag, l_diag,
        u_diag, l_lower, u_lower and pr_diag are computed and kept on the device (no host vector involved)."""
        ptrs = [_ptr(v) for v in (x, xl, xu, zl, zu)]
        assert len({loc for _, loc in ptrs}) == 1, "all five vectors must live on the same side"
        fn = getattr(L.lib(), self._PFX + "_set_aug_diagonal")
        L.check(fn(self._h, *[p for p, _ in ptrs], float(primal_reg), float(dual_reg), ptrs[0][1]),
                self._PFX + "_set_aug_diagonal")

    def set_aug_RR_device(self, x, xl, xu, zl, zu, D_R, pp, zp, nn, zn, zeta, primal_reg, dual_reg):
        """`set_aug_RR!(kkt, solver, RR)` (reference `src/IPM/kernels.jl:72-87`) inside the handle, from device tensors:
        reg = primal_reg + zeta D_R^2, du_diag = -dual_reg - pp/zp - nn/zn, bound terms and pr_diag as
        `set_aug_diagonal!`."""
        ptrs = [_ptr(v) for v in (x, xl, xu, zl, zu, D_R, pp, zp, nn, zn)]
        assert all(loc == L.MNK_DEVICE for _, loc in ptrs), "set_aug_RR_device takes device tensors"
        fn = getattr(L.lib(), self._PFX + "_set_aug_RR")
        L.check(fn(self._h, *[p for p, _ in ptrs], float(zeta), float(primal_reg), float(dual_reg)), self._PFX + "_set_aug_RR")

    def regularize_diagonal_device(self, primal, dual):
        """`regularize_diagonal!(kkt, primal, dual)` (reference `src/KKT/KKTsystem.jl:222-226`) on the handle's own
        reg / pr_diag / du_diag."""
        fn = getattr(L.lib(), self._PFX + "_regularize_diagonal")
        L.check(fn(self._h, float(primal), float(dual)), self._PFX + "_regularize_diagonal")

    def build_kkt_device(self):
        """`build_kkt!` from the diagonals the handle keeps itself (after `set_aug_diagonal_device`)."""
        fn = getattr(L.lib(), self._PFX + "_build")
        L.check(fn(self._h, None, None, L.MNK_DEVICE), self._PFX + "_build")
        self._diag_buffer = None

    def get_diagonals_device(self):
        """Host copies of the handle's pr_diag, du_diag, reg, l_diag, u_diag, l_lower, u_lower (tests)."""
        out = {"pr_diag": np.zeros_like(self.pr_diag), "du_diag": np.zeros_like(self.du_diag),
               "reg": np.zeros_like(self.reg), "l_diag": np.zeros_like(self.l_diag), "u_diag": np.zeros_like(self.u_diag),
               "l_lower": np.zeros_like(self.l_lower), "u_lower": np.zeros_like(self.u_lower)}
        fn = getattr(L.lib(), self._PFX + "_get_diagonals")
        L.check(fn(self._h, *[out[k].ctypes.data for k in ("pr_diag", "du_diag", "reg", "l_diag", "u_diag", "l_lower",
                                                            "u_lower")]), self._PFX + "_get_diagonals")
        return out

    def factorize_kkt(self):
        """`factorize_kkt!` reference `src/KKT/KKTsystem.jl:218-220`."""
        return self.linear_solver.factorize()

    def get_kkt(self):
        return self.aug_com

    def get_hessian(self):
        return self.hess

    def get_jacobian(self):
        return self.jac

    def size(self):
        return (self._order, self._order)

    def _reduce_rhs(self, d):
        """`reduce_rhs!` reference `src/IPM/kernels.jl:182-195`."""
        v = d.values
        v[d.ind_lb] -= d.dual_lb() / self.l_diag
        v[d.ind_ub] -= d.dual_ub() / self.u_diag

    def _finish_aug_solve(self, d):
        """`finish_aug_solve!` reference `src/IPM/kernels.jl:198-204`."""
        v = d.values
        dlb, dub = d.dual_lb(), d.dual_ub()
        dlb[:] = (-dlb + self.l_lower * v[d.ind_lb]) / self.l_diag
        dub[:] = (dub - self.u_lower * v[d.ind_ub]) / self.u_diag

    def _kktmul(self, w, x, alpha, beta):
        """`_kktmul!` reference `src/IPM/kernels.jl:161-180`."""
        wv, xv = w.values, x.values
        w.primal()[:] += alpha * self.reg * x.primal()
        w.dual()[:] += alpha * self.du_diag * x.dual()
        wv[w.ind_lb] -= alpha * x.dual_lb()
        wv[w.ind_ub] += alpha * x.dual_ub()
        w.dual_lb()[:] = beta * w.dual_lb() + alpha * (xv[x.ind_lb] * self.l_lower - x.dual_lb() * self.l_diag)
        w.dual_ub()[:] = beta * w.dual_ub() + alpha * (xv[x.ind_ub] * self.u_lower + x.dual_ub() * self.u_diag)


# ------------------------------------------------------------ SparseCondensedKKTSystem
class SparseCondensedKKTSystem(_KKTCommon):
    _PFX = "mnk_sc"
    """`create_kkt_system(SparseCondensedKKTSystem, cb, linear_solver)` (reference
    `src/KKT/Sparse/condensed.jl:55-133`).  `jac_I/jac_J`, `hess_I/hess_J` are the COO
    sparsity patterns reported by the callback (0-based)."""

    def __init__(self, n, m, jac_I, jac_J, hess_I, hess_J, ind_ineq, ind_lb, ind_ub,
                 ctx: HipContext | None = None, linear_solver=HipLinearSolver,
                 opt_linear_solver: HipSolverOptions | None = None, device_kkt_ops: bool = False, early_reject: bool = True):
        """`device_kkt_ops`: run `solve_kkt!` and `mul!` entirely on the device (`mnk_sc_solve_kkt`,
        `mnk_sc_mul`): the primal-dual vector makes one round trip per call instead of the host doing the
        vector algebra around a device solve.  `early_reject`: a matrix that is not positive definite is reported as such from
        its first non-positive pivot on, without finishing its factorization (its inertia is then a lower bound on num_neg,
        it has no usable factor; `False`: every factorization runs to the end and reports the signs of all pivots)."""
        if len(ind_ineq) != m:
            raise ValueError("SparseCondensedKKTSystem does not support equality constrained NLPs.")
        self.device_kkt_ops = bool(device_kkt_ops)
        self.ctx = ctx or HipContext()
        self.n, self.m = int(n), int(m)
        jI = np.ascontiguousarray(jac_I, dtype=np.int32)
        jJ = np.ascontiguousarray(jac_J, dtype=np.int32)
        hI = np.ascontiguousarray(hess_I, dtype=np.int32)
        hJ = np.ascontiguousarray(hess_J, dtype=np.int32)
        lib = L.lib()
        self._h = C.c_void_p()
        L.check(lib.mnk_sc_create(self.ctx.handle, n, m, len(jI), jI.ctypes.data, jJ.ctypes.data, len(hI),
                                  hI.ctypes.data, hJ.ctypes.data, 0, C.byref(self._h)), "mnk_sc_create")
        s = [C.c_int64() for _ in range(4)]
        L.check(lib.mnk_sc_sizes(self._h, *[C.byref(v) for v in s]), "mnk_sc_sizes")
        self.nnz_jt, self.nnz_hess, self.nnz_aug, self.len_jptr = [v.value for v in s]
        self.jt_colptr, self.jt_rowval = self._structure(L.MNK_SC_JT, m, self.nnz_jt)
        self.hess_colptr, self.hess_rowval = self._structure(L.MNK_SC_HESS, n, self.nnz_hess)
        aug_colptr, aug_rowval = self._structure(L.MNK_SC_AUG, n, self.nnz_aug)
        self.jt_csc_map = self._map(L.MNK_SC_JT, len(jI))
        self.hess_csc_map = self._map(L.MNK_SC_HESS, len(hI))
        # host-visible fields of the reference struct
        self.hess = np.zeros(len(hI))
        self.jac = np.zeros(len(jI))
        nlb, nub = len(ind_lb), len(ind_ub)
        self.reg = np.zeros(n + m)
        self.pr_diag = np.zeros(n + m)
        self.du_diag = np.zeros(m)
        self.l_diag = np.zeros(nlb)
        self.u_diag = np.zeros(nub)
        self.l_lower = np.zeros(nlb)
        self.u_lower = np.zeros(nub)
        self.buffer = np.zeros(m)
        self.buffer2 = np.zeros(m)
        self.ind_ineq = np.asarray(ind_ineq, dtype=np.int64)
        self.ind_lb = np.asarray(ind_lb, dtype=np.int64)
        self.ind_ub = np.asarray(ind_ub, dtype=np.int64)
        self.aug_com = DeviceCSC(self, n, aug_colptr.astype(np.int64), aug_rowval.astype(np.int64))
        self._order = n
        self._host_jt = None
        self._host_h = None
        self._diag_buffer = None
        self.linear_solver = linear_solver(self.aug_com, ctx=self.ctx, opt=opt_linear_solver)
        # is_inertia_correct accepts (n, 0, 0) only (condensed.jl:138-140): "not positive definite" from the static-pivot tier
        # is final, the pivoted tier could only confirm the rejection
        if hasattr(self.linear_solver, "set_option"):
            self.linear_solver.set_option("accept_only_pd", 1)
            # ... and since should_regularize_dual is `true` whatever the counts (condensed.jl:141), nothing but "positive
            # definite or not" is ever read from the inertia of this system: the factorization of a matrix that is not may stop
            # at its first non-positive pivot (as dpotrf does) instead of running to the end (as dsytrf does)
            self.linear_solver.set_option("early_reject", 1 if early_reject else 0)
        L.check(lib.mnk_sc_set_bounds(self._h, nlb, self.ind_lb.ctypes.data, nub, self.ind_ub.ctypes.data, 0),
                "mnk_sc_set_bounds")
        self._spare_args = (linear_solver, opt_linear_solver, early_reject)
        self.spare_solver = None
        _LIVE_OBJECTS.add(self)

    # -- speculative inertia correction (madnlp_jl_amd.ipm_dev.DeviceMadNLPSolver.inertia_correction) ---------------------
    def ensure_spare_solver(self):
        """A second linear solver on the same `aug_com` (created on first use: one more factor in HBM): the trial of
        `inertia_correction!` that is factorized ahead of the verdict on the unperturbed matrix, in the same merged launch."""
        if self.spare_solver is None:
            cls, opt, early_reject = self._spare_args
            self.spare_solver = cls(self.aug_com, ctx=self.ctx, opt=opt)
            self.spare_solver.set_option("accept_only_pd", 1)
            self.spare_solver.set_option("early_reject", 1 if early_reject else 0)
        return self.spare_solver

    def probe_solver(self, m):
        """A solver of order `m` < n on the leading principal block of `aug_com` (cached per order): if that block is not positive
        definite neither is the matrix, and the block costs (m / n)^3 of a factorization where the full static-pivot elimination
        has done 1 - (1 - m / n)^3 of its work by the time it reaches column m."""
        from .linear_solver import LeadingBlock
        if not hasattr(self, "_probes"):
            self._probes = {}
        if m not in self._probes:
            cls, opt, _ = self._spare_args
            ps = cls(LeadingBlock(self.aug_com, m), ctx=self.ctx, opt=opt)
            ps.set_option("accept_only_pd", 1)
            ps.set_option("early_reject", 1)
            self._probes[m] = ps
        return self._probes[m]

    def swap_solvers(self):
        """The spare solver's factor is the one the iteration goes on with."""
        self.linear_solver, self.spare_solver = self.spare_solver, self.linear_solver

    def save_diagonals_device(self):
        L.check(L.lib().mnk_sc_save_diagonals(self._h), "mnk_sc_save_diagonals")

    def restore_diagonals_device(self):
        L.check(L.lib().mnk_sc_restore_diagonals(self._h), "mnk_sc_restore_diagonals")

    # -- helpers -----------------------------------------------------------------------
    def _structure(self, which, ncol, nnz):
        colptr = np.zeros(ncol + 1, dtype=np.int32)
        rowval = np.zeros(max(nnz, 1), dtype=np.int32)
        L.check(L.lib().mnk_sc_get_structure(self._h, which, colptr.ctypes.data, rowval.ctypes.data),
                "mnk_sc_get_structure")
        return colptr, rowval[:nnz]

    def _map(self, which, nnz):
        mp = np.zeros(max(nnz, 1), dtype=np.int64)
        L.check(L.lib().mnk_sc_get_map(self._h, which, mp.ctypes.data), "mnk_sc_get_map")
        return mp[:nnz]

    def _values(self, which, count):
        out = np.zeros(max(count, 1))
        L.check(L.lib().mnk_sc_get_values(self._h, which, out.ctypes.data, L.MNK_HOST), "mnk_sc_get_values")
        return out[:count]

    def get_ptrs(self):
        """(dptr, hptr, jptr) as in the oracle's build_condensed_aug_symbolic."""
        n, nh, Lj = self.n, self.nnz_hess, self.len_jptr
        arrs = [np.zeros(max(k, 1), dtype=np.int32) for k in (n, n, nh, nh, Lj, Lj, Lj, Lj)]
        L.check(L.lib().mnk_sc_get_ptrs(self._h, *[a.ctypes.data for a in arrs]), "mnk_sc_get_ptrs")
        d = (arrs[0][:n], arrs[1][:n])
        h = (arrs[2][:nh], arrs[3][:nh])
        j = tuple(a[:Lj] for a in arrs[4:])
        return d, h, j

    @property
    def jt_csc(self):
        """Host copy of jt_csc (n x m) with current values (fetched lazily)."""
        if self._host_jt is None:
            self._host_jt = sp.csc_matrix((self._values(L.MNK_SC_JT, self.nnz_jt), self.jt_rowval,
                                           self.jt_colptr), shape=(self.n, self.m))
        return self._host_jt

    @property
    def hess_com(self):
        if self._host_h is None:
            self._host_h = sp.csc_matrix((self._values(L.MNK_SC_HESS, self.nnz_hess), self.hess_rowval,
                                          self.hess_colptr), shape=(self.n, self.n))
        return self._host_h

    @property
    def diag_buffer(self):
        if self._diag_buffer is None:
            self._diag_buffer = self._values(L.MNK_SC_DIAGBUF, self.m)
        return self._diag_buffer

    # -- interface ---------------------------------------------------------------------
    def num_variables(self):
        return len(self.pr_diag)

    def initialize(self):
        """sparse `initialize!` reference `src/KKT/Sparse/utils.jl:52-62`."""
        super().initialize()
        self.l_lower[:] = 0.0
        self.u_lower[:] = 0.0
        self.l_diag[:] = 1.0
        self.u_diag[:] = 1.0
        self.compress_hessian()  # hess == 0 -> nonzeros(hess_com) = 0

    def compress_jacobian(self, jac=None):
        """`compress_jacobian!` reference `src/KKT/Sparse/condensed.jl:145-148`.  `jac`
        may be a device tensor with the COO values; default: the host field `self.jac`."""
        p, loc = _ptr(self.jac if jac is None else jac)
        L.check(L.lib().mnk_sc_compress_jacobian(self._h, p, loc), "mnk_sc_compress_jacobian")
        self._host_jt = None

    def compress_hessian(self, hess=None):
        """`compress_hessian!` reference `src/KKT/Sparse/utils.jl:48-50`."""
        p, loc = _ptr(self.hess if hess is None else hess)
        L.check(L.lib().mnk_sc_compress_hessian(self._h, p, loc), "mnk_sc_compress_hessian")
        self._host_h = None

    def build_kkt(self, pr_diag=None, du_diag=None):
        """`build_kkt!` reference `src/KKT/Sparse/condensed.jl:354-366`."""
        pp, loc = _ptr(self.pr_diag if pr_diag is None else pr_diag)
        dp, loc2 = _ptr(self.du_diag if du_diag is None else du_diag)
        assert loc == loc2
        L.check(L.lib().mnk_sc_build(self._h, pp, dp, loc), "mnk_sc_build")
        self._diag_buffer = None
        if self.device_kkt_ops:
            self.upload_barrier_terms()

    def upload_barrier_terms(self):
        """reg, l_diag, u_diag, l_lower, u_lower of the current iterate -> device (for the device-side
        `solve_kkt!` / `mul!`)."""
        L.check(L.lib().mnk_sc_set_barrier_terms(self._h, self.reg.ctypes.data, self.l_diag.ctypes.data,
                                                 self.u_diag.ctypes.data, self.l_lower.ctypes.data,
                                                 self.u_lower.ctypes.data, L.MNK_HOST), "mnk_sc_set_barrier_terms")

    def solve_kkt_device(self, w):
        """`solve_kkt!` on the device (reference `src/IPM/factorization.jl:143-167`); `w` is an
        UnreducedKKTVector (host values) or a device tensor holding its values."""
        p, loc = _ptr(w.values if isinstance(w, UnreducedKKTVector) else w)
        rc = L.lib().mnk_sc_solve_kkt(self._h, self.linear_solver._h, p, loc)
        if rc:
            from .linear_solver import SolveException
            raise SolveException(L.lib().mnk_last_error_string().decode())
        return w

    def mul_device(self, w, x, alpha=1.0, beta=0.0):
        """`mul!(w, kkt, x, alpha, beta)` on the device (reference `src/IPM/factorization.jl:289-308`)."""
        pw, loc = _ptr(w.values if isinstance(w, UnreducedKKTVector) else w)
        px, loc2 = _ptr(x.values if isinstance(x, UnreducedKKTVector) else x)
        assert loc == loc2
        L.check(L.lib().mnk_sc_mul(self._h, pw, px, float(alpha), float(beta), loc), "mnk_sc_mul")
        return w

    def spmv_device(self, which, trans, alpha, x, beta, y):
        """y = alpha op(A) x + beta y on device vectors with A = jt_csc (`MNK_SC_JT`; trans 0: n <- m, 1: m <- n) or
        Symmetric(hess_com, :L) (`MNK_SC_HESS`), the compressed values the handle holds (`mnk_sc_spmv`)."""
        px, lx = _ptr(x)
        py, ly = _ptr(y)
        assert lx == ly == L.MNK_DEVICE
        L.check(L.lib().mnk_sc_spmv(self._h, int(which), int(trans), float(alpha), px, float(beta), py), "mnk_sc_spmv")

    def is_inertia_correct(self, num_pos, num_zero, num_neg):
        """reference `src/KKT/Sparse/condensed.jl:138-140`."""
        return num_zero == 0 and num_pos == self.n

    def should_regularize_dual(self, num_pos, num_zero, num_neg):
        return True

    def jtprod(self, y, x):
        """reference `src/KKT/Sparse/condensed.jl:150-156`."""
        y[:self.n] = self.jt_csc @ x
        y[self.n:] = -x
        return y

    def solve_kkt(self, w):
        """reference `src/IPM/factorization.jl:143-167`; the condensed solve runs on the device."""
        if self.device_kkt_ops:
            return self.solve_kkt_device(w)
        n, m = self.n, self.m
        full = w.values
        wx, ws, wz = full[:n], full[n:n + m], full[n + m:n + 2 * m]
        Ss = self.pr_diag[n:n + m]
        D = self.diag_buffer
        self._reduce_rhs(w)
        self.buffer[:] = D * (wz + ws / Ss)
        wx += self.jt_csc @ self.buffer
        self.linear_solver.solve_linear_system(wx)
        self.buffer2[:] = self.jt_csc.T @ wx
        wz[:] = -self.buffer + D * self.buffer2
        ws[:] = (ws + wz) / Ss
        self._finish_aug_solve(w)
        return w

    def mul(self, w, x, alpha=1.0, beta=0.0):
        """reference `src/IPM/factorization.jl:278-299`."""
        if self.device_kkt_ops:
            return self.mul_device(w, x, alpha, beta)
        n, m = self.n, self.m
        xf, wf = x.values, w.values
        xx, xs, xz = xf[:n], xf[n:n + m], xf[n + m:n + 2 * m]
        wx, ws, wz = wf[:n], wf[n:n + m], wf[n + m:n + 2 * m]
        H = self.hess_com
        hx = H @ xx + H.T @ xx - H.diagonal() * xx
        wx[:] = alpha * hx + beta * wx
        wx += alpha * (self.jt_csc @ xz)
        wz[:] = alpha * (self.jt_csc.T @ xx) + beta * wz
        wz -= alpha * xs
        ws[:] = beta * ws - alpha * xz
        self._kktmul(w, x, alpha, beta)
        return w

    def mul_hess_blk(self, wx, t):
        """reference `src/IPM/factorization.jl:333-338`."""
        n = self.n
        H = self.hess_com
        wx[:n] = H @ t[:n] + H.T @ t[:n] - H.diagonal() * t[:n]
        wx[n:] = 0.0
        wx += t * self.pr_diag
        return wx

    def close(self):
        if getattr(self, "linear_solver", None) is not None:
            self.linear_solver.close()
        if getattr(self, "spare_solver", None) is not None:
            self.spare_solver.close()
            self.spare_solver = None
        for ps in getattr(self, "_probes", {}).values():
            ps.close()
        self._probes = {}
        if self._h:
            L.lib().mnk_sc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------- dense systems
class _DenseBase(_KKTCommon):
    _PFX = "mnk_dc"
    def _create(self, condensed, n, m, ind_ineq, ind_eq, ctx):
        self.ctx = ctx or HipContext()
        ii = np.ascontiguousarray(ind_ineq, dtype=np.int64)
        ie = np.ascontiguousarray(ind_eq, dtype=np.int64)
        self._h = C.c_void_p()
        L.check(L.lib().mnk_dc_create(self.ctx.handle, condensed, n, m, len(ii),
                                      ii.ctypes.data if len(ii) else None,
                                      ie.ctypes.data if len(ie) else None, 0, C.byref(self._h)), "mnk_dc_create")
        self._order = L.lib().mnk_dc_order(self._h)
        _LIVE_OBJECTS.add(self)

    def compress_jacobian(self):
        """no-op, reference `src/KKT/Dense/utils.jl:25-27`."""
        return

    device_kkt_ops = False

    def _setup_device_ops(self, device_kkt_ops):
        """Upload the bound structure once; with `device_kkt_ops` `solve_kkt!` / `mul!` run on the device
        (`mnk_dc_solve_kkt`, `mnk_dc_mul`)."""
        self.device_kkt_ops = bool(device_kkt_ops)
        L.check(L.lib().mnk_dc_set_bounds(self._h, len(self.ind_lb), self.ind_lb.ctypes.data, len(self.ind_ub),
                                          self.ind_ub.ctypes.data, 0), "mnk_dc_set_bounds")

    def upload_barrier_terms(self):
        L.check(L.lib().mnk_dc_set_barrier_terms(self._h, self.reg.ctypes.data, self.l_diag.ctypes.data,
                                                 self.u_diag.ctypes.data, self.l_lower.ctypes.data,
                                                 self.u_lower.ctypes.data, L.MNK_HOST), "mnk_dc_set_barrier_terms")

    def solve_kkt_device(self, w):
        p, loc = _ptr(w.values if isinstance(w, UnreducedKKTVector) else w)
        rc = L.lib().mnk_dc_solve_kkt(self._h, self.linear_solver._h, p, loc)
        if rc:
            from .linear_solver import SolveException
            raise SolveException(L.lib().mnk_last_error_string().decode())
        return w

    def mul_device(self, w, x, alpha=1.0, beta=0.0):
        pw, loc = _ptr(w.values if isinstance(w, UnreducedKKTVector) else w)
        px, loc2 = _ptr(x.values if isinstance(x, UnreducedKKTVector) else x)
        assert loc == loc2
        L.check(L.lib().mnk_dc_mul(self._h, pw, px, float(alpha), float(beta), loc), "mnk_dc_mul")
        return w

    def _upload(self):
        lib = L.lib()
        L.check(lib.mnk_dc_set_hess(self._h, self.hess.ctypes.data, self.hess.shape[0], L.MNK_HOST), "mnk_dc_set_hess")
        if self.m > 0:
            L.check(lib.mnk_dc_set_jac(self._h, self.jac.ctypes.data, self.jac.shape[0], L.MNK_HOST), "mnk_dc_set_jac")

    def build_kkt(self):
        """`build_kkt!` (reference `src/KKT/Dense/condensed.jl:157-186` /
        `src/KKT/Dense/augmented.jl:147-156`).  The callbacks wrote `hess`/`jac` on the host."""
        self._upload()
        L.check(L.lib().mnk_dc_build(self._h, self.pr_diag.ctypes.data, self.du_diag.ctypes.data, L.MNK_HOST),
                "mnk_dc_build")
        if self.device_kkt_ops:
            self.upload_barrier_terms()

    def jtprod(self, y, x):
        """reference `src/KKT/Dense/utils.jl:12-23`."""
        nx = self.hess.shape[0]
        ns = len(self.ind_ineq)
        y[:nx] = self.jac.T @ x
        y[nx:nx + ns] = -x[self.ind_ineq]
        return y

    def mul(self, w, x, alpha=1.0, beta=0.0):
        """reference `src/IPM/factorization.jl:301-324`."""
        if self.device_kkt_ops:
            return self.mul_device(w, x, alpha, beta)
        m, n = self.jac.shape
        wp, xp = w.primal(), x.primal()
        wx, ws = wp[:n], wp[n:]
        xx, xs = xp[:n], xp[n:]
        wy, xy = w.dual(), x.dual()
        Hl = np.tril(self.hess)
        wx[:] = alpha * (Hl @ xx + np.tril(self.hess, -1).T @ xx) + beta * wx
        if m > 0:
            wx += alpha * (self.jac.T @ xy)
            wy[:] = alpha * (self.jac @ xx) + beta * wy
        ws[:] = beta * ws - alpha * xy[self.ind_ineq]
        wy[self.ind_ineq] -= alpha * xs
        self._kktmul(w, x, alpha, beta)
        return w

    def mul_hess_blk(self, wx, t):
        """reference `src/IPM/factorization.jl:326-331`."""
        n = self.hess.shape[0]
        wx[:n] = np.tril(self.hess) @ t[:n] + np.tril(self.hess, -1).T @ t[:n]
        wx[n:] = 0.0
        wx += t * self.pr_diag
        return wx

    def close(self):
        if getattr(self, "linear_solver", None) is not None:
            self.linear_solver.close()
        if self._h:
            L.lib().mnk_dc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DenseKKTSystem(_DenseBase):
    """reference `src/KKT/Dense/augmented.jl:10-94` (order n + ns + m)."""

    def __init__(self, n, m, ind_ineq, ind_lb, ind_ub, ctx=None, linear_solver=HipLinearSolver,
                 opt_linear_solver=None, device_kkt_ops=False):
        ns = len(ind_ineq)
        self.n, self.m, self.ns = n, m, ns
        self._create(0, n, m, ind_ineq, [], ctx)
        self.hess = np.zeros((n, n), order="F")
        self.jac = np.zeros((m, n), order="F")
        self.reg = np.zeros(n + ns)
        self.pr_diag = np.zeros(n + ns)
        self.du_diag = np.zeros(m)
        self.diag_hess = np.zeros(n)
        nlb, nub = len(ind_lb), len(ind_ub)
        self.l_diag, self.u_diag = np.ones(nlb), np.ones(nub)
        self.l_lower, self.u_lower = np.zeros(nlb), np.zeros(nub)
        self.ind_ineq = np.asarray(ind_ineq, dtype=np.int64)
        self.ind_lb = np.asarray(ind_lb, dtype=np.int64)
        self.ind_ub = np.asarray(ind_ub, dtype=np.int64)
        self.aug_com = DeviceDense(self, self._order)
        self.linear_solver = linear_solver(self.aug_com, ctx=self.ctx, opt=opt_linear_solver)
        self._setup_device_ops(device_kkt_ops)

    def num_variables(self):
        return len(self.pr_diag)

    def compress_hessian(self):
        """`compress_hessian!` reference `src/KKT/Dense/augmented.jl:158-161` (diag!)."""
        self.diag_hess[:] = np.diagonal(self.hess)

    def is_inertia_correct(self, num_pos, num_zero, num_neg):
        return num_zero == 0 and num_pos == self.num_variables()

    def should_regularize_dual(self, num_pos, num_zero, num_neg):
        return num_zero != 0

    def solve_kkt(self, w):
        """reference `src/IPM/factorization.jl:41-46`."""
        if self.device_kkt_ops:
            return self.solve_kkt_device(w)
        self._reduce_rhs(w)
        self.linear_solver.solve_linear_system(w.primal_dual())
        self._finish_aug_solve(w)
        return w


class DenseCondensedKKTSystem(_DenseBase):
    """reference `src/KKT/Dense/condensed.jl:10-111` (order n + n_eq)."""

    def __init__(self, n, m, ind_ineq, ind_eq, ind_lb, ind_ub, ctx=None, linear_solver=HipLinearSolver,
                 opt_linear_solver=None, device_kkt_ops=False):
        ns = len(ind_ineq)
        self.n, self.m, self.n_ineq, self.n_eq = n, m, ns, m - ns
        assert self.n_eq == len(ind_eq)
        self._create(1, n, m, ind_ineq, ind_eq, ctx)
        self.hess = np.zeros((n, n), order="F")
        self.jac = np.zeros((m, n), order="F")
        self.reg = np.zeros(n + ns)
        self.pr_diag = np.zeros(n + ns)
        self.du_diag = np.zeros(m)
        nlb, nub = len(ind_lb), len(ind_ub)
        self.l_diag, self.u_diag = np.ones(nlb), np.ones(nub)
        self.l_lower, self.u_lower = np.zeros(nlb), np.zeros(nub)
        self.pd_buffer = np.zeros(n + self.n_eq)
        self.diag_buffer = np.zeros(ns)
        self.buffer = np.zeros(m)
        self.ind_ineq = np.asarray(ind_ineq, dtype=np.int64)
        self.ind_eq = np.asarray(ind_eq, dtype=np.int64)
        self.ind_lb = np.asarray(ind_lb, dtype=np.int64)
        self.ind_ub = np.asarray(ind_ub, dtype=np.int64)
        self.ind_eq_shifted = self.ind_eq + n + ns
        self.ind_ineq_shifted = self.ind_ineq + n + ns
        self.aug_com = DeviceDense(self, self._order)
        self.linear_solver = linear_solver(self.aug_com, ctx=self.ctx, opt=opt_linear_solver)
        if self.n_eq == 0 and hasattr(self.linear_solver, "set_option"):   # is_inertia_correct: num_zero == 0 && num_neg == n_eq
            self.linear_solver.set_option("accept_only_pd", 1)
        self._setup_device_ops(device_kkt_ops)

    def num_variables(self):
        return self.hess.shape[0]

    def compress_hessian(self):
        return

    def build_kkt(self):
        n, ns = self.n, self.n_ineq
        Ss = self.pr_diag[n:n + ns]
        # host copy of diag_buffer for solve_kkt!'s vector algebra (reference condensed.jl:166-168)
        self.diag_buffer[:] = Ss / (1.0 - self.du_diag[self.ind_ineq] * Ss)
        super().build_kkt()

    def is_inertia_correct(self, num_pos, num_zero, num_neg):
        """reference `src/KKT/Dense/condensed.jl:189-191`."""
        return num_zero == 0 and num_neg == self.n_eq

    def should_regularize_dual(self, num_pos, num_zero, num_neg):
        return num_zero != 0

    def solve_kkt(self, w):
        """reference `src/IPM/factorization.jl:190-229`."""
        if self.device_kkt_ops:
            return self.solve_kkt_device(w)
        n, n_eq, ns = self.n, self.n_eq, self.n_ineq
        full = w.values
        wx, ws = full[:n], full[n:n + ns]
        x = self.pd_buffer
        xx, xy = x[:n], x[n:n + n_eq]
        Ss = self.pr_diag[n:n + ns]
        self._reduce_rhs(w)
        wz = full[self.ind_ineq_shifted]
        wy = full[self.ind_eq_shifted]
        self.buffer[:] = 0.0
        self.buffer[self.ind_ineq] = self.diag_buffer * (wz + ws / Ss)
        xx[:] = self.jac.T @ self.buffer
        xx += wx
        xy[:] = wy
        self.linear_solver.solve_linear_system(x)
        wx[:] = xx
        dual = w.dual()
        dual[:] = self.jac @ wx
        full[self.ind_eq_shifted] = xy
        full[self.ind_ineq_shifted] *= self.diag_buffer
        dual -= self.buffer
        ws[:] = (ws + full[self.ind_ineq_shifted]) / Ss
        self._finish_aug_solve(w)
        return w


class ScenarioBatch:
    """n independent sparse condensed KKT systems driven together (BASELINE config C5: scenario batches on one GPU): one
    library call per phase of an interior-point iteration instead of four to six per instance (`mnk_sc_step_batch`,
    `mnk_ls_inertia_batch`, `mnk_ls_solve_batch`).  The factorizations of a step run as ONE batch (merged persistent launch),
    the solves up to four systems per launch.  All value arguments are device tensors; the pointer tables are built once per
    set of tensors (`bind`)."""

    def __init__(self, kkts):
        import ctypes as C
        self.kkts = list(kkts)
        n = self.n = len(self.kkts)
        self._sc = (C.c_void_p * n)(*[k._h.value for k in self.kkts])
        self._ls = (C.c_void_p * n)(*[k.linear_solver._h.value for k in self.kkts])
        self._pos, self._zero, self._neg = (C.c_int64 * n)(), (C.c_int64 * n)(), (C.c_int64 * n)()
        self._tabs = {}

    @staticmethod
    def _table(tensors):
        import ctypes as C
        return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])

    def bind(self, jacs, hesss, prs, dus):
        """Device tensors of the COO Jacobian / Hessian values and of pr_diag / du_diag of every instance (kept by reference:
        the tensors must stay alive and in place)."""
        self._keep = (list(jacs), list(hesss), list(prs), list(dus))
        self._tabs["step"] = tuple(self._table(t) for t in self._keep)

    def step(self):
        """compress_jacobian! + compress_hessian! + build_kkt! + factorize! of every instance; asynchronous."""
        j, h, p, d = self._tabs["step"]
        L.check(L.lib().mnk_sc_step_batch(self.n, self._sc, self._ls, j, h, p, d, L.MNK_DEVICE), "mnk_sc_step_batch")
        for k in self.kkts:
            k._host_jt = k._host_h = k._diag_buffer = None

    def inertia(self):
        L.check(L.lib().mnk_ls_inertia_batch(self.n, self._ls, self._pos, self._zero, self._neg), "mnk_ls_inertia_batch")
        return [(self._pos[i], self._zero[i], self._neg[i]) for i in range(self.n)]

    def solve(self, xs):
        """One right-hand side per instance (device tensors, in place); asynchronous."""
        key = tuple(x.data_ptr() for x in xs)
        tab = self._tabs.get(key)
        if tab is None:
            tab = self._tabs[key] = self._table(xs)
        L.check(L.lib().mnk_ls_solve_batch(self.n, self._ls, tab, L.MNK_DEVICE), "mnk_ls_solve_batch")
