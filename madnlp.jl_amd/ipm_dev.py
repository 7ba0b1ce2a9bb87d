"""Device-resident regular phase of the IPM mirror: the iterate, the multipliers, the KKT vectors and the
right-hand sides live in HBM; per iteration only scalars cross PCIe.  Same algorithm, same order of operations as
`madnlp_jl_amd.ipm.MadNLPSolver` (which documents the reference lines); the vector work goes through

  * the KKT handle:   `set_aug_diagonal!`, `regularize_diagonal!`, `build_kkt!`, `factorize!`, `solve_kkt!`, `mul!`, SpMV
  * `mnk_ipm_*`:      the reductions and elementwise pieces of reference `src/IPM/kernels.jl`, and the loop's plain vector
                      work (`mnk_ipm_vec_*`, `mnk_ipm_get_dot`, `mnk_ipm_gemv`)
  * `mnk_opf_*`:      the callbacks of the polar AC-OPF model (`problems.ACOPFModel`) evaluated on the device
  * torch:            device memory only (allocation, uploads, the final download) -- no torch kernel runs in the loop

Scope: `SparseCondensedKKTSystem` (all constraints relaxed to inequalities, as the reference's preset does) and
`DenseCondensedKKTSystem` (equalities allowed), with models whose callbacks can be evaluated on the device: the AC-OPF NLP
(`DeviceOPFCallbacks`), the sparse QP through the KKT handle's own SpMV on the compressed Jacobian / Hessian, the dense QP
through `mnk_ipm_gemv` on device copies of P and A.  Initialization runs once on the host (the base class) and is uploaded."""
from __future__ import annotations


import ctypes as C

import numpy as np
import torch

from . import _lib as L
from .ipm import EPS, INF, IPMOptions, MadNLPSolver, _pow
from .ipm_device import IPMDeviceKernels


def _up(a, dev, dtype=np.float64):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(dev)


class DeviceQPCallbacks:
    """f = 0.5 x'Hx + q'x, c = Jx with H = Symmetric(hess_com, :L) and J = jt_csc' held by the KKT handle."""

    def __init__(self, nlp, kkt, dev, K):
        self.kkt, self.K, self.n, self.m = kkt, K, nlp.n, nlp.m
        self.q = _up(nlp.q, dev)
        self.jv = _up(nlp.jac_coord(None), dev)
        self.hv = _up(nlp.hess_coord(None, None, 1.0), dev)
        self.hv0 = torch.empty_like(self.hv)
        K.vec_fill(self.hv0, 0.0)
        self.handle_has_H = True
        self._hx = torch.empty(self.n, dtype=torch.float64, device=dev)

    def _hmul(self, x):
        # the SpMV runs on the handle's compressed Lagrangian Hessian: while that holds something else (zero after
        # initialize!(kkt), objective weight 0 in robust!) H is put back for the product
        if not self.handle_has_H:
            self.kkt.compress_hessian(self.hv)
        self.kkt.spmv_device(L.MNK_SC_HESS, 0, 1.0, x, 0.0, self._hx)
        if not self.handle_has_H:
            self.kkt.compress_hessian(self.hv0)
        return self._hx

    def obj(self, x):
        hx = self._hmul(x)
        with self.K.batch():
            a = self.K.get_dot(x, hx)
            b = self.K.get_dot(self.q, x)
        return 0.5 * a[0] + b[0]

    def grad(self, g, x):
        self.K.vec_axpby(g, 1.0, self._hmul(x), 1.0, self.q)

    def cons(self, c, x):
        self.kkt.spmv_device(L.MNK_SC_JT, 1, 1.0, x, 0.0, c)

    def jtprod_x(self, out_x, y):
        self.kkt.spmv_device(L.MNK_SC_JT, 0, 1.0, y, 0.0, out_x)

    def load_jac(self, x=None):
        self.kkt.compress_jacobian(self.jv)

    def load_hess(self, x=None, y=None, sigma=1.0):
        zero = sigma == 0.0            # linear constraints: the Lagrangian Hessian is sigma H
        self.kkt.compress_hessian(self.hv0 if zero else self.hv)
        self.handle_has_H = not zero

    def zero_hess(self):
        self.load_hess(sigma=0.0)

    def close(self):
        pass


class DeviceDenseQPCallbacks:
    """f = 0.5 x'Px + q'x, c = Ax with dense P, A on the device (`DenseQPModel`); Hessian / Jacobian of the KKT handle are
    loaded from these device copies (`mnk_dc_set_hess` / `mnk_dc_set_jac` with device pointers)."""

    def __init__(self, nlp, kkt, dev, K):
        self.kkt, self.K, self.n, self.m = kkt, K, nlp.n, nlp.m
        # column-major device images (a row-major upload of the transpose IS the column-major matrix)
        self.P_cm = _up(np.asarray(nlp.P).T, dev)
        self.A_cm = _up(np.asarray(nlp.A).T, dev)        # (n, m) row-major == (m, n) column-major
        self.q = _up(nlp.q, dev)
        self.P0_cm = torch.empty_like(self.P_cm)
        K.vec_fill(self.P0_cm, 0.0)
        self.handle_has_H = True
        self._px = torch.empty(self.n, dtype=torch.float64, device=dev)

    def _pmul(self, x, out):
        self.K.gemv(0, self.n, self.n, 1.0, self.P_cm, self.n, x, 0.0, out)
        return out

    def obj(self, x):
        px = self._pmul(x, self._px)
        with self.K.batch():
            a = self.K.get_dot(x, px)
            b = self.K.get_dot(self.q, x)
        return 0.5 * a[0] + b[0]

    def grad(self, g, x):
        self._pmul(x, g)
        self.K.vec_axpby(g, 1.0, g, 1.0, self.q)

    def cons(self, c, x):
        if self.m > 0:
            self.K.gemv(0, self.m, self.n, 1.0, self.A_cm, self.m, x, 0.0, c)

    def jtprod_x(self, out_x, y):
        if self.m > 0:
            self.K.gemv(1, self.m, self.n, 1.0, self.A_cm, self.m, y, 0.0, out_x)
        else:
            self.K.vec_fill(out_x, 0.0)

    def load_jac(self, x=None):
        if self.m > 0:
            L.check(L.lib().mnk_dc_set_jac(self.kkt._h, self.A_cm.data_ptr(), self.m, L.MNK_DEVICE), "mnk_dc_set_jac")

    def load_hess(self, x=None, y=None, sigma=1.0):
        zero = sigma == 0.0
        src = self.P0_cm if zero else self.P_cm
        L.check(L.lib().mnk_dc_set_hess(self.kkt._h, src.data_ptr(), self.n, L.MNK_DEVICE), "mnk_dc_set_hess")
        self.handle_has_H = not zero

    def zero_hess(self):
        self.load_hess(sigma=0.0)

    def close(self):
        pass


class DeviceOPFCallbacks:
    """Callbacks of `problems.ACOPFModel` on the device (`mnk_opf_*`, csrc/opf_eval.hip): objective, gradient, constraints,
    Jacobian and Lagrangian-Hessian COO values are computed from the device iterate and handed to the KKT handle's
    compressors without leaving HBM (reference `eval_*_wrapper!`, src/IPM/callbacks.jl:1-96, with a device model)."""

    def __init__(self, nlp, kkt, dev, K):
        self.kkt, self.K, self.n, self.m = kkt, K, nlp.n, nlp.m
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)   # noqa: E731
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
        fr, to, gb, coef, bus, cost = i32(nlp.fr), i32(nlp.to), i32(nlp.gen_bus), f64(nlp.arc_coef), f64(nlp.bus_data), f64(nlp.gen_cost)
        self._h = C.c_void_p()
        L.check(L.lib().mnk_opf_create(K.ctx.handle, nlp.nbus, nlp.ngen, nlp.nbr, fr.ctypes.data, to.ctypes.data, gb.ctypes.data,
                                       coef.ctypes.data, bus.ctypes.data, cost.ctypes.data, C.byref(self._h)), "mnk_opf_create")
        sz = [C.c_int64() for _ in range(4)]
        L.check(L.lib().mnk_opf_sizes(self._h, *[C.byref(v) for v in sz]), "mnk_opf_sizes")
        assert (sz[0].value, sz[1].value, sz[2].value, sz[3].value) == (nlp.n, nlp.m, len(nlp.jac_I), len(nlp.hess_I))
        E = lambda k: torch.empty(k, dtype=torch.float64, device=dev)  # noqa: E731
        self.jv, self.hv, self.hv0, self.terms = E(len(nlp.jac_I)), E(len(nlp.hess_I)), E(len(nlp.hess_I)), E(nlp.ngen)
        K.vec_fill(self.hv0, 0.0)

    def obj(self, x):
        L.check(L.lib().mnk_opf_obj_terms(self._h, x.data_ptr(), self.terms.data_ptr()), "mnk_opf_obj_terms")
        return self.K.get_sum(self.terms)

    def grad(self, g, x):
        L.check(L.lib().mnk_opf_grad(self._h, x.data_ptr(), g.data_ptr()), "mnk_opf_grad")

    def cons(self, c, x):
        L.check(L.lib().mnk_opf_cons(self._h, x.data_ptr(), c.data_ptr()), "mnk_opf_cons")

    def jac_coord(self, x):
        L.check(L.lib().mnk_opf_jac_coord(self._h, x.data_ptr(), self.jv.data_ptr()), "mnk_opf_jac_coord")
        return self.jv

    def hess_coord(self, x, y, sigma=1.0):
        L.check(L.lib().mnk_opf_hess_coord(self._h, x.data_ptr(), y.data_ptr(), float(sigma), self.hv.data_ptr()),
                "mnk_opf_hess_coord")
        return self.hv

    def jtprod_x(self, out_x, y):
        self.kkt.spmv_device(L.MNK_SC_JT, 0, 1.0, y, 0.0, out_x)

    def load_jac(self, x):
        self.kkt.compress_jacobian(self.jac_coord(x))

    def load_hess(self, x, y, sigma=1.0):
        self.kkt.compress_hessian(self.hess_coord(x, y, sigma))

    def zero_hess(self):
        self.kkt.compress_hessian(self.hv0)

    def close(self):
        if self._h:
            L.lib().mnk_opf_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceMadNLPSolver(MadNLPSolver):
    def __init__(self, nlp, kkt_factory, opt: IPMOptions | None = None, device="cuda", sparse=True):
        super().__init__(nlp, kkt_factory, opt, sparse=sparse)
        assert not sparse or self.ns == self.m, "device driver: all-inequality (RelaxEquality) sparse condensed systems"
        self.dev = torch.device(device)
        self._on_device = False

    # ------------------------------------------------------------------ host initialization, then upload
    def _upload(self):
        ctx = self.kkt.linear_solver.ctx
        for name in ("x", "xl", "xu", "zl", "zu", "f", "y", "c", "rhs", "jacl", "x_trial", "c_trial"):
            setattr(self, name, _up(getattr(self, name), self.dev))
        nt, m, nlb, nub = self.n + self.ns, self.m, len(self.ind_lb), len(self.ind_ub)
        self.nt = nt
        self._lw = nt + m + nlb + nub
        self.K = IPMDeviceKernels(nt, self.ind_lb, self.ind_ub, ctx=ctx)
        self.K.set_perturbation_sets(self.ind_llb, self.ind_uub)
        self.dv, self.pv, self.w1v, self.w4v = (self._new_vec(self._lw) for _ in range(4))
        if hasattr(self.nlp, "arc_coef"):
            assert self.sparse, "the AC-OPF callbacks feed the sparse condensed handle"
            cls = DeviceOPFCallbacks
        else:
            cls = DeviceQPCallbacks if self.sparse else DeviceDenseQPCallbacks
        self.cb = cls(self.nlp, self.kkt, self.dev, self.K)
        self.ind_ineq_t = _up(self.ind_ineq, self.dev, np.int64)
        self.kkt.device_kkt_ops = True
        if not self.sparse:          # the dense handle reads Hessian / Jacobian from its own device copies
            self.cb.load_jac()
            self.cb.load_hess()
        self._on_device = True
        self._sync = ctx.synchronize

    # slices of a KKT vector (reference src/KKT/rhs.jl:90-150)
    def _primal(self, v): return v[:self.nt]
    def _dual(self, v): return v[self.nt:self.nt + self.m]
    def _dual_lb(self, v): return v[self.nt + self.m:self.nt + self.m + len(self.ind_lb)]
    def _dual_ub(self, v): return v[self.nt + self.m + len(self.ind_lb):]

    # ------------------------------------------------------------------ callbacks
    def eval_f(self, x):
        if not self._on_device:
            return super().eval_f(x)
        return self.cb.obj(x[:self.n])

    def eval_grad(self, x):
        if not self._on_device:
            return super().eval_grad(x)
        self.cb.grad(self.f[:self.n], x[:self.n])
        self.K.vec_fill(self.f[self.n:], 0.0)

    def eval_cons(self, c, x):
        if not self._on_device:
            return super().eval_cons(c, x)
        self.cb.cons(c, x[:self.n])
        if self.ns == self.m:
            self.K.vec_axpby(c, 1.0, c, -1.0, x[self.n:])      # ind_ineq = all constraints, in order
        else:
            self.K.vec_scatter_axpy(c, self.ind_ineq_t, -1.0, x[self.n:])
        self.K.vec_axpby(c, 1.0, c, -1.0, self.rhs)

    def eval_jac(self, x):
        if not self._on_device:
            return super().eval_jac(x)
        self.cb.load_jac(x[:self.n])

    def eval_lag_hess(self, x, y, is_resto=False):
        if not self._on_device:
            return super().eval_lag_hess(x, y, is_resto)
        self.cb.load_hess(x[:self.n], y, 0.0 if is_resto else 1.0)   # objective weight 0 in robust!

    def jtprod(self, out, y):
        """`jtprod!` reference src/KKT/Sparse/condensed.jl:150-156."""
        self.cb.jtprod_x(out[:self.n], y)
        if self.ns == self.m:
            self.K.vec_axpby(out[self.n:], -1.0, y)
        else:
            self.K.vec_gather(out[self.n:], -1.0, y, self.ind_ineq_t)

    # ------------------------------------------------------------------ factorization glue
    def factorize_wrapper(self):
        if not self._on_device:
            return super().factorize_wrapper()
        self.kkt.build_kkt_device()
        self.kkt.linear_solver.factorize_async()
        self.cnt.factorization_cnt += 1

    def solve_refine(self, x, b, w):
        """Richardson refinement (reference src/LinearSolvers/backsolve.jl:27-76) on device vectors."""
        from .linear_solver import SolveException
        try:
            ok = self._solve_refine(x, b, w)
            self.kkt.linear_solver.check_solve()   # the stream is idle here (the norms synchronized)
        except SolveException:
            # a one-launch solve gave up (its workgroups were not co-resident): the solver has switched to the stepwise
            # solve; the refinement is repeated from the right-hand side
            ok = self._solve_refine(x, b, w)
            self.kkt.linear_solver.check_solve()
        return ok

    def _solve_refine(self, x, b, w):
        # One host synchronization per Richardson step: solve -> x += w -> w = b - K x -> the norms are enqueued back to back
        # and read together; the norm of the right-hand side rides along with the first step's norms (round 4 fetched it on
        # its own in front of the loop: one more stream synchronization per solve_refine, ~25 per AC-OPF run).  A zero
        # right-hand side is then solved once instead of not at all: x = K^-1 0 = 0 exactly, reported as the reference
        # reports it (no step, ratio 0).
        it = self.iterator
        residual_ratio = 0.0
        K = self.K
        K.vec_fill(x, 0.0)
        it.ir = 0
        K.vec_copy(w, b)
        norm_b = None
        while True:
            self.kkt.solve_kkt_device(w)
            K.vec_axpby(x, 1.0, x, 1.0, w)
            K.vec_copy(w, b)
            self.kkt.mul_device(w, x, -1.0, 1.0)
            with K.batch():
                b_b = K.get_norms(b) if norm_b is None else None
                b_w = K.get_norms(w)
                b_x = K.get_norms(x)
            if norm_b is None:
                norm_b = b_b[0]
                if norm_b == 0:
                    K.vec_fill(x, 0.0)
                    break
            norm_w, norm_x = b_w[0], b_x[0]
            residual_ratio = norm_w / (min(norm_x, 1e6 * norm_b) + norm_b)
            it.ir += 1
            if it.ir >= it.richardson_max_iter or residual_ratio < it.richardson_tol:
                break
        it.residual_ratio = residual_ratio
        return residual_ratio < it.richardson_acceptable_tol

    def solve_refine_wrapper(self, d, p, w):
        if not self._on_device:
            return super().solve_refine_wrapper(d, p, w)
        ok = self.solve_refine(d, p, w)
        self.cnt.backsolve_cnt += self.iterator.ir
        return ok

    # ------------------------------------------------------------------ kernels
    def set_aug_diagonal(self):
        o = self.opt
        self.kkt.set_aug_diagonal_device(self.x, self.xl, self.xu, self.zl, self.zu, o.default_primal_regularization,
                                         o.default_dual_regularization)

    def set_aug_rhs(self, c):
        p = self.pv
        self.K.set_aug_rhs(self.f, self.zl, self.zu, self.jacl, c, self.x, self.xl, self.xu, self.mu, self._primal(p),
                           self._dual(p), self._dual_lb(p), self._dual_ub(p))
        self.K.dual_inf_perturbation(self._primal(p), self.mu, self.opt.kappa_d)

    def inf_compl(self, mu, sc):
        return self.K.get_inf_compl(self.x, self.xl, self.xu, self.zl, self.zu, mu, sc)

    def varphi(self, obj, x):
        return self.K.get_varphi(obj, x, self.xl, self.xu, self.mu)

    def alpha_max(self, dx):
        return self.K.get_alpha_max(self.x, self.xl, self.xu, dx, self.tau)

    on_trial = None   # diagnostics: callable(solver, n_trial, inertia, inertia_correct, accepted) after every trial of inertia_correction

    # SPECULATIVE first correction (round 6).  On the AC-OPF run of the bench line 17 of 20 iterations take exactly two trials: the
    # unperturbed matrix is rejected, the first perturbation is accepted -- and that perturbation is known BEFORE the verdict on the
    # unperturbed matrix: after an iteration that needed a correction it is max(min_hessian_perturbation, perturb_dec_fact *
    # del_w_last) (reference src/IPM/solver.jl:633-636), and for a system whose should_regularize_dual is `true` whatever the
    # inertia (SparseCondensedKKTSystem: src/KKT/Sparse/condensed.jl:141) del_c depends on mu alone.  Both trials are then
    # assembled back to back and factorized as ONE batch of two (a second solver on the same KKT handle, one merged launch: the two
    # fill each other's chain-bound ends, and trial 0 stops at its first non-positive pivot).  The speculative factor is used only
    # if trial 0 is rejected, so the sequence of accepted perturbations -- and every bit of the accepted factors -- is the one
    # the sequential loop produces; if trial 0 is accepted after all, reg / pr_diag / du_diag return to the bits they had
    # (`save_diagonals_device`) and the matrix is assembled again.
    #
    # MEASURED, AND OFF BY DEFAULT (profiles/r06_speculative_correction.txt, tools/spec_pair_time.py): on the case1354pegase-shaped
    # system the rejected trial stops at pivot ~2700-3100 of 11 192 -- but a right-looking elimination has done 1 - (1 - 0.25)^3 =
    # 56 % of its flops by then: 4.4 ms alone against 9.2 for the accepted trial, and the merged pair is bulk-bound at the SUM of the
    # two (13.3 ms against 13.7 one after the other: what it saves is one host round trip).  Every wasted speculation costs a whole
    # factorization (+9 ms), and the AC-OPF run ends at 59.7 instead of 65 iterations/s.  The path stays for systems whose
    # rejections come early in the pivot order; `speculate = True` on the solver object arms it.
    speculate = False     # (only ever taken for KKT systems that offer `ensure_spare_solver`)
    speculative_factorizations = 0
    speculative_wasted = 0

    # LEADING-BLOCK PROBE (round 6): lives in the library (`mnk_ls_factorize_sc_async`, option "probe", on by default while early
    # rejection is armed) so that every caller -- this driver, the reference's own inertia_correction! through the Julia glue -- gets
    # it: after a rejection that stopped in the first half of the columns, a matrix that follows an accepted one is first probed
    # through its leading principal block (13 of the AC-OPF run's 17 rejections stop in the same 64 columns; the block of order
    # 3328 is 2.6 % of a factorization's flops where the full elimination has done 63 % by that column).  `probe = False` on the
    # solver object switches it off for A/B runs; `probe_hits` / `probe_misses` read the library's counters.
    probe = True

    @property
    def probe_hits(self):
        return int(self.kkt.linear_solver.get_stat("probe_hits")) if hasattr(self.kkt.linear_solver, "get_stat") else 0

    @property
    def probe_misses(self):
        return int(self.kkt.linear_solver.get_stat("probe_misses")) if hasattr(self.kkt.linear_solver, "get_stat") else 0

    def _can_speculate(self):
        k = self.kkt
        return (self._on_device and self.speculate and self.del_w_last != 0 and hasattr(k, "ensure_spare_solver")
                and k.should_regularize_dual(0, 0, 0) and k.should_regularize_dual(0, 1, 0))

    def inertia_correction(self):
        from .linear_solver import factorize_batch
        o, k = self.opt, self.kkt
        n_trial = 0
        dw_prev = dc_prev = 0.0
        self.del_w = self.del_c = 0.0
        if self._on_device and getattr(self, "_probe_applied", None) != self.probe and hasattr(k.linear_solver, "set_option"):
            k.linear_solver.set_option("probe", 1 if self.probe else 0)
            self._probe_applied = self.probe
        spec = self._can_speculate()
        if spec:
            dw1 = max(o.min_hessian_perturbation, o.perturb_dec_fact * self.del_w_last)
            dc1 = o.jacobian_regularization_value * self.mu ** o.jacobian_regularization_exponent
            spare = k.ensure_spare_solver()
            k.save_diagonals_device()
            with factorize_batch():
                k.build_kkt_device()
                k.linear_solver.factorize_async()       # trial 0: the matrix as it is
                k.regularize_diagonal_device(dw1, dc1)  # trial 1: exactly the sequential loop's first correction (dw1 - 0, dc1 - 0)
                k.build_kkt_device()
                spare.factorize_async()
            self.cnt.factorization_cnt += 1
            self.speculative_factorizations += 1
        else:
            self.factorize_wrapper()
        inertia = k.linear_solver.inertia()
        correct = k.is_inertia_correct(*inertia)
        if spec and correct:
            # trial 0 stands: the handle goes back to the unperturbed system (diagonals bit for bit, matrix and condensation
            # buffers rebuilt from them)
            k.restore_diagonals_device()
            k.build_kkt_device()
            self.speculative_wasted += 1
            spec = False
        ok = self.solve_refine_wrapper(self.dv, self.pv, self.w4v) if correct else False
        if self.on_trial is not None:
            self.on_trial(self, n_trial, inertia, correct, ok)
        if spec:
            # trial 0 rejected: the speculative factorization IS the sequential loop's next trial (the handle's diagonals and
            # matrix are already that trial's)
            self.del_w, self.del_c = dw1, dc1
            dw_prev, dc_prev = dw1, dc1
            k.swap_solvers()
            self.cnt.factorization_cnt += 1
            inertia = k.linear_solver.inertia()
            correct = k.is_inertia_correct(*inertia)
            ok = self.solve_refine_wrapper(self.dv, self.pv, self.w4v) if correct else False
            n_trial = 1
            if self.on_trial is not None:
                self.on_trial(self, n_trial, inertia, correct, ok)
        while not ok:
            if n_trial == 0:
                self.del_w = (o.first_hessian_perturbation if self.del_w_last == 0 else
                              max(o.min_hessian_perturbation, o.perturb_dec_fact * self.del_w_last))
            else:
                self.del_w *= o.perturb_inc_fact_first if self.del_w_last == 0 else o.perturb_inc_fact
                if self.del_w > o.max_hessian_perturbation:
                    self.cnt.k += 1
                    return False
            self.del_c = (o.jacobian_regularization_value * self.mu ** o.jacobian_regularization_exponent
                          if k.should_regularize_dual(*inertia) else 0.0)
            k.regularize_diagonal_device(self.del_w - dw_prev, self.del_c - dc_prev)
            dw_prev, dc_prev = self.del_w, self.del_c
            self.factorize_wrapper()
            inertia = k.linear_solver.inertia()
            correct = k.is_inertia_correct(*inertia)
            ok = self.solve_refine_wrapper(self.dv, self.pv, self.w4v) if correct else False
            n_trial += 1
            if self.on_trial is not None:
                self.on_trial(self, n_trial, inertia, correct, ok)
        if self.del_w != 0:
            self.del_w_last = self.del_w
        return True

    # ------------------------------------------------------------------ filter line search
    def filter_line_search(self):
        o, K = self.opt, self.K
        dx = self._primal(self.dv)
        with K.batch():     # six reductions, one synchronization
            b_n = K.get_norms(self.c)
            b_v = K.get_varphi(self.obj_val, self.x, self.xl, self.xu, self.mu)
            b_d = K.get_varphi_d(self.f, self.x, self.xl, self.xu, dx, self.mu)
            b_a = K.get_alpha_max(self.x, self.xl, self.xu, dx, self.tau)
            b_z = K.get_alpha_z(self.zl, self.zu, self._dual_lb(self.dv), self._dual_ub(self.dv), self.tau)
            b_r = K.get_rel_search_norm(self.x, dx)
        theta, varphi, varphi_d, alpha_max, self.alpha_z, rel_norm = b_n[1], b_v[0], b_d[0], b_a[0], b_z[0], b_r[0]
        if varphi_d < 0:
            if theta <= self.theta_min:
                alpha_min = o.alpha_min_frac * min(o.gamma_theta, o.gamma_phi * theta / (-varphi_d),
                                                   o.delta * _pow(theta, o.s_theta) / _pow(-varphi_d, o.s_phi))
            else:
                alpha_min = o.alpha_min_frac * min(o.gamma_theta, -o.gamma_phi * theta / varphi_d)
        else:
            alpha_min = o.alpha_min_frac * o.gamma_theta
        self.cnt.l = 1
        self.alpha = alpha_max
        small = rel_norm < 10 * EPS
        switching = varphi_d < 0 and self.alpha * _pow(-varphi_d, o.s_phi) > o.delta * 2.0 ** o.s_theta
        armijo = False
        unsuccessful = False
        theta_trial = varphi_trial = 0.0
        norm_dx = None
        while True:
            K.vec_axpby(self.x_trial, 1.0, self.x, self.alpha, dx)
            self.obj_val_trial = self.eval_f(self.x_trial)
            self.eval_cons(self.c_trial, self.x_trial)
            with K.batch():
                b_n = K.get_norms(self.c_trial)
                b_v = K.get_varphi(self.obj_val_trial, self.x_trial, self.xl, self.xu, self.mu)
            theta_trial, varphi_trial = b_n[1], b_v[0]
            armijo = varphi_trial <= varphi + o.eta_phi * self.alpha * varphi_d
            if small:
                break
            ftype = self._ftype(theta, theta_trial, varphi, varphi_trial, switching, armijo)
            if ftype in ("f", "h"):
                break
            if self.cnt.l == 1 and theta_trial >= theta:
                if self._second_order_correction(alpha_max, theta, varphi, theta_trial, varphi_d, switching):
                    theta_trial = K.get_norms(self.c_trial)[1]
                    varphi_trial = self.varphi(self.obj_val_trial, self.x_trial)
                    break
            unsuccessful = True
            self.alpha /= 2
            self.cnt.l += 1
            if self.alpha < alpha_min:
                self.cnt.k += 1
                return "RESTORE"
            if norm_dx is None:
                norm_dx = K.get_norm2(dx)
            if self.alpha * norm_dx < EPS * 10:
                return "SEARCH_DIRECTION_BECOMES_TOO_SMALL"
        if unsuccessful:
            self.cnt.unsuccessful_iterate += 1
            if self.cnt.unsuccessful_iterate >= 4:
                if self.theta_max / 10 > theta_trial:
                    self.theta_max /= 10
                    self.filter = [(self.theta_max, -INF)]
                self.cnt.unsuccessful_iterate = 0
        else:
            self.cnt.unsuccessful_iterate = 0
        if not switching or not armijo:
            self.filter.append(((1 - o.gamma_theta) * theta_trial, varphi_trial - o.gamma_theta * theta_trial))
        return "LINESEARCH_SUCCEEDED"

    def _second_order_correction(self, alpha_max, theta, varphi, theta_trial, varphi_d, switching):
        o, K = self.opt, self.K
        w1 = self.w1v
        wy = self._dual(w1)
        K.vec_axpby(wy, 1.0, self.c_trial, alpha_max, self.c)
        theta_soc_old = theta_trial
        for _ in range(o.max_soc):
            self.set_aug_rhs(wy)
            self.solve_refine_wrapper(w1, self.pv, self.w4v)
            wx = self._primal(w1)
            alpha_soc = self.alpha_max(wx)
            K.vec_axpby(self.x_trial, 1.0, self.x, alpha_soc, wx)
            self.eval_cons(self.c_trial, self.x_trial)
            self.obj_val_trial = self.eval_f(self.x_trial)
            theta_soc = K.get_norms(self.c_trial)[1]
            varphi_soc = self.varphi(self.obj_val_trial, self.x_trial)
            if not self._filter_ok(theta_soc, varphi_soc):
                break
            if theta <= self.theta_min and switching:
                if varphi_soc <= varphi + o.eta_phi * self.alpha * varphi_d:
                    self.alpha = alpha_soc
                    return True
            else:
                suff = (self.m > 0 and theta_soc <= (1 - o.gamma_theta) * theta + 10 * EPS * abs(theta)) or \
                       (varphi_soc <= varphi - o.gamma_phi * theta + 10 * EPS * abs(varphi))
                if suff:
                    self.alpha = alpha_soc
                    return True
            if theta_soc > o.kappa_soc * theta_soc_old:
                break
            theta_soc_old = theta_soc
        return False

    # ------------------------------------------------------------------ vector primitives (see `ipm.MadNLPSolver`): the regular
    # phase, restore! and robust! of the base class run unchanged on device tensors through these
    def _dx(self): return self._primal(self.dv)
    def _dy(self): return self._dual(self.dv)
    def _dzl(self): return self._dual_lb(self.dv)
    def _dzu(self): return self._dual_ub(self.dv)

    def _new_vec(self, n):
        v = torch.empty(n, dtype=torch.float64, device=self.dev)
        self.K.vec_fill(v, 0.0)
        return v

    def _clone(self, v):
        w = torch.empty_like(v)
        self.K.vec_copy(w, v)
        return w

    def _vcopy(self, dst, src): self.K.vec_copy(dst, src)

    def _vaxpy(self, y, a, x): self.K.vec_axpby(y, 1.0, y, a, x)

    def _vfill(self, v, value): self.K.vec_fill(v, value)

    def _theta(self, c): return self.K.get_norms(c)[1]

    def _norm_inf(self, v): return self.K.get_norms(v)[0]

    def _sd_sc(self): return self.K.get_sd_sc(self.y, self.zl, self.zu, self.opt.s_max)

    def _inf_du(self, sd): return self.K.get_inf_du(self.f, self.zl, self.zu, self.jacl, sd)

    def _iteration_scalars(self):
        """the five reductions of the iteration header in ONE synchronization (`mnk_ipm_batch_*`)."""
        K = self.K
        with K.batch():
            b_s = K.get_sd_sc(self.y, self.zl, self.zu, self.opt.s_max)
            b_n = K.get_norms(self.c)
            b_d = K.get_inf_du(self.f, self.zl, self.zu, self.jacl, 1.0)
            b_c = K.get_inf_compl(self.x, self.xl, self.xu, self.zl, self.zu, 0.0, 1.0)
        sd, sc = b_s[0], b_s[1]
        return sd, sc, b_n[0], b_d[0] / sd, b_c[0] / sc

    def _jtprod(self): self.jtprod(self.jacl, self.y)

    def _alpha_z(self, tau): return self.K.get_alpha_z(self.zl, self.zu, self._dzl(), self._dzu(), tau)

    def _bound_dual_axpy(self, a): self.K.bound_dual_axpy(self.zl, self.zu, a, self._dzl(), self._dzu())

    def _bound_dual_fill(self, v): self.K.bound_dual_fill(self.zl, self.zu, v)

    def _adjust_boundary(self): self.K.adjust_boundary(self.x, self.xl, self.xu, self.mu)

    def _reset_bound_dual(self, mu):
        self.K.reset_bound_dual(self.zl, self.zu, self.x, self.xl, self.xu, mu, self.opt.kappa_sigma)

    def _get_F(self):
        return self.K.get_F(self.c, self.f, self.zl, self.zu, self.jacl, self.x, self.xl, self.xu, self.mu)

    def _set_initial_rhs(self):
        p = self.pv
        self.K.set_initial_rhs(self.f, self.zl, self.zu, self._primal(p), self._dual(p), self._dual_lb(p), self._dual_ub(p))

    def _kkt_initialize(self):
        """`initialize!(kkt)` (reference src/KKT/Sparse/utils.jl:52-62) on the handle's device state: zero Hessian, and
        reg = pr_diag = 1, du_diag = 0, l_diag = u_diag = 1, l_lower = u_lower = 0 through the feeder (x = 0, xl = 1,
        xu = -1, zl = zu = 0, primal_reg = 1)."""
        self.kkt.initialize()
        self.cb.zero_hess()
        nt = self.nt
        z = np.zeros(nt)
        self.kkt.set_aug_diagonal_device(z, np.ones(nt), -np.ones(nt), z, z, 1.0, 0.0)

    def _solve_newton(self):
        return self.solve_refine_wrapper(self.dv, self.pv, self.w4v)

    def _rel_search_norm(self): return self.K.get_rel_search_norm(self.x, self._dx())

    # robust restorer (`mnk_ipm_*_R`)
    def _rr_init_vectors(self, RR, mu_R, rho):
        self.K.initialize_robust_restorer(self.x, self.c, mu_R, rho, RR.x_ref, RR.D_R, RR.nn, RR.pp, RR.zp, RR.zn, self.zl, self.zu)
        self.K.vec_fill(RR.f_R, 0.0)
        self.K.vec_fill(self.y, 0.0)

    def _rr_obj_val(self, pp, nn, x):
        RR = self.RR
        return self.K.get_obj_val_R(pp, nn, RR.D_R, x, RR.x_ref, self.opt.rho, RR.zeta)

    def _rr_theta(self, c, pp, nn): return self.K.get_theta_R(c, pp, nn)

    def _rr_inf_pr(self): return self.K.get_inf_pr_R(self.c, self.RR.pp, self.RR.nn)

    def _rr_inf_du(self, sd):
        RR = self.RR
        return self.K.get_inf_du_R(RR.f_R, self.y, self.zl, self.zu, self.jacl, RR.zp, RR.zn, self.opt.rho, sd)

    def _rr_inf_compl(self, mu, sc):
        RR = self.RR
        return self.K.get_inf_compl_R(self.x, self.xl, self.xu, self.zl, self.zu, RR.pp, RR.zp, RR.nn, RR.zn, mu, sc)

    def _rr_varphi(self, obj, x, pp, nn):
        return self.K.get_varphi_R(obj, x, self.xl, self.xu, pp, nn, self.RR.mu_R)

    def _rr_varphi_d(self):
        RR = self.RR
        return self.K.get_varphi_d_R(RR.f_R, self.x, self.xl, self.xu, self._dx(), RR.pp, RR.nn, RR.dpp, RR.dnn, RR.mu_R,
                                     self.opt.rho)

    def _rr_alpha_max(self):
        RR = self.RR
        return self.K.get_alpha_max_R(self.x, self.xl, self.xu, self._dx(), RR.pp, RR.dpp, RR.nn, RR.dnn, RR.tau_R)

    def _rr_alpha_z(self):
        RR = self.RR
        return self.K.get_alpha_z_R(self.zl, self.zu, self._dzl(), self._dzu(), RR.zp, RR.dzp, RR.zn, RR.dzn, RR.tau_R)

    def _rr_set_aug(self):
        o, RR = self.opt, self.RR
        self.kkt.set_aug_RR_device(self.x, self.xl, self.xu, self.zl, self.zu, RR.D_R, RR.pp, RR.zp, RR.nn, RR.zn, RR.zeta,
                                   o.default_primal_regularization, o.default_dual_regularization)

    def _rr_set_rhs(self):
        p, RR = self.pv, self.RR
        self.K.set_aug_rhs_RR(RR.f_R, self.zl, self.zu, self.jacl, self.c, self.y, RR.pp, RR.nn, RR.zp, RR.zn, self.x, self.xl,
                              self.xu, RR.mu_R, self.opt.rho, self._primal(p), self._dual(p), self._dual_lb(p), self._dual_ub(p))

    def _rr_finish(self):
        RR = self.RR
        self.K.finish_aug_solve_RR(RR.dpp, RR.dnn, RR.dzp, RR.dzn, self.y, self._dy(), RR.pp, RR.nn, RR.zp, RR.zn, RR.mu_R,
                                   self.opt.rho)

    def _rr_set_f(self):
        RR = self.RR
        self.K.set_f_RR(RR.f_R, RR.D_R, self.x, RR.x_ref, RR.zeta)

    def _rr_reset_slack_duals(self):
        RR, ks = self.RR, self.opt.kappa_sigma
        self.K.reset_bound_dual_1(RR.zp, RR.pp, RR.mu_R, ks)
        self.K.reset_bound_dual_1(RR.zn, RR.nn, RR.mu_R, ks)

    # ------------------------------------------------------------------ solve!: host initialization, upload, then the base
    # class's phases (regular! / restore! / robust!) on the device tensors
    def solve(self):
        if self.status == "INITIAL":
            self.initialize()          # host (numpy), once
            self._upload()
        return super().solve()

    def host_state(self):
        """x, y, zl, zu on the host (tests, reporting)."""
        self._sync()      # the library's stream may not be torch's: the download below must see the finished loop
        return tuple(v.cpu().numpy() for v in (self.x, self.y, self.zl, self.zu))
