"""Minimal interior-point driver that replays MadNLP's call sequence around the KKT hot path.

The IPM loop itself is NOT part of the hot path (host code stays the reference's Julia); this
driver exists because Julia cannot run in the build image and the parity statement of the north
star is about *primal/dual residuals per iteration*.  It follows the regular phase of the
reference closely enough that iteration counts and residual histories are meaningful
(SURVEY.md Appendix A):

  initialize!            src/IPM/solver.jl:14-77, src/Callbacks/nlpmodels.jl:593-636
  regular!               src/IPM/solver.jl:216-298
  update_barrier!        src/IPM/barrier.jl:12-34 (monotone), src/IPM/kernels.jl:697-713
  set_aug_diagonal!/rhs  src/IPM/kernels.jl:4-27,113-130,818-823
  inertia_correction!    src/IPM/solver.jl:611-670 (InertiaBased)
  solve_refine_wrapper!  src/IPM/factorization.jl:1-19 + backsolve.jl
  filter_line_search!    src/IPM/line_search.jl:6-123 (+ second-order correction solver.jl:547-608)
  restore!               src/IPM/solver.jl:300-411 (soft restoration, get_F kernels.jl:572-610)
  robust!                src/IPM/solver.jl:413-545, src/IPM/restoration.jl:39-76, filter_line_search_RR!
                         src/IPM/line_search.jl:128-222, _update_monotone_RR! src/IPM/barrier.jl:39-84

Not implemented: NLP
scaling (problems used here have gradients below nlp_scaling_max_gradient, so the reference's
scaling factors are 1), inertia-free regularization, quasi-Newton.

It is backend agnostic: `kkt_factory(info)` builds any object with the KKT interface -- the HIP
mirror (`madnlp_jl_amd.kkt`) or, in the tests, the CPU oracle.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

from .backsolve import RichardsonIterator
from .kkt import UnreducedKKTVector

INF = float("inf")
EPS = np.finfo(np.float64).eps


def _pow(a, b):
    """a^b with IEEE overflow to Inf (Python's float power raises OverflowError instead)."""
    try:
        return a ** b
    except OverflowError:
        return INF


@dataclass
class IPMOptions:
    """Defaults of reference src/IPM/options.jl:117-204 and src/IPM/types.jl:66-73."""
    tol: float = 1e-8
    acceptable_tol: float = 1e-6
    acceptable_iter: int = 15
    diverging_iterates_tol: float = 1e20
    max_iter: int = 3000
    s_max: float = 100.0
    kappa_d: float = 1e-5
    bound_relax_factor: float = 1e-8
    default_primal_regularization: float = 0.0
    default_dual_regularization: float = 0.0
    constr_mult_init_max: float = 1e3
    bound_push: float = 1e-2
    bound_fac: float = 1e-2
    dual_initialization: str = "least_squares"  # "zero" for the sparse condensed preset (options.jl:160)
    relax_equality: bool = False                 # RelaxEquality preset of SparseCondensedKKTSystem
    min_hessian_perturbation: float = 1e-20
    first_hessian_perturbation: float = 1e-4
    max_hessian_perturbation: float = 1e20
    perturb_inc_fact_first: float = 1e2
    perturb_inc_fact: float = 8.0
    perturb_dec_fact: float = 1 / 3
    jacobian_regularization_exponent: float = 1 / 4
    jacobian_regularization_value: float = 1e-8
    obj_max_inc: float = 5.0
    max_soc: int = 4
    alpha_min_frac: float = 0.05
    s_theta: float = 1.1
    s_phi: float = 2.3
    eta_phi: float = 1e-4
    kappa_soc: float = 0.99
    gamma_theta: float = 1e-5
    gamma_phi: float = 1e-5
    delta: float = 1.0
    kappa_sigma: float = 1e10
    barrier_tol_factor: float = 10.0
    tau_min: float = 0.99
    mu_init: float = 1e-1
    mu_linear_decrease_factor: float = 0.2
    mu_superlinear_decrease_power: float = 1.5
    rho: float = 1000.0                                  # options.jl:195
    soft_resto_pderror_reduction_factor: float = 0.9999  # options.jl:178
    required_infeasibility_reduction: float = 0.9        # options.jl:179

    @property
    def mu_min(self):
        return min(1e-4, self.tol) / (self.barrier_tol_factor + 1)


@dataclass
class Counters:
    k: int = 0
    l: int = 0
    factorization_cnt: int = 0
    backsolve_cnt: int = 0
    acceptable_cnt: int = 0
    unsuccessful_iterate: int = 0
    restoration_fail_count: int = 0
    t: int = 0


@dataclass
class RobustRestorer:
    """`RobustRestorer` reference src/IPM/types.jl / src/IPM/restoration.jl:1-37 (vectors are created by the solver: numpy
    arrays here, device tensors in the device-resident driver)."""
    obj_val_R: float = 0.0
    theta_ref: float = 0.0
    obj_val_R_trial: float = 0.0
    inf_pr_R: float = 0.0
    inf_du_R: float = 0.0
    inf_compl_R: float = 0.0
    mu_R: float = 0.0
    tau_R: float = 0.0
    zeta: float = 0.0
    filter: list = field(default_factory=list)


@dataclass
class IterRecord:
    k: int
    obj: float
    inf_pr: float
    inf_du: float
    inf_compl: float
    mu: float
    del_w: float
    alpha: float
    ls: int
    phase: str = ""   # "" regular, "r" soft restoration (restore!), "R" robust restoration (robust!)


def parse_indexes(lvar, uvar, lcon, ucon, enforce_equality):
    """`_parse_indexes` reference src/Callbacks/nlpmodels.jl:369-406 (0-based)."""
    m = len(lcon)
    if m > 0 and enforce_equality:
        is_eq = lcon == ucon
        ind_eq = np.nonzero(is_eq)[0]
        ind_ineq = np.nonzero(~is_eq)[0]
    else:
        ind_eq = np.zeros(0, dtype=np.int64)
        ind_ineq = np.arange(m)
    xl = np.concatenate((lvar, lcon[ind_ineq]))
    xu = np.concatenate((uvar, ucon[ind_ineq]))
    ind_llb = np.nonzero((lvar != -INF) & (uvar == INF))[0]
    ind_uub = np.nonzero((lvar == -INF) & (uvar != INF))[0]
    ind_lb = np.nonzero(xl != -INF)[0]
    ind_ub = np.nonzero(xu != INF)[0]
    return dict(xl=xl, xu=xu, ind_eq=ind_eq, ind_ineq=ind_ineq, ind_lb=ind_lb, ind_ub=ind_ub, ind_llb=ind_llb,
                ind_uub=ind_uub)


def _set_initial_bounds(xl, xu, tol):
    """reference src/IPM/kernels.jl:206-218."""
    if tol > 0:
        xl -= np.maximum(1.0, np.abs(xl)) * tol
        xu += np.maximum(1.0, np.abs(xu)) * tol


def _initialize_variables(x, xl, xu, bound_push, bound_fac):
    """reference src/IPM/kernels.jl:638-654."""
    for i in range(len(x)):
        l, u = xl[i], xu[i]
        if l != -INF and u != INF:
            x[i] = min(u - min(bound_push * max(1, abs(u)), bound_fac * (u - l)),
                       max(l + min(bound_push * max(1, abs(l)), bound_fac * (u - l)), x[i]))
        elif l != -INF:
            x[i] = max(l + bound_push * max(1, abs(l)), x[i])
        elif u != INF:
            x[i] = min(u - bound_push * max(1, abs(u)), x[i])


class MadNLPSolver:
    def __init__(self, nlp, kkt_factory, opt: IPMOptions | None = None, sparse=True):
        self.nlp, self.opt, self.sparse = nlp, opt or IPMOptions(), sparse
        o = self.opt
        n, m = nlp.n, nlp.m
        idx = parse_indexes(np.asarray(nlp.lvar, float), np.asarray(nlp.uvar, float), np.asarray(nlp.lcon, float),
                            np.asarray(nlp.ucon, float), enforce_equality=not o.relax_equality)
        self.idx = idx
        self.n, self.m, self.ns = n, m, len(idx["ind_ineq"])
        self.ind_ineq, self.ind_lb, self.ind_ub = idx["ind_ineq"], idx["ind_lb"], idx["ind_ub"]
        self.ind_llb, self.ind_uub = idx["ind_llb"], idx["ind_uub"]
        self.kkt = kkt_factory(dict(n=n, m=m, **idx))
        nt = n + self.ns
        self.x, self.xl, self.xu = np.zeros(nt), idx["xl"].copy(), idx["xu"].copy()
        self.zl, self.zu, self.f = np.zeros(nt), np.zeros(nt), np.zeros(nt)
        self.y, self.c, self.rhs = np.zeros(m), np.zeros(m), np.zeros(m)
        self.jacl = np.zeros(nt)
        self.x_trial, self.c_trial = np.zeros(nt), np.zeros(m)
        V = lambda: UnreducedKKTVector(nt, m, len(self.ind_lb), len(self.ind_ub), self.ind_lb, self.ind_ub)  # noqa: E731
        self.d, self.p, self._w1, self._w4 = V(), V(), V(), V()
        self.iterator = RichardsonIterator(self.kkt, tol=o.tol)
        self.cnt = Counters()
        self.filter = []
        self.history: list[IterRecord] = []
        self.del_w = self.del_c = self.del_w_last = 0.0
        self.alpha = self.alpha_z = 0.0
        self.mu = self.tau = 0.0
        self.obj_val = 0.0
        self.status = "INITIAL"

    # ------------------------------------------------------------------ callbacks (src/IPM/callbacks.jl)
    def eval_f(self, x):
        return self.nlp.obj(x[:self.n])

    def eval_grad(self, x):
        self.f[:self.n] = self.nlp.grad(x[:self.n])
        self.f[self.n:] = 0.0

    def eval_cons(self, c, x):
        c[:] = self.nlp.cons(x[:self.n])
        c[self.ind_ineq] -= x[self.n:]
        c -= self.rhs

    def eval_jac(self, x):
        if self.sparse:
            self.kkt.get_jacobian()[:] = self.nlp.jac_coord(x[:self.n])
        else:
            self.kkt.get_jacobian()[...] = self.nlp.jac_dense(x[:self.n])
        self.kkt.compress_jacobian()

    def eval_lag_hess(self, x, y, is_resto=False):
        """`eval_lag_hess_wrapper!` callbacks.jl:77-95: objective weight 0 in the robust restoration phase."""
        w = 0.0 if is_resto else 1.0
        if self.sparse:
            self.kkt.get_hessian()[:] = self.nlp.hess_coord(x[:self.n], y, w)
        else:
            self.kkt.get_hessian()[...] = self.nlp.hess_dense(x[:self.n], y, w)
        self.kkt.compress_hessian()

    # ------------------------------------------------------------------ views
    @property
    def x_lr(self): return self.x[self.ind_lb]
    @property
    def x_ur(self): return self.x[self.ind_ub]
    @property
    def xl_r(self): return self.xl[self.ind_lb]
    @property
    def xu_r(self): return self.xu[self.ind_ub]
    @property
    def zl_r(self): return self.zl[self.ind_lb]
    @property
    def zu_r(self): return self.zu[self.ind_ub]

    # ------------------------------------------------------------------ initialize! (solver.jl:14-77)
    def initialize(self):
        o, nlp, n = self.opt, self.nlp, self.n
        x0, lvar, uvar = self.x[:n], self.xl[:n], self.xu[:n]
        x0[:] = nlp.x0
        self.y[:] = nlp.y0
        lvar[:] = nlp.lvar
        uvar[:] = nlp.uvar
        lcon, ucon = np.array(nlp.lcon, float), np.array(nlp.ucon, float)
        if o.relax_equality:
            _set_initial_bounds(lcon, ucon, o.bound_relax_factor)
        _set_initial_bounds(lvar, uvar, o.bound_relax_factor)
        _initialize_variables(x0, lvar, uvar, o.bound_push, o.bound_fac)
        con = nlp.cons(x0)
        self.xl[n:] = lcon[self.ind_ineq]
        self.xu[n:] = ucon[self.ind_ineq]
        self.rhs[:] = np.where(lcon == ucon, lcon, 0.0)   # Julia: false * -Inf == -0.0 (`false` is a strong zero)
        self.x[n:] = con[self.ind_ineq]
        sl, su = self.xl[n:], self.xu[n:]
        _set_initial_bounds(sl, su, o.bound_relax_factor)
        xs = self.x[n:]
        _initialize_variables(xs, sl, su, o.bound_push, o.bound_fac)
        self.jacl[:] = 0.0
        self.zl[self.ind_lb] = 1.0
        self.zu[self.ind_ub] = 1.0
        self.kkt.initialize()
        self.eval_jac(self.x)
        self.eval_grad(self.x)
        if o.dual_initialization == "least_squares":
            self._initialize_dual_least_squares()
        else:
            self.y[:] = 0.0
        self.obj_val = self.eval_f(self.x)
        self.eval_cons(self.c, self.x)
        self.eval_lag_hess(self.x, self.y)
        theta = np.abs(self.c).sum()
        self.theta_max = 1e4 * max(1.0, theta)
        self.theta_min = 1e-4 * max(1.0, theta)
        self.mu = o.mu_init
        self.tau = max(o.tau_min, 1 - o.mu_init)
        self.filter = [(self.theta_max, -INF)]
        self.status = "REGULAR"

    def _initialize_dual_least_squares(self):
        """solver.jl:86-97 with set_initial_rhs! (kernels.jl:220-230)."""
        p = self.p
        p.values[:] = 0.0
        p.primal()[:] = -self.f + self.zl - self.zu
        self.factorize_wrapper()
        ok = self.solve_refine_wrapper(self.d, p, self._w4)
        if (not ok) or np.abs(self.d.dual()).max(initial=0.0) > self.opt.constr_mult_init_max:
            self.y[:] = 0.0
        else:
            self.y[:] = self.d.dual()

    # ------------------------------------------------------------------ factorization glue (factorization.jl:1-27)
    def factorize_wrapper(self):
        self.kkt.build_kkt()
        self.kkt.linear_solver.factorize()
        self.cnt.factorization_cnt += 1

    def solve_refine_wrapper(self, d, p, w):
        ok = self.iterator.solve_refine(d, p, w)
        if not ok and self.kkt.linear_solver.improve():
            ok = self.iterator.solve_refine(d, p, w)
        self.cnt.backsolve_cnt += self.iterator.ir
        return ok

    # ------------------------------------------------------------------ kernels (kernels.jl)
    def set_aug_diagonal(self):
        k, o = self.kkt, self.opt
        k.reg[:] = o.default_primal_regularization
        k.du_diag[:] = -o.default_dual_regularization
        k.l_diag[:] = self.xl_r - self.x_lr
        k.u_diag[:] = self.x_ur - self.xu_r
        k.l_lower[:] = self.zl_r
        k.u_lower[:] = self.zu_r
        k.pr_diag[:] = k.reg
        k.pr_diag[self.ind_lb] -= k.l_lower / k.l_diag
        k.pr_diag[self.ind_ub] -= k.u_lower / k.u_diag

    def set_aug_rhs(self, c):
        p = self.p
        p.primal()[:] = -self.f + self.zl - self.zu - self.jacl
        p.dual()[:] = -c
        p.dual_lb()[:] = (self.xl_r - self.x_lr) * self.zl_r + self.mu
        p.dual_ub()[:] = (self.xu_r - self.x_ur) * self.zu_r - self.mu
        px = p.primal()
        px[self.ind_llb] -= self.mu * self.opt.kappa_d   # dual_inf_perturbation! (kernels.jl:818-823)
        px[self.ind_uub] += self.mu * self.opt.kappa_d

    def inf_compl(self, mu, sc):
        a = np.abs((self.x_lr - self.xl_r) * self.zl_r - mu).max(initial=0.0)
        b = np.abs((self.xu_r - self.x_ur) * self.zu_r - mu).max(initial=0.0)
        return max(a, b) / sc

    def varphi(self, obj, x):
        dl = x[self.ind_lb] - self.xl_r
        du = self.xu_r - x[self.ind_ub]
        if (dl < 0).any() or (du < 0).any():
            return INF
        with np.errstate(divide="ignore"):
            return obj - self.mu * (np.log(dl).sum() + np.log(du).sum())

    def alpha_max(self, dx):
        x, xl, xu, tau = self.x, self.xl, self.xu, self.tau
        a = 1.0
        neg, pos = dx < 0, dx > 0
        with np.errstate(invalid="ignore"):
            if neg.any():
                a = min(a, ((-x[neg] + xl[neg]) * tau / dx[neg]).min())
            if pos.any():
                a = min(a, ((-x[pos] + xu[pos]) * tau / dx[pos]).min())
        return a

    # ------------------------------------------------------------------ inertia_correction! (solver.jl:611-670)
    def inertia_correction(self):
        o, k = self.opt, self.kkt
        n_trial = 0
        dw_prev = dc_prev = 0.0
        self.del_w = self.del_c = 0.0
        self.factorize_wrapper()
        inertia = k.linear_solver.inertia()
        ok = self.solve_refine_wrapper(self.d, self.p, self._w4) if k.is_inertia_correct(*inertia) else False
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
            k.regularize_diagonal(self.del_w - dw_prev, self.del_c - dc_prev)
            dw_prev, dc_prev = self.del_w, self.del_c
            self.factorize_wrapper()
            inertia = k.linear_solver.inertia()
            ok = self.solve_refine_wrapper(self.d, self.p, self._w4) if k.is_inertia_correct(*inertia) else False
            n_trial += 1
        if self.del_w != 0:
            self.del_w_last = self.del_w
        return True

    # ------------------------------------------------------------------ barrier (barrier.jl:12-34)
    def update_barrier(self, sc):
        o = self.opt
        inf_compl_mu = self.inf_compl(self.mu, sc)
        while self.mu > max(o.mu_min, o.tol / 10) and \
                max(self.inf_pr, self.inf_du, inf_compl_mu) <= o.barrier_tol_factor * self.mu:
            a = min(99.0 * o.mu_min / o.tol, 0.01)
            mu_new = max(o.mu_min, a * o.tol, min(o.mu_linear_decrease_factor * self.mu,
                                                   self.mu ** o.mu_superlinear_decrease_power))
            inf_compl_mu = self.inf_compl(self.mu, sc)
            self.tau = max(o.tau_min, 1 - self.mu)
            self.mu = mu_new
            self.filter = [(self.theta_max, -INF)]

    # ------------------------------------------------------------------ filter line search (line_search.jl:6-123)
    def _filter_ok(self, theta, varphi):
        if not (math.isfinite(theta) and math.isfinite(varphi)):
            return False
        return all(theta <= tF or varphi <= vF for tF, vF in self.filter)

    def _ftype(self, theta, theta_trial, varphi, varphi_trial, switching, armijo):
        o = self.opt
        if not self._filter_ok(theta_trial, varphi_trial):
            return " "
        if varphi_trial >= varphi and varphi_trial > varphi and \
                math.log10(varphi_trial - varphi) > o.obj_max_inc + max(1.0, math.log10(abs(varphi)) if varphi != 0 else -INF):
            return " "
        if theta <= self.theta_min and switching:
            return "f" if armijo else " "
        suff = (self.m > 0 and theta_trial <= (1 - o.gamma_theta) * theta + 10 * EPS * abs(theta)) or \
               (varphi_trial <= varphi - o.gamma_phi * theta + 10 * EPS * abs(varphi))
        return "h" if suff else " "

    def filter_line_search(self):
        o = self.opt
        dx = self.d.primal()
        theta = np.abs(self.c).sum()
        varphi = self.varphi(self.obj_val, self.x)
        with np.errstate(divide="ignore"):
            varphi_d = float(((self.f - self.mu / (self.x - self.xl) + self.mu / (self.xu - self.x)) * dx).sum())
        alpha_max = self.alpha_max(dx)
        dzl, dzu = self.d.dual_lb(), self.d.dual_ub()
        az = 1.0
        if (dzl < 0).any():
            az = min(az, (-self.zl_r[dzl < 0] * self.tau / dzl[dzl < 0]).min())
        if (dzu < 0).any():
            az = min(az, (-self.zu_r[dzu < 0] * self.tau / dzu[dzu < 0]).min())
        self.alpha_z = az
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
        small = (np.abs(dx) / (1 + np.abs(self.x))).max(initial=0.0) < 10 * EPS
        switching = varphi_d < 0 and self.alpha * _pow(-varphi_d, o.s_phi) > o.delta * 2.0 ** o.s_theta
        armijo = False
        unsuccessful = False
        theta_trial = varphi_trial = 0.0
        while True:
            self.x_trial[:] = self.x + self.alpha * dx
            self.obj_val_trial = self.eval_f(self.x_trial)
            self.eval_cons(self.c_trial, self.x_trial)
            theta_trial = np.abs(self.c_trial).sum()
            varphi_trial = self.varphi(self.obj_val_trial, self.x_trial)
            armijo = varphi_trial <= varphi + o.eta_phi * self.alpha * varphi_d
            if small:
                break
            ftype = self._ftype(theta, theta_trial, varphi, varphi_trial, switching, armijo)
            if ftype in ("f", "h"):
                break
            if self.cnt.l == 1 and theta_trial >= theta:
                if self._second_order_correction(alpha_max, theta, varphi, theta_trial, varphi_d, switching):
                    theta_trial = np.abs(self.c_trial).sum()
                    varphi_trial = self.varphi(self.obj_val_trial, self.x_trial)
                    break
            unsuccessful = True
            self.alpha /= 2
            self.cnt.l += 1
            if self.alpha < alpha_min:
                self.cnt.k += 1
                return "RESTORE"
            if self.alpha * np.linalg.norm(dx) < EPS * 10:
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
        """solver.jl:547-608: reuses the factorization, solves only."""
        o = self.opt
        w1 = self._w1
        # wy IS dual(_w1) in the reference (solver.jl:552-554): the solve below overwrites it, and the
        # next correction starts from that overwritten vector.  Mirrored on purpose.
        wy = w1.dual()
        wy[:] = self.c_trial + alpha_max * self.c
        theta_soc_old = theta_trial
        for _ in range(o.max_soc):
            self.set_aug_rhs(wy)
            self.solve_refine_wrapper(w1, self.p, self._w4)
            wx = w1.primal()
            alpha_soc = self.alpha_max(wx)
            self.x_trial[:] = self.x + alpha_soc * wx
            self.eval_cons(self.c_trial, self.x_trial)
            self.obj_val_trial = self.eval_f(self.x_trial)
            theta_soc = np.abs(self.c_trial).sum()
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

    # ------------------------------------------------------------------ vector primitives shared with the device driver
    # (the restoration phases below are written once against these; `ipm_dev.DeviceMadNLPSolver` overrides them with the
    # `mnk_ipm_*` kernels on device tensors)
    def _dx(self): return self.d.primal()
    def _dy(self): return self.d.dual()
    def _dzl(self): return self.d.dual_lb()
    def _dzu(self): return self.d.dual_ub()

    def _new_vec(self, n): return np.zeros(n)

    def _clone(self, v): return v.copy()

    def _vcopy(self, dst, src): dst[:] = src         # copyto!

    def _vaxpy(self, y, a, x): y += a * x            # axpy!

    def _vfill(self, v, value): v[:] = value         # fill!

    def _theta(self, c): return float(np.abs(c).sum())

    def _norm_inf(self, v): return float(np.abs(v).max(initial=0.0))

    def _sd_sc(self):
        o, nl, nu = self.opt, len(self.ind_lb), len(self.ind_ub)
        nz = np.abs(self.zl_r).sum() + np.abs(self.zu_r).sum()
        sd = max(o.s_max, (np.abs(self.y).sum() + nz) / max(1, self.m + nl + nu)) / o.s_max
        return sd, max(o.s_max, nz / max(1, nl + nu)) / o.s_max

    def _inf_du(self, sd): return float(np.abs(self.f - self.zl + self.zu + self.jacl).max(initial=0.0) / sd)

    def _iteration_scalars(self):
        """sd, sc, inf_pr, inf_du, inf_compl(mu = 0) at the top of an iteration (solver.jl:224-234)."""
        sd, sc = self._sd_sc()
        return sd, sc, self._norm_inf(self.c), self._inf_du(sd), self.inf_compl(0.0, sc)

    def _jtprod(self): self.kkt.jtprod(self.jacl, self.y)

    def _alpha_z(self, tau):
        """`get_alpha_z` kernels.jl:373-388."""
        az = 1.0
        for z, dz in ((self.zl_r, self._dzl()), (self.zu_r, self._dzu())):
            neg = dz < 0
            if neg.any():
                az = min(az, (-z[neg] * tau / dz[neg]).min())
        return float(az)

    def _bound_dual_axpy(self, a):
        self.zl[self.ind_lb] += a * self._dzl()
        self.zu[self.ind_ub] += a * self._dzu()

    def _bound_dual_fill(self, v):
        self.zl[self.ind_lb] = v
        self.zu[self.ind_ub] = v

    def _adjust_boundary(self):
        """`adjust_boundary!` kernels.jl:656-673."""
        c1, c2 = EPS * self.mu, EPS ** 0.75
        xl_r, x_lr = self.xl_r, self.x_lr
        adj = x_lr - xl_r < c1
        self.xl[self.ind_lb[adj]] = xl_r[adj] - c2 * np.maximum(1, np.abs(x_lr[adj]))
        xu_r, x_ur = self.xu_r, self.x_ur
        adj = xu_r - x_ur < c1
        self.xu[self.ind_ub[adj]] = xu_r[adj] + c2 * np.maximum(1, np.abs(x_ur[adj]))

    def _reset_bound_dual(self, mu):
        """the two `reset_bound_dual!` calls on the full primal-sized vectors (kernels.jl:788-800)."""
        ks = self.opt.kappa_sigma
        with np.errstate(divide="ignore", invalid="ignore"):
            dl = self.x - self.xl
            self.zl[:] = np.where(np.isfinite(dl), np.maximum(np.minimum(self.zl, ks * mu / dl), mu / ks / dl), self.zl)
            du = self.xu - self.x
            self.zu[:] = np.where(np.isfinite(du), np.maximum(np.minimum(self.zu, ks * mu / du), mu / ks / du), self.zu)

    def _get_F(self):
        """`get_F` kernels.jl:572-610; the upper-bound term as the reference writes it, `(xu_r - xu_r) * zu_r - mu`."""
        mu = self.mu
        x_lr, xl_r, zl_r, xu_r, x_ur, zu_r = self.x_lr, self.xl_r, self.zl_r, self.xu_r, self.x_ur, self.zu_r
        F3 = np.where((x_lr >= xl_r) & (zl_r >= 0), np.abs((x_lr - xl_r) * zl_r - mu), INF).sum()
        with np.errstate(invalid="ignore"):
            F4 = np.where((xu_r >= x_ur) & (zu_r >= 0), np.abs((xu_r - xu_r) * zu_r - mu), INF).sum()
        return float(np.abs(self.c).sum() + np.abs(self.f - self.zl + self.zu + self.jacl).sum() + F3 + F4)

    def _set_initial_rhs(self):
        """`set_initial_rhs!` kernels.jl:220-230."""
        self.p.values[:] = 0.0
        self.p.primal()[:] = -self.f + self.zl - self.zu

    def _kkt_initialize(self): self.kkt.initialize()

    def _record(self, phase=""):
        self.history.append(IterRecord(self.cnt.k, self.obj_val, self.inf_pr, self.inf_du, self.inf_compl_v, self.mu,
                                       self.del_w, self.alpha, self.cnt.l, phase))

    # robust restorer pieces
    def _rr_init_vectors(self, RR, mu_R, rho):
        """vector part of `initialize_robust_restorer!` restoration.jl:46-70."""
        RR.x_ref[:] = self.x
        with np.errstate(divide="ignore"):
            RR.D_R[:] = np.minimum(1.0, 1.0 / np.abs(RR.x_ref))
        t = (mu_R - rho * self.c) / (2 * rho)
        RR.nn[:] = t + np.sqrt(t * t + mu_R * self.c / (2 * rho))      # populate_RR_nn! kernels.jl:825-829
        RR.pp[:] = self.c + RR.nn
        RR.zp[:] = mu_R / RR.pp
        RR.zn[:] = mu_R / RR.nn
        RR.f_R[:] = 0.0
        self.y[:] = 0.0
        self.zl[self.ind_lb] = np.minimum(rho, self.zl_r)
        self.zu[self.ind_ub] = np.minimum(rho, self.zu_r)

    def _rr_obj_val(self, pp, nn, x):
        RR, o = self.RR, self.opt
        return float((o.rho * (pp + nn)).sum() + (RR.zeta / 2 * RR.D_R ** 2 * (x - RR.x_ref) ** 2).sum())

    def _rr_theta(self, c, pp, nn): return float(np.abs(c - pp + nn).sum())

    def _rr_inf_pr(self): return float(np.abs(self.c - self.RR.pp + self.RR.nn).max(initial=0.0))

    def _rr_inf_du(self, sd):
        RR, rho = self.RR, self.opt.rho
        return float(max(np.abs(RR.f_R - self.zl + self.zu + self.jacl).max(initial=0.0),
                         np.abs(rho - self.y - RR.zp).max(initial=0.0), np.abs(rho + self.y - RR.zn).max(initial=0.0)) / sd)

    def _rr_inf_compl(self, mu, sc):
        RR = self.RR
        return float(max(np.abs((self.x_lr - self.xl_r) * self.zl_r - mu).max(initial=0.0),
                         np.abs((self.xu_r - self.x_ur) * self.zu_r - mu).max(initial=0.0),
                         np.abs(RR.pp * RR.zp - mu).max(initial=0.0), np.abs(RR.nn * RR.zn - mu).max(initial=0.0)) / sc)

    def _rr_varphi(self, obj, x, pp, nn):
        mu = self.RR.mu_R
        dl, du = x[self.ind_lb] - self.xl_r, self.xu_r - x[self.ind_ub]
        if (dl < 0).any() or (du < 0).any() or (pp < 0).any() or (nn < 0).any():
            return -INF
        with np.errstate(divide="ignore"):
            return float(obj - mu * (np.log(dl).sum() + np.log(du).sum() + np.log(pp).sum() + np.log(nn).sum()))

    def _rr_varphi_d(self):
        RR, mu, rho, dx = self.RR, self.RR.mu_R, self.opt.rho, self._dx()
        with np.errstate(divide="ignore"):
            return float(((RR.f_R - mu / (self.x - self.xl) + mu / (self.xu - self.x)) * dx).sum()
                         + ((rho - mu / RR.pp) * RR.dpp).sum() + ((rho - mu / RR.nn) * RR.dnn).sum())

    def _rr_alpha_max(self):
        RR, tau = self.RR, self.RR.tau_R
        keep, self.tau = self.tau, tau
        a = self.alpha_max(self._dx())
        self.tau = keep
        for v, dv in ((RR.pp, RR.dpp), (RR.nn, RR.dnn)):
            neg = dv < 0
            if neg.any():
                a = min(a, (-v[neg] * tau / dv[neg]).min())
        return float(a)

    def _rr_alpha_z(self):
        RR, tau = self.RR, self.RR.tau_R
        a = self._alpha_z(tau)
        for z, dz in ((RR.zp, RR.dzp), (RR.zn, RR.dzn)):
            neg = dz < 0
            if neg.any():
                a = min(a, (-z[neg] * tau / dz[neg]).min())
        return float(a)

    def _rr_set_aug(self):
        """`set_aug_RR!` kernels.jl:72-87."""
        k, o, RR = self.kkt, self.opt, self.RR
        k.reg[:] = o.default_primal_regularization + RR.zeta * RR.D_R ** 2
        k.du_diag[:] = -o.default_dual_regularization - RR.pp / RR.zp - RR.nn / RR.zn
        k.l_lower[:] = self.zl_r
        k.u_lower[:] = self.zu_r
        k.l_diag[:] = self.xl_r - self.x_lr
        k.u_diag[:] = self.x_ur - self.xu_r
        k.pr_diag[:] = k.reg
        k.pr_diag[self.ind_lb] -= k.l_lower / k.l_diag
        k.pr_diag[self.ind_ub] -= k.u_lower / k.u_diag

    def _rr_set_rhs(self):
        """`set_aug_rhs_RR!` kernels.jl:133-158."""
        p, RR, mu, rho, y = self.p, self.RR, self.RR.mu_R, self.opt.rho, self.y
        p.primal()[:] = -RR.f_R + self.zl - self.zu - self.jacl
        p.dual()[:] = -self.c + RR.pp - RR.nn + (mu - (rho - y) * RR.pp) / RR.zp - (mu - (rho + y) * RR.nn) / RR.zn
        p.dual_lb()[:] = (self.xl_r - self.x_lr) * self.zl_r + mu
        p.dual_ub()[:] = (self.xu_r - self.x_ur) * self.zu_r - mu

    def _rr_finish(self):
        """`finish_aug_solve_RR!` kernels.jl:251-257."""
        RR, mu, rho, l, dl = self.RR, self.RR.mu_R, self.opt.rho, self.y, self._dy()
        RR.dzp[:] = rho - l - dl - RR.zp
        RR.dzn[:] = rho + l + dl - RR.zn
        RR.dpp[:] = -RR.pp + mu / RR.zp - (RR.pp / RR.zp) * RR.dzp
        RR.dnn[:] = -RR.nn + mu / RR.zn - (RR.nn / RR.zn) * RR.dzn

    def _rr_set_f(self):
        RR = self.RR
        RR.f_R[:] = RR.zeta * RR.D_R ** 2 * (self.x - RR.x_ref)      # set_f_RR! kernels.jl:106-110

    def _rr_reset_slack_duals(self):
        RR, mu, ks = self.RR, self.RR.mu_R, self.opt.kappa_sigma
        for z, v in ((RR.zp, RR.pp), (RR.zn, RR.nn)):                  # reset_bound_dual!(z, x, mu, ks) kernels.jl:775-786
            with np.errstate(divide="ignore"):
                z[:] = np.maximum(np.minimum(z, (ks * mu) / v), (mu / ks) / v)

    def _rel_search_norm(self): return float((np.abs(self._dx()) / (1 + np.abs(self.x))).max(initial=0.0))

    # ------------------------------------------------------------------ restore! (solver.jl:300-411)
    def restore(self):
        o = self.opt
        self.del_w = 0.0
        w1x, w1y, w2c = self._clone(self.x), self._clone(self.y), self._clone(self.c)   # the backups in _w1 / _w2
        F = self._get_F()
        self.alpha_z = 0.0
        while True:
            alpha_max = self.alpha_max(self._dx())
            self.alpha = min(alpha_max, self._alpha_z(self.tau))
            self._vaxpy(self.x, self.alpha, self._dx())
            self._vaxpy(self.y, self.alpha, self._dy())
            self._bound_dual_axpy(self.alpha)
            self.eval_cons(self.c, self.x)
            self.eval_grad(self.x)
            self.obj_val = self.eval_f(self.x)
            self.eval_jac(self.x)
            self._jtprod()
            F_trial = self._get_F()
            if F_trial > o.soft_resto_pderror_reduction_factor * F:
                self._vcopy(self.x, w1x)
                self._vcopy(self.y, w1y)
                self._vcopy(self.c, w2c)
                return "ROBUST"
            self._adjust_boundary()
            F = F_trial
            theta = self._theta(self.c)
            varphi = self.varphi(self.obj_val, self.x)
            self.cnt.k += 1
            if self._filter_ok(theta, varphi):
                return "REGULAR"
            self.cnt.t += 1
            if self.cnt.k >= o.max_iter:
                return "MAXIMUM_ITERATIONS_EXCEEDED"
            sd, sc, self.inf_pr, self.inf_du, self.inf_compl_v = self._iteration_scalars()
            self._record("r")
            self.eval_lag_hess(self.x, self.y)
            self.set_aug_diagonal()
            self.set_aug_rhs(self.c)
            self.factorize_wrapper()
            self._solve_newton()

    def _solve_newton(self):
        return self.solve_refine_wrapper(self.d, self.p, self._w4)

    # ------------------------------------------------------------------ robust! (solver.jl:413-545)
    def _initialize_robust_restorer(self):
        """`initialize_robust_restorer!` restoration.jl:39-76."""
        o = self.opt
        if getattr(self, "RR", None) is None:
            RR = RobustRestorer()
            nt, m = len(self.x), self.m
            for name, n in (("f_R", nt), ("x_ref", nt), ("D_R", nt)) + tuple((v, m) for v in (
                    "pp", "nn", "zp", "zn", "dpp", "dnn", "dzp", "dzn", "pp_trial", "nn_trial")):
                setattr(RR, name, self._new_vec(n))
            self.RR = RR
        RR = self.RR
        RR.theta_ref = self._theta(self.c)
        RR.mu_R = max(self.mu, self._norm_inf(self.c))
        RR.tau_R = max(o.tau_min, 1 - RR.mu_R)
        RR.zeta = math.sqrt(RR.mu_R)
        self._rr_init_vectors(RR, RR.mu_R, o.rho)
        RR.obj_val_R = self._rr_obj_val(RR.pp, RR.nn, self.x)
        RR.filter = [(self.theta_max, -INF)]
        self.cnt.t = 0
        self.del_w = 0.0

    def _update_monotone_RR(self, sc):
        """`_update_monotone_RR!` barrier.jl:39-84."""
        o, RR = self.opt, self.RR
        inf_compl_mu_R = self._rr_inf_compl(RR.mu_R, sc)
        while RR.mu_R >= o.mu_min and max(RR.inf_pr_R, RR.inf_du_R, inf_compl_mu_R) <= o.barrier_tol_factor * RR.mu_R:
            a = min(99.0 * o.mu_min / o.tol, 0.01)
            RR.mu_R = max(o.mu_min, a * o.tol, min(o.mu_linear_decrease_factor * RR.mu_R,
                                                   RR.mu_R ** o.mu_superlinear_decrease_power))
            inf_compl_mu_R = self._rr_inf_compl(RR.mu_R, sc)
            RR.tau_R = max(o.tau_min, 1 - RR.mu_R)
            RR.zeta = math.sqrt(RR.mu_R)
            RR.filter = [(self.theta_max, -INF)]

    def robust(self):
        o = self.opt
        self._initialize_robust_restorer()
        RR = self.RR
        while True:
            self.eval_jac(self.x)
            self._jtprod()
            sd, sc, self.inf_pr, self.inf_du, self.inf_compl_v = self._iteration_scalars()
            RR.inf_pr_R = self._rr_inf_pr()
            RR.inf_du_R = self._rr_inf_du(sd)
            RR.inf_compl_R = self._rr_inf_compl(0.0, sc)
            self._record("R")
            if max(RR.inf_pr_R, RR.inf_du_R, RR.inf_compl_R) <= o.tol:
                return "INFEASIBLE_PROBLEM_DETECTED"
            if self.cnt.k >= o.max_iter:
                return "MAXIMUM_ITERATIONS_EXCEEDED"
            self._update_monotone_RR(sc)
            self.eval_lag_hess(self.x, self.y, is_resto=True)
            self._rr_set_aug()
            self._rr_set_rhs()
            if not self.inertia_correction():
                return "RESTORATION_FAILED"
            self._rr_finish()
            st = self.filter_line_search_RR()
            if st != "LINESEARCH_SUCCEEDED":
                return st
            self._vcopy(self.x, self.x_trial)
            self._vcopy(self.c, self.c_trial)
            self._vcopy(RR.pp, RR.pp_trial)
            self._vcopy(RR.nn, RR.nn_trial)
            RR.obj_val_R = RR.obj_val_R_trial
            self._rr_set_f()
            self._vaxpy(self.y, self.alpha, self._dy())
            self._vaxpy(RR.zp, self.alpha_z, RR.dzp)
            self._vaxpy(RR.zn, self.alpha_z, RR.dzn)
            self._bound_dual_axpy(self.alpha_z)
            self._reset_bound_dual(RR.mu_R)
            self._rr_reset_slack_duals()
            self._adjust_boundary()
            self.obj_val = self.eval_f(self.x)
            self.eval_grad(self.x)
            theta = self._theta(self.c)
            varphi = self.varphi(self.obj_val, self.x)
            if self._filter_ok(theta, varphi) and theta <= o.required_infeasibility_reduction * RR.theta_ref:
                self._set_initial_rhs()
                self._kkt_initialize()
                self.factorize_wrapper()
                self._solve_newton()
                if self._norm_inf(self._dy()) > o.constr_mult_init_max:
                    self._vfill(self.y, 0.0)
                else:
                    self._vcopy(self.y, self._dy())
                self.cnt.k += 1
                self.cnt.t += 1
                return "REGULAR"
            if self.cnt.k >= o.max_iter:
                return "MAXIMUM_ITERATIONS_EXCEEDED"
            self.cnt.k += 1
            self.cnt.t += 1

    def _ftype_RR(self, theta, theta_trial, varphi, varphi_trial, switching, armijo):
        keep, self.filter = self.filter, self.RR.filter
        try:
            return self._ftype(theta, theta_trial, varphi, varphi_trial, switching, armijo)
        finally:
            self.filter = keep

    def filter_line_search_RR(self):
        """`filter_line_search_RR!` line_search.jl:128-222."""
        o, RR = self.opt, self.RR
        theta_R = self._rr_theta(self.c, RR.pp, RR.nn)
        varphi_R = self._rr_varphi(RR.obj_val_R, self.x, RR.pp, RR.nn)
        varphi_d_R = self._rr_varphi_d()
        alpha_max = self._rr_alpha_max()
        self.alpha_z = self._rr_alpha_z()
        if varphi_d_R < 0:
            if theta_R <= self.theta_min:
                alpha_min = o.alpha_min_frac * min(o.gamma_theta, o.gamma_phi * theta_R / (-varphi_d_R),
                                                   o.delta * _pow(theta_R, o.s_theta) / _pow(-varphi_d_R, o.s_phi))
            else:
                alpha_min = o.alpha_min_frac * min(o.gamma_theta, -o.gamma_phi * theta_R / varphi_d_R)
        else:
            alpha_min = o.alpha_min_frac * o.gamma_theta
        self.alpha = alpha_max
        self.cnt.l = 1
        small = self._rel_search_norm() < 10 * EPS
        switching = varphi_d_R < 0 and self.alpha * _pow(-varphi_d_R, o.s_phi) > o.delta * _pow(theta_R, o.s_theta)
        armijo = False
        while True:
            self._vcopy(self.x_trial, self.x)
            self._vaxpy(self.x_trial, self.alpha, self._dx())
            self._vcopy(RR.pp_trial, RR.pp)
            self._vaxpy(RR.pp_trial, self.alpha, RR.dpp)
            self._vcopy(RR.nn_trial, RR.nn)
            self._vaxpy(RR.nn_trial, self.alpha, RR.dnn)
            RR.obj_val_R_trial = self._rr_obj_val(RR.pp_trial, RR.nn_trial, self.x_trial)
            self.eval_cons(self.c_trial, self.x_trial)
            theta_R_trial = self._rr_theta(self.c_trial, RR.pp_trial, RR.nn_trial)
            varphi_R_trial = self._rr_varphi(RR.obj_val_R_trial, self.x_trial, RR.pp_trial, RR.nn_trial)
            armijo = varphi_R_trial <= varphi_R + o.eta_phi * self.alpha * varphi_d_R
            if small:
                break
            if self._ftype_RR(theta_R, theta_R_trial, varphi_R, varphi_R_trial, switching, armijo) in ("f", "h"):
                break
            self.alpha /= 2
            self.cnt.l += 1
            if self.alpha < alpha_min:
                self.cnt.restoration_fail_count += 1
                if self.cnt.restoration_fail_count >= 4:
                    return "RESTORATION_FAILED"
                # the reference's "second chance": back to the regular phase from the current iterate
                self._vfill(self.y, 0.0)
                self._bound_dual_fill(1.0)
                self.filter = [(self.theta_max, -INF)]
                self.cnt.k += 1
                self.cnt.t += 1
                return "REGULAR"
            if self.alpha < EPS * 10:
                return "SOLVED_TO_ACCEPTABLE_LEVEL" if self.cnt.acceptable_cnt > 0 else "SEARCH_DIRECTION_BECOMES_TOO_SMALL"
        if not switching or not armijo:
            RR.filter.append(((1 - o.gamma_theta) * theta_R_trial, varphi_R_trial - o.gamma_theta * theta_R_trial))
        return "LINESEARCH_SUCCEEDED"

    # ------------------------------------------------------------------ solve! (solver.jl:119-166)
    def solve(self):
        if self.status == "INITIAL":
            self.initialize()
        while self.status in ("REGULAR", "RESTORE", "ROBUST"):
            if self.status == "REGULAR":
                self.status = self.regular()
            if self.status == "RESTORE":
                self.status = self.restore()
            if self.status == "ROBUST":
                self.status = self.robust()
        return self.status

    # ------------------------------------------------------------------ regular! (solver.jl:216-298)
    def regular(self):
        o = self.opt
        while True:
            if self.cnt.k != 0:
                self.eval_jac(self.x)
            self._jtprod()
            sd, sc, self.inf_pr, self.inf_du, self.inf_compl_v = self._iteration_scalars()
            self._record()
            inf_total = max(self.inf_pr, self.inf_du, self.inf_compl_v)
            if inf_total <= o.tol:
                return "SOLVE_SUCCEEDED"
            if inf_total <= o.acceptable_tol:
                if self.cnt.acceptable_cnt < o.acceptable_iter:
                    self.cnt.acceptable_cnt += 1
                else:
                    return "SOLVED_TO_ACCEPTABLE_LEVEL"
            else:
                self.cnt.acceptable_cnt = 0
            if inf_total >= o.diverging_iterates_tol:
                return "DIVERGING_ITERATES"
            if self.cnt.k >= o.max_iter:
                return "MAXIMUM_ITERATIONS_EXCEEDED"
            if self.cnt.k != 0:
                self.eval_lag_hess(self.x, self.y)
            self.update_barrier(sc)
            self.set_aug_diagonal()
            self.set_aug_rhs(self.c)
            if not self.inertia_correction():
                return "ROBUST"
            st = self.filter_line_search()
            if st != "LINESEARCH_SUCCEEDED":
                return st
            self._vcopy(self.x, self.x_trial)
            self._vcopy(self.c, self.c_trial)
            self.obj_val = self.obj_val_trial
            self._adjust_boundary()
            self._vaxpy(self.y, self.alpha, self._dy())
            self._bound_dual_axpy(self.alpha_z)
            self._reset_bound_dual(self.mu)
            self.eval_grad(self.x)
            self.cnt.k += 1
