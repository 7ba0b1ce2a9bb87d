"""Synthetic inputs of the KKT hot path (numpy only).

Real PGLib cases / ExaModels are not available offline, so the sparse systems are
*OPF-shaped*: the variable/constraint counts and sparsity pattern of the polar AC-OPF
model (va, vm per bus; pg, qg per generator; p, q per arc), random values.  The dense
system is shaped like the reference's `DenseDummyQP`
(`lib/MadNLPTests/src/Instances/dummy_qp.jl:79-151`): P = R R' + 100 I, bidiagonal +-1 A.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

OPF_CASES = {
    # name: (nbus, ngen, nbranch)  -- public PGLib case metadata
    "case1354pegase": (1354, 260, 1991),
    "case9241pegase": (9241, 1445, 16049),
    "case118": (118, 54, 186),
    "case30": (30, 6, 41),
}


@dataclass
class SparseKKTProblem:
    name: str
    n: int
    m: int
    jac_I: np.ndarray
    jac_J: np.ndarray
    hess_I: np.ndarray
    hess_J: np.ndarray
    jac: np.ndarray
    hess: np.ndarray
    ind_ineq: np.ndarray
    ind_lb: np.ndarray
    ind_ub: np.ndarray
    reg: np.ndarray
    l_diag: np.ndarray
    u_diag: np.ndarray
    l_lower: np.ndarray
    u_lower: np.ndarray
    du_diag: np.ndarray
    meta: dict = field(default_factory=dict)

    @property
    def pr_diag(self):
        """`_set_aug_diagonal!` (reference src/IPM/kernels.jl:22-27)."""
        p = self.reg.copy()
        p[self.ind_lb] -= self.l_lower / self.l_diag
        p[self.ind_ub] -= self.u_lower / self.u_diag
        return p


def _random_connected_graph(rng, nbus, nbranch, max_deg=12):
    """Spanning tree + extra edges; degrees capped (power-grid-like sparse graph)."""
    deg = np.zeros(nbus, dtype=np.int64)
    fr = np.empty(nbranch, dtype=np.int64)
    to = np.empty(nbranch, dtype=np.int64)
    for i in range(1, nbus):
        # attach to a recent bus (locality) whose degree is still small
        lo = max(0, i - 50)
        for _ in range(20):
            j = int(rng.integers(lo, i))
            if deg[j] < max_deg - 1:
                break
        fr[i - 1], to[i - 1] = j, i
        deg[i] += 1
        deg[j] += 1
    k = nbus - 1
    seen = set(zip(fr[:k].tolist(), to[:k].tolist()))
    while k < nbranch:
        a = int(rng.integers(0, nbus))
        b = a + int(rng.integers(1, 60))
        if b >= nbus or deg[a] >= max_deg or deg[b] >= max_deg or (a, b) in seen:
            if nbranch > 4 * nbus:  # dense request: relax locality
                b = int(rng.integers(0, nbus))
                if b == a or (min(a, b), max(a, b)) in seen:
                    continue
                a, b = min(a, b), max(a, b)
            else:
                continue
        seen.add((a, b))
        fr[k], to[k] = a, b
        deg[a] += 1
        deg[b] += 1
        k += 1
    return fr, to


def _opf_structure(case, seed=None):
    """Graph, variable layout and the COO patterns of the polar AC-OPF model (`opf_shaped` fills them with random values,
    `ACOPFModel` with the power-flow derivatives).  Returns a dict and the generator `rng` after the structure draws."""
    nbus, ngen, nbr = OPF_CASES[case] if isinstance(case, str) else case
    name = case if isinstance(case, str) else f"opf{nbus}"
    if seed is None:
        seed = nbus
    rng = np.random.default_rng(seed)
    fr, to = _random_connected_graph(rng, nbus, nbr)
    gen_bus = rng.integers(0, nbus, ngen)
    # variable layout
    va = np.arange(nbus)
    vm = nbus + np.arange(nbus)
    pg = 2 * nbus + np.arange(ngen)
    qg = 2 * nbus + ngen + np.arange(ngen)
    narc = 2 * nbr
    p = 2 * nbus + 2 * ngen + np.arange(narc)
    q = 2 * nbus + 2 * ngen + narc + np.arange(narc)
    n = 2 * nbus + 2 * ngen + 2 * narc
    arc_f = np.concatenate((fr, to))  # arc a: from-bus
    arc_t = np.concatenate((to, fr))

    rows, cols = [], []
    r = 0
    # reference angle
    rows.append(np.array([r])); cols.append(np.array([va[0]])); r += 1
    # flow definitions: p and q of every arc (4 rows per branch, 5 nz each)
    for own in (p, q):
        rr = r + np.arange(narc)
        rows.append(np.repeat(rr, 5))
        cols.append(np.stack((own, vm[arc_f], vm[arc_t], va[arc_f], va[arc_t]), axis=1).ravel())
        r += narc
    # angle difference limits
    rr = r + np.arange(nbr)
    rows.append(np.repeat(rr, 2)); cols.append(np.stack((va[fr], va[to]), axis=1).ravel()); r += nbr
    # thermal limits on both arc directions
    rr = r + np.arange(narc)
    rows.append(np.repeat(rr, 2)); cols.append(np.stack((p, q), axis=1).ravel()); r += narc
    # active / reactive balance per bus: vm + incident arcs + generators
    for own_arc, own_gen in ((p, pg), (q, qg)):
        rr0 = r
        rows.append(rr0 + np.arange(nbus)); cols.append(vm)
        rows.append(rr0 + arc_f); cols.append(own_arc)
        rows.append(rr0 + gen_bus); cols.append(own_gen)
        r += nbus
    m = r
    jac_I = np.concatenate(rows).astype(np.int32)
    jac_J = np.concatenate(cols).astype(np.int32)
    # Hessian of the Lagrangian: per flow row the lower triangle of the clique (vm_f, vm_t, va_f, va_t) (COO with duplicates
    # across the 4 rows of a branch, like an AD back-end emits), then thermal p^2 / q^2, shunt vm^2 and generation-cost diagonals
    il, jl = np.tril_indices(4)
    clique = np.stack((vm[arc_f], vm[arc_t], va[arc_f], va[arc_t]), axis=1)  # narc x 4
    hI = [clique[:, il].ravel(), clique[:, il].ravel(), p, q, vm, pg]
    hJ = [clique[:, jl].ravel(), clique[:, jl].ravel(), p, q, vm, pg]
    S = dict(name=name, seed=seed, nbus=nbus, ngen=ngen, nbr=nbr, narc=narc, n=n, m=m, fr=fr, to=to, gen_bus=gen_bus,
             arc_f=arc_f, arc_t=arc_t, va=va, vm=vm, pg=pg, qg=qg, p=p, q=q, jac_I=jac_I, jac_J=jac_J,
             hess_I=np.concatenate(hI).astype(np.int32), hess_J=np.concatenate(hJ).astype(np.int32), il=il, jl=jl)
    return S, rng


def opf_shaped(case="case1354pegase", seed=None, sigma_s_decades=8.0, du=0.0, indefinite=False):
    """OPF-shaped sparse condensed KKT inputs (all constraints are inequalities, as under
    MadNLP's RelaxEquality preset for SparseCondensedKKTSystem, reference
    src/IPM/options.jl:146-147)."""
    S, rng = _opf_structure(case, seed)
    name, seed, nbus, ngen, nbr, narc, n, m = (S[k] for k in ("name", "seed", "nbus", "ngen", "nbr", "narc", "n", "m"))
    pg, jac_I, jac_J, il, jl = S["pg"], S["jac_I"], S["jac_J"], S["il"], S["jl"]
    jac = np.clip(rng.standard_normal(len(jac_I)), -100, 100)

    # values: per flow row a rank-one PSD clique, positive diagonals
    hV = []
    for _own in range(2):
        a = rng.standard_normal((narc, 4))
        hV.append((a[:, il] * a[:, jl]).ravel())
    hV.append(rng.random(narc) + 0.1)
    hV.append(rng.random(narc) + 0.1)
    hV.append(rng.random(nbus) + 0.1)
    hV.append(rng.random(ngen) + 0.1)
    hess_I, hess_J = S["hess_I"], S["hess_J"]
    hess = np.concatenate(hV)
    if indefinite:
        # negative curvature on the generator injections (each touches one balance row only, so
        # J'DJ cannot mask it when Sigma_s is moderate): K is indefinite until delta_w > ~50.
        hess_I = np.concatenate((hess_I, pg.astype(np.int32)))
        hess_J = np.concatenate((hess_J, pg.astype(np.int32)))
        hess = np.concatenate((hess, -50.0 * np.ones(ngen)))

    # IPM-like diagonals: every slack has both bounds, a third of x has a lower bound
    ind_lb = np.concatenate((np.arange(0, n, 3), n + np.arange(m)))
    ind_ub = np.concatenate((np.arange(1, n, 3), n + np.arange(m)))
    nlb, nub = len(ind_lb), len(ind_ub)
    dec = sigma_s_decades
    l_diag = -(10.0 ** rng.uniform(-dec / 2, 1, nlb))   # xl - x < 0
    u_diag = -(10.0 ** rng.uniform(-dec / 2, 1, nub))   # x - xu < 0
    l_lower = 10.0 ** rng.uniform(-dec / 2, 1, nlb)     # zl > 0
    u_lower = 10.0 ** rng.uniform(-dec / 2, 1, nub)
    reg = np.full(n + m, 1e-8)
    du_diag = np.full(m, -abs(du))
    return SparseKKTProblem(name, n, m, jac_I, jac_J, hess_I, hess_J, jac, hess, np.arange(m), ind_lb, ind_ub,
                            reg, l_diag, u_diag, l_lower, u_lower, du_diag,
                            meta=dict(nbus=nbus, ngen=ngen, nbranch=nbr, seed=seed))


@dataclass
class DenseKKTProblem:
    name: str
    n: int
    m: int
    hess: np.ndarray
    jac: np.ndarray
    ind_ineq: np.ndarray
    ind_eq: np.ndarray
    ind_lb: np.ndarray
    ind_ub: np.ndarray
    reg: np.ndarray
    l_diag: np.ndarray
    u_diag: np.ndarray
    l_lower: np.ndarray
    u_lower: np.ndarray
    du_diag: np.ndarray
    q: np.ndarray

    @property
    def pr_diag(self):
        p = self.reg.copy()
        p[self.ind_lb] -= self.l_lower / self.l_diag
        p[self.ind_ub] -= self.u_lower / self.u_diag
        return p


def dense_dummy_qp(n=2048, m=512, n_eq=0, seed=1, du_eq=-1e-8):
    """DenseDummyQP-shaped system (reference lib/MadNLPTests/src/Instances/dummy_qp.jl:79-151):
    P = R R' + 100 I (here R is n x n/8 + the 100 I shift keeps P well conditioned and the
    generator fast), A[i,i] = 1, A[i,i+1] = -1, bounds 0 <= x <= 1, 0 <= A x <= 1; the first
    `n_eq` constraints are equalities."""
    rng = np.random.default_rng(seed)
    R = rng.standard_normal((n, max(1, n // 8)))
    P = np.asfortranarray(R @ R.T + 100.0 * np.eye(n))
    A = np.zeros((m, n), order="F")
    i = np.arange(m)
    A[i, i] = 1.0
    A[i, i + 1] = -1.0
    q = rng.standard_normal(n)
    ind_eq = np.arange(n_eq)
    ind_ineq = np.arange(n_eq, m)
    ns = m - n_eq
    ind_lb = np.arange(n + ns)
    ind_ub = np.arange(n + ns)
    l_diag = -(10.0 ** rng.uniform(-4, 0, n + ns))
    u_diag = -(10.0 ** rng.uniform(-4, 0, n + ns))
    l_lower = 10.0 ** rng.uniform(-6, 1, n + ns)
    u_lower = 10.0 ** rng.uniform(-6, 1, n + ns)
    reg = np.zeros(n + ns)
    du_diag = np.zeros(m)
    du_diag[ind_eq] = du_eq
    return DenseKKTProblem(f"dense_dummy_qp_{n}_{m}_{n_eq}", n, m, P, A, ind_ineq, ind_eq, ind_lb, ind_ub, reg,
                           l_diag, u_diag, l_lower, u_lower, du_diag, q)


# ---------------------------------------------------------------------------- NLP models for the IPM driver
class HS15Model:
    """Hock-Schittkowski 15 exactly as the reference's test instance
    (lib/MadNLPTests/src/Instances/hs15.jl:1-103); optimum (0.5, 2), objective 306.5."""
    n, m = 2, 2
    x0 = np.zeros(2)
    y0 = np.zeros(2)
    lvar = np.array([-np.inf, -np.inf])
    uvar = np.array([0.5, np.inf])
    lcon = np.array([1.0, 0.0])
    ucon = np.array([np.inf, np.inf])
    jac_I = np.array([0, 0, 1, 1])
    jac_J = np.array([0, 1, 0, 1])
    hess_I = np.array([0, 1, 1])
    hess_J = np.array([0, 0, 1])

    def obj(self, x):
        return 100.0 * (x[1] - x[0] ** 2) ** 2 + (1.0 - x[0]) ** 2

    def grad(self, x):
        z = x[1] - x[0] ** 2
        return np.array([-400.0 * z * x[0] - 2.0 * (1.0 - x[0]), 200.0 * z])

    def cons(self, x):
        return np.array([x[0] * x[1], x[0] + x[1] ** 2])

    def jac_coord(self, x):
        return np.array([x[1], x[0], 1.0, 2 * x[1]])

    def jac_dense(self, x):
        return np.array([[x[1], x[0]], [1.0, 2 * x[1]]], order="F")

    def hess_coord(self, x, y, w=1.0):
        return np.array([w * (-400.0 * x[1] + 1200.0 * x[0] ** 2 + 2.0), w * (-400.0 * x[0]) + y[0],
                         w * 200.0 + 2.0 * y[1]])

    def hess_dense(self, x, y, w=1.0):
        h = self.hess_coord(x, y, w)
        return np.array([[h[0], h[1]], [h[1], h[2]]], order="F")


class InfeasibleModel:
    """The reference's `infeasible` test problem (lib/MadNLPTests/src/MadNLPTests.jl:120-136): min x^2 s.t. x == 0, x >= 1.
    The reference's own suite requires the exit status INFEASIBLE_PROBLEM_DETECTED (reached through restore! / robust!)."""
    n, m = 1, 1
    x0 = np.zeros(1)
    y0 = np.zeros(1)
    lvar = np.array([1.0])
    uvar = np.array([np.inf])
    lcon = np.array([0.0])
    ucon = np.array([0.0])
    jac_I = np.array([0])
    jac_J = np.array([0])
    hess_I = np.array([0])
    hess_J = np.array([0])
    q = np.zeros(1)          # as a QP 0.5 x'Hx + q'x for the device-resident driver (H = [2], J = [1])

    def obj(self, x):
        return x[0] ** 2

    def grad(self, x):
        return np.array([2.0 * x[0]])

    def cons(self, x):
        return np.array([x[0]])

    def jac_coord(self, x):
        return np.array([1.0])

    def jac_dense(self, x):
        return np.array([[1.0]], order="F")

    def hess_coord(self, x, y, w=1.0):
        return np.array([2.0 * w])

    def hess_dense(self, x, y, w=1.0):
        return np.array([[2.0 * w]], order="F")


class CubicDiskModel:
    """min -x1  s.t.  x2 - x1^3 = 0,  x1^2 + x2^2 <= r.  Small nonconvex NLP whose runs from poor starting points leave the
    regular phase (line-search failure -> restore! / robust!) and come back: used to exercise the restoration state machine.
    Optimum: x1 = the positive root of t^2 + t^6 = r, x2 = x1^3 (`solution()`)."""
    n, m = 2, 2
    lvar = np.array([-np.inf, -np.inf])
    uvar = np.array([np.inf, np.inf])
    jac_I = np.array([0, 0, 1, 1])
    jac_J = np.array([0, 1, 0, 1])
    hess_I = np.array([0, 1, 1])
    hess_J = np.array([0, 0, 1])

    def __init__(self, r, x0):
        self.r = float(r)
        self.x0, self.y0 = np.array(x0, dtype=float), np.zeros(2)
        self.lcon = np.array([0.0, -np.inf])
        self.ucon = np.array([0.0, self.r])

    def solution(self):
        lo, hi = 0.0, max(1.0, self.r)
        for _ in range(200):
            t = 0.5 * (lo + hi)
            lo, hi = (t, hi) if t ** 2 + t ** 6 < self.r else (lo, t)
        return np.array([lo, lo ** 3])

    def obj(self, x):
        return -x[0]

    def grad(self, x):
        return np.array([-1.0, 0.0])

    def cons(self, x):
        return np.array([x[1] - x[0] ** 3, x[0] ** 2 + x[1] ** 2])

    def jac_coord(self, x):
        return np.array([-3.0 * x[0] ** 2, 1.0, 2.0 * x[0], 2.0 * x[1]])

    def jac_dense(self, x):
        return np.array([[-3.0 * x[0] ** 2, 1.0], [2.0 * x[0], 2.0 * x[1]]], order="F")

    def hess_coord(self, x, y, w=1.0):
        return np.array([-6.0 * x[0] * y[0] + 2.0 * y[1], 0.0, 2.0 * y[1]])

    def hess_dense(self, x, y, w=1.0):
        h = self.hess_coord(x, y, w)
        return np.array([[h[0], h[1]], [h[1], h[2]]], order="F")


class WachterBieglerModel:
    """Waechter & Biegler's example (Math. Program. 88, 2000):  min x1  s.t.  x1^2 - x2 - 1 = 0,  x1 - x3 - 1/2 = 0,
    x2, x3 >= 0, started from (-2, 3, 1) -- the point from which pure line-search interior-point steps cannot reach the
    feasible region.  Drives the regular phase into restoration and back (state-machine regression; no reference-held
    answer for this instance)."""
    n, m = 3, 2
    x0 = np.array([-2.0, 3.0, 1.0])
    y0 = np.zeros(2)
    lvar = np.array([-np.inf, 0.0, 0.0])
    uvar = np.array([np.inf, np.inf, np.inf])
    lcon = np.zeros(2)
    ucon = np.zeros(2)
    jac_I = np.array([0, 0, 1, 1])
    jac_J = np.array([0, 1, 0, 2])
    hess_I = np.array([0])
    hess_J = np.array([0])

    def obj(self, x):
        return x[0]

    def grad(self, x):
        return np.array([1.0, 0.0, 0.0])

    def cons(self, x):
        return np.array([x[0] ** 2 - x[1] - 1.0, x[0] - x[2] - 0.5])

    def jac_coord(self, x):
        return np.array([2.0 * x[0], -1.0, 1.0, -1.0])

    def jac_dense(self, x):
        return np.array([[2.0 * x[0], -1.0, 0.0], [1.0, 0.0, -1.0]], order="F")

    def hess_coord(self, x, y, w=1.0):
        return np.array([2.0 * y[0]])

    def hess_dense(self, x, y, w=1.0):
        h = np.zeros((3, 3), order="F")
        h[0, 0] = 2.0 * y[0]
        return h


class LootsmaModel:
    """The reference's `lootsma` test problem (lib/MadNLPTests/src/MadNLPTests.jl:153-194), the fixed
    variable par = 6 substituted:  min x1^3 + 11 x1 - 6 sqrt(x1) + x3  s.t.  -sqrt(x1) - sqrt(x2) + sqrt(x3) >= 0,
    sqrt(x1) + sqrt(x2) + sqrt(x3) >= 4,  0 <= x <= 5,  x0 = 0.  The reference hard-codes the answers its own runs
    must reproduce (`LOOTSMA_X`, `LOOTSMA_Y`, accepted at atol = rtol = sqrt(tol))."""
    n, m = 3, 2
    x0 = np.zeros(3)
    y0 = np.zeros(2)
    lvar = np.zeros(3)
    uvar = 5.0 * np.ones(3)
    lcon = np.array([0.0, 4.0])
    ucon = np.array([np.inf, np.inf])
    jac_I = np.array([0, 0, 0, 1, 1, 1])
    jac_J = np.array([0, 1, 2, 0, 1, 2])
    hess_I = np.array([0, 1, 2])
    hess_J = np.array([0, 1, 2])
    LOOTSMA_X = np.array([0.07415998565403112, 2.9848713863700236, 4.0000304145340415])   # MadNLPTests.jl:177
    LOOTSMA_Y = np.array([-2.000024518601535, -2.0000305441119535])                       # MadNLPTests.jl:182

    def obj(self, x):
        return x[0] ** 3 + 11.0 * x[0] - 6.0 * np.sqrt(x[0]) + x[2]

    def grad(self, x):
        return np.array([3.0 * x[0] ** 2 + 11.0 - 3.0 / np.sqrt(x[0]), 0.0, 1.0])

    def cons(self, x):
        r = np.sqrt(x)
        return np.array([-r[0] - r[1] + r[2], r[0] + r[1] + r[2]])

    def jac_coord(self, x):
        d = 0.5 / np.sqrt(x)
        return np.array([-d[0], -d[1], d[2], d[0], d[1], d[2]])

    def jac_dense(self, x):
        d = 0.5 / np.sqrt(x)
        return np.array([[-d[0], -d[1], d[2]], [d[0], d[1], d[2]]], order="F")

    def hess_coord(self, x, y, w=1.0):
        dd = -0.25 * x ** -1.5  # (sqrt)''
        return np.array([w * (6.0 * x[0] + 1.5 * x[0] ** -1.5) + (-y[0] + y[1]) * dd[0], (-y[0] + y[1]) * dd[1],
                         (y[0] + y[1]) * dd[2]])

    def hess_dense(self, x, y, w=1.0):
        return np.asfortranarray(np.diag(self.hess_coord(x, y, w)))


class DenseQPModel:
    """min 0.5 x'Px + q'x  s.t. 0 <= x <= 1, gl <= Ax <= gu -- the reference's DenseDummyQP
    (lib/MadNLPTests/src/Instances/dummy_qp.jl) with our own seeded RNG."""

    def __init__(self, n=50, m=10, n_eq=0, seed=1):
        P = dense_dummy_qp(n, m, n_eq, seed)
        self.n, self.m = n, m
        # scaled so that gradients stay below nlp_scaling_max_gradient (the reference would scale)
        self.P, self.A, self.q = P.hess / 10.0, P.jac, P.q / 10.0
        self.x0, self.y0 = np.zeros(n), np.zeros(m)
        self.lvar, self.uvar = np.zeros(n), np.ones(n)
        self.lcon, self.ucon = np.zeros(m), np.ones(m)
        self.ucon[:n_eq] = 0.0
        il, jl = np.tril_indices(n)
        self.hess_I, self.hess_J = il, jl
        ji, jj = np.nonzero(self.A)
        self.jac_I, self.jac_J = ji, jj

    def obj(self, x):
        return 0.5 * x @ self.P @ x + self.q @ x

    def grad(self, x):
        return self.P @ x + self.q

    def cons(self, x):
        return self.A @ x

    def jac_dense(self, x):
        return self.A

    def jac_coord(self, x):
        return self.A[self.jac_I, self.jac_J]

    def hess_dense(self, x, y, w=1.0):
        return w * self.P

    def hess_coord(self, x, y, w=1.0):
        return w * self.P[self.hess_I, self.hess_J]


class SparseQPModel:
    """Convex QP with the OPF-shaped sparsity of `opf_shaped`: min 0.5 x'Hx + q'x s.t. cl <= Jx <= cu,
    xl <= x <= xu.  Drives the sparse-condensed path through a full IPM run (inertia correction,
    refinement, line search) at BASELINE config sizes."""

    def __init__(self, case="case118", seed=None):
        P = opf_shaped(case, seed=seed)
        rng = np.random.default_rng((seed or P.meta["seed"]) + 17)
        self.n, self.m = P.n, P.m
        self.jac_I, self.jac_J, self.hess_I, self.hess_J = P.jac_I, P.jac_J, P.hess_I, P.hess_J
        self.jv = P.jac / 4.0
        self.hv = P.hess
        import scipy.sparse as sp
        self.J = sp.csr_matrix((self.jv, (P.jac_I, P.jac_J)), shape=(P.m, P.n))
        lo_i, lo_j = np.maximum(P.hess_I, P.hess_J), np.minimum(P.hess_I, P.hess_J)
        L = sp.csr_matrix((self.hv, (lo_i, lo_j)), shape=(P.n, P.n))
        self.H = L + sp.tril(L, -1).T
        self.q = rng.standard_normal(P.n)
        xs = rng.uniform(-0.5, 0.5, P.n)            # a strictly feasible point defines the bounds
        cs = self.J @ xs
        self.lvar, self.uvar = xs - rng.uniform(0.5, 2, P.n), xs + rng.uniform(0.5, 2, P.n)
        self.lcon, self.ucon = cs - rng.uniform(0.1, 2, P.m), cs + rng.uniform(0.1, 2, P.m)
        self.x0, self.y0 = np.zeros(P.n), np.zeros(P.m)

    def obj(self, x):
        return 0.5 * x @ (self.H @ x) + self.q @ x

    def grad(self, x):
        return self.H @ x + self.q

    def cons(self, x):
        return self.J @ x

    def jac_coord(self, x):
        return self.jv

    def hess_coord(self, x, y, w=1.0):
        return w * self.hv


class ACOPFModel:
    """Polar AC optimal power flow on the synthetic grid of `_opf_structure` (PGLib case sizes; PGLib data are not available
    offline): the model the reference's benchmarks solve (`benchmark/` runs ExaModelsPower's `opf_model` on pglib cases).

      variables    va, vm per bus; pg, qg per generator; p, q per arc (two arcs per branch)
      objective    sum_g c2 pg^2 + c1 pg + c0
      constraints  va_ref = 0;  p_a - P_a(vm, va) = 0,  q_a - Q_a(vm, va) = 0 per arc;  angmin <= va_f - va_t <= angmax per
                   branch;  p_a^2 + q_a^2 <= rate_a^2 per arc;  active / reactive balance per bus
      arc flows    T(u, w, d) = alpha u^2 + u w (beta cos d + gamma sin d) with u = vm at the arc's own end, w = vm at the far
                   end, d = va_own - va_far; `arc_coef[a] = (alpha, beta, gamma)` for P_a, `(alpha', beta', gamma')` for Q_a,
                   derived from g, b, tap, shift and line charging as in the pi-model.

    The callbacks are the host (numpy) evaluation; `csrc/opf_eval.hip` evaluates the same expressions on the device
    (`DeviceOPFCallbacks`).  Constraint and nonzero order are those of `opf_shaped`."""

    def __init__(self, case="case118", seed=None, load=1.0):
        S, _ = _opf_structure(case, seed)
        self.S = S
        self.name = f"acopf_{S['name']}"
        nbus, ngen, nbr, narc = S["nbus"], S["ngen"], S["nbr"], S["narc"]
        self.nbus, self.ngen, self.nbr, self.narc = nbus, ngen, nbr, narc
        self.n, self.m = S["n"], S["m"]
        self.jac_I, self.jac_J, self.hess_I, self.hess_J = S["jac_I"], S["jac_J"], S["hess_I"], S["hess_J"]
        self.fr, self.to, self.gen_bus, self.arc_f, self.arc_t = S["fr"], S["to"], S["gen_bus"], S["arc_f"], S["arc_t"]
        rng = np.random.default_rng(S["seed"] + 101)
        # branches (per unit): series impedance, line charging, a few transformers with tap / phase shift
        r = rng.uniform(0.005, 0.03, nbr)
        x = rng.uniform(0.03, 0.12, nbr)
        g, b = r / (r * r + x * x), -x / (r * r + x * x)
        bc = rng.uniform(0.0, 0.06, nbr)
        trafo = rng.random(nbr) < 0.1
        tap = np.where(trafo, rng.uniform(0.95, 1.05, nbr), 1.0)
        shift = np.where(trafo & (rng.random(nbr) < 0.5), rng.uniform(-0.05, 0.05, nbr), 0.0)
        tr, ti = tap * np.cos(shift), tap * np.sin(shift)
        tm2 = tap * tap
        coef = np.empty((narc, 6))
        # from side: P = (g + g_fr)/tm2 vf^2 + (-g tr + b ti)/tm2 vf vt cos + (-b tr - g ti)/tm2 vf vt sin
        #            Q = -(b + b_fr)/tm2 vf^2 - (-b tr - g ti)/tm2 vf vt cos + (-g tr + b ti)/tm2 vf vt sin
        coef[:nbr, 0] = g / tm2
        coef[:nbr, 1] = (-g * tr + b * ti) / tm2
        coef[:nbr, 2] = (-b * tr - g * ti) / tm2
        coef[:nbr, 3] = -(b + bc / 2) / tm2
        coef[:nbr, 4] = -coef[:nbr, 2]
        coef[:nbr, 5] = coef[:nbr, 1]
        # to side (own end = the branch's to-bus): P = (g + g_to) vt^2 + (-g tr - b ti)/tm2 vt vf cos + (-b tr + g ti)/tm2 vt vf sin
        coef[nbr:, 0] = g
        coef[nbr:, 1] = (-g * tr - b * ti) / tm2
        coef[nbr:, 2] = (-b * tr + g * ti) / tm2
        coef[nbr:, 3] = -(b + bc / 2)
        coef[nbr:, 4] = -coef[nbr:, 2]
        coef[nbr:, 5] = coef[nbr:, 1]
        self.arc_coef = np.ascontiguousarray(coef)
        # buses: loads on ~70 %, a few shunts
        pd = np.where(rng.random(nbus) < 0.7, rng.uniform(0.05, 0.4, nbus), 0.0) * load
        qd = pd * rng.uniform(0.1, 0.4, nbus)
        sh = rng.random(nbus) < 0.05
        gs = np.where(sh, rng.uniform(0.0, 0.02, nbus), 0.0)
        bs = np.where(sh, rng.uniform(-0.1, 0.2, nbus), 0.0)
        self.bus_data = np.ascontiguousarray(np.stack((pd, qd, gs, bs), axis=1))
        # generators: capacity 1.8 x the total load, spread unevenly; quadratic costs
        wgt = rng.uniform(0.5, 1.5, ngen)
        pmax = 1.8 * pd.sum() * wgt / wgt.sum()
        qmax = 0.8 * pmax + 0.2
        self.gen_cost = np.ascontiguousarray(np.stack((rng.uniform(1.0, 8.0, ngen) / np.maximum(pmax, 1.0),
                                                       rng.uniform(10.0, 40.0, ngen), rng.uniform(0.0, 5.0, ngen)), axis=1))
        rate = rng.uniform(2.0, 4.0, nbr) * max(1.0, load)
        rate = np.concatenate((rate, rate))
        ang = np.pi / 6
        # bounds and start
        va, vm, pgi, qgi, p, q = (S[k] for k in ("va", "vm", "pg", "qg", "p", "q"))
        self.lvar, self.uvar = np.full(self.n, -np.inf), np.full(self.n, np.inf)
        self.lvar[vm], self.uvar[vm] = 0.94, 1.06
        self.lvar[pgi], self.uvar[pgi] = 0.0, pmax
        self.lvar[qgi], self.uvar[qgi] = -qmax, qmax
        self.lvar[p], self.uvar[p] = -rate, rate
        self.lvar[q], self.uvar[q] = -rate, rate
        self.lcon, self.ucon = np.zeros(self.m), np.zeros(self.m)
        o = 1 + 2 * narc
        self.lcon[o:o + nbr], self.ucon[o:o + nbr] = -ang, ang
        o += nbr
        self.lcon[o:o + narc], self.ucon[o:o + narc] = -np.inf, rate * rate
        self.x0 = np.zeros(self.n)
        self.x0[vm] = 1.0
        self.x0[pgi] = 0.5 * pmax
        self.y0 = np.zeros(self.m)
        # bus -> arcs / generators incidence (balance rows)
        import scipy.sparse as sp
        self._A_arc = sp.csr_matrix((np.ones(narc), (self.arc_f, np.arange(narc))), shape=(nbus, narc))
        self._A_gen = sp.csr_matrix((np.ones(ngen), (self.gen_bus, np.arange(ngen))), shape=(nbus, ngen))

    # ---- pieces
    def _split(self, x):
        S = self.S
        return x[S["va"]], x[S["vm"]], x[S["pg"]], x[S["qg"]], x[S["p"]], x[S["q"]]

    def _arc_terms(self, x):
        va, vm = x[self.S["va"]], x[self.S["vm"]]
        u, w = vm[self.arc_f], vm[self.arc_t]
        d = va[self.arc_f] - va[self.arc_t]
        return u, w, np.cos(d), np.sin(d)

    def obj(self, x):
        pg = x[self.S["pg"]]
        c = self.gen_cost
        return float((c[:, 0] * pg * pg + c[:, 1] * pg + c[:, 2]).sum())

    def grad(self, x):
        g = np.zeros(self.n)
        pg = x[self.S["pg"]]
        g[self.S["pg"]] = 2.0 * self.gen_cost[:, 0] * pg + self.gen_cost[:, 1]
        return g

    def cons(self, x):
        va, vm, pg, qg, p, q = self._split(x)
        u, w, cs, sn = self._arc_terms(x)
        k = self.arc_coef
        c = np.empty(self.m)
        c[0] = va[0]
        narc, nbr = self.narc, self.nbr
        o = 1
        c[o:o + narc] = p - (k[:, 0] * u * u + u * w * (k[:, 1] * cs + k[:, 2] * sn)); o += narc
        c[o:o + narc] = q - (k[:, 3] * u * u + u * w * (k[:, 4] * cs + k[:, 5] * sn)); o += narc
        c[o:o + nbr] = va[self.fr] - va[self.to]; o += nbr
        c[o:o + narc] = p * p + q * q; o += narc
        pd, qd, gs, bs = self.bus_data.T
        c[o:o + self.nbus] = pd + gs * vm * vm + self._A_arc @ p - self._A_gen @ pg; o += self.nbus
        c[o:o + self.nbus] = qd - bs * vm * vm + self._A_arc @ q - self._A_gen @ qg
        return c

    def jac_coord(self, x):
        va, vm, pg, qg, p, q = self._split(x)
        u, w, cs, sn = self._arc_terms(x)
        k = self.arc_coef
        out = [np.ones(1)]
        for c0 in (0, 3):
            K = k[:, c0 + 1] * cs + k[:, c0 + 2] * sn
            Kp = -k[:, c0 + 1] * sn + k[:, c0 + 2] * cs
            Tu, Tw, Td = 2.0 * k[:, c0] * u + w * K, u * K, u * w * Kp
            out.append(np.stack((np.ones(self.narc), -Tu, -Tw, -Td, Td), axis=1).ravel())
        out.append(np.tile(np.array([1.0, -1.0]), self.nbr))
        out.append(np.stack((2.0 * p, 2.0 * q), axis=1).ravel())
        pd, qd, gs, bs = self.bus_data.T
        out += [2.0 * gs * vm, np.ones(self.narc), -np.ones(self.ngen)]
        out += [-2.0 * bs * vm, np.ones(self.narc), -np.ones(self.ngen)]
        return np.concatenate(out)

    def hess_coord(self, x, y, w=1.0):
        u, wv, cs, sn = self._arc_terms(x)
        k = self.arc_coef
        narc, nbr, nbus = self.narc, self.nbr, self.nbus
        out = []
        for blk, c0 in enumerate((0, 3)):
            yy = -y[1 + blk * narc:1 + (blk + 1) * narc]          # c = own - T
            K = k[:, c0 + 1] * cs + k[:, c0 + 2] * sn
            Kp = -k[:, c0 + 1] * sn + k[:, c0 + 2] * cs
            uwK = u * wv * K
            z = np.zeros(narc)
            # lower triangle of (u, w, tf, tt): (0,0) (1,0) (1,1) (2,0) (2,1) (2,2) (3,0) (3,1) (3,2) (3,3)
            H = np.stack((2.0 * k[:, c0], K, z, wv * Kp, u * Kp, -uwK, -(wv * Kp), -(u * Kp), uwK, -uwK), axis=1)
            out.append((yy[:, None] * H).ravel())
        o = 1 + 2 * narc + nbr
        yth = y[o:o + narc]; o += narc
        out += [2.0 * yth, 2.0 * yth]
        pd, qd, gs, bs = self.bus_data.T
        out.append(2.0 * gs * y[o:o + nbus] - 2.0 * bs * y[o + nbus:o + 2 * nbus])
        out.append(w * 2.0 * self.gen_cost[:, 0])
        return np.concatenate(out)


class TwoStageQPModel:
    """Diagonal-Hessian two-stage stochastic QP -- the reference's `TwoStageQP` / `build_twostage_qp`
    (lib/MadNLPTests/src/Instances/twostage_qp.jl:1-175), the instance its `SchurComplementKKTSystem` tests run on
    (test/schur_test.jl).  Variables `x = [v_1 .. v_ns (nv each), d (nd)]`, constraints `c = [c_1 .. c_ns (nc each)]`,
    `min 0.5 x' diag(H) x + g' x  s.t.  lcon <= A x <= ucon, lvar <= x <= uvar`; constraint row i of scenario k touches the
    scenario's own variables and the design variables only (the block-arrow structure the Schur system needs).
    Argument shapes as in the reference, with the scenario index LAST: hess_v, g_v, lvar_v, uvar_v (nv, ns); hess_d, g_d,
    lvar_d, uvar_d (nd,); A_v (nc, nv, ns); A_d (nc, nd, ns); lcon, ucon (nc, ns)."""

    def __init__(self, ns, nv, nd, nc, hess_v, hess_d, g_v, g_d, A_v, A_d, lcon, ucon, lvar_v, uvar_v, lvar_d, uvar_d,
                 sparse_pattern=False):
        """`sparse_pattern`: the Jacobian's COO pattern holds the nonzero entries of A_v / A_d only and A is kept as a scipy CSR
        matrix (large instances: the dense pattern of the reference's generator has nc (nv + nd) entries per scenario and a dense
        m x n matrix behind `cons`)."""
        f = lambda a, shape: np.asarray(a, dtype=float).reshape(shape)  # noqa: E731
        hess_v, g_v, lvar_v, uvar_v = (f(a, (nv, ns)) for a in (hess_v, g_v, lvar_v, uvar_v))
        hess_d, g_d, lvar_d, uvar_d = (f(a, (nd,)) for a in (hess_d, g_d, lvar_d, uvar_d))
        A_v, A_d = f(A_v, (nc, nv, ns)), f(A_d, (nc, nd, ns))
        lcon, ucon = f(lcon, (nc, ns)), f(ucon, (nc, ns))
        self.ns, self.nv, self.nd, self.nc = ns, nv, nd, nc
        n, m, off = ns * nv + nd, ns * nc, ns * nv
        self.n, self.m = n, m
        # (column-major flattening of the (j, k) arrays = scenario-major vectors)
        self.H_diag = np.concatenate((hess_v.T.ravel(), hess_d))
        self.g0 = np.concatenate((g_v.T.ravel(), g_d))
        self.lvar = np.concatenate((lvar_v.T.ravel(), lvar_d))
        self.uvar = np.concatenate((uvar_v.T.ravel(), uvar_d))
        self.lcon, self.ucon = lcon.T.ravel().copy(), ucon.T.ravel().copy()
        if sparse_pattern:
            import scipy.sparse as sp
            rows, cols, vals = [], [], []
            for k in range(ns):
                iv, jv = np.nonzero(A_v[:, :, k])
                idd, jd = np.nonzero(A_d[:, :, k])
                # (row-major within a scenario, recourse entries of a row before its design entries)
                r = np.concatenate((iv, idd)); c = np.concatenate((k * nv + jv, off + jd))
                v = np.concatenate((A_v[iv, jv, k], A_d[idd, jd, k]))
                o = np.lexsort((c, r))
                rows.append(k * nc + r[o]); cols.append(c[o]); vals.append(v[o])
            self.jac_I, self.jac_J = np.concatenate(rows).astype(np.int64), np.concatenate(cols).astype(np.int64)
            self.jvals = np.concatenate(vals)
            self.A = sp.csr_matrix((self.jvals, (self.jac_I, self.jac_J)), shape=(m, n))
            self.hess_I = self.hess_J = np.arange(n, dtype=np.int64)
            self.x0, self.y0 = np.zeros(n), np.zeros(m)
            return
        A = np.zeros((m, n))
        rows, cols = [], []
        for k in range(ns):
            for i in range(nc):
                r = k * nc + i
                A[r, k * nv:(k + 1) * nv] = A_v[i, :, k]
                A[r, off:] = A_d[i, :, k]
                rows += [r] * (nv + nd)
                cols += list(range(k * nv, (k + 1) * nv)) + list(range(off, n))
        self.A = A
        self.jac_I, self.jac_J = np.array(rows, dtype=np.int64), np.array(cols, dtype=np.int64)
        self.jvals = A[self.jac_I, self.jac_J]
        self.hess_I = self.hess_J = np.arange(n, dtype=np.int64)
        self.x0, self.y0 = np.zeros(n), np.zeros(m)

    def schur_opts(self):
        """the reference's `schur_opts(; ns, nv, nd, nc)` (twostage_qp.jl:189-191)."""
        return dict(ns=self.ns, nv=self.nv, nd=self.nd, nc=self.nc)

    def obj(self, x):
        return 0.5 * x @ (self.H_diag * x) + self.g0 @ x

    def grad(self, x):
        return self.H_diag * x + self.g0

    def cons(self, x):
        return self.A @ x

    def jac_coord(self, x):
        return self.jvals

    def jac_dense(self, x):
        return self.A if isinstance(self.A, np.ndarray) else self.A.toarray()

    def hess_coord(self, x, y, w=1.0):
        return w * self.H_diag

    def hess_dense(self, x, y, w=1.0):
        return w * np.diag(self.H_diag)


def random_twostage_qp(ns=8, nv=40, nd=12, nc=10, nc_eq=4, seed=0, density_v=0.4, density_d=1.0):
    """A larger `TwoStageQPModel` with the first `nc_eq` constraints of every scenario equalities and the rest two-sided
    inequalities around a strictly feasible point (so that both kinds of rows of the Schur system's blocks are exercised).
    `density_v` / `density_d`: fraction of nonzeros in a constraint row's recourse / design part (the condensation of an
    inequality row costs the SQUARE of its length in the reference's pair lists: large instances need sparse rows)."""
    rng = np.random.default_rng(seed)
    hess_v = rng.uniform(0.5, 4.0, (nv, ns))
    hess_d = rng.uniform(0.5, 4.0, nd)
    g_v, g_d = rng.standard_normal((nv, ns)), rng.standard_normal(nd)
    sparse = density_v < 0.4 or density_d < 1.0
    # (sparse patterns: ONE mask for all scenarios -- the reference builds one symbolic block and requires the same local pattern)
    A_v = rng.standard_normal((nc, nv, ns)) * (rng.random((nc, nv, 1) if sparse else (nc, nv, ns)) < density_v)
    for k in range(ns):                       # full row rank of the equality rows
        for i in range(nc):
            A_v[i, (3 * i + (0 if sparse else k)) % nv, k] += 2.0
    A_d = rng.standard_normal((nc, nd, ns)) * 0.5
    if density_d < 1.0:
        A_d = A_d * (rng.random((nc, nd, 1)) < density_d)
    xs_v, xs_d = rng.uniform(-0.5, 0.5, (nv, ns)), rng.uniform(-0.5, 0.5, nd)
    c = np.einsum("ijk,jk->ik", A_v, xs_v) + np.einsum("ijk,j->ik", A_d, xs_d)
    lcon, ucon = c - rng.uniform(0.2, 1.0, (nc, ns)), c + rng.uniform(0.2, 1.0, (nc, ns))
    lcon[:nc_eq], ucon[:nc_eq] = c[:nc_eq], c[:nc_eq]
    return TwoStageQPModel(ns, nv, nd, nc, hess_v, hess_d, g_v, g_d, A_v, A_d, lcon, ucon,
                           xs_v - rng.uniform(1, 3, (nv, ns)), xs_v + rng.uniform(1, 3, (nv, ns)),
                           xs_d - rng.uniform(1, 3, nd), xs_d + rng.uniform(1, 3, nd), sparse_pattern=sparse)
