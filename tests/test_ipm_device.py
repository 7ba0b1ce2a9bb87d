"""Device IPM reductions (`mnk_ipm_*`, SURVEY 8(f).4 second slice) against the oracle restatement of reference
`src/IPM/kernels.jl:263-388,675-695`.  The CPU half pins the oracle on hand-computed values; the GPU half compares the
HIP reductions with it: max/min-type results bit-exact, sum-type results to summation-order rounding."""
import math

import numpy as np
import pytest

from oracle import ipm_kernels as ok


# --------------------------------------------------------------------------- oracle pins (CPU)
def test_oracle_reductions_on_hand_computed_values():
    x = np.array([1.0, 2.0, 3.0]); xl = np.array([0.0, 1.0, -1e300]); xu = np.array([2.0, 1e300, 5.0])
    lb, ub = np.array([0, 1]), np.array([0, 2])
    mu = 0.1
    # varphi = obj - mu (log 1 + log 1) - mu (log 1 + log 2)
    assert ok.get_varphi(7.0, x[lb], xl[lb], xu[ub], x[ub], mu) == pytest.approx(7.0 - mu * math.log(2.0), rel=1e-15)
    assert ok.get_varphi(7.0, np.array([0.5]), np.array([1.0]), xu[ub], x[ub], mu) == float("inf")  # negative slack
    f = np.array([1.0, -2.0, 0.5]); zl = np.array([0.5, 0.25, 0.0]); zu = np.array([0.125, 0.0, 2.0]); jl = np.array([0.0, 1.0, -4.0])
    assert ok.get_inf_du(f, zl, zu, jl, 2.0) == max(abs(1 - .5 + .125), abs(-2 - .25 + 1), abs(.5 + 2 - 4)) / 2.0
    assert ok.get_inf_compl(x[lb], xl[lb], zl[lb], xu[ub], x[ub], zu[ub], mu, 4.0) == \
        max(abs(1 * .5 - mu), abs(1 * .25 - mu), abs(1 * .125 - mu), abs(2 * 2.0 - mu)) / 4.0
    assert ok.get_min_complementarity(x[lb], xl[lb], zl[lb], x[ub], xu[ub], zu[ub]) == 0.125
    assert ok.get_average_complementarity(x[lb], xl[lb], zl[lb], x[ub], xu[ub], zu[ub]) == pytest.approx((.5 + .25 + .125 + 4.0) / 4)
    dx = np.array([-4.0, 0.0, 8.0])
    assert ok.get_alpha_max(x, xl, xu, dx, 0.99) == min(1.0, (-1.0 + 0.0) * 0.99 / -4.0, (-3.0 + 5.0) * 0.99 / 8.0)
    assert ok.get_alpha_max(x, xl, xu, np.zeros(3), 0.99) == 1.0
    assert ok.get_alpha_z(zl[lb], zu[ub], np.array([-1.0, 1.0]), np.array([0.0, -16.0]), 0.5) == min((-0.5) * 0.5 / -1.0, (-2.0) * 0.5 / -16.0)
    assert ok.get_rel_search_norm(x, dx) == 8.0 / 4.0
    assert ok.get_sd(np.array([3.0, -5.0]), zl[lb], zu[ub], 1.0) == (8.0 + .75 + 2.125) / 6
    assert ok.get_sc(zl[lb], zu[ub], 100.0) == 1.0
    assert ok.get_varphi_d(f, x, np.array([0.0, 1.0, -np.inf]), np.array([2.0, np.inf, 5.0]), dx, mu) == \
        pytest.approx((1 - mu / 1 + mu / 1) * -4.0 + (0.5 - 0.0 + mu / 2.0) * 8.0, rel=1e-15)


# --------------------------------------------------------------------------- HIP vs oracle (GPU)
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ctx():
    import madnlp_jl_amd as mj
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    c = mj.HipContext(0)
    yield c
    c.close()


def _data(rng, ntot, nlb, nub, m):
    lb = np.sort(rng.choice(ntot, nlb, replace=False)); ub = np.sort(rng.choice(ntot, nub, replace=False))
    xl = np.full(ntot, -1e300); xu = np.full(ntot, 1e300)
    xl[lb] = -rng.uniform(0.5, 2.0, nlb); xu[ub] = rng.uniform(0.5, 2.0, nub)
    x = rng.uniform(-0.45, 0.45, ntot)
    near = rng.random(ntot) < 0.05   # some entries close to a bound (tiny slacks: the interesting end of the barrier)
    x[near] = np.where(rng.random(near.sum()) < 0.5, xl[near] + 1e-9, xu[near] - 1e-9)
    x = np.clip(x, np.where(xl > -1e299, xl + 1e-12, -1.0), np.where(xu < 1e299, xu - 1e-12, 1.0))
    zl = np.zeros(ntot); zu = np.zeros(ntot)
    zl[lb] = 10.0 ** rng.uniform(-9, 3, nlb); zu[ub] = 10.0 ** rng.uniform(-9, 3, nub)
    f = rng.standard_normal(ntot) * 10.0 ** rng.uniform(-3, 3, ntot)
    jacl = rng.standard_normal(ntot)
    dx = rng.standard_normal(ntot) * 10.0 ** rng.uniform(-4, 1, ntot)
    dx[rng.random(ntot) < 0.1] = 0.0
    dzl = rng.standard_normal(nlb); dzu = rng.standard_normal(nub)
    y = rng.standard_normal(m) * 10.0
    c = rng.standard_normal(m) * 10.0 ** rng.uniform(-8, 0, m)
    return dict(lb=lb, ub=ub, x=x, xl=xl, xu=xu, zl=zl, zu=zu, f=f, jacl=jacl, dx=dx, dzl=dzl, dzu=dzu, y=y, c=c)


@pytest.mark.gpu
@pytest.mark.parametrize("ntot,nlb,nub,m", [(1, 1, 1, 1), (7, 3, 0, 2), (1000, 400, 377, 300), (27838, 16646 + 4000, 16646 + 3000, 16646),
                                             (300001, 150000, 120000, 100000), (50, 0, 0, 0)])
def test_device_reductions_match_the_oracle(ctx, ntot, nlb, nub, m):
    import madnlp_jl_amd as mj
    rng = np.random.default_rng(ntot)
    nlb, nub = min(nlb, ntot), min(nub, ntot)
    d = _data(rng, ntot, nlb, nub, m)
    lb, ub = d["lb"], d["ub"]
    dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in d.items() if k not in ("lb", "ub")}
    K = mj.IPMDeviceKernels(ntot, lb, ub, ctx=ctx)
    x, xl, xu, zl, zu = d["x"], d["xl"], d["xu"], d["zl"], d["zu"]
    mu, tau, sd, sc = 0.01, 0.99, 1.7, 2.3
    # max / min type: bit-exact
    assert K.get_inf_du(dev["f"], dev["zl"], dev["zu"], dev["jacl"], sd) == ok.get_inf_du(d["f"], zl, zu, d["jacl"], sd)
    assert K.get_inf_compl(dev["x"], dev["xl"], dev["xu"], dev["zl"], dev["zu"], mu, sc) == \
        ok.get_inf_compl(x[lb], xl[lb], zl[lb], xu[ub], x[ub], zu[ub], mu, sc)
    assert K.get_min_complementarity(dev["x"], dev["xl"], dev["xu"], dev["zl"], dev["zu"]) == \
        ok.get_min_complementarity(x[lb], xl[lb], zl[lb], x[ub], xu[ub], zu[ub])
    assert K.get_alpha_max(dev["x"], dev["xl"], dev["xu"], dev["dx"], tau) == ok.get_alpha_max(x, xl, xu, d["dx"], tau)
    assert K.get_alpha_z(dev["zl"], dev["zu"], dev["dzl"], dev["dzu"], tau) == ok.get_alpha_z(zl[lb], zu[ub], d["dzl"], d["dzu"], tau)
    assert K.get_rel_search_norm(dev["x"], dev["dx"]) == ok.get_rel_search_norm(x, d["dx"])
    ninf, none = K.get_norms(dev["c"])
    assert ninf == np.abs(d["c"]).max(initial=0.0)
    # sum type: summation-order rounding (relative to the sum of magnitudes)
    assert none == pytest.approx(np.abs(d["c"]).sum(), rel=1e-13, abs=0.0 if m else 1e-300)
    vo = ok.get_varphi(3.5, x[lb], xl[lb], xu[ub], x[ub], mu)
    mag = abs(3.5) + mu * (np.abs(np.log(x[lb] - xl[lb])).sum() + np.abs(np.log(xu[ub] - x[ub])).sum())
    assert abs(K.get_varphi(3.5, dev["x"], dev["xl"], dev["xu"], mu) - vo) <= 1e-13 * max(mag, 1.0)
    terms = (d["f"] - mu / (x - xl) + mu / (xu - x)) * d["dx"]
    assert abs(K.get_varphi_d(dev["f"], dev["x"], dev["xl"], dev["xu"], dev["dx"], mu) -
               ok.get_varphi_d(d["f"], x, xl, xu, d["dx"], mu)) <= 1e-13 * max(np.abs(terms).sum(), 1e-300)
    avg = ok.get_average_complementarity(x[lb], xl[lb], zl[lb], x[ub], xu[ub], zu[ub])
    cmag = (np.abs((x[lb] - xl[lb]) * zl[lb]).sum() + np.abs((xu[ub] - x[ub]) * zu[ub]).sum()) / max(1, nlb + nub)
    # (the reference forms dot(x, z) - dot(xl, z); the device sums (x - xl) z: same value up to cancellation in the former)
    assert abs(K.get_average_complementarity(dev["x"], dev["xl"], dev["xu"], dev["zl"], dev["zu"]) - avg) <= \
        1e-10 * max(cmag + (np.abs(x[lb] * zl[lb]).sum() + np.abs(xu[ub] * zu[ub]).sum()) / max(1, nlb + nub) * 1e-3, 1e-300) + 1e-12 * abs(avg)
    sd_d, sc_d = K.get_sd_sc(dev["y"], dev["zl"], dev["zu"], 100.0)
    assert sd_d == pytest.approx(ok.get_sd(d["y"], zl[lb], zu[ub], 100.0), rel=1e-13)
    assert sc_d == pytest.approx(ok.get_sc(zl[lb], zu[ub], 100.0), rel=1e-13)
    K.close()


@pytest.mark.gpu
def test_device_reductions_edge_cases(ctx):
    """A negative slack makes the barrier objective +Inf (reference `_get_varphi`), a zero step gives alpha = 1, NaN
    propagates through max/min like the reference's `max` / `min`."""
    import madnlp_jl_amd as mj
    ntot = 5
    lb, ub = np.array([0, 2]), np.array([1, 2])
    K = mj.IPMDeviceKernels(ntot, lb, ub, ctx=ctx)
    t = lambda a: torch.tensor(a, dtype=torch.float64, device="cuda")  # noqa: E731
    x = t([0.0, 1.0, 0.5, 0.0, 0.0]); xl = t([0.1, -9.0, 0.0, -9.0, -9.0]); xu = t([9.0, 2.0, 1.0, 9.0, 9.0])
    assert K.get_varphi(1.0, x, xl, xu, 0.1) == float("inf")                 # x[0] < xl[0]
    assert K.get_alpha_max(x, xl, xu, t([0.0] * 5), 0.99) == 1.0
    assert K.get_alpha_z(t([1.0] * 5), t([1.0] * 5), t([0.0, 1.0]), t([2.0, 0.0]), 0.99) == 1.0
    assert math.isnan(K.get_inf_du(t([float("nan"), 0, 0, 0, 0]), x, x, x, 1.0))
    assert math.isnan(K.get_rel_search_norm(x, t([0, 0, float("nan"), 0, 0])))
    K.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["dense_condensed", "sparse_condensed"])
def test_device_kernels_on_the_iterates_of_a_real_ipm_run(ctx, kind):
    """The IPM mirror solves `lootsma` on the HIP back-end; at EVERY call of set_aug_diagonal!, get_inf_compl,
    get_varphi and get_alpha_max the device-side twin is evaluated on the same iterate: the feeder's diagonals and the
    max/min reductions must be bit-identical to the host values, the barrier objective equal to summation rounding."""
    import madnlp_jl_amd as mj
    from madnlp_jl_amd.ipm import IPMOptions, MadNLPSolver
    from madnlp_jl_amd.problems import LootsmaModel
    nlp = LootsmaModel()
    sparse = kind == "sparse_condensed"
    counts = {"diag": 0, "compl": 0, "varphi": 0, "alpha": 0}
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()  # noqa: E731
    fin = lambda a: np.clip(a, -1e300, 1e300)  # noqa: E731  (the library never reads bounds outside ind_lb / ind_ub)

    class Checked(MadNLPSolver):
        _K = None

        def _kern(self):
            if self._K is None:
                self._K = mj.IPMDeviceKernels(len(self.x), self.ind_lb, self.ind_ub, ctx=ctx)
            return self._K

        def set_aug_diagonal(self):
            super().set_aug_diagonal()
            k, o = self.kkt, self.opt
            k.set_aug_diagonal_device(self.x, fin(self.xl), fin(self.xu), self.zl, self.zu,
                                      o.default_primal_regularization, o.default_dual_regularization)
            got = k.get_diagonals_device()
            for name in ("pr_diag", "du_diag", "reg", "l_diag", "u_diag", "l_lower", "u_lower"):
                np.testing.assert_array_equal(got[name], getattr(k, name), err_msg=name)
            counts["diag"] += 1

        def inf_compl(self, mu, sc):
            v = super().inf_compl(mu, sc)
            assert self._kern().get_inf_compl(dev(self.x), dev(fin(self.xl)), dev(fin(self.xu)), dev(self.zl), dev(self.zu), mu, sc) == v
            counts["compl"] += 1
            return v

        def varphi(self, obj, x):
            v = super().varphi(obj, x)
            d = self._kern().get_varphi(obj, dev(x), dev(fin(self.xl)), dev(fin(self.xu)), self.mu)
            assert (d == v) if not np.isfinite(v) else abs(d - v) <= 1e-12 * max(1.0, abs(v))
            counts["varphi"] += 1
            return v

        def alpha_max(self, dx):
            v = super().alpha_max(dx)
            assert self._kern().get_alpha_max(dev(self.x), dev(self.xl), dev(self.xu), dev(dx), self.tau) == v
            counts["alpha"] += 1
            return v

    def factory(info):
        opt = mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN)
        if sparse:
            return mj.SparseCondensedKKTSystem(info["n"], info["m"], nlp.jac_I, nlp.jac_J, nlp.hess_I, nlp.hess_J,
                                               info["ind_ineq"], info["ind_lb"], info["ind_ub"], ctx=ctx, opt_linear_solver=opt)
        return mj.DenseCondensedKKTSystem(info["n"], info["m"], info["ind_ineq"], info["ind_eq"], info["ind_lb"],
                                          info["ind_ub"], ctx=ctx, opt_linear_solver=opt)

    opt = IPMOptions(tol=1e-6 if sparse else 1e-8)
    if sparse:
        opt.relax_equality, opt.dual_initialization = True, "zero"
    s = Checked(nlp, factory, opt, sparse=sparse)
    s.solve()
    assert s.status == "SOLVE_SUCCEEDED"
    assert min(counts.values()) >= 5, counts
    tol = np.sqrt(s.opt.tol)
    assert np.abs(s.x[:3] - nlp.LOOTSMA_X).max() < tol
    if s._K is not None:
        s._K.close()
    s.kkt.close()


def test_oracle_elementwise_pieces_on_hand_computed_values():
    x = np.array([1.0, 2.0, 3.0]); xl = np.array([0.0, 2.0 - 1e-18, -np.inf]); xu = np.array([1.0 + 1e-17, np.inf, 5.0])
    lb, ub = np.array([0, 1]), np.array([0, 2])
    zl = np.array([0.5, 0.25, 0.0]); zu = np.array([0.125, 0.0, 2.0])
    px, py, pzl, pzu = ok.set_aug_rhs(np.array([1.0, 2.0, 3.0]), zl, zu, np.array([0.5, 0.5, 0.5]), np.array([7.0]),
                                      x[lb], xl[lb], zl[lb], xu[ub], x[ub], zu[ub], 0.1)
    np.testing.assert_array_equal(px, [-1 + .5 - .125 - .5, -2 + .25 - 0 - .5, -3 + 0 - 2 - .5])
    np.testing.assert_array_equal(py, [-7.0])
    np.testing.assert_array_equal(pzl, [(0.0 - 1.0) * .5 + .1, (xl[1] - 2.0) * .25 + .1])
    np.testing.assert_array_equal(pzu, [(xu[0] - 1.0) * .125 - .1, (5.0 - 3.0) * 2.0 - .1])
    p = px.copy(); ok.dual_inf_perturbation(p, np.array([1]), np.array([2]), 0.1, 1e-5)
    np.testing.assert_array_equal(p, [px[0], px[1] - 1e-6, px[2] + 1e-6])
    eps = np.finfo(float).eps
    xl2, xu2 = xl.copy(), xu.copy()
    ok.adjust_boundary(x, xl2, xu2, lb, ub, 0.1)   # x[1] - xl[1] = 0 < eps mu: pushed out; x[0] at its upper bound likewise
    assert xl2[0] == 0.0 and xl2[1] == xl[1] - eps ** 0.75 * 2.0 and xu2[0] == xu[0] + eps ** 0.75 * 1.0 and xu2[2] == 5.0
    z = np.array([0.5, 1e12, 0.0, 1e-30]); xx = np.array([1.0, 2.0, 3.0, 4.0]); xb = np.array([0.0, 1.5, -np.inf, 3.0])
    ok.reset_bound_dual(z, xx, xb, 0.1, 1e10)   # kept / clipped from above / unbounded stays 0 / lifted from below
    np.testing.assert_array_equal(z, [0.5, (1e10 * 0.1) / 0.5, 0.0, (0.1 / 1e10) / 1.0])


@pytest.mark.gpu
@pytest.mark.parametrize("ntot,nlb,nub,m", [(7, 3, 2, 2), (1000, 400, 377, 300), (27838, 20646, 19646, 16646)])
def test_device_elementwise_pieces_bit_exact(ctx, ntot, nlb, nub, m):
    """set_aug_rhs!, dual_inf_perturbation!, adjust_boundary!, reset_bound_dual! on device vectors: bit-identical to the
    oracle (elementwise IEEE operations, no contraction)."""
    import madnlp_jl_amd as mj
    rng = np.random.default_rng(ntot + 1)
    d = _data(rng, ntot, nlb, nub, m)
    lb, ub = d["lb"], d["ub"]
    xl = np.where(d["xl"] < -1e299, -np.inf, d["xl"]); xu = np.where(d["xu"] > 1e299, np.inf, d["xu"])
    x, zl, zu, mu = d["x"], d["zl"], d["zu"], 3e-3
    near = rng.choice(nlb, max(1, nlb // 20), replace=False)
    x[lb[near]] = xl[lb[near]] + 1e-20          # entries sitting on their lower bound: adjust_boundary! must fire
    llb = np.setdiff1d(lb, ub); uub = np.setdiff1d(ub, lb)
    K = mj.IPMDeviceKernels(ntot, lb, ub, ctx=ctx)
    K.set_perturbation_sets(llb, uub)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    px, py, pzl, pzu = (torch.empty(n, dtype=torch.float64, device="cuda") for n in (ntot, m, nlb, nub))
    K.set_aug_rhs(t(d["f"]), t(zl), t(zu), t(d["jacl"]), t(d["c"]), t(x), t(xl), t(xu), mu, px, py, pzl, pzu)
    K.dual_inf_perturbation(px, mu, 1e-5)
    opx, opy, opzl, opzu = ok.set_aug_rhs(d["f"], zl, zu, d["jacl"], d["c"], x[lb], xl[lb], zl[lb], xu[ub], x[ub], zu[ub], mu)
    ok.dual_inf_perturbation(opx, llb, uub, mu, 1e-5)
    for got, want in ((px, opx), (py, opy), (pzl, opzl), (pzu, opzu)):
        np.testing.assert_array_equal(got.cpu().numpy(), want)
    dxl, dxu = t(xl), t(xu)
    K.adjust_boundary(t(x), dxl, dxu, mu)
    oxl, oxu = xl.copy(), xu.copy()
    ok.adjust_boundary(x, oxl, oxu, lb, ub, mu)
    assert (oxl != xl).sum() >= len(near)
    np.testing.assert_array_equal(dxl.cpu().numpy(), oxl); np.testing.assert_array_equal(dxu.cpu().numpy(), oxu)
    dzl, dzu = t(zl), t(zu)
    K.reset_bound_dual(dzl, dzu, t(x), t(oxl), t(oxu), mu, 1e10)
    ozl, ozu = zl.copy(), zu.copy()
    ok.reset_bound_dual(ozl, x, oxl, mu, 1e10); ok.reset_bound_dual(ozu, oxu, x, mu, 1e10)
    np.testing.assert_array_equal(dzl.cpu().numpy(), ozl); np.testing.assert_array_equal(dzu.cpu().numpy(), ozu)
    assert (ozl[np.isinf(xl)] == 0).all() and (ozu[np.isinf(xu)] == 0).all()
    K.close()


@pytest.mark.gpu
def test_batched_reductions_equal_the_unbatched_ones(ctx):
    """`mnk_ipm_batch_begin/_end`: the calls of a batch only enqueue; one synchronization; the deferred results are bit-identical
    to the same calls issued one by one (regular and restoration phase); an over-full or nested batch is an error."""
    import madnlp_jl_amd as mj
    rng = np.random.default_rng(99)
    ntot, nlb, nub, m = 5000, 2100, 1900, 1500
    d = _data(rng, ntot, nlb, nub, m)
    g = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in d.items() if k not in ("lb", "ub")}
    pp, zp = torch.rand(m, dtype=torch.float64, device="cuda") + 0.1, torch.rand(m, dtype=torch.float64, device="cuda") + 0.1
    K = mj.IPMDeviceKernels(ntot, d["lb"], d["ub"], ctx=ctx)
    calls = [lambda: K.get_sd_sc(g["y"], g["zl"], g["zu"], 100.0), lambda: K.get_norms(g["c"]),
             lambda: K.get_inf_du(g["f"], g["zl"], g["zu"], g["jacl"], 1.7),
             lambda: K.get_inf_compl(g["x"], g["xl"], g["xu"], g["zl"], g["zu"], 0.01, 2.3),
             lambda: K.get_varphi(3.5, g["x"], g["xl"], g["xu"], 0.01),
             lambda: K.get_varphi_d(g["f"], g["x"], g["xl"], g["xu"], g["dx"], 0.01),
             lambda: K.get_alpha_max(g["x"], g["xl"], g["xu"], g["dx"], 0.99),
             lambda: K.get_alpha_z(g["zl"], g["zu"], g["dzl"], g["dzu"], 0.99),
             lambda: K.get_rel_search_norm(g["x"], g["dx"]),
             lambda: K.get_inf_compl_R(g["x"], g["xl"], g["xu"], g["zl"], g["zu"], pp, zp, pp, zp, 0.01, 2.3),
             lambda: K.get_theta_R(g["c"], pp, zp)]
    one_by_one = [f() for f in calls]
    with K.batch():
        bufs = [f() for f in calls]
    for a, b in zip(one_by_one, bufs):
        a = a if isinstance(a, tuple) else (a,)
        assert tuple(b) == a
    assert K.get_norms(g["c"]) == one_by_one[1]          # plain calls work again after the batch
    with pytest.raises(mj.HipError):
        with K.batch():
            for _ in range(20):
                K.get_norms(g["c"])                      # 2 slots each: the 17th call does not fit
    assert K.get_norms(g["c"]) == one_by_one[1]          # ... and the handle is usable afterwards
    K.close()
