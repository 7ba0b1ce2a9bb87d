"""Restoration-phase kernels on device-resident vectors (`mnk_ipm_*_R`, `mnk_*_set_aug_RR`; SURVEY 8(f).4 third slice)
against the oracle restatement of reference `src/IPM/kernels.jl:72-110,133-158,206-257,390-654,775-786,825-829` and
`src/IPM/restoration.jl:39-76`.  CPU half: the oracle on hand-computed values.  GPU half: max/min-type reductions and the
elementwise pieces bit-exact, sum-type reductions to summation-order rounding."""
import math

import numpy as np
import pytest

from oracle import ipm_kernels as ok

INF = float("inf")


# --------------------------------------------------------------------------- oracle pins (CPU)
def test_oracle_restoration_reductions_on_hand_computed_values():
    p = np.array([1.0, 2.0]); n = np.array([0.5, 0.25]); c = np.array([3.0, -1.0])
    D = np.array([1.0, 0.5, 0.25]); x = np.array([1.0, 3.0, -2.0]); xr = np.array([0.0, 1.0, 2.0])
    # rho (p + n) summed + zeta/2 D^2 (x - x_ref)^2 summed
    assert ok.get_obj_val_R(p, n, D, x, xr, 10.0, 4.0) == 10.0 * 1.5 + 10.0 * 2.25 + 2.0 * (1.0 + 0.25 * 4.0 + 0.0625 * 16.0)
    assert ok.get_theta_R(c, p, n) == abs(3 - 1 + .5) + abs(-1 - 2 + .25)
    assert ok.get_inf_pr_R(c, p, n) == 2.75
    f_R = np.array([1.0, -2.0, 0.5]); zl = np.array([0.5, 0.25, 0.0]); zu = np.array([0.125, 0.0, 2.0]); jl = np.array([0.0, 1.0, -4.0])
    l = np.array([3.0, -5.0]); zp = np.array([1.0, 2.0]); zn = np.array([4.0, 0.5])
    assert ok.get_inf_du_R(f_R, l, zl, zu, jl, zp, zn, 10.0, 2.0) == \
        max(abs(1 - .5 + .125), abs(-2 - .25 + 1), abs(.5 + 2 - 4), abs(10 - 3 - 1), abs(10 + 5 - 2), abs(10 + 3 - 4), abs(10 - 5 - .5)) / 2.0
    lb, ub = np.array([0, 1]), np.array([0, 2])
    xl = np.array([0.0, 1.0, -INF]); xu = np.array([2.0, INF, 5.0]); xx = np.array([1.0, 2.0, 3.0])
    mu = 0.1
    assert ok.get_inf_compl_R(xx[lb], xl[lb], zl[lb], xu[ub], xx[ub], zu[ub], p, zp, n, zn, mu, 4.0) == \
        max(abs(.5 - mu), abs(.25 - mu), abs(.125 - mu), abs(4.0 - mu), abs(1.0 - mu), abs(4.0 - mu), abs(2.0 - mu), abs(.125 - mu)) / 4.0
    dx = np.array([-4.0, 0.0, 8.0]); dpp = np.array([-8.0, 1.0]); dnn = np.array([0.0, -0.125])
    assert ok.get_alpha_max_R(xx, xl, xu, dx, p, dpp, n, dnn, 0.5) == \
        min(1.0, (-1.0 + 0.0) * 0.5 / -4.0, (-3.0 + 5.0) * 0.5 / 8.0, -1.0 * 0.5 / -8.0, -0.25 * 0.5 / -0.125)
    assert ok.get_alpha_max_R(xx, xl, xu, np.zeros(3), p, np.zeros(2), n, np.zeros(2), 0.5) == 1.0
    assert ok.get_alpha_z_R(zl[lb], zu[ub], np.array([-1.0, 1.0]), np.array([0.0, -16.0]), zp, np.array([-4.0, 0.0]), zn,
                            np.array([1.0, -8.0]), 0.5) == min(.25, 2.0 * .5 / 16.0, 1.0 * .5 / 4.0, 0.5 * 0.5 / 8.0)
    assert ok.get_varphi_R(7.0, xx[lb], xl[lb], xu[ub], xx[ub], np.array([1.0, math.e]), np.array([1.0]), mu) == \
        pytest.approx(7.0 - mu * (math.log(2.0) + 1.0), rel=1e-15)
    assert ok.get_varphi_R(7.0, xx[lb], xl[lb], xu[ub], xx[ub], np.array([-1.0]), np.array([1.0]), mu) == -INF
    # get_F: |c|_1 + |f - zl + zu + jacl|_1 + lower complementarity + the reference's upper term |0 * zu - mu|
    assert ok.get_F(c, f_R, zl, zu, jl, xx[lb], xl[lb], zl[lb], xu[ub], xx[ub], zu[ub], mu) == \
        pytest.approx(4.0 + (.625 + 1.25 + 1.5) + (abs(.5 - mu) + abs(.25 - mu)) + 2 * mu, rel=1e-15)
    assert ok.get_F(c, f_R, zl, zu, jl, np.array([-1.0]), np.array([0.0]), np.array([1.0]), xu[ub], xx[ub], zu[ub], mu) == INF
    assert ok.get_varphi_d_R(f_R, xx, xl, xu, dx, p, n, dpp, dnn, mu, 10.0) == \
        pytest.approx((1 - mu / 1 + mu / 1) * -4.0 + (0.5 - 0.0 + mu / 2.0) * 8.0 + (10 - mu / 1) * -8.0 + (10 - mu / 2) * 1.0
                      + (10 - mu / .25) * -0.125, rel=1e-15)


def test_oracle_restoration_elementwise_on_hand_computed_values():
    c = np.array([3.0, -1.0, 0.0]); mu, rho = 0.5, 2.0
    nn = ok.populate_RR_nn(c, mu, rho)
    t = (mu - rho * c) / (2 * rho)
    np.testing.assert_array_equal(nn, t + np.sqrt(t * t + mu * c / (2 * rho)))
    # nn solves rho nn^2 - (mu - rho c) nn - mu c / 2 = 0 (the positive root): pp = c + nn > 0 as well
    np.testing.assert_allclose(rho * nn ** 2 - (mu - rho * c) * nn - mu * c / 2, 0.0, atol=1e-14)
    assert (nn > 0).all() and (c + nn > 0).all()
    x = np.array([4.0, -0.5, 0.0]); zl_r = np.array([5.0, 1.0]); zu_r = np.array([0.5])
    x_ref, D_R, mu_R, nn2, pp, zp, zn, zl2, zu2 = ok.initialize_robust_restorer(x, c, zl_r, zu_r, 0.1, rho)
    np.testing.assert_array_equal(D_R, [0.25, 1.0, 1.0]); assert mu_R == 3.0
    np.testing.assert_array_equal(zl2, [2.0, 1.0]); np.testing.assert_array_equal(zu2, [0.5])
    np.testing.assert_array_equal(pp, c + nn2); np.testing.assert_array_equal(zp, 3.0 / pp); np.testing.assert_array_equal(zn, 3.0 / nn2)
    np.testing.assert_array_equal(ok.set_f_RR(D_R, np.array([5.0, 0.5, 1.0]), x_ref, 2.0), [2.0 * .0625 * 1.0, 2.0 * 1.0, 2.0 * 1.0])
    lb, ub = np.array([0, 2]), np.array([1])
    reg, du, ld, ud, ll, ul, pr = ok.set_aug_RR(np.array([1.0, 3.0]), np.array([0.0, 1.0]), np.array([0.5, 0.25]), np.array([4.0]),
                                                np.array([2.0]), np.array([0.125]), D_R, np.array([1.0, 2.0]), np.array([4.0, 8.0]),
                                                np.array([3.0, 1.0]), np.array([2.0, 0.5]), 2.0, 1e-3, 1e-2, lb, ub)
    np.testing.assert_array_equal(reg, [1e-3 + 2.0 * .0625, 1e-3 + 2.0, 1e-3 + 2.0])
    np.testing.assert_array_equal(du, [-1e-2 - 0.25 - 1.5, -1e-2 - 0.25 - 2.0])
    np.testing.assert_array_equal(ld, [-1.0, -2.0]); np.testing.assert_array_equal(ud, [-2.0])
    np.testing.assert_array_equal(pr, [reg[0] - .5 / -1.0, reg[1] - .125 / -2.0, reg[2] - .25 / -2.0])
    px, py, pzl, pzu = ok.set_aug_rhs_RR(np.array([1.0, 2.0, 3.0]), np.array([.5, .25, 0.0]), np.array([.125, 0.0, 2.0]), np.full(3, .5),
                                         np.array([7.0]), np.array([3.0]), np.array([2.0]), np.array([4.0]), np.array([8.0]), np.array([0.5]),
                                         np.array([1.0]), np.array([0.0]), np.array([.5]), np.array([5.0]), np.array([3.0]), np.array([2.0]), 0.1, 10.0)
    np.testing.assert_array_equal(px, [-1 + .5 - .125 - .5, -2 + .25 - .5, -3 - 2 - .5])
    np.testing.assert_array_equal(py, [-7.0 + 2.0 - 4.0 + (0.1 - 7.0 * 2.0) / 8.0 - (0.1 - 13.0 * 4.0) / 0.5])
    np.testing.assert_array_equal(pzl, [-.5 + .1]); np.testing.assert_array_equal(pzu, [4.0 - .1])
    dpp, dnn, dzp, dzn = ok.finish_aug_solve_RR(np.array([3.0]), np.array([0.5]), np.array([2.0]), np.array([4.0]), np.array([8.0]),
                                                np.array([0.5]), 0.1, 10.0)
    assert dzp[0] == 10 - 3 - .5 - 8 and dzn[0] == 10 + 3 + .5 - .5
    assert dpp[0] == -2.0 + 0.1 / 8.0 - (2.0 / 8.0) * dzp[0] and dnn[0] == -4.0 + 0.1 / 0.5 - (4.0 / 0.5) * dzn[0]
    z = np.array([0.5, 1e12, 1e-30]); ok.reset_bound_dual_1(z, np.array([1.0, 0.5, 2.0]), 0.1, 1e10)
    np.testing.assert_array_equal(z, [0.5, (1e10 * 0.1) / 0.5, (0.1 / 1e10) / 2.0])
    xl, xu = np.array([0.0, -10.0, -INF]), np.array([0.5, INF, 100.0])
    ok.set_initial_bounds(xl, xu, 1e-8)
    np.testing.assert_array_equal(xl, [-1e-8, -10.0 - 1e-7, -INF]); np.testing.assert_array_equal(xu, [0.5 + 1e-8, INF, 100.0 + 1e-6])
    np.testing.assert_array_equal(ok.set_initial_rhs(np.array([1.0, 2.0]), np.array([.5, 0.0]), np.array([0.0, .25])), [-.5, -2.25])
    np.testing.assert_array_equal(ok.set_g_ifr(np.array([1.0, 2.0]), np.array([1.0, 0.0]), np.array([0.0, -INF]), np.array([3.0, 4.0]),
                                               np.array([.5, .5]), 0.1), [1 - .1 / 1 + .1 / 2 + .5, 2 - 0.0 + .1 / 4 + .5])
    got = ok.initialize_variables(np.array([0.0, 5.0, -3.0, 7.0, 0.5]), np.array([1.0, -INF, -INF, 0.0, 0.0]),
                                  np.array([INF, 2.0, INF, 1.0, 1.0]), 1e-2, 1e-2)
    np.testing.assert_array_equal(got, [1.01, 2.0 - 0.02, -3.0, 1.0 - 0.01, 0.5])


# --------------------------------------------------------------------------- HIP vs oracle (GPU)
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ctx():
    import madnlp_jl_amd as mj
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    c = mj.HipContext(0)
    yield c
    c.close()


def _rr_data(rng, ntot, nlb, nub, m):
    from tests.test_ipm_device import _data
    d = _data(rng, ntot, nlb, nub, m)
    d["xl"] = np.where(d["xl"] < -1e299, -np.inf, d["xl"]); d["xu"] = np.where(d["xu"] > 1e299, np.inf, d["xu"])
    d["pp"] = 10.0 ** rng.uniform(-8, 2, m); d["nn"] = 10.0 ** rng.uniform(-8, 2, m)
    d["zp"] = 10.0 ** rng.uniform(-6, 3, m); d["zn"] = 10.0 ** rng.uniform(-6, 3, m)
    for k in ("dpp", "dnn", "dzp", "dzn", "dl"):
        d[k] = rng.standard_normal(m) * 10.0 ** rng.uniform(-3, 2, m)
        d[k][rng.random(m) < 0.1] = 0.0
    d["x_ref"] = d["x"] + rng.standard_normal(ntot) * 1e-2
    d["D_R"] = np.minimum(1.0, 1.0 / np.abs(d["x_ref"]))
    d["f_R"] = rng.standard_normal(ntot)
    return d


SIZES = [(1, 1, 1, 1), (7, 3, 2, 2), (1000, 400, 377, 300), (27838, 20646, 19646, 16646), (300001, 150000, 120000, 100000), (50, 0, 0, 0)]


@pytest.mark.gpu
@pytest.mark.parametrize("ntot,nlb,nub,m", SIZES)
def test_device_restoration_reductions_match_the_oracle(ctx, ntot, nlb, nub, m):
    import madnlp_jl_amd as mj
    rng = np.random.default_rng(1000 + ntot)
    nlb, nub = min(nlb, ntot), min(nub, ntot)
    d = _rr_data(rng, ntot, nlb, nub, m)
    lb, ub = d["lb"], d["ub"]
    g = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in d.items() if k not in ("lb", "ub")}
    K = mj.IPMDeviceKernels(ntot, lb, ub, ctx=ctx)
    x, xl, xu, zl, zu = d["x"], d["xl"], d["xu"], d["zl"], d["zu"]
    mu, tau, sd, sc, rho, zeta = 0.03, 0.97, 1.7, 2.3, 1000.0, 0.2
    # max / min type: bit-exact
    assert K.get_inf_pr_R(g["c"], g["pp"], g["nn"]) == ok.get_inf_pr_R(d["c"], d["pp"], d["nn"])
    assert K.get_inf_du_R(g["f_R"], g["y"], g["zl"], g["zu"], g["jacl"], g["zp"], g["zn"], rho, sd) == \
        ok.get_inf_du_R(d["f_R"], d["y"], zl, zu, d["jacl"], d["zp"], d["zn"], rho, sd)
    assert K.get_inf_compl_R(g["x"], g["xl"], g["xu"], g["zl"], g["zu"], g["pp"], g["zp"], g["nn"], g["zn"], mu, sc) == \
        ok.get_inf_compl_R(x[lb], xl[lb], zl[lb], xu[ub], x[ub], zu[ub], d["pp"], d["zp"], d["nn"], d["zn"], mu, sc)
    assert K.get_alpha_max_R(g["x"], g["xl"], g["xu"], g["dx"], g["pp"], g["dpp"], g["nn"], g["dnn"], tau) == \
        ok.get_alpha_max_R(x, xl, xu, d["dx"], d["pp"], d["dpp"], d["nn"], d["dnn"], tau)
    assert K.get_alpha_z_R(g["zl"], g["zu"], g["dzl"], g["dzu"], g["zp"], g["dzp"], g["zn"], g["dzn"], tau) == \
        ok.get_alpha_z_R(zl[lb], zu[ub], d["dzl"], d["dzu"], d["zp"], d["dzp"], d["zn"], d["dzn"], tau)

    # sum type: summation-order rounding, relative to the sum of magnitudes
    def close(got, want, mag):
        assert abs(got - want) <= 1e-13 * max(mag, 1e-300), (got, want, mag)

    want = ok.get_obj_val_R(d["pp"], d["nn"], d["D_R"], x, d["x_ref"], rho, zeta)
    close(K.get_obj_val_R(g["pp"], g["nn"], g["D_R"], g["x"], g["x_ref"], rho, zeta), want, abs(want))
    want = ok.get_theta_R(d["c"], d["pp"], d["nn"])
    close(K.get_theta_R(g["c"], g["pp"], g["nn"]), want, want)
    logs = mu * (np.abs(np.log(x[lb] - xl[lb])).sum() + np.abs(np.log(xu[ub] - x[ub])).sum() + np.abs(np.log(d["pp"])).sum()
                 + np.abs(np.log(d["nn"])).sum())
    close(K.get_varphi_R(3.5, g["x"], g["xl"], g["xu"], g["pp"], g["nn"], mu),
          ok.get_varphi_R(3.5, x[lb], xl[lb], xu[ub], x[ub], d["pp"], d["nn"], mu), 3.5 + logs)
    want = ok.get_F(d["c"], d["f"], zl, zu, d["jacl"], x[lb], xl[lb], zl[lb], xu[ub], x[ub], zu[ub], mu)
    close(K.get_F(g["c"], g["f"], g["zl"], g["zu"], g["jacl"], g["x"], g["xl"], g["xu"], mu), want, want)
    terms = np.abs((d["f_R"] - mu / (x - xl) + mu / (xu - x)) * d["dx"]).sum() + np.abs((rho - mu / d["pp"]) * d["dpp"]).sum() \
        + np.abs((rho - mu / d["nn"]) * d["dnn"]).sum()
    close(K.get_varphi_d_R(g["f_R"], g["x"], g["xl"], g["xu"], g["dx"], g["pp"], g["nn"], g["dpp"], g["dnn"], mu, rho),
          ok.get_varphi_d_R(d["f_R"], x, xl, xu, d["dx"], d["pp"], d["nn"], d["dpp"], d["dnn"], mu, rho), terms)
    K.close()


@pytest.mark.gpu
def test_device_restoration_reductions_edge_cases(ctx):
    """Negative slack or negative pp / nn: the barrier objective is -Inf (obj - Inf); a violated guard makes get_F +Inf; zero
    steps give alpha = 1; NaN propagates like the reference's max / min."""
    import madnlp_jl_amd as mj
    K = mj.IPMDeviceKernels(3, np.array([0, 1]), np.array([2]), ctx=ctx)
    t = lambda a: torch.tensor(a, dtype=torch.float64, device="cuda")  # noqa: E731
    x, xl, xu = t([1.0, 2.0, 3.0]), t([0.0, 1.0, -INF]), t([INF, INF, 5.0])
    one2 = t([1.0, 1.0])
    assert K.get_varphi_R(1.0, x, xl, xu, t([1.0, -1.0]), one2, 0.1) == -INF
    assert K.get_varphi_R(1.0, t([-1.0, 2.0, 3.0]), xl, xu, one2, one2, 0.1) == -INF
    z3 = t([0.5, 0.5, 0.5])
    assert K.get_F(t([1.0]), z3, z3, z3, z3, t([-1.0, 2.0, 3.0]), xl, xu, 0.1) == INF         # x_lr < xl_r
    assert K.get_F(t([1.0]), z3, t([0.5, -0.5, 0.5]), z3, z3, x, xl, xu, 0.1) == INF           # zl_r < 0
    zero3, zero2 = t([0.0] * 3), t([0.0] * 2)
    assert K.get_alpha_max_R(x, xl, xu, zero3, one2, zero2, one2, zero2, 0.99) == 1.0
    assert K.get_alpha_z_R(z3, z3, zero2, t([0.0]), one2, zero2, one2, zero2, 0.99) == 1.0
    assert math.isnan(K.get_inf_pr_R(t([float("nan"), 0.0]), one2, one2))
    assert math.isnan(K.get_inf_compl_R(x, xl, xu, z3, z3, t([float("nan"), 1.0]), one2, one2, one2, 0.1, 1.0))
    K.close()


@pytest.mark.gpu
@pytest.mark.parametrize("ntot,nlb,nub,m", SIZES[1:4])
def test_device_restoration_elementwise_pieces(ctx, ntot, nlb, nub, m):
    """populate_RR_nn!, initialize_robust_restorer!, set_f_RR!, set_aug_rhs_RR!, finish_aug_solve_RR!, reset_bound_dual!,
    set_initial_bounds!, set_initial_rhs!, set_aug_rhs_ifr!, set_g_ifr!, initialize_variables! on device vectors:
    bit-identical to the oracle (IEEE +, -, *, /, sqrt elementwise, no contraction)."""
    import madnlp_jl_amd as mj
    rng = np.random.default_rng(2000 + ntot)
    d = _rr_data(rng, ntot, nlb, nub, m)
    lb, ub = d["lb"], d["ub"]
    x, xl, xu, zl, zu, c, y = d["x"], d["xl"], d["xu"], d["zl"], d["zu"], d["c"], d["y"]
    mu_R, rho, zeta = max(0.1, np.abs(c).max()), 1000.0, 0.3
    K = mj.IPMDeviceKernels(ntot, lb, ub, ctx=ctx)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()  # noqa: E731
    e = lambda n: torch.empty(n, dtype=torch.float64, device="cuda")  # noqa: E731
    eq = lambda got, want: np.testing.assert_array_equal(got.cpu().numpy(), want)  # noqa: E731

    nn = e(m); K.populate_RR_nn(nn, t(c), mu_R, rho)
    eq(nn, ok.populate_RR_nn(c, mu_R, rho))
    x_ref, D_R, nn, pp, zp, zn = e(ntot), e(ntot), e(m), e(m), e(m), e(m)
    gzl, gzu = t(zl), t(zu)
    rho_s = 5.0                               # small enough that the cap z_r = min(rho, z_r) fires
    K.initialize_robust_restorer(t(x), t(c), mu_R, rho_s, x_ref, D_R, nn, pp, zp, zn, gzl, gzu)
    o = ok.initialize_robust_restorer(x, c, zl[lb], zu[ub], 0.1, rho_s)
    assert o[2] == mu_R
    ozl, ozu = zl.copy(), zu.copy(); ozl[lb] = o[7]; ozu[ub] = o[8]
    for got, want in ((x_ref, o[0]), (D_R, o[1]), (nn, o[3]), (pp, o[4]), (zp, o[5]), (zn, o[6]), (gzl, ozl), (gzu, ozu)):
        eq(got, want)
    assert (ozl != zl).any() or nlb < 10     # the cap at rho_s fired somewhere (multipliers range up to 1e3)
    f_R = e(ntot); K.set_f_RR(f_R, t(d["D_R"]), t(x), t(d["x_ref"]), zeta)
    eq(f_R, ok.set_f_RR(d["D_R"], x, d["x_ref"], zeta))
    px, py, pzl, pzu = e(ntot), e(m), e(nlb), e(nub)
    K.set_aug_rhs_RR(t(d["f_R"]), t(zl), t(zu), t(d["jacl"]), t(c), t(y), t(d["pp"]), t(d["nn"]), t(d["zp"]), t(d["zn"]), t(x), t(xl),
                     t(xu), mu_R, rho, px, py, pzl, pzu)
    for got, want in zip((px, py, pzl, pzu), ok.set_aug_rhs_RR(d["f_R"], zl, zu, d["jacl"], c, y, d["pp"], d["nn"], d["zp"], d["zn"],
                                                               x[lb], xl[lb], zl[lb], xu[ub], x[ub], zu[ub], mu_R, rho)):
        eq(got, want)
    dpp, dnn, dzp, dzn = e(m), e(m), e(m), e(m)
    K.finish_aug_solve_RR(dpp, dnn, dzp, dzn, t(y), t(d["dl"]), t(d["pp"]), t(d["nn"]), t(d["zp"]), t(d["zn"]), mu_R, rho)
    for got, want in zip((dpp, dnn, dzp, dzn), ok.finish_aug_solve_RR(y, d["dl"], d["pp"], d["nn"], d["zp"], d["zn"], mu_R, rho)):
        eq(got, want)
    z = t(d["zp"]); K.reset_bound_dual_1(z, t(d["pp"]), mu_R, 1e4)
    oz = d["zp"].copy(); ok.reset_bound_dual_1(oz, d["pp"], mu_R, 1e4)
    assert (oz != d["zp"]).any(); eq(z, oz)
    gxl, gxu = t(xl), t(xu); K.set_initial_bounds(gxl, gxu, 1e-8)
    oxl, oxu = xl.copy(), xu.copy(); ok.set_initial_bounds(oxl, oxu, 1e-8)
    eq(gxl, oxl); eq(gxu, oxu)
    K.set_initial_bounds(gxl, gxu, 0.0); eq(gxl, oxl)        # tol = 0 keeps the bounds
    K.set_initial_rhs(t(d["f"]), t(zl), t(zu), px, py, pzl, pzu)
    eq(px, ok.set_initial_rhs(d["f"], zl, zu))
    assert not py.any() and not pzl.any() and not pzu.any()
    K.set_aug_rhs_ifr(t(c), px, py, pzl, pzu)
    assert not px.any() and not pzl.any() and not pzu.any(); eq(py, -c)
    gg = e(ntot); K.set_g_ifr(gg, t(d["f"]), t(x), t(xl), t(xu), t(d["jacl"]), 0.01)
    eq(gg, ok.set_g_ifr(d["f"], x, xl, xu, d["jacl"], 0.01))
    x0 = rng.standard_normal(ntot) * 3.0
    gx = t(x0); K.initialize_variables(gx, t(xl), t(xu), 1e-2, 1e-2)
    want = ok.initialize_variables(x0, xl, xu, 1e-2, 1e-2)
    assert (want != x0).any(); eq(gx, want)
    K.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["dense_condensed", "sparse_condensed"])
def test_set_aug_RR_inside_the_kkt_handle(ctx, kind):
    """`set_aug_RR!` + `_set_aug_diagonal!` on the handle's own diagonals: bit-identical to the oracle; `build_kkt!` from
    them equals the oracle's `build_kkt!` from the same diagonals (sparse condensed: bit-exact nonzeros) and the system
    factorizes with the oracle's inertia."""
    import madnlp_jl_amd as mj
    from madnlp_jl_amd.problems import dense_dummy_qp, opf_shaped
    from oracle import dense as odense
    from oracle.lapack_cpu import BUNCHKAUFMAN, LapackCPUSolver
    from tests.test_hip_round2 import _hip_sc, _iterate, _oracle_sc
    rng = np.random.default_rng(77)
    if kind == "sparse_condensed":
        P = opf_shaped("case118", du=0.0)
        ko, kh = _oracle_sc(P), _hip_sc(P, ctx, mj.BUNCHKAUFMAN)
        kh.compress_jacobian(); kh.compress_hessian()
    else:
        P = dense_dummy_qp(96, 40, 8, seed=3)
        ko = odense.DenseCondensedKKTSystem(P.n, P.m, P.ind_ineq, P.ind_eq, P.ind_lb, P.ind_ub, lambda A: LapackCPUSolver(A, BUNCHKAUFMAN))
        kh = mj.DenseCondensedKKTSystem(P.n, P.m, P.ind_ineq, P.ind_eq, P.ind_lb, P.ind_ub, ctx=ctx)
        for k in (ko, kh):
            k.hess[...] = P.hess
            k.jac[...] = P.jac
            k.compress_jacobian(); k.compress_hessian()
        kh._upload()
    npr, m = len(ko.pr_diag), len(ko.du_diag)
    lb, ub = np.asarray(ko.ind_lb), np.asarray(ko.ind_ub)
    x, xl, xu, zl, zu = _iterate(rng, npr, lb, ub)
    D_R = np.minimum(1.0, 1.0 / np.abs(x)); pp, nn = 10.0 ** rng.uniform(-3, 1, m), 10.0 ** rng.uniform(-3, 1, m)
    zp, zn = 10.0 ** rng.uniform(-2, 2, m), 10.0 ** rng.uniform(-2, 2, m)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()  # noqa: E731
    kh.set_aug_RR_device(t(x), t(xl), t(xu), t(zl), t(zu), t(D_R), t(pp), t(zp), t(nn), t(zn), 0.7, 1e-8, 1e-9)
    got = kh.get_diagonals_device()
    want = ok.set_aug_RR(x[lb], xl[lb], zl[lb], xu[ub], x[ub], zu[ub], D_R, pp, zp, nn, zn, 0.7, 1e-8, 1e-9, lb, ub)
    names = ("reg", "du_diag", "l_diag", "u_diag", "l_lower", "u_lower", "pr_diag")
    for name, w in zip(names, want):
        np.testing.assert_array_equal(got[name], w, err_msg=name)
        getattr(ko, name)[:] = w
    kh.build_kkt_device()
    ko.build_kkt()
    if kind == "sparse_condensed":
        np.testing.assert_array_equal(kh.aug_com.nzval, ko.aug_com.nzval)
    kh.linear_solver.factorize()
    ko.linear_solver.factorize()
    assert kh.linear_solver.inertia() == ko.linear_solver.inertia()
    kh.close()
