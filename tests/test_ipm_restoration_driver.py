"""Restoration phases of the IPM mirror (`restore!` reference src/IPM/solver.jl:300-411, `robust!` :413-545,
`filter_line_search_RR!` src/IPM/line_search.jl:128-222) -- the callers of the KKT hot path when the regular phase fails.

Pin: the reference's own `infeasible` test problem (lib/MadNLPTests/src/MadNLPTests.jl:120-136) must end with
INFEASIBLE_PROBLEM_DETECTED; everything else here checks this mirror against itself (oracle back-end vs HIP back-end, host
vectors vs device-resident vectors) and against closed-form optima -- the reference cannot be run in this image, so the
restoration TRAJECTORIES are not pinned to it."""
import numpy as np
import pytest

from madnlp_jl_amd.ipm import IPMOptions, MadNLPSolver
from madnlp_jl_amd.problems import CubicDiskModel, InfeasibleModel, WachterBieglerModel

from test_ipm_oracle import run


def phases(s):
    return "".join(h.phase or "." for h in s.history)


@pytest.mark.parametrize("kind", ["dense", "dense_condensed", "sparse_condensed"])
def test_infeasible_problem_is_detected_like_the_reference_requires(kind):
    s = run(kind, InfeasibleModel(), tol=1e-8 if kind != "sparse_condensed" else 1e-6)
    assert s.status == "INFEASIBLE_PROBLEM_DETECTED", s.status
    assert "R" in phases(s)                                   # reached through robust!
    RR = s.RR
    assert max(RR.inf_pr_R, RR.inf_du_R, RR.inf_compl_R) <= s.opt.tol
    assert abs(s.x[0] - 1.0) < 1e-6 and abs(s.c[0] - 1.0) < 1e-5     # the least-infeasible point x = 1 (c = x - 0)
    # the slack decomposition of the restoration problem: c = pp - nn at the solution
    assert abs(s.c[0] - RR.pp[0] + RR.nn[0]) <= s.opt.tol


@pytest.mark.parametrize("kind,r,x0,expect", [
    ("dense", 2.636, [-4.9627, -2.877], "r"),                 # soft restoration (restore!) accepted, back to regular
    ("dense_condensed", 2.636, [-4.9627, -2.877], "r"),
    ("sparse_condensed", 2.636, [-4.9627, -2.877], "R"),      # robust! for a few iterations, back to regular
    ("dense", 1.0178, [-1.7068, 1.069], "R"),
    ("dense_condensed", 1.0178, [-1.7068, 1.069], "R"),
])
def test_runs_that_leave_the_regular_phase_come_back_and_converge(kind, r, x0, expect):
    nlp = CubicDiskModel(r, x0)
    s = run(kind, nlp, tol=1e-8 if kind != "sparse_condensed" else 1e-6, max_iter=300)
    assert s.status == "SOLVE_SUCCEEDED", (s.status, phases(s))
    ph = phases(s)
    assert expect in ph and ph[-1] == ".", ph                 # the phase was entered and the run ended in the regular phase
    np.testing.assert_allclose(s.x[:2], nlp.solution(), atol=10 * np.sqrt(s.opt.tol) * 1e-2)
    assert max(s.inf_pr, s.inf_du, s.inf_compl_v) <= s.opt.tol
    # iteration counter keeps counting through the phases
    ks = [h.k for h in s.history]
    assert ks == sorted(ks) and len(set(ks)) == len(ks)


@pytest.mark.parametrize("kind", ["dense", "dense_condensed"])
def test_waechter_biegler_start_walks_robust_regular_robust(kind):
    """State-machine regression (this mirror's own behaviour, not a reference-held answer): from (-2, 3, 1) the regular phase
    gives up, robust! brings the iterate back into the filter, the regular phase fails again, and the second robust! run ends
    at (-1, 0, 0) -- a stationary point of the l1 infeasibility under the bounds (c = (0, -3/2): moving x1 up trades
    |c2| for twice as much |c1|) -- with INFEASIBLE_PROBLEM_DETECTED."""
    s = run(kind, WachterBieglerModel(), tol=1e-8, max_iter=300)
    ph = phases(s)
    assert s.status == "INFEASIBLE_PROBLEM_DETECTED", (s.status, ph)
    first, last = ph.index("R"), ph.rindex(".")
    assert first < last < len(ph) - 1 and ph.endswith("R"), ph        # R ... . ... R
    np.testing.assert_allclose(s.x[:3], [-1.0, 0.0, 0.0], atol=1e-6)
    np.testing.assert_allclose(s.c, [0.0, -1.5], atol=1e-6)


def test_one_sided_constraint_with_infinite_lower_bound_has_a_finite_rhs():
    """`rhs .= (lcon .== ucon) .* lcon` (reference src/IPM/solver.jl:38): Julia's `false * -Inf` is -0.0, not NaN."""
    nlp = CubicDiskModel(1.5, [0.5, 0.5])
    s = run("dense", nlp, tol=1e-8)
    assert np.isfinite(s.rhs).all() and s.status == "SOLVE_SUCCEEDED"


# --------------------------------------------------------------------------- HIP back-end (GPU)
torch = pytest.importorskip("torch")


class Forced:
    """Test hook: the regular phase reports a line-search failure at iteration 2 (-> restore!) and an inertia-correction
    failure at iteration 5 (-> robust!), once each, so that a convex QP walks through both restoration phases."""
    _f1 = _f2 = False

    def filter_line_search(self):
        st = super().filter_line_search()
        if self.cnt.k == 2 and not self._f1 and st == "LINESEARCH_SUCCEEDED":
            self._f1 = True
            self.cnt.k += 1
            return "RESTORE"
        return st

    def inertia_correction(self):
        ok = super().inertia_correction()
        if self.status == "REGULAR" and self.cnt.k >= 5 and not self._f2:
            self._f2 = True
            return False
        return ok


def _hip_factory(mj, nlp, ctx, kind="sparse_condensed", device_kkt_ops=True):
    def factory(info):
        opt = mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN)
        if kind == "sparse_condensed":
            return mj.SparseCondensedKKTSystem(info["n"], info["m"], nlp.jac_I, nlp.jac_J, nlp.hess_I, nlp.hess_J, info["ind_ineq"],
                                               info["ind_lb"], info["ind_ub"], ctx=ctx, opt_linear_solver=opt,
                                               device_kkt_ops=device_kkt_ops)
        if kind == "dense_condensed":
            return mj.DenseCondensedKKTSystem(info["n"], info["m"], info["ind_ineq"], info["ind_eq"], info["ind_lb"], info["ind_ub"],
                                              ctx=ctx, opt_linear_solver=opt)
        return mj.DenseKKTSystem(info["n"], info["m"], info["ind_ineq"], info["ind_lb"], info["ind_ub"], ctx=ctx, opt_linear_solver=opt)
    return factory


def _options(kind="sparse_condensed", **kw):
    o = IPMOptions(tol=1e-6 if kind == "sparse_condensed" else 1e-8, **kw)
    if kind == "sparse_condensed":
        o.relax_equality, o.dual_initialization = True, "zero"
    return o


@pytest.fixture()
def one_stream_ctx():
    import madnlp_jl_amd as mj
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    st = torch.cuda.Stream()
    torch.cuda.set_stream(st)  # torch ops and the library share ONE stream
    ctx = mj.HipContext(0, stream=st.cuda_stream)
    yield ctx
    torch.cuda.set_stream(torch.cuda.default_stream())
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["dense", "dense_condensed", "sparse_condensed"])
def test_hip_backend_through_the_restoration_phases(one_stream_ctx, kind):
    """The host mirror on the HIP KKT systems: the `infeasible` problem takes the oracle back-end's path step for step; the
    nonconvex runs (sensitive to rounding: LDL^T here, LAPACK Bunch-Kaufman there) must leave the regular phase, come back
    and converge to the closed-form optimum, like the oracle back-end's runs."""
    import madnlp_jl_amd as mj

    def hip_run(nlp):
        s = MadNLPSolver(nlp, _hip_factory(mj, nlp, one_stream_ctx, kind, device_kkt_ops=False), _options(kind, max_iter=300),
                         sparse=kind == "sparse_condensed")
        s.solve()
        return s

    nlp = InfeasibleModel()
    so, sh = run(kind, nlp, tol=1e-8 if kind != "sparse_condensed" else 1e-6), hip_run(nlp)
    assert sh.status == so.status == "INFEASIBLE_PROBLEM_DETECTED"
    assert phases(sh) == phases(so)
    for a, b in zip(sh.history, so.history):
        for fld in ("inf_pr", "inf_du", "inf_compl", "mu"):
            assert abs(getattr(a, fld) - getattr(b, fld)) <= 1e-6 * abs(getattr(b, fld)) + 1e-9, (a.k, fld)
    sh.kkt.close()
    for r, x0 in ((2.636, [-4.9627, -2.877]), (1.0178, [-1.7068, 1.069])):
        nlp = CubicDiskModel(r, x0)
        sh = hip_run(nlp)
        assert sh.status == "SOLVE_SUCCEEDED", (sh.status, phases(sh))
        np.testing.assert_allclose(sh.x[:2], nlp.solution(), atol=1e-5)
        sh.kkt.close()
    # at least one of the two runs went through a restoration phase on this back-end too
    assert any(c in phases(sh) for c in "rR") or kind == "sparse_condensed"


def _compare(sd, sh):
    assert sd.status == sh.status, (sd.status, sh.status)
    assert phases(sd) == phases(sh), (phases(sd), phases(sh))
    assert (sd.cnt.k, sd.cnt.factorization_cnt, sd.cnt.backsolve_cnt) == (sh.cnt.k, sh.cnt.factorization_cnt, sh.cnt.backsolve_cnt)
    for a, b in zip(sd.history, sh.history):
        assert a.k == b.k
        for fld in ("inf_pr", "inf_du", "inf_compl", "mu"):
            va, vb = getattr(a, fld), getattr(b, fld)
            assert abs(va - vb) <= 1e-5 * abs(vb) + 1e-9, (a.k, a.phase, fld, va, vb)
    x, y, zl, zu = sd.host_state()
    np.testing.assert_allclose(x, sh.x, rtol=0, atol=1e-7 * max(1.0, np.abs(sh.x).max()))
    np.testing.assert_allclose(y, sh.y, rtol=0, atol=1e-6 * max(1.0, np.abs(sh.y).max()))


@pytest.mark.gpu
def test_device_resident_driver_detects_the_infeasible_problem(one_stream_ctx):
    """robust! entirely on device-resident vectors (`mnk_ipm_*_R`, `mnk_sc_set_aug_RR`): same status and history as the
    host mirror on the reference's `infeasible` problem."""
    import madnlp_jl_amd as mj
    from madnlp_jl_amd.ipm_dev import DeviceMadNLPSolver
    nlp = InfeasibleModel()
    sh = MadNLPSolver(nlp, _hip_factory(mj, nlp, one_stream_ctx), _options(), sparse=True)
    sh.solve()
    sd = DeviceMadNLPSolver(nlp, _hip_factory(mj, nlp, one_stream_ctx), _options())
    sd.solve()
    assert sh.status == "INFEASIBLE_PROBLEM_DETECTED"
    _compare(sd, sh)
    for name in ("pp", "nn", "zp", "zn"):
        np.testing.assert_allclose(getattr(sd.RR, name).cpu().numpy(), getattr(sh.RR, name), rtol=1e-6, atol=1e-9)
    sh.kkt.close(); sd.kkt.close(); sd.K.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["case30", "case118"])
def test_device_resident_driver_through_forced_restore_and_robust(one_stream_ctx, case):
    """A convex QP pushed through restore! (iteration 2) and robust! (iteration 5) by the `Forced` hook: the device-resident
    driver and the host mirror take the same path (phases, counts, residual history) and both converge."""
    import madnlp_jl_amd as mj
    from madnlp_jl_amd.ipm_dev import DeviceMadNLPSolver
    from madnlp_jl_amd.problems import SparseQPModel
    nlp = SparseQPModel(case)

    class FH(Forced, MadNLPSolver):
        pass

    class FD(Forced, DeviceMadNLPSolver):
        pass

    sh = FH(nlp, _hip_factory(mj, nlp, one_stream_ctx), _options(), sparse=True)
    sh.solve()
    sd = FD(nlp, _hip_factory(mj, nlp, one_stream_ctx), _options())
    sd.solve()
    assert sh.status == "SOLVE_SUCCEEDED", (sh.status, phases(sh))
    ph = phases(sh)
    assert "R" in ph and sh._f1 and sh._f2, ph
    _compare(sd, sh)
    sh.kkt.close(); sd.kkt.close(); sd.K.close()
