"""Device-resident IPM regular phase (`madnlp_jl_amd.ipm_dev.DeviceMadNLPSolver`) against the host mirror
on the same convex QP: same status, same iteration / factorization / back-solve counts, same iterate history."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ["case30", "case118"])
def test_device_resident_ipm_reproduces_the_host_mirror(case):
    import madnlp_jl_amd as mj
    from madnlp_jl_amd.ipm import IPMOptions, MadNLPSolver
    from madnlp_jl_amd.ipm_dev import DeviceMadNLPSolver
    from madnlp_jl_amd.problems import SparseQPModel
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    st = torch.cuda.Stream()
    torch.cuda.set_stream(st)  # torch ops and the library share ONE stream
    ctx = mj.HipContext(0, stream=st.cuda_stream)
    try:
        nlp = SparseQPModel(case)

        def factory(info):
            return mj.SparseCondensedKKTSystem(info["n"], info["m"], nlp.jac_I, nlp.jac_J, nlp.hess_I, nlp.hess_J,
                                               info["ind_ineq"], info["ind_lb"], info["ind_ub"], ctx=ctx,
                                               opt_linear_solver=mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN),
                                               device_kkt_ops=True)

        def options():
            o = IPMOptions(tol=1e-6)
            o.relax_equality, o.dual_initialization = True, "zero"
            return o

        sh = MadNLPSolver(nlp, factory, options(), sparse=True)
        sh.solve()
        sd = DeviceMadNLPSolver(nlp, factory, options())
        sd.solve()
        assert sd.status == sh.status == "SOLVE_SUCCEEDED"
        assert (sd.cnt.k, sd.cnt.factorization_cnt, sd.cnt.backsolve_cnt) == (sh.cnt.k, sh.cnt.factorization_cnt, sh.cnt.backsolve_cnt)
        x, y, zl, zu = sd.host_state()
        np.testing.assert_allclose(x, sh.x, rtol=0, atol=1e-7 * max(1.0, np.abs(sh.x).max()))
        np.testing.assert_allclose(y, sh.y, rtol=0, atol=1e-6 * max(1.0, np.abs(sh.y).max()))
        assert abs(sd.obj_val - sh.obj_val) <= 1e-8 * max(1.0, abs(sh.obj_val))
        for a, b in zip(sd.history, sh.history):
            assert a.k == b.k
            for fld in ("inf_pr", "inf_du", "inf_compl", "mu"):
                va, vb = getattr(a, fld), getattr(b, fld)
                assert abs(va - vb) <= 1e-5 * abs(vb) + 1e-9, (a.k, fld, va, vb)  # rtol 1e-5 above 1e-9
        sh.kkt.close(); sd.kkt.close(); sd.K.close()
    finally:
        torch.cuda.set_stream(torch.cuda.default_stream())
        ctx.close()


@pytest.mark.parametrize("n,m,n_eq", [(50, 10, 0), (20, 15, 2), (200, 60, 8)])
def test_device_resident_ipm_on_the_dense_condensed_system(n, m, n_eq):
    """The same driver on `DenseCondensedKKTSystem` (reference DenseDummyQP, equalities allowed): Hessian / Jacobian of the
    handle are loaded from device copies, callbacks are matrix-vector products on the device; same status, counts and
    residual history as the host mirror."""
    import madnlp_jl_amd as mj
    from madnlp_jl_amd.ipm import IPMOptions, MadNLPSolver
    from madnlp_jl_amd.ipm_dev import DeviceMadNLPSolver
    from madnlp_jl_amd.problems import DenseQPModel
    st = torch.cuda.Stream()
    torch.cuda.set_stream(st)
    ctx = mj.HipContext(0, stream=st.cuda_stream)
    try:
        nlp = DenseQPModel(n, m, n_eq)

        def factory(info):
            return mj.DenseCondensedKKTSystem(info["n"], info["m"], info["ind_ineq"], info["ind_eq"], info["ind_lb"], info["ind_ub"],
                                              ctx=ctx, opt_linear_solver=mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN),
                                              device_kkt_ops=True)

        sh = MadNLPSolver(nlp, factory, IPMOptions(tol=1e-8), sparse=False)
        sh.solve()
        sd = DeviceMadNLPSolver(nlp, factory, IPMOptions(tol=1e-8), sparse=False)
        sd.solve()
        assert sd.status == sh.status == "SOLVE_SUCCEEDED"
        assert (sd.cnt.k, sd.cnt.factorization_cnt) == (sh.cnt.k, sh.cnt.factorization_cnt)
        x, y, zl, zu = sd.host_state()
        np.testing.assert_allclose(x, sh.x, rtol=0, atol=1e-7 * max(1.0, np.abs(sh.x).max()))
        np.testing.assert_allclose(y, sh.y, rtol=0, atol=1e-6 * max(1.0, np.abs(sh.y).max()))
        for a, b in zip(sd.history, sh.history):
            assert a.k == b.k
            for fld in ("inf_pr", "inf_du", "inf_compl", "mu"):
                va, vb = getattr(a, fld), getattr(b, fld)
                assert abs(va - vb) <= 1e-5 * abs(vb) + 1e-9, (a.k, fld, va, vb)
        sh.kkt.close(); sd.kkt.close(); sd.K.close()
    finally:
        torch.cuda.set_stream(torch.cuda.default_stream())
        ctx.close()
