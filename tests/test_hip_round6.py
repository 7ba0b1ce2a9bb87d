"""Round 6 (-m gpu): the speculative first correction of inertia_correction! -- same accepted perturbations, same factors, same
trajectory as the sequential loop."""
import numpy as np
import pytest
import torch

import madnlp_jl_amd as mj
from madnlp_jl_amd.problems import ACOPFModel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    c = mj.HipContext(0)
    yield c
    c.close()


def test_speculative_inertia_correction_reproduces_the_sequential_loop(ctx):
    """`DeviceMadNLPSolver.speculate`: after an iteration that needed a correction, the unperturbed matrix and the matrix with the
    next perturbation (known in advance: reference src/IPM/solver.jl:633-636, and del_c depends on mu alone for a KKT system whose
    should_regularize_dual is `true`, src/KKT/Sparse/condensed.jl:141) are factorized as one batch of two on two solvers of the
    same KKT handle; the speculative factor is used only if the unperturbed matrix is rejected.  On the polar AC-OPF NLP of
    case1354pegase (17 of 20 iterations take a correction): per iteration the same del_w, the same residuals to the last bit
    (a member of a merged launch produces the bits of a lone launch: tests/test_hip_round5.py), the same counts -- speculative
    factorizations that were not needed are counted apart -- and the handle's diagonals return to their bits after a wasted one."""
    from madnlp_jl_amd.ipm import IPMOptions
    from madnlp_jl_amd.ipm_dev import DeviceMadNLPSolver
    nlp = ACOPFModel("case1354pegase")

    def factory(info):
        return mj.SparseCondensedKKTSystem(info["n"], info["m"], nlp.jac_I, nlp.jac_J, nlp.hess_I, nlp.hess_J, info["ind_ineq"],
                                           info["ind_lb"], info["ind_ub"], ctx=ctx,
                                           opt_linear_solver=mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN), device_kkt_ops=True)
    runs = []
    for spec in (False, True):
        o = IPMOptions(tol=1e-6)
        o.relax_equality, o.dual_initialization = True, "zero"
        s = DeviceMadNLPSolver(nlp, factory, o)
        s.speculate = spec
        s.solve()
        runs.append((s.status, s.cnt.k, s.cnt.factorization_cnt, s.cnt.backsolve_cnt, s.obj_val,
                     [(h.k, h.del_w, h.inf_pr, h.inf_du, h.mu) for h in s.history], s.speculative_factorizations, s.speculative_wasted,
                     s.host_state()[0].copy()))
        s.cb.close(); s.K.close(); s.kkt.close()
    seq, spc = runs
    assert seq[0] == spc[0] == "SOLVE_SUCCEEDED"
    assert seq[1:4] == spc[1:4], (seq[1:4], spc[1:4])           # iterations, factorizations (used trials), back-solves
    assert seq[5] == spc[5]                                       # del_w and the residual history, bit for bit
    assert seq[4] == spc[4] and np.array_equal(seq[8], spc[8])
    assert seq[6] == 0 and spc[6] >= 10 and 0 <= spc[7] <= spc[6] // 3


def test_a_solver_of_smaller_order_factors_the_leading_block_of_the_kkt_matrix(ctx):
    """`mnk_ls_factorize_sc_async` with a solver of order m < n (`SparseCondensedKKTSystem.probe_solver`): the factor, the inertia
    and a solve are those of the leading principal block K[:m, :m] -- checked against LAPACK on that block (oracle/lapack_cpu.py)
    for a positive definite system, and the verdict "not positive definite" for an indefinite one whose full factorization is
    rejected in the same columns."""
    from madnlp_jl_amd.problems import opf_shaped
    from oracle.lapack_cpu import BUNCHKAUFMAN, LapackCPUSolver
    from tests.test_hip_c5 import _hip_sc
    for indefinite in (False, True):
        P = opf_shaped("case1354pegase", indefinite=indefinite, sigma_s_decades=2.0, du=1e-8)
        k = _hip_sc(P, ctx, mj.BUNCHKAUFMAN)
        k.compress_jacobian(); k.compress_hessian(); k.set_aug_diagonal(); k.build_kkt()
        m = 3328
        ps = k.probe_solver(m)
        ps.factorize()
        Kd = k.aug_com.to_dense()[:m, :m]
        Kf = Kd + np.tril(Kd, -1).T
        ref = LapackCPUSolver(np.asfortranarray(np.tril(Kf)), BUNCHKAUFMAN).factorize()
        ine_ref = tuple(int(v) for v in ref.inertia())
        ine = ps.inertia()
        if not indefinite:
            assert ine == ine_ref == (m, 0, 0)
            b = np.random.default_rng(3).standard_normal(m)
            x = ps.solve_linear_system(b.copy())
            assert np.abs(Kf @ x - b).max() / (np.abs(Kf).sum(1).max() * np.abs(x).max() + np.abs(b).max()) <= 1e-13
        else:
            assert ine_ref[2] >= 1 and ine != (m, 0, 0) and sum(ine) == m       # rejected (early: the counts behind the stop are "negative")
            k.linear_solver.factorize()
            full = k.linear_solver.inertia()
            assert not k.is_inertia_correct(*full)
            assert ps.get_stat("early_reject_col") == k.linear_solver.get_stat("early_reject_col") < m
        k.close()


def test_leading_block_probes_leave_the_interior_point_run_as_it_is(ctx):
    """The library's leading-block probe (`mnk_ls_set_option(ls, "probe", .)`, default on while early rejection is armed;
    `DeviceMadNLPSolver.probe` switches it): after a rejection that stopped in the first half of the columns, a matrix that follows an
    accepted one is first probed through its leading principal block.  On the AC-OPF NLP of case1354pegase: same status, iterations,
    perturbation sequence and optimum with and without the probes (a probe's verdict is the full factorization's in exact arithmetic;
    both runs are compared to 1e-9 -- the leading block is summed in another chunk order than the same columns of the full matrix),
    most rejections are taken by probes, and the run is not slower."""
    from madnlp_jl_amd.ipm import IPMOptions
    from madnlp_jl_amd.ipm_dev import DeviceMadNLPSolver
    import time
    nlp = ACOPFModel("case1354pegase")

    def factory(info):
        return mj.SparseCondensedKKTSystem(info["n"], info["m"], nlp.jac_I, nlp.jac_J, nlp.hess_I, nlp.hess_J, info["ind_ineq"],
                                           info["ind_lb"], info["ind_ub"], ctx=ctx,
                                           opt_linear_solver=mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN), device_kkt_ops=True)
    runs = {}
    for probe in (False, True, False, True):
        o = IPMOptions(tol=1e-6)
        o.relax_equality, o.dual_initialization = True, "zero"
        s = DeviceMadNLPSolver(nlp, factory, o)
        s.probe = probe
        s.initialize(); s._upload(); torch.cuda.synchronize()
        t0 = time.perf_counter(); s.solve(); torch.cuda.synchronize(); wall = time.perf_counter() - t0
        runs[probe] = (s.status, s.cnt.k, s.cnt.factorization_cnt, s.obj_val, [h.del_w for h in s.history], s.probe_hits, s.probe_misses, wall)
        s.cb.close(); s.K.close(); s.kkt.close()
    off, on = runs[False], runs[True]
    assert off[0] == on[0] == "SOLVE_SUCCEEDED" and off[1] == on[1]
    assert abs(off[2] - on[2]) <= 2 and off[4] == on[4]
    assert abs(off[3] - on[3]) <= 1e-9 * abs(off[3])
    assert off[5] == 0 and on[5] >= 8 and on[6] <= on[5] // 2
    assert on[7] <= 1.02 * off[7], (on[7], off[7])


def test_schur_assembly_rejects_what_the_reference_rejects_and_shards_over_ranks(ctx):
    """`mnk_schur_set_structure`: the layout errors of the reference's `_build_schur_symbolic` (schur.jl:140-236, test/schur_test.jl:176-236)
    come back as errors of the call; and scenarios sharded over two stages (`local_scen`, `own_design` on one of them: the partition
    of `schur.shard`) assemble the blocks of the single-stage assembly bit for bit, their S0 parts adding up to its S0."""
    from madnlp_jl_amd.problems import random_twostage_qp
    from madnlp_jl_amd.schur import SchurDenseStage
    ns, nv, nd, nc = 2, 1, 1, 1
    n, m = ns * nv + nd, ns * nc
    mk = lambda nsl=ns, blk=2: SchurDenseStage([np.eye(blk)] * nsl, [np.zeros((nd, blk))] * nsl, np.eye(nd), nd, blk, ctx=ctx)  # noqa: E731
    hI, hJ, jI, jJ = [0, 1, 2], [0, 1, 2], [0, 0, 1, 1], [0, 2, 1, 2]
    st = mk()
    st.set_structure(n, m, nv, nc, hI, hJ, jI, jJ, [], [0, 1])                      # fine: both rows equalities
    with pytest.raises(ValueError, match="couples two scenarios"):
        st.set_structure(n, m, nv, nc, [0, 1, 2, 1], [0, 1, 2, 0], jI, jJ, [], [0, 1])
    with pytest.raises(ValueError, match="another scenario"):
        st.set_structure(n, m, nv, nc, hI, hJ, [0, 0, 0, 1, 1], [0, 1, 2, 1, 2], [], [0, 1])
    with pytest.raises(ValueError, match="different numbers"):
        st.set_structure(n, m, nv, nc, hI, hJ, jI, jJ, [1], [0])
    with pytest.raises(ValueError, match="do not match"):
        st.set_structure(n + 1, m, nv, nc, hI, hJ, jI, jJ, [], [0, 1])
    st.close()
    # sharding: 4 scenarios on one stage vs 2 + 2 on two
    nlp = random_twostage_qp(ns=4, nv=12, nd=5, nc=6, nc_eq=2, seed=7)
    rng = np.random.default_rng(0)
    n, m, blk = nlp.n, nlp.m, nlp.nv + 2
    ind_eq = np.array([k * nlp.nc + i for k in range(nlp.ns) for i in range(2)])
    ind_ineq = np.array([k * nlp.nc + i for k in range(nlp.ns) for i in range(2, nlp.nc)])
    hess = rng.uniform(0.5, 2.0, len(nlp.hess_I)); jac = nlp.jvals.copy()
    pr = rng.uniform(0.1, 1.0, n + len(ind_ineq)); du = -rng.uniform(1e-3, 1e-2, m)
    args = (n, m, nlp.nv, nlp.nc, nlp.hess_I, nlp.hess_J, nlp.jac_I, nlp.jac_J, ind_ineq, ind_eq)
    one = SchurDenseStage([np.eye(blk)] * 4, [np.zeros((nlp.nd, blk))] * 4, np.eye(nlp.nd), nlp.nd, blk, ctx=ctx)
    one.set_structure(*args)
    one.assemble(hess, jac, pr, du)
    parts, s0 = {}, np.zeros((nlp.nd, nlp.nd))
    for rank in (0, 1):
        loc = [k for k in range(4) if k % 2 == rank]
        sh = SchurDenseStage([np.eye(blk)] * 2, [np.zeros((nlp.nd, blk))] * 2, np.eye(nlp.nd), nlp.nd, blk, ctx=ctx)
        sh.set_structure(*args, ns_global=4, local_scen=loc, own_design=rank == 0)
        sh.assemble(hess, jac, pr, du)
        for i, k in enumerate(loc):
            parts[k] = sh.get_block(i)
        s0 += sh.get_s0()
        sh.close()
    for k in range(4):
        A, Cd = one.get_block(k)
        assert np.array_equal(A, parts[k][0]) and np.array_equal(Cd, parts[k][1]), k
        # (both triangles are assembled; (j1 D) j2 and (j2 D) j1 round differently, as in the oracle's scatter)
        assert np.abs(A - A.T).max() <= 4e-16 * np.abs(A).max() and np.abs(A).max() > 0
    S0 = one.get_s0()
    assert np.abs(s0 - S0).max() <= 1e-14 * np.abs(S0).max()
    one.close()
