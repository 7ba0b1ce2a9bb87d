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
