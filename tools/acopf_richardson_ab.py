"""Per-iteration Richardson steps / residual ratios of the host IPM driver on the HIP back-end, scalar vs MFMA explicit
inverses.  usage: MNK_OPTIONS=linv_mfma=0|1 python tools/acopf_richardson_ab.py case118"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import madnlp_jl_amd as mj  # noqa: E402
from madnlp_jl_amd.ipm import MadNLPSolver  # noqa: E402
from madnlp_jl_amd.problems import ACOPFModel  # noqa: E402
from madnlp_jl_amd import backsolve  # noqa: E402
from tests.test_acopf import _options  # noqa: E402

case = sys.argv[1]
nlp = ACOPFModel(case)
ctx = mj.HipContext(0)
log = []
orig = backsolve.RichardsonIterator.solve_refine
def wrapped(self, x, b, w):
    ok = orig(self, x, b, w)
    log.append((self.ir, self.residual_ratio))
    return ok
backsolve.RichardsonIterator.solve_refine = wrapped
def factory(info):
    return mj.SparseCondensedKKTSystem(info["n"], info["m"], nlp.jac_I, nlp.jac_J, nlp.hess_I, nlp.hess_J, info["ind_ineq"], info["ind_lb"], info["ind_ub"], ctx=ctx,
                                       opt_linear_solver=mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN), device_kkt_ops=True)
sh = MadNLPSolver(nlp, factory, _options(), sparse=True)
sh.solve()
print(case, "MNK_OPTIONS", os.environ.get("MNK_OPTIONS"), sh.status, (sh.cnt.k, sh.cnt.factorization_cnt, sh.cnt.backsolve_cnt))
print(" ".join(f"{ir}:{rr:.0e}" for ir, rr in log))
