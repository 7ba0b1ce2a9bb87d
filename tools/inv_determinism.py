"""Are factorize + solve bit-reproducible across solver instances?  (A/B of the explicit-inverse kernels.)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import madnlp_jl_amd as mj  # noqa: E402
from madnlp_jl_amd.ipm import IPMOptions, MadNLPSolver  # noqa: E402
from madnlp_jl_amd.problems import ACOPFModel  # noqa: E402
from tests.test_ipm_oracle import oracle_factory  # noqa: E402

case, itn = sys.argv[1], int(sys.argv[2])
nlp = ACOPFModel(case)
o = IPMOptions(tol=1e-6)
o.relax_equality, o.dual_initialization = True, "zero"
o.max_iter = itn
so = MadNLPSolver(nlp, oracle_factory("sparse_condensed", nlp), o, sparse=True)
so.solve()
L = so.kkt.aug_com.to_dense()
K = L + np.tril(L, -1).T
N = K.shape[0]
b = np.random.default_rng(0).standard_normal(N)
ctx = mj.HipContext(0)
for alg in ("LDL", mj.BUNCHKAUFMAN):
    for mfma in (0, 1):
        xs = []
        for rep in range(6):
            ls = mj.HipLinearSolver(np.asfortranarray(K), ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=alg))
            ls.set_option("linv_mfma", mfma)
            ls.set_option("accept_only_pd", 1)
            ls.factorize()
            if rep % 2: ls.factorize()
            xs.append(ls.solve_linear_system(b.copy()))
            algo = ls.get_stat("panel_algo")
            ls.close()
        print(alg, "linv_mfma", mfma, "panel_algo", algo, "distinct solutions:", len({x.tobytes() for x in xs}),
              "max rel diff %.2e" % max(np.abs(x - xs[0]).max() / np.abs(xs[0]).max() for x in xs))
