"""A/B of the two explicit-inverse kernels on the condensed KKT matrix of an AC-OPF interior-point iteration, with right-hand
sides of the kind the IPM produces.  usage: python tools/acopf_inv_ab.py case118 8"""
import os, sys
import numpy as np
import torch
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
Kl = K.astype(np.longdouble)
rng = np.random.default_rng(1)
xt = {"ones": np.ones(N), "random": rng.standard_normal(N), "scaled": rng.standard_normal(N) / np.sqrt(np.abs(np.diag(K)))}
ctx = mj.HipContext(0)
for name, x0 in xt.items():
    b = np.asarray(Kl @ x0.astype(np.longdouble), dtype=np.float64)
    for mfma in (0, 1):
        ls = mj.HipLinearSolver(np.asfortranarray(K), ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm="LDL"))
        ls.set_option("linv_mfma", mfma)
        ls.factorize()
        x = ls.solve_linear_system(b.copy())
        r = np.asarray(b - Kl @ x.astype(np.longdouble), dtype=np.float64)
        # three steps of iterative refinement: how fast does the residual fall?
        hist = [np.abs(r).max() / np.abs(b).max()]
        for _ in range(3):
            x = x + ls.solve_linear_system(r.copy())
            r = np.asarray(b - Kl @ x.astype(np.longdouble), dtype=np.float64)
            hist.append(np.abs(r).max() / np.abs(b).max())
        print(f"{name:7s} linv_mfma={mfma}: relative residual after 0..3 refinement steps " + " ".join(f"{h:.2e}" for h in hist)
              + f"   componentwise backward error {np.max(np.abs(r) / (np.abs(K) @ np.abs(x) + np.abs(b))):.2e}")
        ls.close()
ctx.close()
