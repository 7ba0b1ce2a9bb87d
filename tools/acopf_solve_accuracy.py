"""Accuracy of ONE solve with the condensed KKT matrix of an AC-OPF interior-point iteration: HipLinearSolver (static-pivot
LDL^T, inverse-based block substitution) against LAPACK dsytrf/dsytrs and a long-double reference.
usage: python tools/acopf_solve_accuracy.py case30 4"""
import os
import sys

import numpy as np
import scipy.linalg as sl
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
d = np.diag(K)
print(f"{case} iteration {itn}: N={N} diag range {d.min():.3e} .. {d.max():.3e}  cond_2 ~ {np.linalg.cond(K):.3e}  min eig {np.linalg.eigvalsh(K).min():.3e}")
rng = np.random.default_rng(0)
b = rng.standard_normal(N)
Kl, bl = K.astype(np.longdouble), b.astype(np.longdouble)
# long-double reference by refinement of the LAPACK solution
lu = sl.lu_factor(K)
x = sl.lu_solve(lu, b).astype(np.longdouble)
for _ in range(6):
    r = bl - Kl @ x
    x = x + sl.lu_solve(lu, r.astype(np.float64)).astype(np.longdouble)
xref = x
res_ref = np.abs(bl - Kl @ xref).max()


def report(name, xs):
    r = np.abs(b - K @ xs).max() / (np.abs(K).sum(axis=1).max() * np.abs(xs).max() + np.abs(b).max())
    fe = float(np.abs(xs - xref).max() / np.abs(xref).max())
    print(f"{name:34s} backward err {r:.2e}   forward err {fe:.2e}")


ldl, dd, perm = sl.ldl(K)  # noqa: F841
xs = sl.solve(K, b, assume_a="sym")
report("LAPACK dsysv (Bunch-Kaufman)", xs)
c = sl.cho_factor(K, lower=True) if np.linalg.eigvalsh(K).min() > 0 else None
if c is not None:
    report("LAPACK dpotrf/dpotrs", sl.cho_solve(c, b))
st = torch.cuda.Stream()
ctx = mj.HipContext(0, stream=st.cuda_stream)
Kd = torch.from_numpy(np.asfortranarray(K)).cuda()
torch.cuda.synchronize()
for alg in (mj.BUNCHKAUFMAN, mj.LDL, mj.CHOLESKY):
  for mfma in (1, 0):   # 256 x 256 inverses from the MFMA kernel / from the scalar block substitution
    try:
        ls = mj.HipLinearSolver(Kd, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=alg))
        ls.set_option("linv_mfma", mfma)
        ls.factorize()
        xs = ls.solve_linear_system(b.copy())
        report(f"HipLinearSolver {alg} linv_mfma={mfma} inertia {ls.inertia()}", xs)
        ls.close()
    except Exception as e:
        print(alg, "failed:", repr(e)[:200])
ctx.close()
