"""End-to-end IPM run on the HIP path at BASELINE size: an OPF-shaped convex QP (case1354pegase shape,
N = 11192 condensed KKT) through the host mirror of MadNLP's regular phase.  Reports the measured
factorizations and back-solves per iteration (the n_f, n_s of SURVEY 8(d)) and the wall-clock split."""
import json
import os
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "8")  # the pool's boxes cap the process at 16 CPUs: idle BLAS pools must not spin on 256
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import madnlp_jl_amd as mj  # noqa: E402
from madnlp_jl_amd.ipm import IPMOptions, MadNLPSolver  # noqa: E402
from madnlp_jl_amd.problems import SparseQPModel  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "case1354pegase"
device_ops = (sys.argv[2] != "host") if len(sys.argv) > 2 else True
nlp = SparseQPModel(case)
ctx = mj.HipContext(0)


def factory(info):
    return mj.SparseCondensedKKTSystem(info["n"], info["m"], nlp.jac_I, nlp.jac_J, nlp.hess_I, nlp.hess_J, info["ind_ineq"],
                                       info["ind_lb"], info["ind_ub"], ctx=ctx,
                                       opt_linear_solver=mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN),
                                       device_kkt_ops=device_ops)


opt = IPMOptions(tol=1e-6)
opt.relax_equality, opt.dual_initialization = True, "zero"
s = MadNLPSolver(nlp, factory, opt, sparse=True)
t0 = time.perf_counter()
s.solve()
ctx.synchronize()
wall = time.perf_counter() - t0
k = max(1, s.cnt.k)
out = {"case": case, "n": nlp.n, "m": nlp.m, "status": s.status, "iterations": s.cnt.k,
       "factorizations": s.cnt.linear_solver_factorize_cnt if hasattr(s.cnt, "linear_solver_factorize_cnt") else getattr(s.cnt, "factorization_cnt", None),
       "backsolves": getattr(s.cnt, "backsolve_cnt", None), "wall_s": wall, "ms_per_iteration_wall": 1e3 * wall / k,
       "device_kkt_ops": device_ops,
       "note": "wall time includes the host-side Python driver (callbacks, line search, vector algebra)"}
fc, bc = out["factorizations"], out["backsolves"]
if fc is not None:
    out["n_f_per_iteration"] = fc / k
if bc is not None:
    out["n_s_per_iteration"] = bc / k
print(json.dumps(out))
