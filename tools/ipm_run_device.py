"""End-to-end IPM run with DEVICE-RESIDENT vectors and callbacks (madnlp_jl_amd.ipm_dev) next to the host mirror on the same
problem: iterations, counts, wall clock.
usage: ipm_run_device.py [case]             convex QP with the OPF sparsity, sparse condensed (default case1354pegase)
       ipm_run_device.py acopf [case]       polar AC-OPF NLP on the synthetic grid (callbacks: mnk_opf_*)
       ipm_run_device.py dense n m [n_eq]   DenseDummyQP, DenseCondensedKKTSystem (C2: 2048 512)
env IPM_DEVICE_ONLY=1: skip the host mirror (kernel profiles of the device-resident loop alone)"""
import json
import os
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "8")  # the pool's boxes cap the process at 16 CPUs: idle BLAS pools must not spin on 256
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import madnlp_jl_amd as mj  # noqa: E402
from madnlp_jl_amd.ipm import IPMOptions, MadNLPSolver  # noqa: E402
from madnlp_jl_amd.ipm_dev import DeviceMadNLPSolver  # noqa: E402
from madnlp_jl_amd.problems import ACOPFModel, DenseQPModel, SparseQPModel  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "case1354pegase"
dense = case == "dense"
if dense:
    dn, dm = int(sys.argv[2]), int(sys.argv[3])
    dne = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    nlp = DenseQPModel(dn, dm, dne)
    case = f"dense n={dn} m={dm} n_eq={dne}"
elif case == "acopf":
    grid = sys.argv[2] if len(sys.argv) > 2 else "case1354pegase"
    nlp = ACOPFModel(grid)
    case = f"acopf {grid}"
else:
    nlp = SparseQPModel(case)
st = torch.cuda.Stream()   # the library's stream; torch's current stream stays the default one (no torch kernel runs in the loop)
ctx = mj.HipContext(0, stream=st.cuda_stream)


def factory(info):
    if dense:
        return mj.DenseCondensedKKTSystem(info["n"], info["m"], info["ind_ineq"], info["ind_eq"], info["ind_lb"], info["ind_ub"],
                                          ctx=ctx, opt_linear_solver=mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN),
                                          device_kkt_ops=True)
    return mj.SparseCondensedKKTSystem(info["n"], info["m"], nlp.jac_I, nlp.jac_J, nlp.hess_I, nlp.hess_J, info["ind_ineq"],
                                       info["ind_lb"], info["ind_ub"], ctx=ctx,
                                       opt_linear_solver=mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN),
                                       device_kkt_ops=True)


def options():
    if dense:
        return IPMOptions(tol=1e-8)
    o = IPMOptions(tol=1e-6)
    o.relax_equality, o.dual_initialization = True, "zero"
    return o


drivers = (("host mirror (numpy vectors and callbacks, device KKT ops)", MadNLPSolver),
           ("device-resident vectors and callbacks", DeviceMadNLPSolver))
if os.environ.get("IPM_DEVICE_ONLY"):
    drivers = drivers[1:]
for label, cls in drivers:
    for rep in range(2):  # second run: warm
        s = cls(nlp, factory, options(), sparse=not dense)
        s.initialize()
        if cls is DeviceMadNLPSolver:
            s._upload()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.solve()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        rec = {"case": case, "driver": label, "run": rep, "n": nlp.n, "m": nlp.m, "status": s.status, "iterations": s.cnt.k,
               "factorizations": s.cnt.factorization_cnt, "backsolves": s.cnt.backsolve_cnt, "wall_s_regular_phase": wall,
               "ms_per_iteration_wall": 1e3 * wall / max(1, s.cnt.k), "it_per_s": s.cnt.k / wall, "obj": float(s.obj_val)}
        print(json.dumps(rec), flush=True)
        if cls is DeviceMadNLPSolver:
            s.cb.close()
            s.K.close()
        s.kkt.close()
ctx.close()
