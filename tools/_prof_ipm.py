import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import madnlp_jl_amd as mj
from madnlp_jl_amd.ipm import IPMOptions
from madnlp_jl_amd.ipm_dev import DeviceMadNLPSolver
from madnlp_jl_amd.problems import SparseQPModel
nlp = SparseQPModel("case1354pegase")
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
ctx = mj.HipContext(0, stream=st.cuda_stream)
def factory(info):
    return mj.SparseCondensedKKTSystem(info["n"], info["m"], nlp.jac_I, nlp.jac_J, nlp.hess_I, nlp.hess_J, info["ind_ineq"], info["ind_lb"], info["ind_ub"], ctx=ctx,
                                       opt_linear_solver=mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN), device_kkt_ops=True)
walls = []; worst = 0
for rep in range(8):
    o = IPMOptions(tol=1e-6); o.relax_equality, o.dual_initialization = True, "zero"
    s = DeviceMadNLPSolver(nlp, factory, o); s.initialize(); s._upload()
    ls = s.kkt.linear_solver; f = ls.inertia; log = []
    def g():
        t0 = time.perf_counter(); r = f(); log.append(1e3 * (time.perf_counter() - t0)); return r
    ls.inertia = g
    torch.cuda.synchronize(); t0 = time.perf_counter(); s.solve(); torch.cuda.synchronize(); walls.append(1e3 * (time.perf_counter() - t0) / s.cnt.k)
    worst = max(worst, max(log)); s.K.close(); s.kkt.close()
print("ms/iter per run:", " ".join("%.2f" % w for w in walls), " worst inertia wait %.1f ms" % worst)
