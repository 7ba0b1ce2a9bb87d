import cProfile, pstats, os, sys, io, time
sys.path.insert(0, os.getcwd())
import torch
import madnlp_jl_amd as mj
from madnlp_jl_amd.ipm import IPMOptions
from madnlp_jl_amd.ipm_dev import DeviceMadNLPSolver
from madnlp_jl_amd.problems import ACOPFModel
nlp = ACOPFModel("case1354pegase")
st = torch.cuda.Stream(); ctx = mj.HipContext(0, stream=st.cuda_stream)
def factory(info):
    return mj.SparseCondensedKKTSystem(info["n"], info["m"], nlp.jac_I, nlp.jac_J, nlp.hess_I, nlp.hess_J, info["ind_ineq"], info["ind_lb"], info["ind_ub"], ctx=ctx,
                                       opt_linear_solver=mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN), device_kkt_ops=True)
def opts():
    o = IPMOptions(tol=1e-6); o.relax_equality, o.dual_initialization = True, "zero"; return o
sd = DeviceMadNLPSolver(nlp, factory, opts()); sd.solve()   # warm
sd2 = DeviceMadNLPSolver(nlp, factory, opts())
pr = cProfile.Profile(); t0=time.perf_counter(); pr.enable(); sd2.solve(); pr.disable(); print("wall", time.perf_counter()-t0, sd2.cnt.k, sd2.cnt.factorization_cnt)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
