import sys, numpy as np, torch
sys.path.insert(0,'/root/repo')
import madnlp_jl_amd as mj
from madnlp_jl_amd.problems import ACOPFModel
from madnlp_jl_amd.ipm import IPMOptions, MadNLPSolver
from madnlp_jl_amd.ipm_dev import DeviceMadNLPSolver
from tests.test_ipm_oracle import oracle_factory
case=sys.argv[1]
nlp=ACOPFModel(case)
st=torch.cuda.Stream(); ctx=mj.HipContext(0, stream=st.cuda_stream)
def opt():
    o=IPMOptions(tol=float(sys.argv[2]) if len(sys.argv)>2 else 1e-6); o.relax_equality, o.dual_initialization = True, "zero"; return o
def factory(info):
    return mj.SparseCondensedKKTSystem(info["n"], info["m"], nlp.jac_I, nlp.jac_J, nlp.hess_I, nlp.hess_J, info["ind_ineq"], info["ind_lb"], info["ind_ub"], ctx=ctx, opt_linear_solver=mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN), device_kkt_ops=True)
runs={}
so=MadNLPSolver(nlp, oracle_factory("sparse_condensed", nlp), opt(), sparse=True); so.solve(); runs["oracle"]=so
sh=MadNLPSolver(nlp, factory, opt(), sparse=True); sh.solve(); runs["host+hip"]=sh
sd=DeviceMadNLPSolver(nlp, factory, opt()); sd.solve(); runs["device"]=sd
for k,s in runs.items():
    print(k, s.status, s.cnt.k, s.cnt.factorization_cnt, s.cnt.backsolve_cnt, s.obj_val)
for i in range(max(len(s.history) for s in runs.values())):
    row=[]
    for k,s in runs.items():
        if i<len(s.history):
            h=s.history[i]; row.append(f"{h.inf_pr:.3e} {h.inf_du:.3e} dw {h.del_w:.1e} a {h.alpha:.3e}")
    print(i, " | ".join(row))
