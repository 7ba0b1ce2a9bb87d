import sys, time, os
sys.path.insert(0, "/root/repo")
import torch, numpy as np
import madnlp_jl_amd as mj
from madnlp_jl_amd.ipm import IPMOptions
from madnlp_jl_amd.ipm_dev import DeviceMadNLPSolver
from madnlp_jl_amd.problems import ACOPFModel
nlp = ACOPFModel("case1354pegase")
mode = sys.argv[1]
if mode == "own":
    st = torch.cuda.Stream(); ctx = mj.HipContext(0, stream=st.cuda_stream)
else:
    ctx = mj.HipContext(0)
def factory(info):
    return mj.SparseCondensedKKTSystem(info["n"], info["m"], nlp.jac_I, nlp.jac_J, nlp.hess_I, nlp.hess_J, info["ind_ineq"], info["ind_lb"], info["ind_ub"], ctx=ctx,
        opt_linear_solver=mj.HipSolverOptions(lapack_algorithm="BUNCHKAUFMAN", outer_block=512), device_kkt_ops=True)
for rep in range(2):
    o = IPMOptions(tol=1e-6); o.relax_equality, o.dual_initialization = True, "zero"
    s = DeviceMadNLPSolver(nlp, factory, o); s.initialize(); s._upload(); torch.cuda.synchronize()
    if rep == 1:
        lsx = s.kkt.linear_solver
        orig = lsx.inertia
        def logged():
            t = time.perf_counter(); r = orig(); dt = time.perf_counter() - t
            print("   it", s.cnt.k, "inertia", r, "growth %.3g" % lsx.get_stat("growth"), "sign changes", lsx.get_stat("sign_changes"), "bk_count", lsx.get_stat("bk_count"), "del_w %.2g" % s.del_w, "inertia() took %.1f ms" % (1e3 * dt))
            return r
        lsx.inertia = logged
    t0 = time.perf_counter(); s.solve(); torch.cuda.synchronize(); w = time.perf_counter() - t0
    ls = s.kkt.linear_solver
    print(mode, rep, s.status, s.cnt.k, s.cnt.factorization_cnt, s.cnt.backsolve_cnt, f"{w:.3f}s", "algo", ls.get_stat("panel_algo"), "bk", ls.bk_info()[1], "growth", ls.get_stat("growth"))
    s.cb.close(); s.K.close(); s.kkt.close()
