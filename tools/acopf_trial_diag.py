"""Per trial of inertia_correction! in the device-resident AC-OPF run: the verdict, where a rejected trial stopped, and the smallest
DIAGONAL entry of the condensed KKT matrix -- is "some K_ii <= 0" (an O(N) look at the matrix) already the reason of the rejections?
usage: python tools/acopf_trial_diag.py [case]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import madnlp_jl_amd as mj  # noqa: E402
from madnlp_jl_amd.ipm import IPMOptions  # noqa: E402
from madnlp_jl_amd.ipm_dev import DeviceMadNLPSolver  # noqa: E402
from madnlp_jl_amd.problems import ACOPFModel  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "case1354pegase"
nlp = ACOPFModel(case)
ctx = mj.HipContext(0)


def factory(info):
    return mj.SparseCondensedKKTSystem(info["n"], info["m"], nlp.jac_I, nlp.jac_J, nlp.hess_I, nlp.hess_J, info["ind_ineq"], info["ind_lb"],
                                       info["ind_ub"], ctx=ctx, opt_linear_solver=mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN),
                                       device_kkt_ops=True)


o = IPMOptions(tol=1e-6)
o.relax_equality, o.dual_initialization = True, "zero"
s = DeviceMadNLPSolver(nlp, factory, o)
diag_pos = None
last_probe = (0, 0)


def on_trial(solver, n_trial, inertia, correct, accepted):
    global diag_pos
    A = solver.kkt.aug_com
    if diag_pos is None:
        colptr, rowval = np.asarray(A.colptr), np.asarray(A.rowval)
        diag_pos = np.array([colptr[j] + int(np.nonzero(rowval[colptr[j]:colptr[j + 1]] == j)[0][0]) for j in range(A.n)])
    vals = A.nzval
    if n_trial == 0 and not correct and solver.cnt.k in (4, 10) and os.environ.get("SAVE_TRIALS"):
        np.savez_compressed(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", f"acopf_k{solver.cnt.k}_trial0.npz"),
                            colptr=np.asarray(A.colptr), rowval=np.asarray(A.rowval), nzval=np.asarray(vals), n=A.n)
    d = vals[diag_pos]
    nneg = int((d <= 0).sum())
    ls = solver.kkt.linear_solver
    global last_probe
    hm = (int(ls.get_stat("probe_hits")), int(ls.get_stat("probe_misses")))
    probe = "probe HIT " if hm[0] > last_probe[0] else "probe miss" if hm[1] > last_probe[1] else "          "
    last_probe = hm
    print(f"{probe} k={solver.cnt.k:2d} trial {n_trial}: del_w={solver.del_w:9.3e} inertia {inertia} {'ok ' if correct else 'REJ'} stop col {int(ls.get_stat('early_reject_col')) if not correct else -1:6d}"
          f"  min diag {d.min():10.3e}  entries <= 0: {nneg}  first at {int(np.nonzero(d <= 0)[0][0]) if nneg else -1}", flush=True)


s.on_trial = on_trial
s.solve()
print(s.status, s.cnt.k, s.cnt.factorization_cnt)
