"""What a SPECULATIVE first correction of inertia_correction! costs against the sequential pair (round 6): trial 0 (the matrix as it is,
not positive definite: rejected at its first non-positive pivot) and trial 1 (regularized, accepted) of a case1354pegase-shaped
condensed KKT system -- one after the other with the inertia fetched in between, or assembled back to back and factorized as ONE
batch of two (a second solver on the same KKT handle).
usage: python tools/spec_pair_time.py [case] [dw]"""
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import madnlp_jl_amd as mj  # noqa: E402
from madnlp_jl_amd.problems import opf_shaped  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "case1354pegase"
dw = float(sys.argv[2]) if len(sys.argv) > 2 else 1e4
ctx = mj.HipContext(0)
P = opf_shaped(case, indefinite=True, sigma_s_decades=2.0, du=1e-8)
k = mj.SparseCondensedKKTSystem(P.n, P.m, P.jac_I, P.jac_J, P.hess_I, P.hess_J, P.ind_ineq, P.ind_lb, P.ind_ub, ctx=ctx,
                                opt_linear_solver=mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN))
for f in ("reg", "l_diag", "u_diag", "l_lower", "u_lower", "du_diag"):
    getattr(k, f)[:] = getattr(P, f)
k.jac[:] = P.jac
k.hess[:] = P.hess
k.compress_jacobian(); k.compress_hessian(); k.set_aug_diagonal()
spare = k.ensure_spare_solver()
pr0 = k.pr_diag.copy(); reg0 = k.reg.copy()


def reset():
    k.pr_diag[:] = pr0; k.reg[:] = reg0


def sequential():
    reset()
    k.build_kkt(); k.linear_solver.factorize_async()
    i0 = k.linear_solver.inertia()
    k.regularize_diagonal(dw, 0.0)
    k.build_kkt(); k.linear_solver.factorize_async()
    i1 = k.linear_solver.inertia()
    return i0, i1


def speculative():
    reset()
    with mj.factorize_batch():
        k.build_kkt(); k.linear_solver.factorize_async()
        k.regularize_diagonal(dw, 0.0)
        k.build_kkt(); spare.factorize_async()
    i0 = k.linear_solver.inertia()
    i1 = spare.inertia()
    return i0, i1


def alone(full):
    reset()
    if full:
        k.regularize_diagonal(dw, 0.0)
    k.build_kkt(); k.linear_solver.factorize_async()
    return k.linear_solver.inertia()


for name, fn in (("trial 0 alone (rejected early)", lambda: alone(False)), ("trial 1 alone (accepted)", lambda: alone(True)),
                 ("sequential pair", sequential), ("speculative pair (one batch of two)", speculative)):
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    print(f"{name:40s}: median {np.median(ts):7.3f} ms  min {min(ts):7.3f}  inertia {out}  schedule {k.linear_solver.get_stat('panel_algo')} / {spare.get_stat('panel_algo')}"
          f"  early rejects {k.linear_solver.get_stat('early_rejects')} / {spare.get_stat('early_rejects')}  fall-backs {k.linear_solver.get_stat('pp_fallbacks')} / {spare.get_stat('pp_fallbacks')}")
k.close()
