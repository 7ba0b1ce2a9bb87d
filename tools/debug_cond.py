"""Why does Cholesky of the case1354pegase-shaped K fail on the GPU but not in LAPACK? (conditioning probe)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import madnlp_jl_amd as mj
from madnlp_jl_amd.problems import opf_shaped
from oracle import kernels as okern
from oracle.sparse_condensed import SparseCondensedKKTSystem as OSC
from oracle.lapack_cpu import LapackCPUSolver, CHOLESKY

P = opf_shaped("case1354pegase", du=1e-8)
ko = OSC(P.n, P.m, P.jac_I, P.jac_J, P.hess_I, P.hess_J, P.ind_ineq, P.ind_lb, P.ind_ub, lambda A: LapackCPUSolver(A, CHOLESKY))
for f in ("reg", "l_diag", "u_diag", "l_lower", "u_lower", "du_diag"):
    getattr(ko, f)[:] = getattr(P, f)
ko.jac[:] = P.jac; ko.hess[:] = P.hess
ko.compress_jacobian(); ko.compress_hessian(); okern.set_aug_diagonal(ko); ko.build_kkt()
Kd = ko.aug_com.to_dense()
ko.linear_solver.factorize()
Lr = np.tril(ko.linear_solver.fact)
d = np.diag(Lr)
print("lapack info", ko.linear_solver.info, "min/max pivot L_jj", d.min(), d.max(), "argmin", d.argmin())
print("diag K range", np.diag(Kd).min(), np.diag(Kd).max())
for la in (0, 1):
    for alg in (mj.CHOLESKY, mj.LDL):
        M = mj.HipLinearSolver(np.asfortranarray(Kd), opt=mj.HipSolverOptions(lapack_algorithm=alg, lookahead=bool(la)))
        M.factorize()
        print("lookahead", la, alg, "info", M.info, "inertia", M.inertia())
        if M.info == 0 and alg == mj.CHOLESKY:
            Lg, _ = M.get_factor(); Lg = np.tril(Lg)
            print("   max |Lg-Lr|/|Lr|", np.abs(Lg - Lr).max() / np.abs(Lr).max(), " min pivot", np.diag(Lg).min())
        if alg == mj.LDL:
            Lg, D = M.get_factor()
            print("   D min/max", D.min(), D.max(), "argmin", D.argmin(), "lapack d^2 there", d[D.argmin()]**2)
        M.close()
