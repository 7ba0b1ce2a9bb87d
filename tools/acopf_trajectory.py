"""Where the device-resident AC-OPF run and the oracle golden part ways (per-iteration
del_w and counts).  Runs `DeviceMadNLPSolver` on the case1354pegase-sized polar AC-OPF NLP with an instrumented
`inertia_correction!` (reference src/IPM/solver.jl:611-670): every trial of every iteration is logged -- del_w, the inertia
the HIP solver reports, which tier produced the factor, Richardson steps and residual ratio, whether the step was accepted --
and printed next to the golden's history (`tests/golden/acopf_case1354_oracle.json`).  With REPLAY=1 the condensed matrix of
every trial that the HIP back-end REJECTED (wrong inertia or refinement failure), and of every trial of an iteration whose
del_w differs from the golden's, is copied to the host and factorized by dsytrf (the oracle's LapackCPUSolver): the two
inertia verdicts side by side.
With PIVOTS=1 (and REPLAY=1, EARLY_REJECT=0) the first matrix on which the static tier counts one negative pivot and dsytrf
none is taken apart: the pivot in question from the HIP factor, from an unpivoted LDL' in numpy fp64 and in 80-bit long double.
usage: [REPLAY=1 [PIVOTS=1 EARLY_REJECT=0]] python tools/acopf_trajectory.py [case] > profiles/r05_acopf_trajectory.txt"""
import json
import os
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "8")
import sys
import time

import numpy as np
import scipy.sparse as sp
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import madnlp_jl_amd as mj  # noqa: E402
from madnlp_jl_amd.ipm import IPMOptions  # noqa: E402
from madnlp_jl_amd.ipm_dev import DeviceMadNLPSolver  # noqa: E402
from madnlp_jl_amd.problems import ACOPFModel  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "case1354pegase"
replay = os.environ.get("REPLAY", "0") != "0"
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "acopf_case1354_oracle.json"))) if case == "case1354pegase" else None
nlp = ACOPFModel(case)
st = torch.cuda.Stream()
ctx = mj.HipContext(0, stream=st.cuda_stream)


def factory(info):
    return mj.SparseCondensedKKTSystem(info["n"], info["m"], nlp.jac_I, nlp.jac_J, nlp.hess_I, nlp.hess_J, info["ind_ineq"],
                                       info["ind_lb"], info["ind_ub"], ctx=ctx,
                                       opt_linear_solver=mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN),
                                       device_kkt_ops=True, early_reject=os.environ.get("EARLY_REJECT", "1") != "0")


class Traced(DeviceMadNLPSolver):
    """`inertia_correction!` with a log of its trials (the `on_trial` hook of ipm_dev.DeviceMadNLPSolver)."""

    def __init__(self, *a, **k):
        self.trials = []
        self.on_trial = Traced._log
        super().__init__(*a, **k)

    def _log(self, n_trial, inertia, correct, ok):
        M = self.kkt.linear_solver
        mat = None
        if replay and (not ok):
            A = self.kkt.aug_com
            mat = (np.asarray(A.colptr).copy(), np.asarray(A.rowval).copy(), A.nzval.copy())
        self.trials.append(dict(k=self.cnt.k, trial=n_trial, del_w=self.del_w, del_c=self.del_c, inertia=tuple(int(v) for v in inertia),
                                correct=bool(correct), ok=bool(ok), ir=int(self.iterator.ir) if correct else 0,
                                rr=float(self.iterator.residual_ratio) if correct else float("nan"), bk=bool(M.bk_info()[0]),
                                growth=M.get_stat("growth"), algo=M.get_stat("panel_algo"), mat=mat,
                                stop=int(M.get_stat("early_reject_col")) if not correct else -1))


o = IPMOptions(tol=gold["tol"] if gold else 1e-6)
o.relax_equality, o.dual_initialization = True, "zero"
sd = Traced(nlp, factory, o)
t0 = time.perf_counter()
sd.solve()
wall = time.perf_counter() - t0
print(f"# {case}: n = {nlp.n}, m = {nlp.m}; device-resident run: {sd.status}, {sd.cnt.k} iterations, "
      f"{sd.cnt.factorization_cnt} factorizations, {sd.cnt.backsolve_cnt} back-solves, objective {sd.obj_val:.12g}, {wall:.3f} s")
if gold:
    print(f"# golden (oracle back-end, dsytrf): {gold['iterations']} iterations, {gold['factorizations']} factorizations, "
          f"{gold['backsolves']} back-solves, objective {gold['objective']:.12g}")
by_k = {}
for t in sd.trials:
    by_k.setdefault(t["k"], []).append(t)
gh = {r["k"]: r for r in gold["history"]} if gold else {}
hist = {r.k: r for r in sd.history}
print("# k | trials (del_w: inertia tier R=Richardson steps/ratio verdict) | del_w accepted HIP | del_w golden | inf_pr HIP / golden | inf_du HIP / golden")
first_div = None
for kk in sorted(by_k):
    tr = by_k[kk]
    cells = []
    for t in tr:
        verdict = "ok" if t["ok"] else ("WRONG-INERTIA" if not t["correct"] else "REFINE-FAIL")
        stop = f" stopped at pivot {t['stop']} of {nlp.n}" if t.get("stop", -1) >= 0 and not t["correct"] else ""
        cells.append(f"{t['del_w']:.3g}: {t['inertia']} {'BK' if t['bk'] else 'static'} g={t['growth']:.2g} R={t['ir']}/{t['rr']:.1e} {verdict}{stop}")
    # the history record written AFTER iteration kk's step is record kk + 1; del_w of iteration kk is stored there
    hrec, grec = hist.get(kk + 1), gh.get(kk + 1)
    dw_h = tr[-1]["del_w"]
    dw_g = grec["del_w"] if grec else float("nan")
    same = grec is not None and abs(dw_h - dw_g) <= 1e-12 * max(1.0, abs(dw_g))
    if grec is not None and not same and first_div is None:
        first_div = kk
    print(f"{kk:3d} | " + " ; ".join(cells) + f" | {dw_h:.4g} | {dw_g:.4g}{'' if same else '  <-- differs'} | "
          + (f"{hrec.inf_pr:.3e} / {grec['inf_pr']:.3e} | {hrec.inf_du:.3e} / {grec['inf_du']:.3e}" if hrec and grec else "-"))
print(f"# first iteration whose accepted del_w differs from the golden's: {first_div}")
extra = sum(len(v) - 1 for v in by_k.values())
print(f"# trials beyond the first, summed over the iterations: {extra}; wrong-inertia verdicts {sum(1 for t in sd.trials if not t['correct'])}, "
      f"refinement failures {sum(1 for t in sd.trials if t['correct'] and not t['ok'])}, pivoted-tier factorizations {sum(1 for t in sd.trials if t['bk'])}")

if replay:
    from oracle.lapack_cpu import BUNCHKAUFMAN, LapackCPUSolver
    print("# replay: the matrices the HIP back-end rejected, factorized by dsytrf on the host (oracle LapackCPUSolver)")
    for t in sd.trials:
        if t["mat"] is None:
            continue
        colptr, rowval, nz = t["mat"]
        n = len(colptr) - 1
        Kl = sp.csc_matrix((nz, rowval, colptr), shape=(n, n))
        dense = np.asfortranarray((Kl + sp.tril(Kl, -1).T).toarray())
        t1 = time.perf_counter()
        ref = LapackCPUSolver(dense, BUNCHKAUFMAN).factorize()
        ine = ref.inertia()
        # a second opinion that does not depend on a factorization's pivoting: the count of negative eigenvalues
        print(f"k={t['k']} trial {t['trial']} del_w={t['del_w']:.3g}: HIP {t['inertia']} ({'pivoted' if t['bk'] else 'static'} tier, "
              f"{'wrong inertia' if not t['correct'] else 'refinement failed'}) | dsytrf {tuple(int(v) for v in ine)} "
              f"({time.perf_counter() - t1:.1f} s) | max|K| {np.abs(nz).max():.3e} min|diag| {np.abs(Kl.diagonal()).min():.3e}")

def ldl_nopivot(A, dtype):
    """Unpivoted LDL' of a dense symmetric matrix in `dtype`, right-looking by 64-column blocks (the static order of the
    HIP tier, another summation order); returns D."""
    A = np.array(A, dtype=dtype, order="F")
    n = A.shape[0]
    d = np.zeros(n, dtype=dtype)
    for j0 in range(0, n, 64):
        j1 = min(n, j0 + 64)
        for j in range(j0, j1):
            d[j] = A[j, j]
            A[j + 1:, j] /= d[j]
            if j + 1 < j1:
                A[j + 1:, j + 1:j1] -= np.outer(A[j + 1:, j] * d[j], A[j + 1:j1, j])
        if j1 < n:
            W = A[j1:, j0:j1] * d[j0:j1]
            A[j1:, j1:] -= W @ A[j1:, j0:j1].T
    return d


if os.environ.get("PIVOTS", "0") != "0" and replay:
    # The two pivots side by side: the first rejected matrix on which the static tier counts ONE negative pivot and dsytrf none.
    from oracle.lapack_cpu import BUNCHKAUFMAN, LapackCPUSolver
    import scipy.linalg as sla
    for t in sd.trials:
        if t["mat"] is None or t["inertia"][2] != 1 or t["inertia"][1] != 0:
            continue
        colptr, rowval, nz = t["mat"]
        n = len(colptr) - 1
        Kl = sp.csc_matrix((nz, rowval, colptr), shape=(n, n))
        dense = np.asfortranarray((Kl + sp.tril(Kl, -1).T).toarray())
        if tuple(int(v) for v in LapackCPUSolver(dense, BUNCHKAUFMAN).factorize().inertia()) != (n, 0, 0):
            continue
        ls = mj.HipLinearSolver(dense, ctx, mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN))
        ls.set_option("accept_only_pd", 1); ls.set_option("early_reject", 0)
        ls.factorize()
        Lh, Dh = ls.get_factor()
        pv = int(np.argmin(Dh))
        lead = dense[:pv + 1, :pv + 1]
        d64 = ldl_nopivot(lead, np.float64)
        t1 = time.perf_counter()
        d80 = ldl_nopivot(lead, np.longdouble)
        sub = float(np.sum(Lh[pv, :pv] ** 2 * Dh[:pv]))
        ev = sla.eigvalsh(lead, subset_by_index=[0, 1])
        print(f"# pivots side by side: k={t['k']} trial {t['trial']} del_w={t['del_w']:.3g}; static order, pivot index {pv} of {n} "
              f"(variable block: {'qg' if pv >= 2 * nlp.nbus + nlp.ngen else 'pg' if pv >= 2 * nlp.nbus else 'vm' if pv >= nlp.nbus else 'va'})")
        print(f"#   K[p,p] = {dense[pv, pv]:.17e}   sum_j L[p,j]^2 D[j] (HIP factor) = {sub:.17e}   max|K| = {np.abs(nz).max():.3e}")
        print(f"#   D[p]: HIP static LDL' (fp64 MFMA, left-looking 128-tiles) {Dh[pv]:+.6e} | numpy fp64 right-looking {float(d64[pv]):+.6e} | "
              f"80-bit long double right-looking {float(d80[pv]):+.6e} ({time.perf_counter() - t1:.0f} s) | dsytrf: all {n} pivots positive")
        print(f"#   negative D entries: HIP {int((Dh < 0).sum())}, numpy fp64 {int((d64 < 0).sum())}, long double {int((d80 < 0).sum())} (leading block); "
              f"two smallest eigenvalues of the leading {pv + 1} x {pv + 1} block (dsyevr): {ev[0]:+.3e}, {ev[1]:+.3e}")
        np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"acopf_k{t['k']}_trial{t['trial']}_matrix.npz"), colptr=colptr, rowval=rowval, nz=nz,
                            pivot=pv, d_hip=Dh[pv], d_fp64=float(d64[pv]), d_ld=float(d80[pv]))
        ls.close()
        break
sd.cb.close(); sd.K.close(); sd.kkt.close()
