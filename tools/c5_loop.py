"""Many rounds of the C5 step on one GPU in one process, every instance checked every round (inertia, backward error of a
solve against the oracle's sparse K): a hunt for intermittent wrong results.
usage: python tools/c5_loop.py [rounds] [nb] [key=value ...]   (solver options, e.g. dag_fill=0 prefill=0)"""
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import madnlp_jl_amd as mj  # noqa: E402
from madnlp_jl_amd.problems import OPF_CASES, opf_shaped  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 50
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 16
opts = dict(a.split("=") for a in sys.argv[3:])
dev = torch.device("cuda", 0)
base = OPF_CASES["case1354pegase"][0]
st = torch.cuda.Stream(dev)
ctx = mj.HipContext(0, stream=st.cuda_stream)
insts = []
for i in range(nb):
    P = opf_shaped("case1354pegase", seed=base + i, du=1e-8)
    kh = mj.SparseCondensedKKTSystem(P.n, P.m, P.jac_I, P.jac_J, P.hess_I, P.hess_J, P.ind_ineq, P.ind_lb, P.ind_ub, ctx=ctx,
                                     opt_linear_solver=mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN))
    for k, v in opts.items():
        kh.linear_solver.set_option(k, float(v))
    din = dict(jac=torch.from_numpy(P.jac).to(dev), hess=torch.from_numpy(P.hess).to(dev), pr=torch.from_numpy(P.pr_diag).to(dev),
               du=torch.from_numpy(P.du_diag).to(dev), rhs=torch.from_numpy(np.random.default_rng(base + i).standard_normal(P.n)).to(dev))
    din["x"] = torch.empty_like(din["rhs"])
    insts.append((P, kh, din))
torch.cuda.synchronize()
Ks = None
bad = 0
with torch.cuda.stream(st):
    import time
    walls = []
    for rnd in range(rounds):
        st.synchronize(); t_round = time.perf_counter()
        for (_, kh, din) in insts:
            kh.compress_jacobian(din["jac"]); kh.compress_hessian(din["hess"]); kh.build_kkt(din["pr"], din["du"])
            kh.linear_solver.factorize_async()
        for idx, (P, kh, din) in enumerate(insts):
            M = kh.linear_solver
            inertia = M.inertia()
            din["x"].copy_(din["rhs"])
            M.solve_linear_system(din["x"])
        st.synchronize()
        walls.append(1e3 * (time.perf_counter() - t_round))
        if Ks is None:   # (the matrices do not change from round to round)
            Ks = []
            for (P, kh, din) in insts:
                a = kh.aug_com
                Kl = sp.csc_matrix((a.nzval.copy(), a.rowval, a.colptr), shape=(P.n, P.n))
                Ks.append((Kl + sp.tril(Kl, -1).T).tocsr())
        for idx, (P, kh, din) in enumerate(insts):
            M = kh.linear_solver
            x, b = din["x"].cpu().numpy(), din["rhs"].cpu().numpy()
            K = Ks[idx]
            bw = np.abs(K @ x - b).max() / (abs(K).sum(axis=1).max() * np.abs(x).max() + np.abs(b).max())
            ine = M.inertia()
            if ine != (P.n, 0, 0) or not bw <= 1e-13 or (M.get_stat("pp_fallbacks") != 0 and rnd == rounds - 1):
                bad += 1
                print(f"round {rnd} instance {idx}: inertia {ine} backward error {bw:.2e} panel_algo {M.get_stat('panel_algo')} "
                      f"fallbacks {M.get_stat('pp_fallbacks')} site {M.get_stat('timeout_site')} growth {M.get_stat('growth'):.3g}", flush=True)
fb = sum(int(kh.linear_solver.get_stat("pp_fallbacks")) for (_, kh, _d) in insts)
w = np.array(walls[2:])
slow = [(i + 2, round(float(v), 1)) for i, v in enumerate(w) if v > 1.3 * np.median(w)]
print(f"rounds {rounds} x {nb} instances, options {opts}: {bad} bad results; fall-backs to schedule 1 (bounded waits that expired): {fb}; round wall time median {np.median(w):.1f} ms, max {w.max():.1f} ms, "
      f"rounds slower than 1.3 x median: {slow}")
