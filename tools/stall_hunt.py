"""Hunt for the rare stall of the task-DAG schedule: rounds of 16 back-to-back factorizations of
case1354pegase-shaped systems on one context, option dag_debug = 1 (and dag_fill = 0: the task list is then the one
tools/dag_tasks.py mirrors), until one of them runs into its bounded wait; the progress words the time-out left and what the
chain's strips were waiting for go to gpurun_out/stall/state_<k>.json for offline analysis (tools/stall_analyze.py).
usage: python tools/stall_hunt.py [max_rounds] [events]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import madnlp_jl_amd as mj  # noqa: E402
from madnlp_jl_amd import _lib as L  # noqa: E402
from madnlp_jl_amd.problems import OPF_CASES, opf_shaped  # noqa: E402

max_rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 400
want = int(sys.argv[2]) if len(sys.argv) > 2 else 1
nb = 16
dev = torch.device("cuda", 0)
base = OPF_CASES["case1354pegase"][0]
st = torch.cuda.Stream(dev)
ctx = mj.HipContext(0, stream=st.cuda_stream)
insts = []
for i in range(nb):
    P = opf_shaped("case1354pegase", seed=base + i, du=1e-8)
    kh = mj.SparseCondensedKKTSystem(P.n, P.m, P.jac_I, P.jac_J, P.hess_I, P.hess_J, P.ind_ineq, P.ind_lb, P.ind_ub, ctx=ctx,
                                     opt_linear_solver=mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN))
    for k, v in (("dag_fill", 0.0), ("dag_debug", 1.0)):
        kh.linear_solver.set_option(k, v)
    din = dict(jac=torch.from_numpy(P.jac).to(dev), hess=torch.from_numpy(P.hess).to(dev), pr=torch.from_numpy(P.pr_diag).to(dev),
               du=torch.from_numpy(P.du_diag).to(dev))
    insts.append((P, kh, din))
torch.cuda.synchronize()
out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "stall")
os.makedirs(out_dir, exist_ok=True)
seen = [0.0] * nb
events = 0
t0 = time.perf_counter()
with torch.cuda.stream(st):
    for rnd in range(max_rounds):
        for (_, kh, din) in insts:
            kh.compress_jacobian(din["jac"]); kh.compress_hessian(din["hess"]); kh.build_kkt(din["pr"], din["du"])
            kh.linear_solver.factorize_async()
        for idx, (P, kh, din) in enumerate(insts):
            M = kh.linear_solver
            inertia = M.inertia()
            fb = M.get_stat("pp_fallbacks")
            if fb != seen[idx]:
                seen[idx] = fb
                Np = (P.n + 127) // 128 * 128
                ntile = Np // 128
                nflags = 2 + Np // 64 + 2 * ntile * ntile
                flags = np.zeros(nflags, dtype=np.int32)
                chain = np.zeros(8 * 128 + 16 * 1024, dtype=np.int32)
                have = C.c_int(0)
                L.check(L.lib().mnk_ls_debug_dag_state(M._h, flags.ctypes.data, nflags, chain.ctypes.data, chain.size, C.byref(have)), "state")
                rec = {"round": rnd, "instance": idx, "N": P.n, "Np": Np, "ntile": ntile, "site": M.get_stat("timeout_site"),
                       "have": have.value, "inertia": inertia, "dag_ntasks": M.get_stat("dag_ntasks"),
                       "flags": flags.tolist(), "chain": chain[:128].reshape(-1, 8).tolist(),
                       # (a diagnostic build -DMNK_DIAG_BULK_DBG=1 only, zeros otherwise: 16 words per bulk workgroup --
                       # task, stage (1 grabbed / 2 waiting for rows / 3 waiting for the chunk order / 4 published), the waited
                       # words, target, value seen, polls >> 18, tasks grabbed, -, the four front words as seen)
                       "bulk": chain[8 * 128:].reshape(-1, 16).tolist()}
                path = os.path.join(out_dir, f"state_{events}.json")
                json.dump(rec, open(path, "w"))
                print(f"round {rnd} instance {idx}: time-out at site {rec['site']}, state -> {path}; chain strips waiting: "
                      f"{[c for c in rec['chain'] if c[0] != 0]}", flush=True)
                events += 1
        if events >= want:
            break
print(f"{rnd + 1} rounds x {nb} factorizations in {time.perf_counter() - t0:.1f} s: {events} time-outs")
