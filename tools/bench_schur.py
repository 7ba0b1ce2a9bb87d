"""Timing record of the dense S stage on one MI355X: ns scenario blocks of order blk, nd design
variables.  usage: python tools/bench_schur.py [ns blk nd]"""
import json
import os
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "8")  # the pool's boxes cap the process at 16 CPUs: idle BLAS pools must not spin on 256
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import madnlp_jl_amd as mj  # noqa: E402
from madnlp_jl_amd.schur import SchurDenseStage  # noqa: E402
from tests.test_schur import two_stage_blocks  # noqa: E402

ns, blk, nd = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (16, 512, 256)
nv, nc = blk * 3 // 4, blk - blk * 3 // 4
A, Cs, S0, blk = two_stage_blocks(ns, nv, nc, nd, seed=1)
ctx = mj.HipContext(0)
st = SchurDenseStage(A, Cs, S0, nd, blk, ctx=ctx)


def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    return float(np.mean(ts))


tb = timeit(st.build_kkt)
tf = timeit(st.factorize_kkt)
rk = torch.randn(ns, blk, dtype=torch.float64, device="cuda")
rd = torch.randn(nd, dtype=torch.float64, device="cuda")
ts = timeit(lambda: st.solve(rk, rd))
flops_acc = 2.0 * ns * nd * nd * blk
print(json.dumps({"config": f"Schur dense S stage: ns={ns} blk={blk} nd={nd} (one rank)", "ms_build_kkt": tb,
                  "ms_factorize_S": tf, "ms_solve": ts, "inertia_S": st.inertia(),
                  "note": "build = ns x (blocked LDL^T of A_k + right-side block triangular solve of the nd rows of C_dk on MFMA "
                          "+ MFMA accumulation S -= (C L^-T D^-1)(C L^-T)'); solve = 2 single-RHS solves per scenario + one with S",
                  "S_accumulation_gflop": flops_acc / 1e9}))
