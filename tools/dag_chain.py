"""Chain-only timeline of one factorization under the task-DAG schedule (option dag_trace): per strip-column when its last
diagonal block was published and, for a few strips, start / front wait / band-tile wait / prologue / end.
usage: python tools/dag_chain.py [N] [LDL|CHOLESKY]"""
import os
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "8")
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import madnlp_jl_amd as mj  # noqa: E402
from madnlp_jl_amd import _lib as L  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 11192
alg = sys.argv[2] if len(sys.argv) > 2 else "LDL"
band = int(os.environ.get("DAG_BAND", "16"))
cus2 = int(os.environ.get("MNK_DAG_CUS2", "96"))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    ctx = mj.HipContext(0, stream=s.cuda_stream)
    g = torch.Generator(device="cuda").manual_seed(N)
    R = torch.randn(N, 96, dtype=torch.float64, device="cuda", generator=g)
    A = R @ R.T + torch.diag(torch.rand(N, dtype=torch.float64, device="cuda", generator=g) * 10 + 1.0)
    ls = mj.HipLinearSolver(A, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=alg, panel_algo=5, single_rows=0))
    ls.set_option("dag_min_rows", 0)
    ls.set_option("dag_band", band)
    ls.factorize(); s.synchronize()
    ls.set_option("dag_fill", 0)   # (the trace is indexed by task: the list without the zero-fill tasks is the one tools/dag_tasks.py mirrors)
    ls.set_option("dag_trace", 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s); ls.factorize(); e1.record(s); s.synchronize()
    print(f"N={N} {alg}: factorize (traced) {e0.elapsed_time(e1):.3f} ms")
Np = (N + 127) // 128 * 128
nsc = (Np + 255) // 256
ntasks, js2 = int(ls.get_stat("dag_ntasks")), int(ls.get_stat("dag_js2"))
v = np.zeros(ntasks * 8 + 4096 * 8, dtype=np.uint64)
L.check(L.lib().mnk_ls_debug_solve_trace(ls._h, v.ctypes.data, v.size), "trace")
ch = v[ntasks * 8:].astype(np.float64)
t0 = ch[ch > 0].min()
us = lambda x: (x - t0) / 100.0  # noqa: E731
print(f"second phase from strip-column {js2}; per strip: start + front wait f + band-tile wait a + prologue p > end, relative to the previous D3")
prev = 0.0
for Js in range(nsc):
    grid = min(band, Np // 64) if Js < js2 else (Np - 256 * js2) // 64
    base = Js * 128 if Js < js2 else js2 * 128 + (Js - js2) * 8 * grid
    nst = min(grid, (Np - 256 * Js) // 64)
    c = ch[base: base + 8 * nst].reshape(nst, 8)
    if c[0, 0] == 0:
        continue
    d3 = us(c[min(3, nst - 1), 2])
    sel = [t for t in (0, 3, 4, 8, 12, 15, 16, 32, 64, nst - 1) if t < nst]
    strips = " ".join(f"{t}:{us(c[t,0])-prev:.0f}+{us(c[t,5])-us(c[t,0]):.0f}f+{us(c[t,1])-us(c[t,5]):.0f}a+{us(c[t,4])-us(c[t,1]):.0f}p>{us(c[t,2])-prev:.0f}" for t in sorted(set(sel)) if c[t, 0] > 0)
    print(f"Js={Js:2d} D3 {d3:7.0f} (+{d3-prev:4.0f}) | {strips}")
    prev = d3
