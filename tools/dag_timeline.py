"""Per-strip timeline of the persistent pivot chain of ONE factorization under the task-DAG schedule (option dag_trace):
when every strip of every band started, what it waited for, when it was done, and when the bulk kernel delivered the rows
that enter the band.  usage: python tools/dag_timeline.py [N] [LDL|CHOLESKY]   (env MNK_DAG_CHUNK / MNK_DAG_BAND as the run)"""
import os
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "8")
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import madnlp_jl_amd as mj  # noqa: E402
from madnlp_jl_amd import _lib as L  # noqa: E402
from dag_tasks import dag_tasks  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 11192
alg = sys.argv[2] if len(sys.argv) > 2 else "CHOLESKY"
chunk = int(os.environ.get("MNK_DAG_CHUNK", "12"))
band = int(os.environ.get("MNK_DAG_BAND", "16"))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    ctx = mj.HipContext(0, stream=s.cuda_stream)
    g = torch.Generator(device="cuda").manual_seed(N)
    R = torch.randn(N, 96, dtype=torch.float64, device="cuda", generator=g)
    A = R @ R.T + torch.diag(torch.rand(N, dtype=torch.float64, device="cuda", generator=g) * 10 + 1.0)
    ls = mj.HipLinearSolver(A, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=alg, panel_algo=5, single_rows=0))
    ls.set_option("dag_min_rows", 0)
    ls.factorize()
    s.synchronize()
    ls.set_option("dag_trace", 1)
    ls.factorize()
    s.synchronize()
Np = (N + 127) // 128 * 128
ntile = Np // 128
ts = dag_tasks(ntile, chunk, band // 2)
nt = len(ts)
nsc = (Np + 255) // 256
tr = np.zeros(nt * 8 + 4096 * 8, dtype=np.uint64)
L.check(L.lib().mnk_ls_debug_solve_trace(ls._h, tr.ctypes.data, tr.size), "trace")
bulk = tr[: nt * 8].reshape(nt, 8).astype(np.float64)
chain = tr[nt * 8: nt * 8 + nsc * band * 8].reshape(nsc, band, 8).astype(np.float64)
t0 = min(bulk[:, 0][bulk[:, 0] > 0].min(), chain[:, :, 0][chain[:, :, 0] > 0].min())
us = lambda x: (x - t0) / 100.0  # wall_clock64: 100 MHz  # noqa: E731
print(f"N={N} {alg}: {nt} bulk tasks, chunk {chunk}, band {band}; span {us(max(bulk[:, 5].max(), chain[:, :, 2].max())):.0f} us")
closing = {}
for k, t in enumerate(ts):
    if t[1] == 1:
        closing[(t[3], t[2])] = (us(bulk[k, 0]), us(bulk[k, 3]), us(bulk[k, 4]), us(bulk[k, 5]))   # grab, acc, diag, end
print("Js: D3 ready (end of strip 3) | strip t: start+wait -> end, all relative to the previous strip-column's D3 | rows entering: closing-task end")
prev = 0.0
for Js in range(nsc):
    c = chain[Js]
    if c[0, 0] == 0:
        continue
    nst = int((c[:, 0] > 0).sum())
    d3 = us(c[min(3, nst - 1), 2])
    # strip t: start, +front wait, +af wait, +prologue, > end
    strips = " ".join(f"{t}:{us(c[t,0])-prev:.0f}+{us(c[t,5])-us(c[t,0]):.0f}f+{us(c[t,1])-us(c[t,5]):.0f}a+{us(c[t,4])-us(c[t,1]):.0f}p>{us(c[t,2])-prev:.0f}" for t in (0, 3, 4, 8, 12, 15) if t < nst and c[t, 0] > 0)
    rows = [closing.get((r, cc)) for r in range(2 * Js + band // 2 - 2, 2 * Js + band // 2) for cc in (2 * Js - 2, 2 * Js - 1)]
    rows = [x for x in rows if x]
    rin = f" | rows-in: acc {max(x[1] for x in rows)-prev:.0f} diag {max(x[2] for x in rows)-prev:.0f} end {max(x[3] for x in rows)-prev:.0f}" if rows else ""
    print(f"Js={Js:2d} D3 {d3:7.0f} (+{d3-prev:4.0f}) | {strips}{rin}")
    prev = d3
