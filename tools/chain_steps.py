"""The pivot chain's hand-over, taken apart on the schedule's own trace (option dag_trace): for the diagonal strips t = 1..3 of
every strip-column the time from the previous leaf's end (block t - 1 published) to this strip seeing the flag, from there to the
start of its own leaf, and the leaf itself; for strip 0 the way across the strip-column boundary.
usage: [MNK_LIBPATH=...] python tools/chain_steps.py [N] [LDL|CHOLESKY]"""
import os
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "8")
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import madnlp_jl_amd as mj  # noqa: E402
from madnlp_jl_amd import _lib as L  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
alg = sys.argv[2] if len(sys.argv) > 2 else "LDL"
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    ctx = mj.HipContext(0, stream=s.cuda_stream)
    g = torch.Generator(device="cuda").manual_seed(N)
    R = torch.randn(N, 96, dtype=torch.float64, device="cuda", generator=g)
    A = R @ R.T + torch.diag(torch.rand(N, dtype=torch.float64, device="cuda", generator=g) * 10 + 1.0)
    ls = mj.HipLinearSolver(A, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=alg, panel_algo=5, single_rows=0))
    ls.set_option("dag_min_rows", 0)
    ls.factorize(); s.synchronize()
    ls.set_option("dag_fill", 0)
    ls.set_option("dag_trace", 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s); ls.factorize(); e1.record(s); s.synchronize()
    print(f"{os.path.basename(os.environ.get('MNK_LIBPATH', 'libmadnlp_hip.so'))} N={N} {alg}: factorize (traced) {e0.elapsed_time(e1):.3f} ms")
Np = (N + 127) // 128 * 128
nsc = (Np + 255) // 256
ntasks, js2 = int(ls.get_stat("dag_ntasks")), int(ls.get_stat("dag_js2"))
band = 16
v = np.zeros(ntasks * 8 + 4096 * 8, dtype=np.uint64)
L.check(L.lib().mnk_ls_debug_solve_trace(ls._h, v.ctypes.data, v.size), "trace")
ch = v[ntasks * 8:].astype(np.float64)
t0 = ch[ch > 0].min()
us = lambda x: (x - t0) / 100.0  # noqa: E731
rows = []
prev_end = None
for Js in range(nsc):
    grid = min(band, Np // 64) if Js < js2 else (Np - 256 * js2) // 64
    base = Js * 128 if Js < js2 else js2 * 128 + (Js - js2) * 8 * grid
    nst = min(grid, (Np - 256 * Js) // 64)
    c = ch[base: base + 8 * nst].reshape(nst, 8)
    if c[0, 0] == 0:
        continue
    line = [f"Js={Js:2d}"]
    for t in range(min(4, nst)):
        end, leaf0, seen = us(c[t, 2]), us(c[t, 7]), us(c[t, 6]) if c[t, 6] > 0 else None
        if t == 0:
            line.append(f"strip 0: leaf starts {leaf0 - prev_end:5.1f} after the previous D3, leaf {end - leaf0:5.1f}" if prev_end is not None else f"strip 0: leaf {end - leaf0:5.1f}")
        else:
            line.append(f"| {t}: flag seen +{seen - prev_end:4.1f}, leaf starts +{leaf0 - seen:4.1f}, leaf {end - leaf0:5.1f}")
            rows.append((seen - prev_end, leaf0 - seen, end - leaf0))
        prev_end = end
    print(" ".join(line))
if rows:
    a = np.array(rows)
    print(f"mean over the strips 1..3: previous leaf's end -> flag seen {a[:,0].mean():.2f} us, -> own leaf starts {a[:,1].mean():.2f} us, leaf {a[:,2].mean():.2f} us")
