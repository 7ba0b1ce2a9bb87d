"""Per-task timeline of ONE factorization under the task-DAG schedule (option dag_trace): what every chain launch waited
for and when the bulk tasks it depends on were finished.  usage: python tools/dag_timeline.py [N] [LDL|CHOLESKY]"""
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
alg = sys.argv[2] if len(sys.argv) > 2 else "CHOLESKY"
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
# the task list, rebuilt as dag_build_tasks does
tasks = []
for Jt in range(ntile):
    Js = Jt // 2
    for I in range(2 * Js + 4, ntile):
        tasks.append((0, I, Jt, Jt))
        if (Jt & 1) and I < 2 * Js + 8:
            Jb = Js + 2
            for cc in (2 * Jb, 2 * Jb + 1):
                if cc < ntile and I >= cc:
                    tasks.append((1, I, cc, 2 * Jb - 2))
nt = len(tasks)
nsc = Np // 256 + 1
tr = np.zeros(nt * 8 + nsc * 32, dtype=np.uint64)
L.check(L.lib().mnk_ls_debug_solve_trace(ls._h, tr.ctypes.data, tr.size), "trace")
bulk = tr[: nt * 8].reshape(nt, 8).astype(np.float64)
chain = tr[nt * 8:].reshape(nsc, 8, 4).astype(np.float64)
t0 = min(bulk[:, 0][bulk[:, 0] > 0].min(), chain[:, :, 0][chain[:, :, 0] > 0].min())
us = lambda x: (x - t0) / 100.0  # wall_clock64: 100 MHz  # noqa: E731
print(f"N={N} {alg}: {nt} bulk tasks, span {us(max(bulk[:, 5].max(), chain[:, :, 2].max())):.0f} us")
fin = {}   # (I, J) -> (type, grab, acc_done, diag_ready, end, nwait, wait_us)
for k, (ty, I, J, kend) in enumerate(tasks):
    b = bulk[k]
    fin[(ty, I, J)] = (us(b[0]), us(b[3]), us(b[4]) if ty == 0 else float("nan"), us(b[5]), int(b[6]), b[7] / 100.0, us(b[1]), us(b[2]))
print("launch: start | per strip: wait-end(+), end(+) relative to the launch start | band tiles: acc end; rows below: finalize end")
for Js in range(min(nsc, Np // 256)):
    c = chain[Js]
    if c[0, 0] == 0:
        continue
    st = us(c[:, 0][c[:, 0] > 0].min())
    en = us(c[:, 2].max())
    strips = " ".join(f"{t}:{us(c[t,1])-st:.0f}/{us(c[t,2])-st:.0f}" for t in range(8) if c[t, 0] > 0)
    line = f"Js={Js:2d} start {st:7.0f} dur {en-st:5.0f} | {strips}"
    # band tiles accumulated (BANDACC) for this launch and the bulk tiles whose rows enter the band (strips 4..7)
    ba = [fin.get((1, r, cc)) for r in range(2 * Js, 2 * Js + 4) for cc in (2 * Js, 2 * Js + 1)]
    ba = [x for x in ba if x]
    if ba:
        line += f" | bandacc grab {min(x[0] for x in ba)-st:.0f}..{max(x[0] for x in ba)-st:.0f} lastwait {max(x[7] for x in ba)-st:.0f} end {max(x[3] for x in ba)-st:.0f}"
    rows = [fin.get((0, r, cc)) for r in (2 * Js + 2, 2 * Js + 3) for cc in (2 * Js - 2, 2 * Js - 1)]
    rows = [x for x in rows if x]
    if rows:
        line += f" | rows-in: acc {max(x[1] for x in rows)-st:.0f} diag {max(x[2] for x in rows)-st:.0f} end {max(x[3] for x in rows)-st:.0f}"
    print(line)
# bulk summary: time in waits vs compute
dur = bulk[:, 5] - bulk[:, 0]
print(f"bulk tasks: mean duration {dur.mean()/100:.0f} us, mean wait {bulk[:,7].mean()/100:.0f} us, mean finalize {np.nanmean([f[3]-f[2] for f in fin.values() if f[2]==f[2]]):.1f} us, "
      f"mean diag wait {np.nanmean([f[2]-f[1] for f in fin.values() if f[2]==f[2]]):.1f} us")
