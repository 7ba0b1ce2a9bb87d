"""Per-step time line of the pivot chain's strips (diagnostic build -DMNK_DIAG_STEP_TRACE=1 of factor.hip: tools/build_alt.sh steptr
"-DMNK_DIAG_STEP_TRACE=1" factor.hip): for every strip of the band and every step j of a strip-column, when it saw block j, when its
substitution was done, when its rows were published and when its updates were done -- relative to the publication of block j.
usage: MNK_LIBPATH=madnlp.jl_amd/lib/libmadnlp_hip_steptr.so python tools/chain_steps2.py [N] [LDL|CHOLESKY] [Js ...]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import madnlp_jl_amd as mj  # noqa: E402
from madnlp_jl_amd import _lib as L  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
alg = sys.argv[2] if len(sys.argv) > 2 else "LDL"
show = [int(a) for a in sys.argv[3:]] or [4, 5]
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
    ls.factorize(); s.synchronize()
Np = (N + 127) // 128 * 128
nsc = (Np + 255) // 256
ntasks, js2 = int(ls.get_stat("dag_ntasks")), int(ls.get_stat("dag_js2"))
G = min(16, Np // 64) if js2 > 0 else Np // 64   # (single phase: every row a strip of the chain)
v = np.zeros(ntasks * 8 + 4096 * 8, dtype=np.uint64)
L.check(L.lib().mnk_ls_debug_solve_trace(ls._h, v.ctypes.data, v.size), "trace")
ch = v[ntasks * 8:].astype(np.float64)
t0 = ch[ch > 0].min()
us = lambda x: (x - t0) / 100.0 if x > 0 else float("nan")  # noqa: E731
for Js in show:
    if Js >= nsc or js2 > 0:
        continue   # (the diagnostic stamps are laid out for single-phase schedules)
    nst = min(G, (Np - 256 * Js) // 64)
    c1 = ch[Js * 8 * G: Js * 8 * G + 8 * nst].reshape(nst, 8)
    c2 = ch[2048 * 8 + 2 * Js * 8 * G: 2048 * 8 + 2 * Js * 8 * G + 16 * nst].reshape(nst, 16)
    pub = [us(c1[j, 2]) for j in range(min(4, nst))]   # block j published
    print(f"strip-column {Js}: blocks published at " + " ".join(f"{p:7.1f}" for p in pub) + " us; per strip and step: seen / substituted / published / updated, relative to the block's publication")
    for t in range(nst):
        line = [f"  strip {t:2d}: start {us(c1[t, 0]) - pub[0]:6.1f} front {us(c1[t, 1]) - pub[0]:6.1f} prologue done {us(c1[t, 4]) - pub[0]:6.1f} |"]
        for j in range(min(4, nst)):
            if j < t:
                a = [us(c2[t, 4 * j + k]) - pub[j] for k in range(4)]
                line.append(f"j={j}: {a[0]:5.1f} {a[1]:5.1f} {a[2]:5.1f} {a[3]:5.1f} |")
            elif j == t:
                line.append(f"j={j}: leaf {us(c1[t, 7]) - (pub[j - 1] if j > 0 else pub[0]):5.1f} .. {us(c1[t, 2]) - (pub[j - 1] if j > 0 else pub[0]):5.1f} after block {max(j - 1, 0)} |")
        print(" ".join(line))
    c3 = v[ntasks * 8 + 3072 * 8 + Js * 8 * G: ntasks * 8 + 3072 * 8 + Js * 8 * G + 8 * nst].astype(np.float64).reshape(nst, 8)
    for t in range(nst):
        n = c3[t, 0]
        if n > 0:
            print(f"  strip {t:2d} prologue: {int(n):2d} tile steps; per step (us at 2.4 GHz): loads awaited + LDS write {c3[t, 1] / n / 2400:5.2f}, barrier {c3[t, 2] / n / 2400:5.2f}, "
                  f"next loads issued {c3[t, 3] / n / 2400:5.2f}, 64 products {c3[t, 4] / n / 2400:5.2f}, sum {(c3[t, 1:5].sum()) / n / 2400:5.2f}")
ls.close()
