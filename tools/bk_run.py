"""One factorization that takes the pivoted (blocked Bunch-Kaufman) tier, for kernel traces: random symmetric indefinite
matrix of order N (static pivoting breaks down on its exact zeros).  usage: python tools/bk_run.py [N] [bk_panel_wgs]"""
import time
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import madnlp_jl_amd as mj  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
rng = np.random.default_rng(N)
A = rng.standard_normal((N, N))
A = A + A.T
A[np.arange(0, N, 7), np.arange(0, N, 7)] = 0.0
st = torch.cuda.Stream()
ctx = mj.HipContext(0, stream=st.cuda_stream)
ls = mj.HipLinearSolver(torch.from_numpy(np.asfortranarray(A)).cuda(), ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN))
if len(sys.argv) > 2:
    ls.set_option("bk_panel_wgs", float(sys.argv[2]))
ls.factorize()
inert = ls.inertia()
st.synchronize()
t0 = time.perf_counter()
ls.factorize()
inert = ls.inertia()
ms = 1e3 * (time.perf_counter() - t0)
b = rng.standard_normal(N)
x = ls.solve_linear_system(b.copy())
res = np.abs(A @ x - b).max() / (np.abs(A).sum(axis=1).max() * np.abs(x).max() + np.abs(b).max())
print(f"N={N} inertia {inert} pivoted tier taken {ls.bk_info()[0]} multi-workgroup panels {ls.get_stat('bk_panel_multi')} "
      f"(fallbacks {ls.get_stat('bk_mw_fallbacks')}) transfer + both tiers + inertia {ms:.1f} ms, backward error {res:.2e}")
ls.close()
ctx.close()
