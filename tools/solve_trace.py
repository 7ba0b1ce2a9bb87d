"""Critical-path trace of the persistent solve's forward sweep (diagnostics)."""
import ctypes as C
import os
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "8")  # the pool's boxes cap the process at 16 CPUs: idle BLAS pools must not spin on 256
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import madnlp_jl_amd as mj  # noqa: E402
from madnlp_jl_amd import _lib as L  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 11192
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    ctx = mj.HipContext(0, stream=s.cuda_stream)
    R = torch.randn(N, 64, dtype=torch.float64, device="cuda")
    A = R @ R.T + N * torch.eye(N, dtype=torch.float64, device="cuda")
    ls = mj.HipLinearSolver(A, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm="LDL"))
    ls.factorize()
    L.check(L.lib().mnk_ls_set_option(ls._h, b"solve_trace", 1.0), "set_option")
    x = torch.randn(N, dtype=torch.float64, device="cuda")
    for _ in range(3):
        ls.solve_linear_system(x)
    s.synchronize()
    nb = (N + 127) // 128 * 2
    tr = np.zeros(nb * 8, dtype=np.uint64)
    L.check(L.lib().mnk_ls_debug_solve_trace(ls._h, tr.ctypes.data, tr.size), "trace")
    tr = tr.reshape(nb, 8).astype(np.int64)
    t0 = tr[0, 4]
    print("blk  wait_start  y_seen  bfin_pub  peers_seen  y_pub   (us since block 0 published y)   step period")
    prev = None
    for i in range(0, nb, 1):
        if (i % 4 in (0, 3) and (i < 24 or i >= nb - 12)) or 80 <= i < 100:
            row = (tr[i, :5] - t0) / 100.0
            per = "" if prev is None or i % 4 else f"{row[4]-prev:6.2f}"
            print(f"{i:4d} " + " ".join(f"{v:9.2f}" for v in row) + "   " + per)
        if i % 4 == 0:
            prev = (tr[i, 4] - t0) / 100.0
    fw = (tr[nb - 1, 4] - tr[0, 4]) / 100.0
    print(f"forward sweep: {fw:.1f} us for {nb//4} steps = {fw/(nb//4):.2f} us/step")
