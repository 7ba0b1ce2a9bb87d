"""Microbenchmarks used while tuning (not part of the bench contract): fp64 MFMA tile
kernel rate, per-phase factorization time.  Run on the GPU box:
    python tools/microbench.py [N ...]"""
import os
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "8")  # the pool's boxes cap the process at 16 CPUs: idle BLAS pools must not spin on 256
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import madnlp_jl_amd as mj  # noqa: E402
from madnlp_jl_amd import _lib as L  # noqa: E402


def ev_time(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts)), float(np.min(ts))


def main():
    st = torch.cuda.Stream()   # a non-default stream: its handle is non-NULL and torch events see it
    torch.cuda.set_stream(st)
    ctx = mj.HipContext(0, stream=st.cuda_stream)
    lib = mj.lib()
    print("device:", torch.cuda.get_device_name(0))
    # ---- gemm_nt rate -------------------------------------------------------------
    for (M, N, K, mode) in [(8192, 8192, 512, 0), (8192, 8192, 512, 2), (8192, 8192, 64, 0), (11264, 448, 64, 0),
                            (11264, 64, 64, 1), (4096, 4096, 2048, 0), (16384, 16384, 512, 2)]:
        A = torch.randn(M * K + 512, dtype=torch.float64, device="cuda")
        B = torch.randn(N * K + 512, dtype=torch.float64, device="cuda")
        Cm = torch.randn(M * N + 512, dtype=torch.float64, device="cuda")
        f = lambda: L.check(lib.mnk_gemm_nt(ctx.handle, mode, M, N, K, A.data_ptr(), M, B.data_ptr(), N,
                                            Cm.data_ptr(), M))
        med, mn = ev_time(f)
        flops = 2.0 * M * N * K * (0.5 if mode == 2 else 1.0)
        print(f"gemm_nt mode={mode} M={M} N={N} K={K}: {med:.3f} ms (min {mn:.3f})  "
              f"{flops / med / 1e9:.1f} TFLOP/s  frac_of_78.6={flops / med / 1e9 / 78.6:.2f}")
        del A, B, Cm
    # ---- factorization ------------------------------------------------------------
    sizes = [int(a) for a in sys.argv[1:]] or [2048, 4096, 11192]
    for N in sizes:
        g = torch.Generator(device="cuda").manual_seed(1)
        R = torch.randn(N, 64, dtype=torch.float64, device="cuda", generator=g)
        Am = R @ R.T + N * torch.eye(N, dtype=torch.float64, device="cuda")
        Am = Am.t().contiguous()  # symmetric: layout irrelevant
        for alg in (mj.CHOLESKY, mj.LDL):
            for nbo in (256, 512, 1024):
                ls = mj.HipLinearSolver(Am, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=alg, outer_block=nbo))
                med, mn = ev_time(lambda: ls.factorize(), reps=3, warm=1)
                x = torch.randn(N, dtype=torch.float64, device="cuda")
                smed, smn = ev_time(lambda: ls.solve_linear_system(x), reps=3, warm=1)
                fl = N ** 3 / 3
                print(f"factorize N={N} {alg} nbo={nbo}: {med:.2f} ms (min {mn:.2f}) "
                      f"{fl / med / 1e9:.1f} TFLOP/s frac={fl / med / 1e9 / 78.6:.2f} | solve {smed:.3f} ms "
                      f"({8.0 * N * N / smed / 1e6:.0f} GB/s) inertia={ls.inertia()}")
                ls.close()
        del Am, R


if __name__ == "__main__":
    main()
