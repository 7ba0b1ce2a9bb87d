"""Run a few factorizations + solves of an N x N SPD matrix (for rocprofv3 --kernel-trace --stats)."""
import os
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "8")  # the pool's boxes cap the process at 16 CPUs: idle BLAS pools must not spin on 256
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import madnlp_jl_amd as mj  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
alg = sys.argv[2] if len(sys.argv) > 2 else "CHOLESKY"
nbo = int(sys.argv[3]) if len(sys.argv) > 3 else 512
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    ctx = mj.HipContext(0, stream=s.cuda_stream)
    R = torch.randn(N, 64, dtype=torch.float64, device="cuda")
    A = R @ R.T + N * torch.eye(N, dtype=torch.float64, device="cuda")
    ls = mj.HipLinearSolver(A, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=alg, outer_block=nbo))
    x = torch.randn(N, dtype=torch.float64, device="cuda")
    for _ in range(reps):
        a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        a.record(s)
        ls.factorize_async() if False else ls.factorize()
        b.record(s)
        ls.solve_linear_system(x)
        c.record(s)
        s.synchronize()
        print(f"N={N} {alg} nbo={nbo}: factorize {a.elapsed_time(b):.3f} ms  solve {b.elapsed_time(c):.3f} ms  "
              f"{N**3/3/a.elapsed_time(b)/1e9:.2f} TFLOP/s")
    print("inertia", ls.inertia())
