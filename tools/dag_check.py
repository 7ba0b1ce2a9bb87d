"""A/B of the factorization schedules on dense matrices: panel_algo 5 (task-DAG, dag.hip) against 4 (one trailing update
per outer panel): backward error of a solve, inertia, factor bits (repeatability), time per factorize!.
usage: python tools/dag_check.py [N ...]   (env MNK_* overrides apply)"""
import os
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "8")
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import madnlp_jl_amd as mj  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [3200, 4096, 5000, 11192]
reps = int(os.environ.get("DAG_REPS", "4"))
algos = [int(a) for a in os.environ.get("DAG_ALGOS", "4,5").split(",")]
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    ctx = mj.HipContext(0, stream=s.cuda_stream)
    for N in sizes:
        for alg in [a for a in ("LDL", "CHOLESKY") if os.environ.get("DAG_ONLY", a) == a]:
            g = torch.Generator(device="cuda").manual_seed(N)
            R = torch.randn(N, 96, dtype=torch.float64, device="cuda", generator=g)
            A = R @ R.T + torch.diag(torch.rand(N, dtype=torch.float64, device="cuda", generator=g) * 10 + 1.0)
            if alg == "LDL" and not os.environ.get("DAG_SPD"):  # quasi-definite: negative trailing block
                h = N // 3
                A[N - h:, N - h:] = -(A[N - h:, N - h:] + 2.0 * torch.eye(h, dtype=torch.float64, device="cuda") * N ** 0.5)
            An = torch.linalg.matrix_norm(A, ord=float("inf")).item()
            b = torch.randn(N, dtype=torch.float64, device="cuda", generator=g)
            ref = None
            for pa in algos:
                ls = mj.HipLinearSolver(A, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=alg, panel_algo=pa, single_rows=int(os.environ.get("DAG_SINGLE", "0")) if pa == 5 else int(os.environ.get("DAG_SINGLE4", "0"))))
                ls.set_option("dag_min_rows", 0)
                ts = []
                facs = []
                for _ in range(reps):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(s)
                    ls.factorize()
                    e1.record(s)
                    s.synchronize()
                    ts.append(e0.elapsed_time(e1))
                    facs.append(np.tril(ls.get_factor()[0]) if N <= 5000 and len(facs) < 2 else None)
                x = b.clone()
                ls.solve_linear_system(x)
                s.synchronize()
                res = (torch.abs(A @ x - b).max() / (An * torch.abs(x).max() + torch.abs(b).max())).item()
                same = all(f is None or np.array_equal(f, facs[0]) for f in facs[1:])
                inert = ls.inertia()
                used = ls.get_stat("panel_algo")
                print(f"N={N:6d} {alg:8s} algo={pa} (ran {used:.0f}): factorize min {min(ts):8.3f} ms  med {sorted(ts)[len(ts)//2]:8.3f} ms "
                      f"{N**3/3/min(ts)/1e9:6.2f} TF/s  bwd err {res:.2e}  inertia {inert}  repeatable {same}", flush=True)
                if ref is None:
                    ref = inert
                elif inert != ref:
                    print("   !!! inertia differs from the first algorithm", flush=True)
                ls.close()
