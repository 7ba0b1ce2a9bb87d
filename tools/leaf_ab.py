"""A/B of two builds of the library on the task-DAG schedule: factorize! time and a hash of the factor's bits per order /
algorithm (the four-wave leaf of round 5 against the one-wave leaf: MNK_LIBPATH=madnlp.jl_amd/lib/libmadnlp_hip_leaf1.so).
usage: [MNK_LIBPATH=...] python tools/leaf_ab.py [N ...]"""
import hashlib
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import madnlp_jl_amd as mj  # noqa: E402
Ns = [int(a) for a in sys.argv[1:]] or [2048, 6100, 11192]
s = torch.cuda.Stream()
tag = os.path.basename(os.environ.get("MNK_LIBPATH", "libmadnlp_hip.so"))
with torch.cuda.stream(s):
    ctx = mj.HipContext(0, stream=s.cuda_stream)
    for N in Ns:
        for alg in ("LDL", "CHOLESKY"):
            g = torch.Generator(device="cuda").manual_seed(N)
            R = torch.randn(N, 96, dtype=torch.float64, device="cuda", generator=g)
            A = R @ R.T + torch.diag(torch.rand(N, dtype=torch.float64, device="cuda", generator=g) * 10 + 1.0)
            if alg == "LDL":
                n1 = 2 * N // 3
                A[n1:, n1:].neg_()
            s.synchronize()
            ls = mj.HipLinearSolver(A, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=alg, panel_algo=5))
            for _ in range(3):
                ls.factorize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(10):
                ls.factorize()
            e1.record(s); s.synchronize()
            Lf, D = ls.get_factor_device()
            h = hashlib.sha256(torch.tril(Lf).contiguous().cpu().numpy().tobytes() + D.cpu().numpy().tobytes()).hexdigest()[:16]
            b = torch.randn(N, dtype=torch.float64, device="cuda", generator=g)
            x = b.clone(); s.synchronize(); ls.solve_linear_system(x); ls.check_solve(); s.synchronize()
            bw = ((A @ x - b).abs().max() / (A.abs().sum(dim=1).max() * x.abs().max() + b.abs().max())).item()
            print(f"{tag:28s} N={N:6d} {alg:8s}: factorize {e0.elapsed_time(e1)/10:8.3f} ms  inertia {ls.inertia()}  factor sha {h}  backward error {bw:.1e}  "
                  f"algo {ls.get_stat('panel_algo')} fallbacks {ls.get_stat('pp_fallbacks')}", flush=True)
            ls.close()
