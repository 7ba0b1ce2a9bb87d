"""Event-timed solve_linear_system! on a device vector (one-launch persistent solve) of a dense SPD-like system.
usage: python tools/solve_time.py [N] [LDL|CHOLESKY]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import madnlp_jl_amd as mj  # noqa: E402
N = int(sys.argv[1]) if len(sys.argv) > 1 else 11192
alg = sys.argv[2] if len(sys.argv) > 2 else "LDL"
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    ctx = mj.HipContext(0, stream=s.cuda_stream)
    g = torch.Generator(device="cuda").manual_seed(N)
    R = torch.randn(N, 96, dtype=torch.float64, device="cuda", generator=g)
    A = R @ R.T + torch.diag(torch.rand(N, dtype=torch.float64, device="cuda", generator=g) * 10 + 1.0)
    s.synchronize()
    ls = mj.HipLinearSolver(A, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=alg))
    ls.factorize()
    b = torch.randn(N, dtype=torch.float64, device="cuda", generator=g)
    x = b.clone()
    for _ in range(5):
        x.copy_(b); ls.solve_linear_system(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    e0.record(s)
    for _ in range(reps):
        ls.solve_linear_system(x)
    e1.record(s); s.synchronize()
    x.copy_(b); ls.solve_linear_system(x); s.synchronize()
    res = (A @ x - b).abs().max().item() / (A.abs().sum(dim=1).max().item() * x.abs().max().item() + b.abs().max().item())
    sha = ""
    if os.environ.get("SOLVE_HASH"):
        import hashlib
        sha = "  solution sha " + hashlib.sha256(x.cpu().numpy().tobytes()).hexdigest()[:16]
    print(f"N={N} {alg}: solve {e0.elapsed_time(e1)/reps:.4f} ms  backward error {res:.1e}{sha}")
