"""Event-timed factorize! of a dense quasi-definite matrix with the task-DAG schedule (diagnostic runs: results not checked).
usage: python tools/dag_time.py [N] [LDL|CHOLESKY]"""
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
    ls = mj.HipLinearSolver(A, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=alg, panel_algo=5))
    if os.environ.get("DAG_BAND"):
        ls.set_option("dag_band", int(os.environ["DAG_BAND"]))
    if os.environ.get("DAG_JS2"):   # strip-column at which every remaining row joins the band (a large value: never -- band + bulk kernel throughout)
        ls.set_option("dag_js2", int(os.environ["DAG_JS2"]))
    if os.environ.get("DAG_MIN_ROWS"):
        ls.set_option("dag_min_rows", int(os.environ["DAG_MIN_ROWS"]))
    for _ in range(3):
        ls.factorize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(10):
        ls.factorize()
    e1.record(s); s.synchronize()
    print(f"N={N} {alg} band={os.environ.get('DAG_BAND','16')} cus={os.environ.get('MNK_DAG_CUS','16')}: factorize {e0.elapsed_time(e1)/10:.3f} ms  panel_algo {ls.get_stat('panel_algo')} fallbacks {ls.get_stat('pp_fallbacks')} bulk_wgs {ls.get_stat('dag_bulk_wgs')} js2 {ls.get_stat('dag_js2')}")
