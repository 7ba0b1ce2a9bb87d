"""Other BASELINE.json configurations, for the record (not the bench.py contract line):
  C2  synthetic DenseCondensedKKTSystem n=2048, m=512 (n_eq = 0 and 64)
  C4  case9241pegase-shaped sparse condensed KKT, N = 85568 (58.7 GB factor)
usage: python tools/bench_configs.py [c2] [c4]"""
import json
import os
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "8")  # the pool's boxes cap the process at 16 CPUs: idle BLAS pools must not spin on 256
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import madnlp_jl_amd as mj  # noqa: E402
from madnlp_jl_amd.problems import dense_dummy_qp, opf_shaped  # noqa: E402


def timeit(fn, sync, reps=5, warm=1):
    for _ in range(warm):
        fn()
    sync()
    ts = []
    for _ in range(reps):
        sync()
        t0 = time.perf_counter()
        fn()
        sync()
        ts.append(1e3 * (time.perf_counter() - t0))
    return float(np.mean(ts)), float(np.min(ts))


def c2(ctx, sync):
    for n_eq in (0, 64):
        P = dense_dummy_qp(2048, 512, n_eq)
        k = mj.DenseCondensedKKTSystem(P.n, P.m, P.ind_ineq, P.ind_eq, P.ind_lb, P.ind_ub, ctx=ctx)
        for f in ("reg", "l_diag", "u_diag", "l_lower", "u_lower", "du_diag"):
            getattr(k, f)[:] = getattr(P, f)
        k.hess[...] = P.hess
        k.jac[...] = P.jac
        k.set_aug_diagonal()
        k._upload()
        lib = mj.lib()
        dpr, ddu = torch.from_numpy(k.pr_diag).cuda(), torch.from_numpy(k.du_diag).cuda()
        build = lambda: mj._lib.check(lib.mnk_dc_build(k._h, dpr.data_ptr(), ddu.data_ptr(), 1))  # noqa: E731
        tb = timeit(build, sync)
        tf = timeit(k.linear_solver.factorize_async, sync)
        x = torch.randn(k._order, dtype=torch.float64, device="cuda")
        ts = timeit(lambda: k.linear_solver.solve_linear_system(x), sync)
        N = k._order
        print(json.dumps({"config": f"C2 dense-condensed n=2048 m=512 n_eq={n_eq} (N={N})", "ms_build": tb[0],
                          "ms_factorize": tf[0], "ms_factorize_min": tf[1], "ms_solve": ts[0], "inertia": k.linear_solver.inertia(),
                          "build_tflops": 512 * 2048 ** 2 / tb[0] / 1e9, "fact_tflops": N ** 3 / 3 / tf[0] / 1e9,
                          "it_per_s_nf1_ns2": 1e3 / (tb[0] + tf[0] + 2 * ts[0])}))
        k.close()


def c4(ctx, sync):
    t0 = time.time()
    P = opf_shaped("case9241pegase", du=1e-8)
    k = mj.SparseCondensedKKTSystem(P.n, P.m, P.jac_I, P.jac_J, P.hess_I, P.hess_J, P.ind_ineq, P.ind_lb, P.ind_ub,
                                    ctx=ctx, opt_linear_solver=mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN,
                                                                                       outer_block=int(os.environ.get("C4_OUTER_BLOCK", "0"))))
    setup = time.time() - t0
    if os.environ.get("C4_DAG_MAX_ROWS"):   # (experiment: the task-DAG schedule above its shipped upper bound of 24 576 rows)
        k.linear_solver.set_option("dag_max_rows", int(os.environ["C4_DAG_MAX_ROWS"]))
    dj, dh = torch.from_numpy(P.jac).cuda(), torch.from_numpy(P.hess).cuda()
    dp, dd = torch.from_numpy(P.pr_diag).cuda(), torch.from_numpy(P.du_diag).cuda()

    def assemble():
        k.compress_jacobian(dj); k.compress_hessian(dh); k.build_kkt(dp, dd)
    ta = timeit(assemble, sync, reps=3)
    tf = timeit(k.linear_solver.factorize_async, sync, reps=2, warm=1)
    inertia = k.linear_solver.inertia()
    x = torch.randn(P.n, dtype=torch.float64, device="cuda")
    b = x.clone()
    ts = timeit(lambda: k.linear_solver.solve_linear_system(x), sync, reps=2, warm=0)
    N = P.n
    print(json.dumps({"config": f"C4 case9241pegase-shaped sparse-condensed N={N} nnzK={k.nnz_aug} len_jptr={k.len_jptr}",
                      "setup_s": setup, "ms_assemble": ta[0], "ms_factorize": tf[0], "ms_solve": ts[0],
                      "inertia": inertia, "schedule_panel_algo": k.linear_solver.get_stat("panel_algo"), "outer_block": int(os.environ.get("C4_OUTER_BLOCK", "0")), "pp_fallbacks": k.linear_solver.get_stat("pp_fallbacks"),
                      "fact_tflops": N ** 3 / 3 / tf[0] / 1e9,
                      "frac_fp64_peak": N ** 3 / 3 / tf[0] / 1e9 / 78.6,
                      "it_per_s_nf1_ns2": 1e3 / (ta[0] + tf[0] + 2 * ts[0])}))
    k.close()


if __name__ == "__main__":
    st = torch.cuda.Stream()
    torch.cuda.set_stream(st)
    ctx = mj.HipContext(0, stream=st.cuda_stream)
    sync = torch.cuda.synchronize
    which = sys.argv[1:] or ["c2", "c4"]
    if "c2" in which:
        c2(ctx, sync)
    if "c4" in which:
        c4(ctx, sync)
