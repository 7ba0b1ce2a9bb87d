"""Batches of small factorizations (mnk_factorize_batch_*: the pivot chains of several systems side by side in one launch):
`count` dense-condensed KKT systems of order n (DenseDummyQP shapes, config C2 = 2048 / 512) factorized one after the other
and as one batch; event-timed.  usage: python tools/bench_small_batch.py [n m n_eq count]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import madnlp_jl_amd as mj  # noqa: E402
from madnlp_jl_amd.problems import dense_dummy_qp  # noqa: E402

n, m, n_eq, count = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (2048, 512, 0, 16)
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(dev)
torch.cuda.set_stream(st)
ctx = mj.HipContext(0, stream=st.cuda_stream)
ks = []
for i in range(count):
    P = dense_dummy_qp(n=n, m=m, n_eq=n_eq, seed=10 + i)
    k = mj.DenseCondensedKKTSystem(P.n, P.m, P.ind_ineq, P.ind_eq, P.ind_lb, P.ind_ub, ctx=ctx,
                                   opt_linear_solver=mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN))
    for f in ("reg", "l_diag", "u_diag", "l_lower", "u_lower", "du_diag"):
        getattr(k, f)[:] = getattr(P, f)
    k.hess[...] = P.hess
    k.jac[...] = P.jac
    k.set_aug_diagonal()
    k.build_kkt()
    ks.append(k)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        out.append(a.elapsed_time(b))
    return sum(out) / len(out)


def one_by_one():
    for k in ks:
        k.linear_solver.factorize_async()


def batched():
    with mj.factorize_batch():
        for k in ks:
            k.linear_solver.factorize_async()


t1 = timed(one_by_one)
t2 = timed(batched)
ok = all(k.linear_solver.inertia() == (n, 0, n_eq) for k in ks)
print(json.dumps({"config": f"{count} dense-condensed KKT systems n={n} m={m} n_eq={n_eq} (order {n + n_eq}), BUNCHKAUFMAN tier 1",
                  "ms_one_by_one": t1, "ms_batched": t2, "ms_per_system_batched": t2 / count, "inertia_ok": ok,
                  "schedule": ks[0].linear_solver.get_stat("panel_algo")}))
