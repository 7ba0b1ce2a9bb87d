"""BASELINE config C5 on one GPU: independent instances on one context and on several."""

import numpy as np
import pytest
import scipy.sparse as sp

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import madnlp_jl_amd as mj  # noqa: E402
from madnlp_jl_amd import _lib as L  # noqa: E402
from madnlp_jl_amd.problems import OPF_CASES, opf_shaped  # noqa: E402
from oracle import kernels as okern  # noqa: E402
from oracle import sparse_condensed as osc  # noqa: E402
from oracle.lapack_cpu import BUNCHKAUFMAN, CHOLESKY, LapackCPUSolver  # noqa: E402



def _oracle_sc(P, alg=CHOLESKY):
    k = osc.SparseCondensedKKTSystem(P.n, P.m, P.jac_I, P.jac_J, P.hess_I, P.hess_J, P.ind_ineq, P.ind_lb,
                                     P.ind_ub, lambda A: LapackCPUSolver(A, alg))
    for f in ("reg", "l_diag", "u_diag", "l_lower", "u_lower", "du_diag"):
        getattr(k, f)[:] = getattr(P, f)
    k.jac[:] = P.jac
    k.hess[:] = P.hess
    k.compress_jacobian()
    k.compress_hessian()
    okern.set_aug_diagonal(k)
    k.build_kkt()
    return k


def _hip_sc(P, ctx, alg, **opt):
    k = mj.SparseCondensedKKTSystem(P.n, P.m, P.jac_I, P.jac_J, P.hess_I, P.hess_J, P.ind_ineq, P.ind_lb, P.ind_ub,
                                    ctx=ctx, opt_linear_solver=mj.HipSolverOptions(lapack_algorithm=alg, **opt))
    for f in ("reg", "l_diag", "u_diag", "l_lower", "u_lower", "du_diag"):
        getattr(k, f)[:] = getattr(P, f)
    k.jac[:] = P.jac
    k.hess[:] = P.hess
    return k


def _full(ko):
    Kl = sp.csc_matrix((ko.aug_com.nzval, ko.aug_com.rowval, ko.aug_com.colptr), shape=(ko.n, ko.n))
    return (Kl + sp.tril(Kl, -1).T).tocsr()


def _bwd(K, x, b):
    return np.abs(K @ x - b).max() / (abs(K).sum(axis=1).max() * np.abs(x).max() + np.abs(b).max())



# --------------------------------------------------------------------------- C5: 16 scenarios on one GPU
@pytest.mark.parametrize("nctx,nb", [(1, 16), (4, 8)])
def test_c5_batch_on_one_gpu_matches_oracle(nctx, nb):
    """BASELINE config C5, one GPU's share: independent case1354pegase-shaped scenarios (seeds 1354 + i, the
    seeds bench.py uses on rank 0), driven exactly like `bench.py --batch 16` (device-resident inputs, asynchronous
    factorization, all of them enqueued before the first inertia fetch): 16 back to back on ONE context (the bench
    default) and 8 spread over 4 contexts / streams (their persistent factorizations and solves take turns on the device,
    chained by the arbiter of common.h -- every one of them on the task-DAG schedule).  Every instance: condensed KKT bit-exact vs the
    oracle, inertia (N, 0, 0), backward error of the solve <= 1e-13 against the oracle's sparse K."""
    dev = torch.device("cuda", 0)
    base = OPF_CASES["case1354pegase"][0]
    streams = [torch.cuda.Stream(dev) for _ in range(nctx)]
    ctxs = [mj.HipContext(0, stream=s.cuda_stream) for s in streams]
    insts = []
    for i in range(nb):
        P = opf_shaped("case1354pegase", seed=base + i, du=1e-8)
        kh = mj.SparseCondensedKKTSystem(P.n, P.m, P.jac_I, P.jac_J, P.hess_I, P.hess_J, P.ind_ineq, P.ind_lb, P.ind_ub,
                                         ctx=ctxs[i % nctx],
                                         opt_linear_solver=mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN))
        din = dict(jac=torch.from_numpy(P.jac).to(dev), hess=torch.from_numpy(P.hess).to(dev),
                   pr=torch.from_numpy(P.pr_diag).to(dev), du=torch.from_numpy(P.du_diag).to(dev),
                   rhs=torch.from_numpy(np.random.default_rng(base + i).standard_normal(P.n)).to(dev))
        din["x"] = torch.empty_like(din["rhs"])
        insts.append((P, kh, streams[i % nctx], din))
    torch.cuda.synchronize()
    for rep in range(2):  # twice: buffers are reused across iterations
        for (_, kh, st, din) in insts:
            with torch.cuda.stream(st):
                kh.compress_jacobian(din["jac"]); kh.compress_hessian(din["hess"]); kh.build_kkt(din["pr"], din["du"])
                kh.linear_solver.factorize_async()
        for (P, kh, st, din) in insts:
            with torch.cuda.stream(st):
                assert kh.linear_solver.inertia() == (P.n, 0, 0)
                din["x"].copy_(din["rhs"])
                kh.linear_solver.solve_linear_system(din["x"])
    torch.cuda.synchronize()
    seen = set()
    for (P, kh, st, din) in insts:
        ko = _oracle_sc(P)
        got = kh.aug_com.nzval
        np.testing.assert_array_equal(got, ko.aug_com.nzval)
        seen.add(got.tobytes()[:4096])
        K = _full(ko)
        x = din["x"].cpu().numpy()
        b = din["rhs"].cpu().numpy()
        assert _bwd(K, x, b) <= 1e-13
        kh.linear_solver.check_solve()
        assert kh.linear_solver.get_stat("panel_algo") == 5.0 and kh.linear_solver.get_stat("pp_fallbacks") == 0.0   # the task-DAG schedule on every context: persistent operations take turns (device arbiter)
    assert len(seen) == nb, "the scenarios must be different problems"
    for (_, kh, _, _) in insts:
        kh.close()
    for c in ctxs:
        c.close()


