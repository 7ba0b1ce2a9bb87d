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
        for idx, (P, kh, st, din) in enumerate(insts):
            with torch.cuda.stream(st):
                M = kh.linear_solver
                assert M.inertia() == (P.n, 0, 0), (rep, idx, M.get_stat("panel_algo"), M.get_stat("pp_fallbacks"),
                                                    M.get_stat("timeout_site"), M.get_stat("growth"), M.bk_info()[:2])
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
        # the task-DAG schedule on every context: persistent operations take turns (device arbiter)
        assert (kh.linear_solver.get_stat("panel_algo"), kh.linear_solver.get_stat("pp_fallbacks")) == (5.0, 0.0), kh.linear_solver.get_stat("timeout_site")
    assert len(seen) == nb, "the scenarios must be different problems"
    for (_, kh, _, _) in insts:
        kh.close()
    for c in ctxs:
        c.close()




# --------------------------------------------------------------------------- batches: one merged launch
def _make_instances(specs, ctxs, streams, dev):
    insts = []
    for i, (case, seed) in enumerate(specs):
        P = opf_shaped(case, seed=seed, du=1e-8)
        kh = mj.SparseCondensedKKTSystem(P.n, P.m, P.jac_I, P.jac_J, P.hess_I, P.hess_J, P.ind_ineq, P.ind_lb, P.ind_ub,
                                         ctx=ctxs[i % len(ctxs)],
                                         opt_linear_solver=mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN))
        din = dict(jac=torch.from_numpy(P.jac).to(dev), hess=torch.from_numpy(P.hess).to(dev),
                   pr=torch.from_numpy(P.pr_diag).to(dev), du=torch.from_numpy(P.du_diag).to(dev),
                   rhs=torch.from_numpy(np.random.default_rng(seed).standard_normal(P.n)).to(dev))
        din["x"] = torch.empty_like(din["rhs"])
        insts.append((P, kh, streams[i % len(streams)], din))
    return insts


def _front(kh, st, din):
    with torch.cuda.stream(st):
        kh.compress_jacobian(din["jac"]); kh.compress_hessian(din["hess"]); kh.build_kkt(din["pr"], din["du"])
        kh.linear_solver.factorize_async()


@pytest.mark.parametrize("nctx,nb", [(1, 5), (2, 6)])
def test_batched_factorizations_are_bit_identical_to_lone_ones(nctx, nb):
    """mnk_factorize_batch_begin / _end: the factorize! calls of a block are queued and launched
    together -- ONE bulk kernel drains the merged task queue of all instances beside TWO pivot chains (even instances on one
    CU partition, odd ones on the other; chain i + 2 follows chain i), each instance's chain-bound ends filled with its
    neighbours' trailing updates.  Independent case1354pegase-shaped scenarios (N = 11 192), on one context and spread over
    two contexts / streams; three rounds (the factor buffers alternate: from the second round on the scatter lands in the
    buffer that the previous round's queue zeroed with its DAG_FILL tasks).  Every instance and round: the task-DAG schedule
    ran, no fall-back, inertia (N, 0, 0), the factor's bits (L and D) equal those of a lone factorize! of the same matrix
    on the same solver, backward error of a solve <= 1e-13 against the oracle's sparse K."""
    dev = torch.device("cuda", 0)
    base = OPF_CASES["case1354pegase"][0]
    streams = [torch.cuda.Stream(dev) for _ in range(nctx)]
    ctxs = [mj.HipContext(0, stream=s.cuda_stream) for s in streams]
    insts = _make_instances([("case1354pegase", base + 100 + i) for i in range(nb)], ctxs, streams, dev)
    torch.cuda.synchronize()
    ref = []
    for (P, kh, st, din) in insts:      # lone factorizations first
        _front(kh, st, din)
        assert kh.linear_solver.inertia() == (P.n, 0, 0)
        Lf, D = kh.linear_solver.get_factor_device()
        ref.append((torch.tril(Lf).clone(), D.clone()))
    for rnd in range(3):
        with mj.factorize_batch():
            for (_, kh, st, din) in insts:
                _front(kh, st, din)
        for (P, kh, st, din) in insts:
            with torch.cuda.stream(st):
                assert kh.linear_solver.inertia() == (P.n, 0, 0)
                din["x"].copy_(din["rhs"])
                kh.linear_solver.solve_linear_system(din["x"])
        torch.cuda.synchronize()
        for i, (P, kh, st, din) in enumerate(insts):
            assert kh.linear_solver.get_stat("panel_algo") == 5.0 and kh.linear_solver.get_stat("pp_fallbacks") == 0.0
            Lf, D = kh.linear_solver.get_factor_device()
            assert torch.equal(torch.tril(Lf), ref[i][0]) and torch.equal(D, ref[i][1]), (rnd, i)
            kh.linear_solver.check_solve()
    for (P, kh, st, din) in insts:
        ko = _oracle_sc(P)
        np.testing.assert_array_equal(kh.aug_com.nzval, ko.aug_com.nzval)
        assert _bwd(_full(ko), din["x"].cpu().numpy(), din["rhs"].cpu().numpy()) <= 1e-13
    for (_, kh, _, _) in insts:
        kh.close()
    for c in ctxs:
        c.close()


def test_batch_with_mixed_orders_and_a_forgotten_end():
    """A batch groups what it can: two instances of N = 11 192 and two of N = 8400 form two merged launches, a case118-sized
    system (below the schedule's window) is not queued at all; and a call that needs a queued solver's factor before
    mnk_factorize_batch_end (here: `inertia`) launches what is queued instead of answering from a stale factor."""
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(dev)
    ctx = mj.HipContext(0, stream=st.cuda_stream)
    specs = [("case1354pegase", 7), ((1000, 200, 1500), 8), ("case1354pegase", 9), ((1000, 200, 1500), 10), ("case118", 11)]
    insts = _make_instances(specs, [ctx], [st], dev)
    assert [P.n for (P, *_r) in insts][:4] == [11192, 8400, 11192, 8400] and insts[4][0].n < 1536
    L.check(L.lib().mnk_factorize_batch_begin(), "begin")
    try:
        for (_, kh, s_, din) in insts:
            _front(kh, s_, din)
        # no _end yet: the inertia of a queued solver launches the queue
        assert insts[0][1].linear_solver.inertia() == (insts[0][0].n, 0, 0)
        for (_, kh, s_, din) in insts[:2]:     # a second round of two of them goes into the (still open) batch
            _front(kh, s_, din)
    finally:
        L.check(L.lib().mnk_factorize_batch_end(), "end")
    assert L.lib().mnk_factorize_batch_end() != 0          # no batch open: an error, not a crash
    for (P, kh, s_, din) in insts:
        with torch.cuda.stream(s_):
            assert kh.linear_solver.inertia() == (P.n, 0, 0)
            din["x"].copy_(din["rhs"])
            kh.linear_solver.solve_linear_system(din["x"])
    torch.cuda.synchronize()
    for (P, kh, s_, din) in insts:
        ko = _oracle_sc(P)
        np.testing.assert_array_equal(kh.aug_com.nzval, ko.aug_com.nzval)
        assert _bwd(_full(ko), din["x"].cpu().numpy(), din["rhs"].cpu().numpy()) <= 1e-13
        if P.n > 6200:
            assert kh.linear_solver.get_stat("panel_algo") == 5.0 and kh.linear_solver.get_stat("pp_fallbacks") == 0.0
        kh.close()
    ctx.close()


@pytest.mark.parametrize("nctx", [1, 2])
def test_batched_solves_are_bit_identical_to_lone_ones(nctx):
    """mnk_solve_batch_begin / _end: the single-right-hand-side solves of a block on device vectors are queued and run up to
    four independent systems per launch (one solve is bound by its chain of hops, not by HBM).  Six case1354pegase-shaped
    instances (launches of four and two) plus a SECOND right-hand side of the first instance inside the same block (it must
    come after the first one, in a later launch), on one context and on two: every solution's bits equal those of a lone
    solve of the same system (the block -> workgroup map only decides who computes what), backward error <= 1e-13."""
    dev = torch.device("cuda", 0)
    base = OPF_CASES["case1354pegase"][0]
    streams = [torch.cuda.Stream(dev) for _ in range(nctx)]
    ctxs = [mj.HipContext(0, stream=s.cuda_stream) for s in streams]
    insts = _make_instances([("case1354pegase", base + 200 + i) for i in range(6)], ctxs, streams, dev)
    torch.cuda.synchronize()
    lone = []
    for (P, kh, st, din) in insts:
        _front(kh, st, din)
        with torch.cuda.stream(st):
            assert kh.linear_solver.inertia() == (P.n, 0, 0)
            din["x"].copy_(din["rhs"])
            kh.linear_solver.solve_linear_system(din["x"])
            kh.linear_solver.check_solve()
            lone.append(din["x"].clone())
    torch.cuda.synchronize()
    rhs2 = insts[0][3]["rhs"] * 3.0 + 1.0
    x2_ref = rhs2.clone()
    with torch.cuda.stream(insts[0][2]):
        insts[0][1].linear_solver.solve_linear_system(x2_ref)
        insts[0][1].linear_solver.check_solve()
    x2 = rhs2.clone()
    torch.cuda.synchronize()
    for rnd in range(3):          # (the publication buffers of every solver alternate from solve to solve)
        for (_, kh, st, din) in insts:
            with torch.cuda.stream(st):
                din["x"].copy_(din["rhs"])
        x2.copy_(rhs2)
        torch.cuda.synchronize()
        with mj.solve_batch():
            for (_, kh, st, din) in insts:
                with torch.cuda.stream(st):
                    kh.linear_solver.solve_linear_system(din["x"])
            with torch.cuda.stream(insts[0][2]):
                insts[0][1].linear_solver.solve_linear_system(x2)
        for (_, kh, st, din) in insts:
            kh.linear_solver.check_solve()
        torch.cuda.synchronize()
        for i, (P, kh, st, din) in enumerate(insts):
            assert torch.equal(din["x"], lone[i]), (rnd, i)
        assert torch.equal(x2, x2_ref), rnd
    for (P, kh, st, din) in insts:
        ko = _oracle_sc(P)
        assert _bwd(_full(ko), din["x"].cpu().numpy(), din["rhs"].cpu().numpy()) <= 1e-13
        kh.close()
    for c in ctxs:
        c.close()


def test_batched_solves_of_small_and_mixed_systems():
    """Orders that leave fewer 64-row blocks than a launch's share of the CUs, two different orders in one block (two groups),
    a host-vector solve inside the block (not queued: runs at once, after the queued solves of its solver)."""
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(dev)
    ctx = mj.HipContext(0, stream=st.cuda_stream)
    g = torch.Generator(device=dev).manual_seed(5)
    sols, mats, refs, rhs = [], [], [], []
    for N in (3001, 3001, 3001, 1700, 1700):
        R = torch.randn(N, 40, dtype=torch.float64, device=dev, generator=g)
        A = R @ R.T
        A.diagonal().add_(float(N))
        n1 = 2 * N // 3
        A[n1:, n1:].neg_()
        mats.append(A)
        rhs.append(torch.randn(N, dtype=torch.float64, device=dev, generator=g))
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        for A, b in zip(mats, rhs):
            M = mj.HipLinearSolver(A, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=mj.LDL))
            M.factorize()
            x = b.clone()
            M.solve_linear_system(x)
            M.check_solve()
            sols.append(M)
            refs.append(x.clone())
        xs = [b.clone() for b in rhs]
        st.synchronize()
        with mj.solve_batch():
            for M, x in zip(sols, xs):
                M.solve_linear_system(x)
            xh = sols[1].solve_linear_system(rhs[1].cpu().numpy().copy())     # host vector: at once, behind solver 1's queued solve
        for M in sols:
            M.check_solve()
        st.synchronize()
        for x, r in zip(xs, refs):
            assert torch.equal(x, r)
        assert np.array_equal(xh, refs[1].cpu().numpy())
        for M in sols:
            M.close()
    ctx.close()


def test_scenario_batch_array_entry_points_match_the_per_instance_calls():
    """`ScenarioBatch` (mnk_sc_step_batch / mnk_ls_inertia_batch / mnk_ls_solve_batch): one library call per phase of an
    iteration for n independent instances.  Five case1354pegase-shaped scenarios, two rounds: the condensed matrix of every
    instance bit-exact vs the oracle, inertia, the solution's bits equal those of the per-instance calls on the same solver,
    backward error <= 1e-13."""
    dev = torch.device("cuda", 0)
    base = OPF_CASES["case1354pegase"][0]
    st = torch.cuda.Stream(dev)
    ctx = mj.HipContext(0, stream=st.cuda_stream)
    insts = _make_instances([("case1354pegase", base + 300 + i) for i in range(5)], [ctx], [st], dev)
    torch.cuda.synchronize()
    ref = []
    for (P, kh, s_, din) in insts:
        _front(kh, s_, din)
        with torch.cuda.stream(s_):
            assert kh.linear_solver.inertia() == (P.n, 0, 0)
            din["x"].copy_(din["rhs"])
            kh.linear_solver.solve_linear_system(din["x"])
            kh.linear_solver.check_solve()
            ref.append(din["x"].clone())
    sb = mj.ScenarioBatch([it[1] for it in insts])
    sb.bind([it[3]["jac"] for it in insts], [it[3]["hess"] for it in insts], [it[3]["pr"] for it in insts],
            [it[3]["du"] for it in insts])
    xs = [it[3]["x"] for it in insts]
    for rnd in range(2):
        sb.step()
        assert sb.inertia() == [(P.n, 0, 0) for (P, *_r) in insts]
        with torch.cuda.stream(st):
            torch._foreach_copy_(xs, [it[3]["rhs"] for it in insts])
        st.synchronize()
        sb.solve(xs)
        for (_, kh, _, _) in insts:
            kh.linear_solver.check_solve()
            assert kh.linear_solver.get_stat("panel_algo") == 5.0 and kh.linear_solver.get_stat("pp_fallbacks") == 0.0
        torch.cuda.synchronize()
        for x, r in zip(xs, ref):
            assert torch.equal(x, r), rnd
    for (P, kh, s_, din) in insts:
        ko = _oracle_sc(P)
        np.testing.assert_array_equal(kh.aug_com.nzval, ko.aug_com.nzval)
        assert _bwd(_full(ko), din["x"].cpu().numpy(), din["rhs"].cpu().numpy()) <= 1e-13
        kh.close()
    ctx.close()


@pytest.mark.parametrize("n,m,n_eq,count", [(2048, 512, 0, 7), (2048, 512, 64, 6), (1900, 300, 60, 3), (500, 120, 0, 9), (700, 100, 30, 5)])
def test_batched_small_factorizations(n, m, n_eq, count):
    """Batches of SMALL systems (reference: the scenario loop `src/KKT/Schur/schur.jl:927-1001`, the uniform
    batch of `lib/MadNLPGPU/ext/MadNLPGPUCUDAExt/cudss.jl:139-143`): a system of up to 6144 rows is bound by its own pivot
    chain and uses Np / 64 CUs; in a batch the chains of several systems run side by side in ONE launch (pchain_multi_kernel),
    their band tiles accumulated by one bulk launch.  Dense condensed KKT systems (DenseDummyQP shapes, with and without
    equality rows): inertia (n, 0, n_eq); systems inside the schedule's window (N >= 1536): factor bits equal those of a lone
    factorize!; below it (a lone factorize! takes the launch-per-panel schedule there): solutions agree to 1e-10 and the
    backward error is <= 1e-13."""
    from madnlp_jl_amd.problems import dense_dummy_qp
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(dev)
    ctx = mj.HipContext(0, stream=st.cuda_stream)
    rng = np.random.default_rng(n + count)
    ks, Ks, bs = [], [], []
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
        bs.append(rng.standard_normal(k.linear_solver.n))
    N = ks[0].linear_solver.n
    lone = []
    for k, b in zip(ks, bs):
        k.linear_solver.factorize_async()
        assert k.linear_solver.inertia() == (n, 0, n_eq)
        Lf, D = k.linear_solver.get_factor_device()
        lone.append((torch.tril(Lf).clone(), D.clone(), k.linear_solver.solve_linear_system(b.copy())))
        Ks.append(k.aug_com.to_host())
    for rnd in range(2):
        with mj.factorize_batch():
            for k in ks:
                k.linear_solver.factorize_async()
        for i, (k, b) in enumerate(zip(ks, bs)):
            assert k.linear_solver.inertia() == (n, 0, n_eq), (rnd, i)
            assert k.linear_solver.get_stat("panel_algo") == 5.0 and k.linear_solver.get_stat("pp_fallbacks") == 0.0
            x = k.linear_solver.solve_linear_system(b.copy())
            Kl = np.tril(Ks[i])
            Kf = Kl + np.tril(Kl, -1).T
            res = np.abs(Kf @ x - b).max() / (np.abs(Kf).sum(axis=1).max() * np.abs(x).max() + np.abs(b).max())
            assert res <= 1e-13, (rnd, i, res)
            Lf, D = k.linear_solver.get_factor_device()
            if N >= 1536:
                assert torch.equal(torch.tril(Lf), lone[i][0]) and torch.equal(D, lone[i][1]), (rnd, i)
                assert np.array_equal(x, lone[i][2])
            else:
                assert np.abs(x - lone[i][2]).max() <= 1e-10 * np.abs(x).max()
    for k in ks:
        k.close()
    ctx.close()
