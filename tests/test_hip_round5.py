"""GPU parity tests added in round 5 (all through the C ABI).

  * a solve on the pivoted (Bunch-Kaufman) tier uses the first publication buffer of the one-launch solve
    as scratch; the next static-tier solve must not poll stale words;
  * the Schur stage's forward / backward steps inside a solve batch the CALLER opened;
  * VERDICT r4 "weak" 1: the step `bench.py --batch 16` / `--gpus N` actually runs -- 16 case1354pegase-shaped instances
    through `factorize_batch` + `ScenarioBatch.step / inertia / solve` + `solve_batch`, on one and on four contexts --
    against the oracle (condensed matrix bit-exact, inertia, backward error) and against lone factorizations (factor bits);
  * VERDICT r4 "weak" 3: the Schur stage at the sizes its timing records quote (ns = 16 and 128, blk 512, nd 256).

Tolerances (fp64): backward errors <= 1e-13 relative to |A||x| + |b| for the static-pivot tier, 1e-11 for the pivoted tier
and the Schur stage (two different factorizations of indefinite blocks); integer results (inertia) exact; bit equality
where the schedule promises it."""
import numpy as np
import pytest
import scipy.sparse as sp

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import madnlp_jl_amd as mj  # noqa: E402
from madnlp_jl_amd.problems import OPF_CASES, opf_shaped  # noqa: E402
from oracle.lapack_cpu import BUNCHKAUFMAN, LapackCPUSolver  # noqa: E402
from oracle.schur import SchurDenseStage as OracleStage  # noqa: E402
from tests.test_hip_c5 import _bwd, _full, _make_instances, _oracle_sc  # noqa: E402
from tests.test_hip_round4 import _bwd_sym, _not_quasi_definite_sparse  # noqa: E402
from tests.test_schur import assemble, two_stage_blocks  # noqa: E402


@pytest.fixture(scope="module")
def ctx():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    c = mj.HipContext(0)
    yield c
    c.close()


def _quasi_definite_sparse(rng, n1, n2):
    """[[H, B'], [B, -C]] with H, C positive diagonal-dominant: the static-pivot LDL' exists (inertia (n1, 0, n2)); the
    SAME sparsity pattern family as `_not_quasi_definite_sparse` plus a leading diagonal, so one solver serves both."""
    A = _not_quasi_definite_sparse(rng, n1, n2).tolil()
    for j in range(n1):
        A[j, j] = rng.uniform(2.0, 3.0)
    for i in range(n2):
        A[n1 + i, n1 + i] = -rng.uniform(2.0, 3.0)
    A = sp.csc_matrix(A)
    A.sort_indices()
    return A


def test_static_then_pivoted_then_static_solves_on_device_vectors(ctx):
    """ADVICE r4 (high).  The one-launch solve trusts a host flag that says "publication buffer b holds the sentinel"; the
    pivoted tier's solve permutes through the memory of buffer 0.  Sequence on ONE solver (BUNCHKAUFMAN, the default):
    static factor + two device-vector solves (buffer 0 is then booked clean and next in line), a matrix that takes the
    pivoted tier + solve (scratch lands in buffer 0), static factor + solve -- every solution against the matrix (backward
    error) and against dsytrs of the oracle's LapackCPUSolver."""
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(77)
    n1, n2 = 300, 420
    N = n1 + n2
    Aq, Ap = _quasi_definite_sparse(rng, n1, n2), _not_quasi_definite_sparse(rng, n1, n2)

    def triple(A):
        return (A.indptr.copy(), A.indices.copy(), A.data.copy())

    def ref_solve(A, b):
        dense = np.asfortranarray((A + sp.tril(A, -1).T).toarray())
        return LapackCPUSolver(dense, BUNCHKAUFMAN).factorize().solve_linear_system(b.copy())

    M = mj.HipLinearSolver(triple(Aq), ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN))
    for rnd in range(2):
        # static tier, an EVEN number of one-launch solves
        M.A = triple(Aq)
        M.factorize()
        assert not M.bk_info()[0] and M.inertia() == (n1, 0, n2)
        for i in range(2):
            b = rng.standard_normal(N)
            x = torch.from_numpy(b.copy()).to(dev)
            torch.cuda.synchronize()
            M.solve_linear_system(x)
            M.check_solve()
            assert _bwd_sym(Aq, x.cpu().numpy(), b) <= 1e-13
        # pivoted tier: its permutation scratch is the first publication buffer
        M.A = triple(Ap)
        M.factorize()
        assert M.bk_info()[0], "this matrix must take the pivoted tier"
        b = rng.standard_normal(N)
        x = torch.from_numpy(b.copy()).to(dev)
        torch.cuda.synchronize()
        M.solve_linear_system(x)
        M.check_solve()
        assert _bwd_sym(Ap, x.cpu().numpy(), b) <= 1e-11
        # back on the static tier: the next one-launch solve polls buffer 0
        M.A = triple(Aq)
        M.factorize()
        assert not M.bk_info()[0]
        for i in range(3):
            b = rng.standard_normal(N)
            x = torch.from_numpy(b.copy()).to(dev)
            torch.cuda.synchronize()
            M.solve_linear_system(x)
            M.check_solve()
            xs = x.cpu().numpy()
            assert _bwd_sym(Aq, xs, b) <= 1e-13, (rnd, i, _bwd_sym(Aq, xs, b))
            xr = ref_solve(Aq, b)
            assert np.abs(xs - xr).max() <= 1e-9 * np.abs(xr).max(), (rnd, i)
    M.close()


def _schur_residual(A, Cs, S0, xk, xd, bk, bd):
    """Backward error of the block-arrow system without assembling it (ns x blk + nd rows)."""
    ns = len(A)
    num, rowsum = 0.0, 0.0
    rd = S0 @ xd - bd
    rs_d = np.abs(S0).sum(axis=1)
    for k in range(ns):
        r = A[k] @ xk[k] + Cs[k].T @ xd - bk[k]
        num = max(num, np.abs(r).max())
        rowsum = max(rowsum, (np.abs(A[k]).sum(axis=1) + np.abs(Cs[k]).sum(axis=0)).max())
        rd += Cs[k] @ xk[k]
        rs_d += np.abs(Cs[k]).sum(axis=1)
    num = max(num, np.abs(rd).max())
    rowsum = max(rowsum, rs_d.max())
    xmax = max(np.abs(xk).max(), np.abs(xd).max())
    return num / (rowsum * xmax + max(np.abs(bk).max(), np.abs(bd).max()))


def test_schur_stage_inside_a_solve_batch_of_the_caller(ctx):
    """ADVICE r4 (medium).  mnk_schur_forward / _backward queue their scenarios' solves in a solve batch of their own and
    read the results right behind it; begin / end pairs nest, so inside a batch the CALLER opened their `end` used to be a
    no-op and the contribution / subtraction kernels ran on unsolved vectors.  The whole Schur solve inside
    `with mj.solve_batch():`, together with an unrelated solver's queued solve: same bits as outside the batch."""
    from madnlp_jl_amd.schur import SchurDenseStage
    dev = torch.device("cuda", 0)
    ns, nv, nc, nd = 6, 300, 84, 64
    A, Cs, S0, blk = two_stage_blocks(ns, nv, nc, nd, seed=19)
    sh = SchurDenseStage(A, Cs, S0, nd, blk, ctx=ctx)
    sh.build_kkt()
    sh.factorize_kkt()
    assert sh.inertia() == (nd, 0, 0)
    rng = np.random.default_rng(3)
    b = rng.standard_normal(ns * blk + nd)
    bk, bd = b[:ns * blk].reshape(ns, blk), b[ns * blk:]
    rk0, rd0 = torch.from_numpy(bk.copy()).to(dev), torch.from_numpy(bd.copy()).to(dev)
    sh.solve(rk0, rd0)
    # an unrelated solver with a solve of its own in the caller's batch
    g = torch.Generator(device=dev).manual_seed(5)
    R = torch.randn(1700, 40, dtype=torch.float64, device=dev, generator=g)
    Ao = R @ R.T
    Ao.diagonal().add_(1700.0)
    bo = torch.randn(1700, dtype=torch.float64, device=dev, generator=g)
    torch.cuda.synchronize()
    Mo = mj.HipLinearSolver(Ao, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=mj.CHOLESKY))
    Mo.factorize()
    xo_ref = bo.clone()
    Mo.solve_linear_system(xo_ref)
    Mo.check_solve()
    rk1, rd1 = torch.from_numpy(bk.copy()).to(dev), torch.from_numpy(bd.copy()).to(dev)
    xo = bo.clone()
    torch.cuda.synchronize()
    with mj.solve_batch():
        Mo.solve_linear_system(xo)          # queued by the caller's batch
        sh.solve(rk1, rd1)                  # the stage's inner batches must still run before their results are read
    Mo.check_solve()
    torch.cuda.synchronize()
    assert torch.equal(rk1, rk0) and torch.equal(rd1, rd0)
    assert torch.equal(xo, xo_ref)
    assert _schur_residual(A, Cs, S0, rk1.cpu().numpy(), rd1.cpu().numpy(), bk, bd) <= 1e-11
    Mo.close()
    sh.close()


@pytest.mark.parametrize("ns,blk,nd", [(16, 512, 256), (128, 512, 256)])
def test_hip_schur_stage_at_the_sizes_of_its_timing_records(ctx, ns, blk, nd):
    """VERDICT r4 "weak" 3 (reference src/KKT/Schur/schur.jl:927-1058).  `profiles/r0x_schur_stage.jsonl` quote ns = 16 and
    ns = 128 scenario blocks of order 512 with 256 design variables: grouped launches per 64-column step, per-lane partial
    sums, more than 32 systems = several solve launches.  At exactly those sizes: S within 1e-11 |S| of the oracle's (host:
    ns Bunch-Kaufman factorizations of order 512), every scenario's inertia, the inertia of S, and the backward error of the
    solution of the block-arrow system (computed block by block -- the assembled matrix of ns = 128 would be 35 GB)."""
    from madnlp_jl_amd.schur import SchurDenseStage
    nv, nc = blk * 3 // 4, blk - blk * 3 // 4
    A, Cs, S0, blk_ = two_stage_blocks(ns, nv, nc, nd, seed=ns + nd)
    assert blk_ == blk
    so = OracleStage(A, Cs, S0)
    S_o = so.build_local()
    sh = SchurDenseStage(A, Cs, S0, nd, blk, ctx=ctx)
    for rnd in range(2):       # (twice: the handle's buffers are reused)
        S_h = sh.build_kkt().cpu().numpy().reshape((nd, nd), order="F")
        assert np.abs(S_h - S_o).max() <= 1e-11 * np.abs(S_o).max(), (rnd, np.abs(S_h - S_o).max() / np.abs(S_o).max())
        for k in range(ns):
            assert sh.scenario_inertia(k) == (nv, 0, nc), (rnd, k)
        sh.factorize_kkt()
        assert sh.inertia() == so.factorize(S_o) == (nd, 0, 0)
        rng = np.random.default_rng(5 + rnd)
        b = rng.standard_normal(ns * blk + nd)
        bk, bd = b[:ns * blk].reshape(ns, blk), b[ns * blk:]
        rk, rd = torch.from_numpy(bk.copy()).cuda(), torch.from_numpy(bd.copy()).cuda()
        sh.solve(rk, rd)
        xk, xd = rk.cpu().numpy(), rd.cpu().numpy()
        res = _schur_residual(A, Cs, S0, xk, xd, bk, bd)
        assert res <= 1e-11, (rnd, res)
        # the oracle's solve of the same system (conditioning-limited forward agreement)
        rko, rdo = bk.copy(), bd.copy()
        rdo += so.forward(rko); so.solve_s(rdo); so.backward(rko, rdo)
        scale = max(np.abs(rko).max(), np.abs(rdo).max())
        assert max(np.abs(xk - rko).max(), np.abs(xd - rdo).max()) <= 1e-6 * scale
    sh.close()


@pytest.mark.parametrize("nctx", [1, 4])
def test_the_multi_instance_step_of_bench_py_with_16_instances(nctx):
    """VERDICT r4 "weak" 1 (reference: concurrent distinct instances, src/KKT/Schur/schur.jl:953).  The step
    `bench.py --batch 16` runs on every rank of `--gpus N`, statement for statement: 16 case1354pegase-shaped instances (seeds
    1354 + i), one `ScenarioBatch` per context, all of them inside ONE `factorize_batch` (16 task queues merged with period
    ntile / 2, two pivot chains at a time), the inertia of every instance, then the solves of all instances inside ONE
    `solve_batch` (four systems per launch); two rounds.  Checked per instance: the condensed matrix bit-exact vs the oracle,
    inertia (N, 0, 0), schedule 5 without a fall-back, the factor's bits (L and D) equal to those of a LONE factorize! of the
    same matrix on the same solver, the solution's bits equal to a lone solve's, backward error <= 1e-13 against the oracle's K."""
    dev = torch.device("cuda", 0)
    nb = 16
    base = OPF_CASES["case1354pegase"][0]
    streams = [torch.cuda.Stream(dev) for _ in range(nctx)]
    ctxs = [mj.HipContext(0, stream=s.cuda_stream) for s in streams]
    insts = _make_instances([("case1354pegase", base + i) for i in range(nb)], ctxs, streams, dev)
    torch.cuda.synchronize()
    # lone factorizations and solves first: the bits the batch must reproduce
    refL, refD, refx = [], [], []
    for (P, kh, st, din) in insts:
        with torch.cuda.stream(st):
            kh.compress_jacobian(din["jac"]); kh.compress_hessian(din["hess"]); kh.build_kkt(din["pr"], din["du"])
            kh.linear_solver.factorize_async()
            assert kh.linear_solver.inertia() == (P.n, 0, 0)
            Lf, D = kh.linear_solver.get_factor_device()
            refL.append(torch.tril(Lf).clone()); refD.append(D.clone())
            din["x"].copy_(din["rhs"])
            kh.linear_solver.solve_linear_system(din["x"])
            kh.linear_solver.check_solve()
            refx.append(din["x"].clone())
    torch.cuda.synchronize()
    sbatches = []
    for (c, st) in zip(ctxs, streams):
        mine = [it for it in insts if it[2] is st]
        sb = mj.ScenarioBatch([it[1] for it in mine])
        sb.bind([it[3]["jac"] for it in mine], [it[3]["hess"] for it in mine], [it[3]["pr"] for it in mine],
                [it[3]["du"] for it in mine])
        sbatches.append((sb, st, mine, [it[3]["x"] for it in mine], [it[3]["rhs"] for it in mine]))
    for rnd in range(2):
        with mj.factorize_batch():
            for (sb, st, mine, xs, rhss) in sbatches:
                sb.step()
        for (sb, st, mine, xs, rhss) in sbatches:
            assert sb.inertia() == [(it[0].n, 0, 0) for it in mine], rnd
        with mj.solve_batch():
            for (sb, st, mine, xs, rhss) in sbatches:
                with torch.cuda.stream(st):
                    torch._foreach_copy_(xs, rhss)
                sb.solve(xs)
        torch.cuda.synchronize()
        for i, (P, kh, st, din) in enumerate(insts):
            M = kh.linear_solver
            M.check_solve()
            assert (M.get_stat("panel_algo"), M.get_stat("pp_fallbacks")) == (5.0, 0.0), (rnd, i, M.get_stat("timeout_site"))
            Lf, D = M.get_factor_device()
            assert torch.equal(torch.tril(Lf), refL[i]) and torch.equal(D, refD[i]), (rnd, i)
            del Lf, D
            assert torch.equal(din["x"], refx[i]), (rnd, i)
    seen = set()
    for (P, kh, st, din) in insts:
        ko = _oracle_sc(P)
        got = kh.aug_com.nzval
        np.testing.assert_array_equal(got, ko.aug_com.nzval)
        seen.add(got.tobytes()[:4096])
        assert _bwd(_full(ko), din["x"].cpu().numpy(), din["rhs"].cpu().numpy()) <= 1e-13
    assert len(seen) == nb, "the scenarios must be different problems"
    for (_, kh, _, _) in insts:
        kh.close()
    for c in ctxs:
        c.close()


def test_a_foreign_pageable_copy_beside_the_persistent_schedule_costs_at_most_its_bound(ctx):
    """VERDICT r4 item 4.  A copy between PAGEABLE host memory and the device issued by other code
    of the process (here: a second thread calling hipMemcpy through the runtime library directly, the way a host framework
    next to the solver would) while a pivot chain + bulk kernel pair is in flight stops that pair; the schedule's bounded
    waits then expire, the factorization is redone with one launch per piece and the solver stays there for 16
    factorizations.  What that costs is the bound: ~0.1 s since round 5 (1 s before).  Twenty factorizations beside a thread
    that copies all the time: every factor right (inertia, backward error of a solve), no factorize! call longer than
    0.25 s, at most two fall-backs (the second when the schedule is tried again), and `stall_ms_total` says what they cost."""
    import ctypes
    import threading
    import time
    dev = torch.device("cuda", 0)
    N = 8000
    g = torch.Generator(device=dev).manual_seed(8000)
    R = torch.randn(N, 48, dtype=torch.float64, device=dev, generator=g)
    A = R @ R.T
    A.diagonal().add_(float(N))
    n1 = 2 * N // 3
    A[n1:, n1:].neg_()
    b = torch.randn(N, dtype=torch.float64, device=dev, generator=g)
    anorm = A.abs().sum(dim=1).max().item()
    torch.cuda.synchronize()
    M = mj.HipLinearSolver(A, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=mj.LDL))
    for _ in range(3):
        M.factorize()
    assert M.get_stat("panel_algo") == 5.0 and M.get_stat("pp_fallbacks") == 0.0 and M.get_stat("stall_ms_total") == 0.0
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipMemcpy.restype = ctypes.c_int
    hip.hipSetDevice.argtypes = [ctypes.c_int]
    dst = torch.empty(1 << 20, dtype=torch.float64, device=dev)
    src = np.ones(1 << 20)                       # pageable host memory
    stop = threading.Event()
    copies = [0]

    def copier():
        hip.hipSetDevice(0)
        while not stop.is_set():
            assert hip.hipMemcpy(dst.data_ptr(), src.ctypes.data, src.nbytes, 1) == 0   # hipMemcpyHostToDevice
            copies[0] += 1
            time.sleep(0.002)

    th = threading.Thread(target=copier)
    th.start()
    worst = 0.0
    try:
        for it in range(20):
            t0 = time.perf_counter()
            M.factorize()
            worst = max(worst, time.perf_counter() - t0)
            assert M.inertia() == (n1, 0, N - n1), it
            x = b.clone()
            torch.cuda.synchronize()
            M.solve_linear_system(x)
            M.check_solve()
            assert ((A @ x - b).abs().max() / (anorm * x.abs().max() + b.abs().max())).item() <= 1e-13, it
    finally:
        stop.set()
        th.join(timeout=30)
    assert copies[0] > 0
    fb, stall = M.get_stat("pp_fallbacks"), M.get_stat("stall_ms_total")
    assert worst <= 0.25, (worst, fb, stall)
    assert fb <= 2.0, (fb, stall)
    assert stall <= 250.0 * max(fb, 1.0) and (fb > 0) == (stall > 0.0), (fb, stall)
    assert M.get_stat("stall_ms_process") >= stall
    M.close()


def test_bench_py_on_two_gpus_over_rccl():
    """the N > 1 path of bench.py as the driver launches it -- one rank per GPU, 16 independent
    instances per rank through the batch API, barrier + all-reduce(MAX) of the step time + one all-gather of the per-rank
    phase times on the nccl (= RCCL) backend.  Skipped below two visible devices (the pool's boxes have one).  Checks the
    JSON line: two ranks, two distinct per-rank rows, whole-job value within 10 % of twice the single-rank `--batch 16` value
    (independent instances: weak scaling with no data-path collective)."""
    import json
    import os
    import subprocess
    import sys
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-c4", "--no-ipm-loop"]

    def run(extra):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.pop("WORLD_SIZE", None)
        res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *extra, *common], capture_output=True, text=True,
                             timeout=900, cwd=root, env=env)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
        lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, res.stdout[-2000:]
        return json.loads(lines[0])

    one = run(["--gpus", "1", "--batch", "16"])
    two = run(["--gpus", "2"])
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and two["config"]["batch_per_gpu"] == 16
    rows = two["per_rank_ms"]
    assert len(rows) == 2 and rows[0] != rows[1]
    assert abs(two["value"] - 2.0 * one["value"]) <= 0.10 * 2.0 * one["value"], (one["value"], two["value"])
    # round 6: the line carries its own denominator (rank 0's GPU alone on the same 16 instances, measured in the same run)
    assert abs(two["per_gpu_value"] - two["value"] / 2) <= 1e-9 * two["value"]
    assert two["efficiency_vs_c5_shape_per_gpu"] >= 0.95, (two["per_gpu_value"], two["c5_shape_per_gpu"])
    assert two["roofline"]["pp_fallbacks"] == 0.0


def test_first_factorizations_of_several_threads_next_to_a_process_full_of_contexts():
    """round 4 demoted this run to a tool because it lost a factorization in one run of two -- a CHILD
    process whose four host threads each start with their first factorization (N = 8000, task-DAG schedule), beside a PARENT
    process that holds eight live contexts which have used every schedule and therefore every CU-masked stream pair the library
    keeps per device (task-DAG pair, deep band, batch pair, small-batch partitions, look-ahead pairs): about a dozen idle
    hardware queues, and the device runs only so many side by side, all processes together.  A host process that goes idle
    next to other GPU processes now gives them back (`mnk_release_idle_streams`; the last context of a device to be
    destroyed does the same); the child then runs clean, and the parent's contexts make their streams again with their next
    factorization -- same bits as before the release."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dev = torch.device("cuda", 0)
    keep = []
    g = torch.Generator(device=dev).manual_seed(99)
    for i in range(8):
        st = torch.cuda.Stream(dev)
        c = mj.HipContext(0, stream=st.cuda_stream)
        keep.append((st, c))
    # every schedule once: large system (task-DAG pair), deep band (N = 2048), launch-per-panel (panel_algo 4: look-ahead pair),
    # a batch of two large systems (batch pair), a batch of small systems (small-batch partition)
    def spd(N):
        R = torch.randn(N, 32, dtype=torch.float64, device=dev, generator=g)
        A = R @ R.T
        A.diagonal().add_(float(N))
        return A
    mats = {N: spd(N) for N in (7000, 2048, 3000, 700)}
    torch.cuda.synchronize()
    sols = []
    for i, (st, c) in enumerate(keep):
        N = (7000, 2048, 3000, 700)[i % 4]
        M = mj.HipLinearSolver(mats[N], ctx=c, opt=mj.HipSolverOptions(lapack_algorithm=mj.CHOLESKY, panel_algo=4 if N == 3000 else 5))
        M.factorize()
        assert M.inertia() == (N, 0, 0)
        sols.append((N, M))
    big = [M for (N, M) in sols if N == 7000]
    small = [M for (N, M) in sols if N == 700]
    for grp in (big, small):
        with mj.factorize_batch():
            for M in grp:
                M.factorize()
        for M in grp:
            assert M.inertia()[1:] == (0, 0)
    ref = {}
    for i, (N, M) in enumerate(sols):
        M.factorize()
        Lf, D = M.get_factor_device()
        ref[i] = (torch.tril(Lf).clone(), D.clone())
    torch.cuda.synchronize()
    if not os.environ.get("MNK_TEST_KEEP_QUEUES"):   # (diagnostic: the run round 4 could not make pass)
        mj.release_idle_streams(0)
    child = [sys.executable, os.path.join(root, "tools", "thread_stress.py"), "8000", "4", "3"]
    for rep in range(2):
        out = subprocess.run(child, capture_output=True, text=True, timeout=600, cwd=root)
        assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
        last = (out.stdout.strip().splitlines() or [""])[-1]
        m = re.search(r"fallbacks ([0-9.]+), wrong solves (\d+)", last)
        assert m is not None and float(m.group(1)) == 0.0 and int(m.group(2)) == 0, (rep, last)
    # the parent goes on: streams are made again, the factors are the same bits
    for i, (N, M) in enumerate(sols):
        M.factorize()
        assert M.inertia() == (N, 0, 0)
        if N >= 1536 and not (N == 3000):
            assert M.get_stat("pp_fallbacks") == 0.0
        Lf, D = M.get_factor_device()
        assert torch.equal(torch.tril(Lf), ref[i][0]) and torch.equal(D, ref[i][1]), (i, N)
    for (N, M) in sols:
        M.close()
    for (st, c) in keep:
        c.close()


def test_schur_build_in_chunks_gives_the_same_bits(ctx):
    """the grouped Schur build works on chunks of scenarios with reused X / V / P buffers (memory no longer
    grows with ns x nd^2); the partial sums are added to S chunk by chunk in scenario order -- the same additions in the same
    order, so S must come out bit-identical whatever the chunk size (own process per size: MNK_SCHUR_CHUNK is read at creation)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import hashlib, sys
import numpy as np
sys.path.insert(0, %r)
import madnlp_jl_amd as mj
from madnlp_jl_amd.schur import SchurDenseStage
from tests.test_schur import two_stage_blocks
A, Cs, S0, blk = two_stage_blocks(11, 150, 50, 70, seed=23)
st = SchurDenseStage(A, Cs, S0, 70, blk, ctx=mj.HipContext(0))
S = st.build_kkt().cpu().numpy()
st.factorize_kkt()
print("SHA", hashlib.sha256(S.tobytes()).hexdigest(), st.inertia())
''' % root
    out = []
    for chunk in ("3", "4", "64"):
        env = dict(os.environ, MNK_SCHUR_CHUNK=chunk)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env, cwd=root)
        assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-1500:]
        out.append([ln for ln in r.stdout.splitlines() if ln.startswith("SHA")][0])
    assert out[0] == out[1] == out[2], out
    assert "(70, 0, 0)" in out[0]


@pytest.mark.parametrize("case", ["case118", "case1354pegase"])
def test_early_rejection_of_matrices_that_are_not_positive_definite(ctx, case):
    """Round 5: the sparse condensed KKT system accepts positive definite matrices only and regularizes its dual block whatever
    the counts (reference src/KKT/Sparse/condensed.jl:138-141), so the static-pivot LDL' of a matrix that is NOT positive
    definite may stop at its first non-positive pivot, as dpotrf does (option early_reject, on for this KKT type), instead of
    running to the end as dsytrf does.  On OPF-shaped systems with an indefinite Hessian block: the verdict
    (`is_inertia_correct`) equals LAPACK's for every regularization of a ladder; a rejected matrix reports num_neg >= 1, and a
    solve with it completes the factorization first (the full run's bits); it costs less than a full factorization; and the next positive definite matrix on the SAME
    solver factors and solves as if nothing had happened (bit-identical to a solver that never rejected anything)."""
    import time
    from tests.test_hip_c5 import _hip_sc
    P = opf_shaped(case, indefinite=True, sigma_s_decades=2.0, du=1e-8)
    ko = _oracle_sc(P, BUNCHKAUFMAN)
    kh = _hip_sc(P, ctx, mj.BUNCHKAUFMAN)                 # early_reject on (the default of the KKT type)
    kf = mj.SparseCondensedKKTSystem(P.n, P.m, P.jac_I, P.jac_J, P.hess_I, P.hess_J, P.ind_ineq, P.ind_lb, P.ind_ub, ctx=ctx,
                                     opt_linear_solver=mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN), early_reject=False)
    for f in ("reg", "l_diag", "u_diag", "l_lower", "u_lower", "du_diag"):
        getattr(kf, f)[:] = getattr(P, f)
    kf.jac[:] = P.jac
    kf.hess[:] = P.hess
    from oracle import kernels as okern
    for k in (kh, kf):
        k.compress_jacobian(); k.compress_hessian(); k.set_aug_diagonal()
    rng = np.random.default_rng(1)
    b = rng.standard_normal(P.n)
    dw_prev, seen_reject, seen_accept = 0.0, 0, 0
    t_rej, t_full = [], []
    for dw in (0.0, 1e-4, 1e-2, 1.0, 1e2, 1e4):
        okern.regularize_diagonal(ko, dw - dw_prev, 0.0)
        for k in (kh, kf):
            k.regularize_diagonal(dw - dw_prev, 0.0)
        dw_prev = dw
        ko.build_kkt(); ko.linear_solver.factorize()
        ok_ref = ko.is_inertia_correct(*ko.linear_solver.inertia())
        verdicts = []
        for k, tl in ((kh, t_rej), (kf, t_full)):
            k.build_kkt()
            k.linear_solver.factorize()            # (warm)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            k.linear_solver.factorize()
            ine = k.linear_solver.inertia()
            tl.append(time.perf_counter() - t0)
            verdicts.append((k.is_inertia_correct(*ine), ine))
        (ok_h, ine_h), (ok_f, ine_f) = verdicts
        assert ok_h == ok_f == ok_ref, (dw, ine_h, ine_f, ko.linear_solver.inertia())
        assert sum(ine_h) == sum(ine_f) == P.n
        if ok_ref:
            seen_accept += 1
            assert ine_h == ine_f == (P.n, 0, 0)
            xh, xf = kh.linear_solver.solve_linear_system(b.copy()), kf.linear_solver.solve_linear_system(b.copy())
            assert np.array_equal(xh, xf), dw          # the same factor, bit for bit, after any number of rejections
            K = _full(ko)
            assert _bwd(K, xh, b) <= 1e-13
        else:
            seen_reject += 1
            assert ine_h[2] >= 1 and ine_f[2] + ine_f[1] >= 1
            assert ine_h[0] <= ine_f[0] + 64            # (pivots behind the stopping block count as negative)
            # a caller that solves all the same (MadNLP's multiplier initialization and restoration phases factorize and solve
            # without asking for the inertia) gets the factorization completed behind its back: the full run's bits
            redone = kh.linear_solver.get_stat("early_reject_redone")
            xh, xf = kh.linear_solver.solve_linear_system(b.copy()), kf.linear_solver.solve_linear_system(b.copy())
            assert np.array_equal(xh, xf), dw
            assert kh.linear_solver.get_stat("early_reject_redone") == redone + 1
            assert kh.linear_solver.inertia() == ine_f      # and the complete counts
    assert seen_reject >= 1 and seen_accept >= 1
    assert kh.linear_solver.get_stat("early_rejects") >= seen_reject and kf.linear_solver.get_stat("early_rejects") == 0
    assert kh.linear_solver.get_stat("pp_fallbacks") == 0
    if case == "case1354pegase":    # at N = 11 192 the saving is visible in wall time (rejections stop somewhere in the matrix)
        rej = [t for t, (dw) in zip(t_rej, range(len(t_rej)))][:seen_reject]
        full = t_full[:seen_reject]
        assert sum(rej) <= sum(full) * 1.02, (rej, full)
    kh.close(); kf.close()


@pytest.mark.parametrize("alg", ["BUNCHKAUFMAN", "CHOLESKY"])
def test_a_batch_in_which_some_instances_are_indefinite(alg):
    """A factorization batch (mnk_factorize_batch_begin / _end: the C5 shape with real interior-point loops) in which some
    scenarios' trial matrices are indefinite while their neighbours' are fine.  Six case1354pegase-shaped instances on one
    context, two of them with an indefinite Hessian block, alternating rounds.  The indefinite members DIE EARLY in the merged
    launch -- rejected at their first non-positive pivot (BUNCHKAUFMAN: early rejection, armed in batches too) or broken down
    (CHOLESKY) -- and their tasks of the merged bulk queue are dropped; the other four are factored as if they were alone:
    the task-DAG schedule, no fall-back, L and D bit-identical to lone factorizations, in every round; and the two get positive
    definite values on the SAME solvers in between.  (The first version of this test found round 4's batch kernel deciding
    "is this member dead" per THREAD: a member dying between two waves' looks split the workgroup between two tasks -- wrong
    tiles and spurious breakdowns in OTHER members in one round of three, once a memory fault; tools/dbg_batch_reject.py.)"""
    from tests.test_hip_c5 import _front
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(dev)
    ctx = mj.HipContext(0, stream=st.cuda_stream)
    bad = {1, 4}
    insts = []
    for i in range(6):
        # (the indefinite variant has Hessian entries of its own: a "bad" instance lives on that pattern, and its positive
        # definite values are the same matrix with delta_w = 100 on the primal diagonal -- K is indefinite below ~50)
        P = opf_shaped("case1354pegase", seed=4000 + i, du=1e-8, **(dict(indefinite=True, sigma_s_decades=2.0) if i in bad else {}))
        kh = mj.SparseCondensedKKTSystem(P.n, P.m, P.jac_I, P.jac_J, P.hess_I, P.hess_J, P.ind_ineq, P.ind_lb, P.ind_ub, ctx=ctx,
                                         opt_linear_solver=mj.HipSolverOptions(lapack_algorithm=getattr(mj, alg)))
        mk = lambda pr: dict(jac=torch.from_numpy(P.jac).to(dev), hess=torch.from_numpy(P.hess).to(dev),  # noqa: E731
                             pr=torch.from_numpy(pr).to(dev), du=torch.from_numpy(P.du_diag).to(dev))
        insts.append(dict(n=P.n, kh=kh, good=mk(P.pr_diag + 100.0 if i in bad else P.pr_diag), bad=mk(P.pr_diag) if i in bad else None))
    torch.cuda.synchronize()
    ref = []
    for it in insts:      # lone factorizations of the positive definite values
        _front(it["kh"], st, it["good"])
        assert it["kh"].linear_solver.inertia() == (it["n"], 0, 0)
        Lf, D = it["kh"].linear_solver.get_factor_device()
        ref.append((torch.tril(Lf).clone(), D.clone()))
    for rnd in range(6):
        use_bad = rnd % 2 == 0
        before = [it["kh"].linear_solver.get_stat("early_rejects") for it in insts]
        with mj.factorize_batch():
            for it in insts:
                _front(it["kh"], st, it["bad"] if (use_bad and it["bad"] is not None) else it["good"])
        for i, it in enumerate(insts):
            M = it["kh"].linear_solver
            with torch.cuda.stream(st):
                ine = M.inertia()
            assert M.get_stat("panel_algo") == 5.0 and M.get_stat("pp_fallbacks") == 0.0, (rnd, i)
            if use_bad and i in bad:
                assert not it["kh"].is_inertia_correct(*ine) and sum(ine) == it["n"], (rnd, i, ine)
                assert ine[2] >= 1 or alg == "CHOLESKY"               # (a Cholesky breakdown reports (0, n, 0))
                assert M.get_stat("early_rejects") == before[i] + (1 if alg == "BUNCHKAUFMAN" else 0)
            else:
                assert ine == (it["n"], 0, 0), (rnd, i, ine)
                assert M.get_stat("early_rejects") == before[i]
                Lf, D = M.get_factor_device()
                assert torch.equal(torch.tril(Lf), ref[i][0]) and torch.equal(D, ref[i][1]), (rnd, i)
    for it in insts:
        it["kh"].close()
    ctx.close()


@pytest.mark.parametrize("n,m,count", [(2048, 512, 7), (700, 100, 5)])
def test_a_batch_of_small_systems_in_which_one_breaks_down(n, m, count):
    """The small-system form of the test above (pchain_multi_kernel: the members' pivot chains side by side in one launch, one
    bulk launch for their band tiles): dense condensed KKT systems factored by CHOLESKY, member 2 with a Hessian that is not
    positive definite in alternating rounds -- it breaks down and dies early; the others keep inertia (n, 0, 0), backward
    error <= 1e-13 and, inside the schedule's window, the bits of their lone factorizations."""
    from madnlp_jl_amd.problems import dense_dummy_qp
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(dev)
    ctx = mj.HipContext(0, stream=st.cuda_stream)
    rng = np.random.default_rng(n + count)
    ks, Ks, bs, hess = [], [], [], []
    for i in range(count):
        P = dense_dummy_qp(n=n, m=m, n_eq=0, seed=30 + i)
        k = mj.DenseCondensedKKTSystem(P.n, P.m, P.ind_ineq, P.ind_eq, P.ind_lb, P.ind_ub, ctx=ctx,
                                       opt_linear_solver=mj.HipSolverOptions(lapack_algorithm=mj.CHOLESKY))
        for f in ("reg", "l_diag", "u_diag", "l_lower", "u_lower", "du_diag"):
            getattr(k, f)[:] = getattr(P, f)
        k.hess[...] = P.hess
        k.jac[...] = P.jac
        k.set_aug_diagonal()
        k.build_kkt()
        ks.append(k)
        hess.append(P.hess.copy())
        bs.append(rng.standard_normal(k.linear_solver.n))
    N = ks[0].linear_solver.n
    lone = []
    for k in ks:
        k.linear_solver.factorize_async()
        assert k.linear_solver.inertia() == (n, 0, 0)
        Lf, D = k.linear_solver.get_factor_device()
        lone.append((torch.tril(Lf).clone(), D.clone()))
        Ks.append(k.aug_com.to_host())
    for rnd in range(6):
        bad = rnd % 2 == 0
        Hb = hess[2].copy()
        if bad:
            Hb[n // 3, n // 3] -= 1e6          # one strongly negative direction a third of the way in
        ks[2].hess[...] = Hb
        ks[2].build_kkt()
        with mj.factorize_batch():
            for k in ks:
                k.linear_solver.factorize_async()
        for i, (k, b) in enumerate(zip(ks, bs)):
            ine = k.linear_solver.inertia()
            if bad and i == 2:
                assert ine != (n, 0, 0) and not k.is_inertia_correct(*ine), (rnd, ine)
                continue
            assert ine == (n, 0, 0), (rnd, i, ine)
            assert k.linear_solver.get_stat("pp_fallbacks") == 0.0
            x = k.linear_solver.solve_linear_system(b.copy())
            Kl = np.tril(Ks[i])
            Kf = Kl + np.tril(Kl, -1).T
            assert np.abs(Kf @ x - b).max() / (np.abs(Kf).sum(axis=1).max() * np.abs(x).max() + np.abs(b).max()) <= 1e-13, (rnd, i)
            if N >= 1536:
                Lf, D = k.linear_solver.get_factor_device()
                assert torch.equal(torch.tril(Lf), lone[i][0]) and torch.equal(D, lone[i][1]), (rnd, i)
    for k in ks:
        k.close()
    ctx.close()
