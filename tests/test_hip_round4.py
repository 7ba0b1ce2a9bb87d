"""GPU parity tests added in round 4 (all through the C ABI):

  * the default (task-DAG) schedule across its whole window of matrix orders, both algorithms, against LAPACK where a
    host factorization takes seconds and against size-independent identities everywhere (VERDICT r3 "weak" 1);
  * a sparse (CSC / KKT-handle) source, BUNCHKAUFMAN, a matrix that takes the pivoted tier, factorized twice on the same
    solver: the background zero-fill must never hand a used buffer to the next transfer;
  * persistent schedules of several contexts take turns on the device instead of degrading to one launch per piece;
  * the one-launch solve on the caller's own device vector (no staging copies), alternating publication buffers;
  * options pinned by the environment are visible to the caller.

Tolerances (fp64): backward errors <= 1e-13 relative to |A||x| + |b|; factor probes 1e-12 |A|; inertia exact; factor bits
identical across repetitions (the schedule is deterministic).
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as sp

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import madnlp_jl_amd as mj  # noqa: E402
from oracle.lapack_cpu import BUNCHKAUFMAN, CHOLESKY, LapackCPUSolver  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    c = mj.HipContext(0)
    yield c
    c.close()


def _dev_matrix(N, alg, dev):
    """Symmetric test matrix built ON THE DEVICE (a 24 576-row matrix is 4.8 GB): A = R R' + N I with 48 random columns --
    SPD; for LDL' the trailing third is negated, [[A11, A21'], [A21, -A22]]: quasi-definite, the unpivoted LDL' exists and
    the inertia is (n1, 0, N - n1) by construction (the structure of the KKT systems of the path)."""
    g = torch.Generator(device=dev).manual_seed(1000 + N)
    R = torch.randn(N, 48, dtype=torch.float64, device=dev, generator=g)
    A = R @ R.T
    A.diagonal().add_(float(N))
    n1 = N
    if alg == mj.LDL:
        n1 = (2 * N) // 3
        A[n1:, n1:].neg_()
    return A, n1


@pytest.mark.parametrize("alg", [mj.CHOLESKY, mj.LDL])
@pytest.mark.parametrize("N", [1280, 1600, 5376, 5400, 6100, 6200, 7777, 9000, 12345, 16384, 24576, 30720])
def test_default_schedule_across_its_range(ctx, N, alg):
    """panel_algo = 5 serves 1280 <= N <= 30 720 (round 6; 1536 ... 24 576 before): all rows in the chain's band up to 5376 rows, band + bulk + tile-closing
    tasks above.  A drop-in solver receives arbitrary N; round 3 tested the large-system mode at N = 11 192 only.  Per
    order and algorithm: the schedule that ran is 5 with no fall-back; inertia = the constructed one (and LAPACK's --
    dpotrf / dsytrf through the oracle's LapackCPUSolver -- up to N = 9000, where the host factorization takes seconds);
    backward error of a solve <= 1e-13; (L D L') v = A v on probe vectors to 1e-12 |A|; two factorizations give the same
    bits."""
    dev = torch.device("cuda", 0)
    A, n1 = _dev_matrix(N, alg, dev)
    g = torch.Generator(device=dev).manual_seed(7)
    b = torch.randn(N, dtype=torch.float64, device=dev, generator=g)
    x = b.clone()
    V = torch.randn(N, 3, dtype=torch.float64, device=dev, generator=g)
    torch.cuda.synchronize()   # A, b, x were produced on torch's stream, the solver works on the context's
    M = mj.HipLinearSolver(A, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=alg))
    M.factorize()
    assert (M.get_stat("panel_algo"), M.get_stat("pp_fallbacks")) == (5.0, 0.0), M.get_stat("timeout_site")
    want = (n1, 0, N - n1)
    assert M.inertia() == want
    if N <= 9000:
        Ah = np.asfortranarray(A.cpu().numpy())
        ref = LapackCPUSolver(Ah, CHOLESKY if alg == mj.CHOLESKY else BUNCHKAUFMAN).factorize()
        assert ref.inertia() == want
    M.solve_linear_system(x)
    M.check_solve()
    anorm = A.abs().sum(dim=1).max()
    bwd = ((A @ x - b).abs().max() / (anorm * x.abs().max() + b.abs().max())).item()
    assert bwd <= 1e-13, bwd
    Lf, D = M.get_factor_device()
    if alg == mj.CHOLESKY:
        Lt = torch.tril(Lf)
        LV = Lt @ (Lt.T @ V)
    else:
        Lt = torch.tril(Lf, -1)
        T = Lt.T @ V + V
        T *= D[:, None]
        LV = Lt @ T + T
    err = ((LV - A @ V).abs().max() / (anorm * V.abs().max())).item()
    assert err <= 1e-12, err
    del Lt, LV
    M.factorize()
    L2, D2 = M.get_factor_device()
    assert torch.equal(torch.tril(Lf), torch.tril(L2)) and torch.equal(D, D2), "the schedule must be deterministic"
    assert M.get_stat("panel_algo") == 5.0 and M.get_stat("pp_fallbacks") == 0.0
    M.close()


def _not_quasi_definite_sparse(rng, n1, n2):
    """[[0, B'], [B, C]] as a lower-triangular CSC triple: a zero leading block -- the static-pivot LDL' meets an exact zero
    pivot at once -- with a banded B of full column rank and a diagonal C."""
    N = n1 + n2
    rows, cols, vals = [], [], []
    for j in range(n1):                       # B: n2 x n1, three diagonals (n2 >= n1)
        for d in (0, 1, 5):
            i = (j + d) % n2
            rows.append(n1 + i); cols.append(j); vals.append(rng.uniform(0.5, 1.5) * (1 if d == 0 else 0.3))
    for i in range(n2):
        rows.append(n1 + i); cols.append(n1 + i); vals.append(rng.uniform(-1.0, 1.0))
    A = sp.csc_matrix((vals, (rows, cols)), shape=(N, N))
    A.sum_duplicates()
    A.sort_indices()
    return A


def _bwd_sym(Al, x, b):
    K = (Al + sp.tril(Al, -1).T).tocsr()
    return np.abs(K @ x - b).max() / (abs(K).sum(axis=1).max() * np.abs(x).max() + np.abs(b).max())


def test_csc_source_that_takes_the_pivoted_tier_twice_on_one_solver(ctx):
    """ADVICE r3 (high).  Sparse sources are scattered into a zeroed dense buffer; with `prefill` the zeros are written in
    the background into a second buffer and the next transfer swaps the two.  The pivoted tier transfers the matrix a
    SECOND time for the same factorize!, i.e. swaps again: the spare buffer then holds the discarded static factor and must
    not count as zeroed -- round 3 scattered the next matrix over the old L.  A CSC matrix that is not quasi-definite,
    factorized three times on one solver (the normal IPM sequence: wrong inertia -> regularize -> refactorize), each time
    against dsytrf on the same matrix: inertia, backward error, and the dense image of what the device factored."""
    rng = np.random.default_rng(41)
    n1, n2 = 300, 420
    N = n1 + n2
    A = _not_quasi_definite_sparse(rng, n1, n2)
    M = mj.HipLinearSolver((A.indptr.copy(), A.indices.copy(), A.data.copy()), ctx=ctx,
                           opt=mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN))
    b = rng.standard_normal(N)
    for rep in range(3):
        Ar = A.copy()
        if rep:                                   # the IPM's regularization between the calls: another matrix, same pattern
            Ar.data = Ar.data * (1.0 + 0.01 * rep)
        M.A = (Ar.indptr.copy(), Ar.indices.copy(), Ar.data.copy())
        M.factorize()
        dense = np.asfortranarray((Ar + sp.tril(Ar, -1).T).toarray())
        ref = LapackCPUSolver(dense, BUNCHKAUFMAN).factorize()
        assert M.bk_info()[:2] == (True, rep + 1), "every one of these factorizations should take the pivoted tier"
        assert M.inertia() == ref.inertia(), rep
        x = M.solve_linear_system(b.copy())
        assert _bwd_sym(Ar, x, b) <= 1e-11, (rep, _bwd_sym(Ar, x, b))
    M.close()


def test_kkt_handle_source_that_takes_the_pivoted_tier_twice(ctx):
    """The same sequence with the matrix living in a sparse condensed KKT handle (mnk_ls_factorize_sc): an OPF-shaped
    system whose FIRST pivot is cancelled (pr_diag[0] is set so that K[0, 0] is zero, or a rounding error away from it):
    the static-pivot LDL' breaks down -- or grows by 1e15 -- at once and BUNCHKAUFMAN (without the accept_only_pd shortcut
    of the condensed wrappers) refactors with pivoting; twice in a row with different diagonals, then a regular positive
    definite matrix on the same solver.  Every factorization: inertia = dsytrf's on the same matrix, backward error."""
    from madnlp_jl_amd.problems import opf_shaped
    P = opf_shaped("case30", du=1e-8)
    k = mj.SparseCondensedKKTSystem(P.n, P.m, P.jac_I, P.jac_J, P.hess_I, P.hess_J, P.ind_ineq, P.ind_lb, P.ind_ub, ctx=ctx,
                                    opt_linear_solver=mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN))
    for f in ("reg", "l_diag", "u_diag", "l_lower", "u_lower", "du_diag"):
        getattr(k, f)[:] = getattr(P, f)
    k.jac[:] = P.jac
    k.hess[:] = P.hess
    k.linear_solver.set_option("accept_only_pd", 0)
    rng = np.random.default_rng(3)
    b = rng.standard_normal(P.n)
    tiers = []
    for rep in range(3):
        k.set_aug_diagonal()
        k.pr_diag[:P.n] += 0.1 * rep                                  # the IPM's delta_w between the calls
        k.compress_jacobian(); k.compress_hessian(); k.build_kkt()
        if rep < 2:
            assert k.aug_com.rowval[k.aug_com.colptr[0]] == 0         # first stored entry of column 0 is the diagonal
            for _ in range(3):
                k.pr_diag[0] -= k.aug_com.nzval[k.aug_com.colptr[0]]  # K[0, 0] -> 0 (up to the rounding of its sum)
                k.build_kkt()
            assert abs(k.aug_com.nzval[k.aug_com.colptr[0]]) <= 1e-9 * np.abs(k.aug_com.nzval).max()
        k.linear_solver.factorize()
        Al = k.aug_com.to_scipy()
        dense = np.asfortranarray((Al + sp.tril(Al, -1).T).toarray())
        ref = LapackCPUSolver(dense, BUNCHKAUFMAN).factorize()
        assert k.linear_solver.inertia() == ref.inertia(), rep
        tiers.append(bool(k.linear_solver.bk_info()[0]))
        x = k.linear_solver.solve_linear_system(b.copy())
        xr = ref.solve_linear_system(b.copy())
        assert _bwd_sym(Al, x, b) <= 1e3 * _bwd_sym(Al, xr, b) + 1e-14, (rep, _bwd_sym(Al, x, b), _bwd_sym(Al, xr, b))
    assert tiers == [True, True, False], tiers
    k.close()


def test_persistent_schedules_of_several_contexts_take_turns(ctx):
    """VERDICT r3 'missing' 1.  Round 3 sent every solver to one launch per piece as soon as a second context was alive on
    the device.  Now the persistent operations of one process are chained by the device arbiter (mnk_persist_begin).  The
    reference drives distinct solver instances from concurrent host threads (src/KKT/Schur/schur.jl:953): four threads,
    each with its own context, stream and solver, factorize and solve concurrently (ctypes releases the GIL inside the
    calls) -- every factorization on the task-DAG schedule, no fall-back, the same factor bits as a solver that has the
    device to itself, correct solves.  (Queued asynchronous factorizations of several contexts: tests/test_hip_c5.py.)"""
    import threading
    dev = torch.device("cuda", 0)
    N = 4100
    A, n1 = _dev_matrix(N, mj.LDL, dev)
    torch.cuda.synchronize()
    alone = mj.HipLinearSolver(A, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=mj.LDL))
    alone.factorize()
    assert alone.get_stat("panel_algo") == 5.0
    Lref, Dref = alone.get_factor_device()
    Lref = torch.tril(Lref)
    alone.close()
    anorm = A.abs().sum(dim=1).max()
    torch.cuda.synchronize()
    errors = []

    def worker(i):
        try:
            st = torch.cuda.Stream(dev)
            c = mj.HipContext(0, stream=st.cuda_stream)
            with torch.cuda.stream(st):
                M = mj.HipLinearSolver(A, ctx=c, opt=mj.HipSolverOptions(lapack_algorithm=mj.LDL))
                g = torch.Generator(device=dev).manual_seed(100 + i)
                for rep in range(3):
                    M.factorize()
                    assert (M.get_stat("panel_algo"), M.get_stat("pp_fallbacks")) == (5.0, 0.0), (i, rep, M.get_stat("timeout_site"))
                    assert M.inertia() == (n1, 0, N - n1)
                    b = torch.randn(N, dtype=torch.float64, device=dev, generator=g)
                    x = b.clone()
                    M.solve_linear_system(x)
                    M.check_solve()
                    st.synchronize()
                    bwd = ((A @ x - b).abs().max() / (anorm * x.abs().max() + b.abs().max())).item()
                    assert bwd <= 1e-13, (i, rep, bwd)
                Lf, D = M.get_factor_device()
                assert torch.equal(torch.tril(Lf), Lref) and torch.equal(D, Dref), i
                M.close()
            c.close()
        except BaseException as e:  # noqa: BLE001  (reported by the main thread)
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    assert not any(t.is_alive() for t in threads)


def test_solve_on_the_callers_device_vector_alternating_buffers(ctx):
    """The one-launch solve reads the right-hand side from and writes the solution to the caller's device vector (N
    entries, not padded) and clears the publication buffer of the NEXT solve in passing: five solves in a row with
    different right-hand sides, an N that is not a multiple of 64, against the host path of the same solver (which stages
    through the padded work vector) -- identical bits -- and against the matrix."""
    dev = torch.device("cuda", 0)
    N = 3001
    A, _ = _dev_matrix(N, mj.LDL, dev)
    torch.cuda.synchronize()
    M = mj.HipLinearSolver(A, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=mj.LDL))
    M.factorize()
    g = torch.Generator(device=dev).manual_seed(11)
    anorm = A.abs().sum(dim=1).max()
    for i in range(5):
        b = torch.randn(N, dtype=torch.float64, device=dev, generator=g)
        guard = torch.full((N + 64,), 7.25, dtype=torch.float64, device=dev)   # nothing behind entry N - 1 may be touched
        guard[:N] = b
        torch.cuda.synchronize()   # (torch's stream produced the vectors, the solver works on the context's)
        M.solve_linear_system(guard[:N])
        M.check_solve()
        assert (guard[N:] == 7.25).all()
        xh = M.solve_linear_system(b.cpu().numpy().copy())
        assert np.array_equal(guard[:N].cpu().numpy(), xh), i
        x = guard[:N]
        assert ((A @ x - b).abs().max() / (anorm * x.abs().max() + b.abs().max())).item() <= 1e-13
    M.close()


def test_options_pinned_by_the_environment_are_reported():
    """mnk_ls_set_option returns success for a key that MNK_OPTIONS pinned without applying it.  The pinned set
    is visible through get_stat("pinned:<key>") and the first ignored call says so on stderr (own process: the environment
    is read when a solver is created)."""
    code = (
        "import numpy as np, madnlp_jl_amd as mj\n"
        "A = np.asfortranarray(np.eye(200) * 3.0)\n"
        "M = mj.HipLinearSolver(A, opt=mj.HipSolverOptions(lapack_algorithm=mj.LDL))\n"
        "print('PINNED', M.get_stat('pinned:panel_algo'), M.get_stat('pinned:dag_chunk'))\n"
        "M.set_option('panel_algo', 4)\n"
        "M.factorize()\n"
        "print('ALGO', M.get_stat('panel_algo'))\n"
    )
    env = dict(os.environ, MNK_OPTIONS="panel_algo=1", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "PINNED 1.0 0.0" in r.stdout and "ALGO 1.0" in r.stdout
    assert "ignored" in r.stderr and "panel_algo" in r.stderr


def _indefinite(kind, N, rng):
    if kind == "random":
        S = rng.standard_normal((N, N))
        return np.asfortranarray((S + S.T) / 2)
    n1 = 2 * N // 3                          # [[H, J'], [J, 0]] with an indefinite H: needs 2x2 pivots
    H = rng.standard_normal((n1, n1)); H = (H + H.T) / 2
    J = rng.standard_normal((N - n1, n1))
    A = np.zeros((N, N)); A[:n1, :n1] = H; A[n1:, :n1] = J; A[:n1, n1:] = J.T
    return np.asfortranarray(A)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,N", [("random", 700), ("saddle", 1000), ("random", 2500)])
def test_multi_workgroup_bunchkaufman_panels_take_the_pivots_of_the_one_workgroup_panels(ctx, kind, N):
    """the pivoted tier's panels are factored by a workgroup per 256 rows (virtual positions, one or
    two message rounds per column; csrc/bk.hip, host model tools/bk_mw_model.py).  Same pivoting rule, same order of the
    column updates: the permutation, the 1x1 / 2x2 pattern and the factor must equal the one-workgroup panels' (option
    bk_panel_wgs = 1), and both reconstruct P A P' = L D L'."""
    rng = np.random.default_rng(N)
    A = _indefinite(kind, N, rng)
    out = []
    for wgs in (0, 1):
        M = mj.HipLinearSolver(A, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN))
        M.set_option("bk_panel_wgs", wgs)
        M.factorize()
        inertia = M.inertia()
        assert M.bk_info()[0] and M.get_stat("bk_panel_multi") == (1.0 if wgs == 0 else 0.0) and M.get_stat("bk_mw_fallbacks") == 0
        err, perm, doff = _bk_reconstruct_r4(M, A)
        assert err <= 1e-11 * np.abs(A).max() * max(1, N / 16)
        Lg, D = M.get_factor()
        out.append((inertia, perm.copy(), doff.copy(), np.tril(Lg, -1), D.copy()))
        M.close()
    (i0, p0, o0, L0, D0), (i1, p1, o1, L1, D1) = out
    assert i0 == i1 and np.array_equal(p0, p1) and np.array_equal(o0 != 0, o1 != 0)
    assert np.abs(D0 - D1).max() <= 1e-9 * np.abs(D1).max() and np.abs(L0 - L1).max() <= 1e-9 * max(1.0, np.abs(L1).max())


def _bk_reconstruct_r4(M, A):
    active, count, perm, doff = M.bk_info()
    assert active
    Lg, D = M.get_factor()
    N = A.shape[0]
    Lu = np.tril(Lg, -1) + np.eye(N)
    Dm = np.diag(D)
    for k in np.nonzero(doff)[0]:
        Dm[k + 1, k] = Dm[k, k + 1] = doff[k]
    Af = np.tril(A) + np.tril(A, -1).T
    return np.abs(Lu @ Dm @ Lu.T - Af[np.ix_(perm, perm)]).max(), perm, doff


@pytest.mark.gpu
def test_pivoted_tier_that_starts_with_one_workgroup_panels_and_goes_on_with_many(ctx):
    """Orders beyond 256 rows x the number of CUs start with one-workgroup panels (physical interchanges inside the panel)
    and switch to the multi-workgroup kernel (virtual positions, per-panel permutation kernels) once the trailing matrix fits:
    the two kinds must compose.  Exercised at N = 1500 with the cap bk_max_wgs = 3 (panels of more than 768 rows: one
    workgroup): same pivots and factor as either kind alone."""
    N = 1500
    A = _indefinite("random", N, np.random.default_rng(11))
    out = []
    for cap in (3, 0):
        M = mj.HipLinearSolver(A, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN))
        M.set_option("bk_max_wgs", cap)
        M.factorize()
        inertia = M.inertia()
        err, perm, doff = _bk_reconstruct_r4(M, A)
        assert M.bk_info()[0] and err <= 1e-11 * np.abs(A).max() * max(1, N / 16)
        Lg, D = M.get_factor()
        out.append((inertia, perm.copy(), doff != 0, D.copy(), np.tril(Lg, -1)))
        M.close()
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
    assert np.abs(out[0][3] - out[1][3]).max() <= 1e-9 * np.abs(out[1][3]).max()
    assert np.abs(out[0][4] - out[1][4]).max() <= 1e-9 * max(1.0, np.abs(out[1][4]).max())


@pytest.mark.gpu
def test_blocked_bunchkaufman_at_the_bench_order(ctx):
    """... and at the bench's order, N = 11 192 (random symmetric indefinite: nearly every column needs the partner
    search): inertia = dsytrf's, backward error within 1e3 x dsytrs's, and the tier's time is printed."""
    import time
    N = 11192
    rng = np.random.default_rng(7)
    A = _indefinite("random", N, rng)
    b = rng.standard_normal(N)
    ref = LapackCPUSolver(A, BUNCHKAUFMAN).factorize()
    xr = ref.solve_linear_system(b.copy())
    nrm = np.abs(A).sum(axis=1).max()
    bwd = lambda x: np.abs(A @ x - b).max() / (nrm * np.abs(x).max() + np.abs(b).max())
    M = mj.HipLinearSolver(A, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN))
    M.factorize()
    assert M.inertia() == ref.inertia() and M.bk_info()[0] and M.get_stat("bk_panel_multi") == 1.0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    M.factorize()
    inertia = M.inertia()
    ms = 1e3 * (time.perf_counter() - t0)
    assert inertia == ref.inertia() and M.get_stat("bk_mw_fallbacks") == 0
    x = M.solve_linear_system(b.copy())
    assert bwd(x) <= 1e3 * bwd(xr) + 1e-15, (bwd(x), bwd(xr))
    print(f"N={N}: transfer + static tier + pivoted tier + inertia {ms:.1f} ms; backward error {bwd(x):.1e} (dsytrs {bwd(xr):.1e})")
    assert ms < 1000.0
    M.close()


@pytest.mark.gpu
def test_multi_workgroup_panel_that_gives_up_is_redone_with_one_workgroup_per_panel(ctx):
    """The multi-workgroup panels wait for each other's messages with a bound (another process' kernels on the CUs): when a
    round expires the factorization is void, the matrix is transferred again and factored with one workgroup per panel, and
    the solver stays there.  Forced here by a workgroup that never takes part (option debug_bk_missing)."""
    N = 700
    A = _indefinite("random", N, np.random.default_rng(5))
    ref = LapackCPUSolver(A, BUNCHKAUFMAN).factorize()
    M = mj.HipLinearSolver(A, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN))
    M.set_option("bk_spin_limit", 20000)
    M.set_option("debug_bk_missing", 1)
    M.factorize()
    assert M.inertia() == ref.inertia()
    assert M.bk_info()[0] and M.get_stat("bk_mw_fallbacks") == 1 and M.get_stat("bk_panel_multi") == 0.0
    err, perm, doff = _bk_reconstruct_r4(M, A)
    assert err <= 1e-11 * np.abs(A).max() * max(1, N / 16)
    M.set_option("debug_bk_missing", -1)
    M.factorize()                       # (stays on the one-workgroup panels)
    assert M.inertia() == ref.inertia() and M.get_stat("bk_mw_fallbacks") == 1 and M.get_stat("bk_panel_multi") == 0.0
    M.close()


@pytest.mark.gpu
def test_time_out_of_the_task_dag_schedule_with_a_sparse_source_is_redone_on_a_clean_buffer(ctx):
    """The zero-fill of the second factor buffer rides on the task queue of the factorization (DAG_FILL tasks) -- and a
    factorization whose `info` becomes non-zero drops the rest of its queue.  The redo after a time-out transfers the matrix
    again: it must not swap the half-zeroed buffer (the previous factor) in.  case1354pegase-shaped condensed system through
    its KKT handle: a good factorization (the buffers now hold a factor), then one whose chain gives up (option
    debug_pp_missing) -- the redo's inertia and solve must be right, and so must the next factorization's."""
    from madnlp_jl_amd.problems import OPF_CASES, opf_shaped
    P = opf_shaped("case1354pegase", seed=OPF_CASES["case1354pegase"][0] + 7, du=1e-8)
    k = mj.SparseCondensedKKTSystem(P.n, P.m, P.jac_I, P.jac_J, P.hess_I, P.hess_J, P.ind_ineq, P.ind_lb, P.ind_ub, ctx=ctx,
                                    opt_linear_solver=mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN))
    k.jac[:] = P.jac; k.hess[:] = P.hess; k.pr_diag[:] = P.pr_diag; k.du_diag[:] = P.du_diag
    k.compress_jacobian(); k.compress_hessian(); k.build_kkt()
    M = k.linear_solver
    M.set_option("dag_spin_limit", 1 << 17)
    a = k.aug_com
    Kl = sp.csc_matrix((a.nzval.copy(), a.rowval, a.colptr), shape=(P.n, P.n))
    K = (Kl + sp.tril(Kl, -1).T).tocsr()
    b = np.random.default_rng(3).standard_normal(P.n)
    knorm = abs(K).sum(axis=1).max()
    bwd = lambda x: np.abs(K @ x - b).max() / (knorm * np.abs(x).max() + np.abs(b).max())
    for rnd, missing in enumerate((-1, -1, 5, -1, -1)):
        M.set_option("debug_pp_missing", missing)
        k.build_kkt()
        M.factorize()
        assert M.inertia() == (P.n, 0, 0), (rnd, M.inertia(), M.get_stat("pp_fallbacks"))
        x = M.solve_linear_system(b.copy())
        assert bwd(x) <= 1e-13, (rnd, bwd(x))
        assert M.get_stat("pp_fallbacks") == (0.0 if rnd < 2 else 1.0)
    k.close()


@pytest.mark.gpu
def test_every_workgroup_of_the_bulk_grid_is_resident_at_launch(ctx):
    """A persistent grid must not exceed what the hardware places at launch: workgroups the dispatcher starts in mid-kernel
    were what the one-in-~3000 time-out of the task-DAG schedule came from (csrc/ls.hip
    mnk_ctx_bulk_wgs).  With the 16-CU chain mask two shader engines of every XCD keep 7 of their 8 CUs, and the grid is
    3 x 7 x 32 = 672 -- not 3 x 240.  Checked on the device with the schedule's own per-workgroup statistics (option
    dag_trace): every workgroup of the grid takes its first task within 2 ms of the first one (the late ones of the old
    grid started 3.6 and 7 ms into the 9 ms kernel, or never), every one works, and no slot beyond the grid is used."""
    import ctypes as C

    from madnlp_jl_amd import _lib as L
    N = 11192
    A, _ = _dev_matrix(N, mj.BUNCHKAUFMAN, torch.device("cuda", 0))
    torch.cuda.synchronize()
    ls = mj.HipLinearSolver(A, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN, panel_algo=5))
    ls.set_option("dag_fill", 0)
    ls.factorize()
    nwg = int(ls.get_stat("dag_bulk_wgs"))
    props = torch.cuda.get_device_properties(0)
    if props.multi_processor_count == 256:
        assert nwg == 672
    assert 0 < nwg <= 3 * (props.multi_processor_count - 16)
    ls.set_option("dag_trace", 1)
    ls.factorize()
    assert ls.inertia() == (N, 0, 0) and ls.get_stat("panel_algo") == 5.0 and ls.get_stat("pp_fallbacks") == 0.0
    ntasks = int(ls.get_stat("dag_ntasks"))
    tr = np.zeros(ntasks * 8 + 4096 * 8 + 1024 * 8, dtype=np.uint64)
    L.check(L.lib().mnk_ls_debug_solve_trace(ls._h, tr.ctypes.data, tr.size), "trace")
    w = tr[ntasks * 8 + 4096 * 8:].reshape(1024, 8).astype(np.int64)   # {first task taken, exit, ticks waited, tasks, ...} at 100 MHz
    assert np.all(w[:nwg, 3] > 0), f"workgroups without a task: {np.nonzero(w[:nwg, 3] == 0)[0][:8]}"
    assert not w[nwg:].any()
    first = w[:nwg, 0]
    late_ms = (first - first.min()) / 1e5
    assert late_ms.max() <= 2.0, f"workgroups {np.nonzero(late_ms > 2.0)[0][:8]} started {late_ms.max():.2f} ms after the first"
    ls.set_option("dag_trace", 0)
    ls.close()
