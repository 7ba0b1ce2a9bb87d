"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the
same seeded inputs.  Tolerances (fp64, stated per test):
  * assembly kernels (compress, condensation, densify)      bit-exact
  * dense-condensed build (MFMA Gram product)                1e-12 relative to |K|max per entry
  * factorization                                            backward error |A - L L'| <= 1e-13 |A|
  * solves                                                   backward error <= 1e-13 (cond-free)
"""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import madnlp_jl_amd as mj  # noqa: E402
from madnlp_jl_amd import _lib as L  # noqa: E402
from madnlp_jl_amd.problems import dense_dummy_qp, opf_shaped  # noqa: E402
from oracle import hs15  # noqa: E402
from oracle import dense as odense  # noqa: E402
from oracle import kernels as okern  # noqa: E402
from oracle import sparse_condensed as osc  # noqa: E402
from oracle.lapack_cpu import BUNCHKAUFMAN, CHOLESKY, LapackCPUSolver  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ctx():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    c = mj.HipContext(0)
    yield c
    c.close()


def solcmp(x, sol, atol=1e-4, rtol=1e-4):
    aerr = np.linalg.norm(x - sol, np.inf)
    return aerr < atol or aerr / np.linalg.norm(sol, np.inf) < rtol


# --------------------------------------------------------------------------- MFMA tile kernel
@pytest.mark.parametrize("mode", [0, 1, 2, 4])
@pytest.mark.parametrize("M,N,K", [(64, 64, 16), (128, 128, 64), (320, 192, 48), (576, 576, 512), (256, 64, 64),
                                   (1088, 64, 192)])
def test_gemm_nt_tiles(ctx, mode, M, N, K):
    """Asymmetric random operands (a transposed or row/col-swapped MFMA fragment map cannot pass)."""
    if mode in (2, 4) and M != N:
        pytest.skip("lower-only modes are for square diagonal-aligned C")
    rng = np.random.default_rng(M * 7 + N * 3 + K + mode)
    A = rng.standard_normal((M, K))
    B = rng.standard_normal((N, K)) + 0.5
    C0 = rng.standard_normal((M, N))
    # +256 rows of slack behind every operand, as the library's own buffers have
    def dev(a):
        t = torch.zeros(a.size + 256, dtype=torch.float64, device="cuda")
        t[:a.size] = torch.from_numpy(np.asfortranarray(a).ravel(order="F")).cuda()
        return t
    dA, dB, dC = dev(A), dev(B), dev(C0)
    L.check(mj.lib().mnk_gemm_nt(ctx.handle, mode, M, N, K, dA.data_ptr(), M, dB.data_ptr(), N, dC.data_ptr(), M))
    ctx.synchronize()
    got = dC[:M * N].cpu().numpy().reshape((M, N), order="F")
    P = A @ B.T
    if mode == 0:
        ref = C0 - P
    elif mode == 1:
        ref = P
    else:
        ref = C0 - P if mode == 2 else C0 + P
        # only the wave tiles touching the lower triangle are defined: 64 x 64 (mode 2; mode 4 on large matrices), 32 x 32 where
        # mode 4 takes the 64 x 64 workgroup tiling (few tiles: the Gram product of the dense condensed system) -- the 32-row
        # rule is the part both tilings define
        g = 64 if mode == 2 else 32
        bi, bj = np.arange(M)[:, None] // g, np.arange(N)[None, :] // g
        mask = bi >= bj
        got, ref = np.where(mask, got, 0.0), np.where(mask, ref, 0.0)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12 * K)


# --------------------------------------------------------------------------- linear solver
@pytest.mark.parametrize("alg", [mj.CHOLESKY, mj.LDL, mj.BUNCHKAUFMAN])
@pytest.mark.parametrize("as_csc", [False, True])
def test_linear_solver_known_answer(ctx, alg, as_csc):
    """reference test/matrix_test.jl:21-30, MadNLPTests.test_linear_solver (:24-51)."""
    dense = np.array([[1.0, 0.0], [0.1, 2.0]], order="F")  # Array(sparse(row,col,val)): lower stored
    A = (np.array([0, 2, 3]), np.array([0, 1, 1]), np.array([1.0, 0.1, 2.0])) if as_csc else dense
    M = mj.HipLinearSolver(A, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=alg))
    assert "HIP" in M.introduce()
    assert M.improve() is False
    M.factorize()
    assert M.is_inertia() and M.inertia() == (2, 0, 0)
    x = M.solve_linear_system(np.array([1.0, 3.0]))
    assert solcmp(x, np.array([0.8542713567839195, 1.4572864321608041]))
    np.testing.assert_allclose(x, [0.8542713567839195, 1.4572864321608041], rtol=1e-14)
    M.close()


def _spd(rng, N, cond_decades=6):
    Q, _ = np.linalg.qr(rng.standard_normal((N, N)))
    w = 10.0 ** rng.uniform(-cond_decades / 2, cond_decades / 2, N)
    A = (Q * w) @ Q.T
    return np.asfortranarray((A + A.T) / 2)


@pytest.mark.parametrize("N", [1, 2, 63, 64, 65, 128, 200, 577, 1100])
def test_cholesky_vs_lapack(ctx, N):
    rng = np.random.default_rng(N)
    A = _spd(rng, N)
    A_upper_garbage = A.copy(order="F")
    A_upper_garbage[np.triu_indices(N, 1)] = np.nan  # 'L' storage: the upper triangle must never be read
    M = mj.HipLinearSolver(A_upper_garbage, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=mj.CHOLESKY, outer_block=256))
    M.factorize()
    assert M.info == 0 and M.inertia() == (N, 0, 0)
    Lg, _ = M.get_factor()
    Lg = np.tril(Lg)
    ref = LapackCPUSolver(A, CHOLESKY).factorize()
    Lr = np.tril(ref.fact)
    # backward error of the factorization (tolerance 1e-13 * |A|)
    assert np.abs(Lg @ Lg.T - A).max() <= 1e-13 * np.abs(A).max() * max(1, N / 64)
    # and agreement with LAPACK's factor, conditioning-scaled
    assert np.abs(Lg - Lr).max() <= 1e-9 * np.abs(Lr).max()
    b = rng.standard_normal(N)
    x = M.solve_linear_system(b.copy())
    xr = ref.solve_linear_system(b.copy())
    res = np.abs(A @ x - b).max() / (np.abs(A).max() * np.abs(x).max() + np.abs(b).max())
    res_ref = np.abs(A @ xr - b).max() / (np.abs(A).max() * np.abs(xr).max() + np.abs(b).max())
    assert res <= 1e-13 and res <= 50 * res_ref + 1e-15
    # multiple right-hand sides (reference loops over columns, linearsolvers.jl:102-110)
    Bm = np.asfortranarray(rng.standard_normal((N, 3)))
    X = M.solve_linear_system(Bm.copy(order="F"))
    assert np.abs(A @ X - Bm).max() / (np.abs(A).max() * np.abs(X).max()) <= 1e-13
    M.close()


def test_cholesky_not_positive_definite_reports_inertia_not_exception(ctx):
    """reference lapack_common.jl:96-98: failed Cholesky => inertia (0, N, 0), no throw."""
    rng = np.random.default_rng(3)
    N = 300
    A = _spd(rng, N)
    A[150, 150] = -1.0
    M = mj.HipLinearSolver(A, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=mj.CHOLESKY))
    M.factorize()
    assert M.info > 0 and M.info <= 151
    assert M.inertia() == (0, N, 0)
    ref = LapackCPUSolver(A, CHOLESKY).factorize()
    assert ref.info == M.info  # same failing pivot as dpotrf
    M.close()


@pytest.mark.parametrize("N,nneg", [(5, 2), (130, 17), (700, 300)])
def test_ldl_inertia_vs_bunch_kaufman(ctx, N, nneg):
    """Quasi-definite matrices [[H, J'],[J, -D]] (the shape of a regularized KKT system):
    LDL' without pivoting exists and sign(D) is the inertia Bunch-Kaufman reports."""
    rng = np.random.default_rng(N + nneg)
    npos = N - nneg
    H = _spd(rng, npos, 4)
    J = rng.standard_normal((nneg, npos))
    A = np.zeros((N, N), order="F")
    A[:npos, :npos] = H
    A[npos:, :npos] = J
    A[:npos, npos:] = J.T
    A[npos:, npos:] = -np.diag(10.0 ** rng.uniform(-6, 0, nneg))
    M = mj.HipLinearSolver(A, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN, outer_block=128))
    M.factorize()
    ref = LapackCPUSolver(A, BUNCHKAUFMAN).factorize()
    assert M.inertia() == ref.inertia() == (npos, 0, nneg)
    Lg, D = M.get_factor()
    Lg = np.tril(Lg, -1) + np.eye(N)
    assert np.abs((Lg * D) @ Lg.T - A).max() <= 1e-11 * np.abs(A).max()
    b = rng.standard_normal(N)
    x = M.solve_linear_system(b.copy())
    xr = ref.solve_linear_system(b.copy())
    res = np.abs(A @ x - b).max() / (np.abs(A).max() * np.abs(x).max() + np.abs(b).max())
    assert res <= 1e-11
    assert np.abs(x - xr).max() <= 1e-7 * np.abs(xr).max()
    M.close()


def test_ldl_zero_pivot_is_reported_as_num_zero(ctx):
    """A singular (2,2) block (du_diag = 0 with dependent equality rows) must surface as
    num_zero > 0 so that the IPM adds delta_c (reference src/IPM/solver.jl:636-666)."""
    A = np.zeros((4, 4), order="F")
    A[:2, :2] = np.eye(2)
    A[2, 0] = A[0, 2] = 1.0
    A[3, 0] = A[0, 3] = 1.0  # two identical equality rows -> Schur complement [[-1,-1],[-1,-1]]
    M = mj.HipLinearSolver(A, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=mj.LDL))
    M.factorize()
    pos, zero, neg = M.inertia()
    assert zero == 1 and pos == 2 and neg == 1
    M.close()


# --------------------------------------------------------------------------- KKT systems, HS15
def _make(kind, ctx):
    if kind == "sparse_condensed":
        return mj.SparseCondensedKKTSystem(hs15.N, hs15.M, hs15.JAC_I, hs15.JAC_J, hs15.HESS_I, hs15.HESS_J,
                                           hs15.IND_INEQ, hs15.IND_LB, hs15.IND_UB, ctx=ctx)
    if kind == "dense_condensed":
        return mj.DenseCondensedKKTSystem(hs15.N, hs15.M, hs15.IND_INEQ, hs15.IND_EQ, hs15.IND_LB, hs15.IND_UB,
                                          ctx=ctx)
    return mj.DenseKKTSystem(hs15.N, hs15.M, hs15.IND_INEQ, hs15.IND_LB, hs15.IND_UB, ctx=ctx)


@pytest.mark.parametrize("kind", ["sparse_condensed", "dense_condensed", "dense"])
def test_kkt_system_hs15(ctx, kind):
    """reference test/kkt_test.jl:27-48 -> MadNLPTests.test_kkt_system (:53-110), on the HIP path."""
    kkt = _make(kind, ctx)
    m, p = kkt.size()
    assert m == p
    kkt.initialize()
    x0, y0 = np.zeros(2), np.zeros(2)
    if kind == "sparse_condensed":
        kkt.get_jacobian()[:] = hs15.jac_coord(x0)
        kkt.get_hessian()[:] = hs15.hess_coord(x0, y0)
    else:
        kkt.get_jacobian()[...] = hs15.jac_dense(x0)
        kkt.get_hessian()[...] = hs15.hess_dense(x0, y0)
    kkt.compress_jacobian()
    kkt.compress_hessian()
    kkt.l_lower[:] = 1e-3
    kkt.u_lower[:] = 1e-3
    kkt.set_aug_diagonal()
    kkt.build_kkt()
    kkt.linear_solver.factorize()
    x = mj.UnreducedKKTVector.from_kkt(kkt)
    x.values[:] = 1.0
    assert kkt.solve_kkt(x) is x
    y = x.copy()
    y.values[:] = 0.0
    assert kkt.mul(y, x) is y
    np.testing.assert_allclose(y.values, np.ones(len(x.values)), rtol=0, atol=1e-13)
    ni, mi, pi = kkt.linear_solver.inertia()
    assert kkt.is_inertia_correct(ni, mi, pi)
    gold = json.load(open(os.path.join(GOLDEN, "hs15_kkt.json")))
    np.testing.assert_allclose(x.values, gold["oracle_solve_kkt_ones"], rtol=1e-12, atol=1e-14)
    if kind == "dense_condensed":
        np.testing.assert_allclose(kkt.aug_com.to_host(), np.diag(gold["K_condensed_diag"]), rtol=1e-15)
    if kind == "sparse_condensed":
        np.testing.assert_allclose(kkt.aug_com.to_dense(), np.diag(gold["K_condensed_diag"]), rtol=1e-15)
    kkt.regularize_diagonal(1.0, 1.0)
    kkt.close()


# --------------------------------------------------------------------------- sparse condensed
def _oracle_sc(P, alg=CHOLESKY):
    k = osc.SparseCondensedKKTSystem(P.n, P.m, P.jac_I, P.jac_J, P.hess_I, P.hess_J, P.ind_ineq, P.ind_lb,
                                     P.ind_ub, lambda A: LapackCPUSolver(A, alg))
    for f in ("reg", "l_diag", "u_diag", "l_lower", "u_lower", "du_diag"):
        getattr(k, f)[:] = getattr(P, f)
    k.jac[:] = P.jac
    k.hess[:] = P.hess
    return k


def _hip_sc(P, ctx, alg=mj.CHOLESKY, **opt):
    k = mj.SparseCondensedKKTSystem(P.n, P.m, P.jac_I, P.jac_J, P.hess_I, P.hess_J, P.ind_ineq, P.ind_lb, P.ind_ub,
                                    ctx=ctx, opt_linear_solver=mj.HipSolverOptions(lapack_algorithm=alg, **opt))
    for f in ("reg", "l_diag", "u_diag", "l_lower", "u_lower", "du_diag"):
        getattr(k, f)[:] = getattr(P, f)
    k.jac[:] = P.jac
    k.hess[:] = P.hess
    return k


@pytest.mark.parametrize("case,du", [("case30", 0.0), ("case118", 1e-8), ("case1354pegase", 1e-8)])
def test_sparse_condensed_assembly_bit_exact(ctx, case, du):
    P = opf_shaped(case, du=du)
    ko, kh = _oracle_sc(P), _hip_sc(P, ctx)
    for k in (ko, kh):
        k.compress_jacobian()
        k.compress_hessian()
        okern.set_aug_diagonal(k) if k is ko else k.set_aug_diagonal()
        k.build_kkt()
    np.testing.assert_array_equal(kh.pr_diag, ko.pr_diag)
    np.testing.assert_array_equal(kh._values(L.MNK_SC_JT, kh.nnz_jt), ko.jt_csc.nzval)
    np.testing.assert_array_equal(kh._values(L.MNK_SC_HESS, kh.nnz_hess), ko.hess_com.nzval)
    np.testing.assert_array_equal(kh.diag_buffer, ko.diag_buffer)
    np.testing.assert_array_equal(kh.aug_com.nzval, ko.aug_com.nzval)   # bit-exact condensation
    # device inputs (values already resident in HBM) give the same bits
    dj = torch.from_numpy(P.jac).cuda()
    dh = torch.from_numpy(P.hess).cuda()
    dp = torch.from_numpy(ko.pr_diag).cuda()
    dd = torch.from_numpy(ko.du_diag).cuda()
    kh.compress_jacobian(dj); kh.compress_hessian(dh); kh.build_kkt(dp, dd)
    np.testing.assert_array_equal(kh.aug_com.nzval, ko.aug_com.nzval)
    kh.close()


@pytest.mark.parametrize("case,alg", [("case30", "CHOLESKY"), ("case118", "CHOLESKY"), ("case118", "BUNCHKAUFMAN"),
                                      ("case1354pegase", "CHOLESKY")])
def test_sparse_condensed_factorize_solve(ctx, case, alg):
    P = opf_shaped(case, du=1e-8)
    ko, kh = _oracle_sc(P, alg), _hip_sc(P, ctx, alg)
    for k in (ko, kh):
        k.compress_jacobian(); k.compress_hessian()
        okern.set_aug_diagonal(k) if k is ko else k.set_aug_diagonal()
        k.build_kkt()
        k.linear_solver.factorize()
    assert kh.linear_solver.inertia() == ko.linear_solver.inertia() == (P.n, 0, 0)
    assert kh.is_inertia_correct(*kh.linear_solver.inertia())
    # densified matrix == oracle's (factor of the same bits): compare the solve through residuals
    Kd = ko.aug_com.to_dense()
    Kfull = Kd + np.tril(Kd, -1).T
    rng = np.random.default_rng(5)
    b = rng.standard_normal(P.n)
    xh = kh.linear_solver.solve_linear_system(b.copy())
    xo = ko.linear_solver.solve_linear_system(b.copy())
    nrm = np.abs(Kfull).sum(axis=1).max()
    rh = np.abs(Kfull @ xh - b).max() / (nrm * np.abs(xh).max() + np.abs(b).max())
    ro = np.abs(Kfull @ xo - b).max() / (nrm * np.abs(xo).max() + np.abs(b).max())
    assert rh <= 1e-13, (rh, ro)
    # full solve_kkt! + mul! identity (MadNLPTests.jl:85-98) on the IPM-like diagonals
    for k, V in ((ko, okern.UnreducedKKTVector), (kh, mj.UnreducedKKTVector)):
        x = V.from_kkt(k)
        x.values[:] = 1.0
        k.solve_kkt(x)
        y = x.copy(); y.values[:] = 0.0
        k.mul(y, x)
        k._res = np.abs(y.values - 1.0).max() / max(1.0, np.abs(x.values).max())
        k._x = x.values.copy()
    assert kh._res <= max(1e-9, 10 * ko._res), (kh._res, ko._res)
    # Richardson refinement converges to the same acceptable tolerance on both paths
    from oracle.backsolve import RichardsonIterator as ORich
    for k, R, V in ((ko, ORich, okern.UnreducedKKTVector), (kh, mj.RichardsonIterator, mj.UnreducedKKTVector)):
        it = R(k, tol=1e-8)
        bb = V.from_kkt(k); bb.values[:] = np.random.default_rng(9).standard_normal(len(bb.values))
        xx, ww = V.from_kkt(k), V.from_kkt(k)
        k._ok = it.solve_refine(xx, bb, ww)
        k._rr = it.residual_ratio
    assert kh._ok == ko._ok
    assert kh._rr <= max(1e-10, 100 * ko._rr), (kh._rr, ko._rr)
    kh.close()


def test_sparse_condensed_indefinite_hessian_triggers_regularization(ctx):
    """Negative curvature: both paths must report wrong inertia, then accept after
    regularize_diagonal! with the same delta_w (reference src/IPM/solver.jl:636-666)."""
    P = opf_shaped("case118", indefinite=True, sigma_s_decades=2.0)
    ko, kh = _oracle_sc(P, BUNCHKAUFMAN), _hip_sc(P, ctx, mj.BUNCHKAUFMAN)
    outcomes = []
    for k in (ko, kh):
        k.compress_jacobian(); k.compress_hessian()
        okern.set_aug_diagonal(k) if k is ko else k.set_aug_diagonal()
        seq = []
        dw_prev = 0.0
        for dw in (0.0, 1e-4, 1e-2, 1.0, 1e2):
            okern.regularize_diagonal(k, dw - dw_prev, 0.0) if k is ko else k.regularize_diagonal(dw - dw_prev, 0.0)
            dw_prev = dw
            k.build_kkt()
            k.linear_solver.factorize()
            ok = k.is_inertia_correct(*k.linear_solver.inertia())
            seq.append(ok)
            if ok:
                break
        outcomes.append(seq)
    assert outcomes[0] == outcomes[1] and outcomes[0][0] is False and outcomes[0][-1] is True
    kh.close()


# --------------------------------------------------------------------------- dense condensed
def _dense_pair(P, ctx, alg_o, alg_h, condensed=True):
    fac = lambda A: LapackCPUSolver(A, alg_o)  # noqa: E731
    opt = mj.HipSolverOptions(lapack_algorithm=alg_h, outer_block=256)
    if condensed:
        ko = odense.DenseCondensedKKTSystem(P.n, P.m, P.ind_ineq, P.ind_eq, P.ind_lb, P.ind_ub, fac)
        kh = mj.DenseCondensedKKTSystem(P.n, P.m, P.ind_ineq, P.ind_eq, P.ind_lb, P.ind_ub, ctx=ctx,
                                        opt_linear_solver=opt)
    else:
        ko = odense.DenseKKTSystem(P.n, P.m, P.ind_ineq, P.ind_lb, P.ind_ub, fac)
        kh = mj.DenseKKTSystem(P.n, P.m, P.ind_ineq, P.ind_lb, P.ind_ub, ctx=ctx, opt_linear_solver=opt)
    for k in (ko, kh):
        for f in ("reg", "l_diag", "u_diag", "l_lower", "u_lower", "du_diag"):
            getattr(k, f)[:] = getattr(P, f)
        k.hess[...] = P.hess
        k.jac[...] = P.jac
    return ko, kh


@pytest.mark.parametrize("n,m,n_eq", [(10, 5, 0), (50, 10, 0), (20, 15, 2), (300, 100, 0), (333, 77, 13),
                                      (2048, 512, 0), (2048, 512, 64)])
def test_dense_condensed_build_and_solve(ctx, n, m, n_eq):
    """Sizes from reference test/madnlp_dense.jl:8-53 plus BASELINE.json config C2."""
    P = dense_dummy_qp(n, m, n_eq)
    ko, kh = _dense_pair(P, ctx, BUNCHKAUFMAN, mj.BUNCHKAUFMAN)
    for k in (ko, kh):
        okern.set_aug_diagonal(k) if k is ko else k.set_aug_diagonal()
        k.compress_hessian(); k.compress_jacobian()
        k.build_kkt()
    Ko, Kh = ko.aug_com, kh.aug_com.to_host()
    # entries are sums of <= m products: 1e-12 relative to the largest entry
    assert np.abs(Kh - Ko).max() <= 1e-12 * np.abs(Ko).max()
    assert np.array_equal(Kh, Kh.T)  # both triangles written, exactly mirrored
    for k in (ko, kh):
        k.linear_solver.factorize()
    assert kh.linear_solver.inertia() == ko.linear_solver.inertia() == (n, 0, n_eq)
    assert kh.is_inertia_correct(*kh.linear_solver.inertia())
    for k, V in ((ko, okern.UnreducedKKTVector), (kh, mj.UnreducedKKTVector)):
        x = V.from_kkt(k)
        x.values[:] = 1.0
        k.solve_kkt(x)
        y = x.copy(); y.values[:] = 0.0
        k.mul(y, x)
        k._res = np.abs(y.values - 1.0).max() / max(1.0, np.abs(x.values).max())
    assert kh._res <= max(1e-10, 10 * ko._res), (kh._res, ko._res)
    kh.close()


@pytest.mark.parametrize("n,m", [(10, 5), (50, 10), (120, 40)])
def test_dense_augmented_build_and_solve(ctx, n, m):
    P = dense_dummy_qp(n, m, 0)
    ko, kh = _dense_pair(P, ctx, BUNCHKAUFMAN, mj.BUNCHKAUFMAN, condensed=False)
    for k in (ko, kh):
        okern.set_aug_diagonal(k) if k is ko else k.set_aug_diagonal()
        k.compress_hessian(); k.compress_jacobian()
        k.build_kkt()
    np.testing.assert_array_equal(kh.aug_com.to_host(), ko.aug_com)  # pure scatter: bit-exact
    for k in (ko, kh):
        k.linear_solver.factorize()
    assert kh.linear_solver.inertia() == ko.linear_solver.inertia() == (n + m, 0, m)
    for k, V in ((ko, okern.UnreducedKKTVector), (kh, mj.UnreducedKKTVector)):
        x = V.from_kkt(k)
        x.values[:] = 1.0
        k.solve_kkt(x)
        y = x.copy(); y.values[:] = 0.0
        k.mul(y, x)
        k._res = np.abs(y.values - 1.0).max() / max(1.0, np.abs(x.values).max())
    assert kh._res <= max(1e-10, 10 * ko._res), (kh._res, ko._res)
    kh.close()


def test_spmv_pieces(ctx):
    """Device SpMV building blocks of solve_kkt!/mul! (reference src/IPM/factorization.jl:157-160,289-293)."""
    P = opf_shaped("case118")
    ko, kh = _oracle_sc(P), _hip_sc(P, ctx)
    for k in (ko, kh):
        k.compress_jacobian(); k.compress_hessian()
    rng = np.random.default_rng(2)
    xm, xn = rng.standard_normal(P.m), rng.standard_normal(P.n)
    lib = mj.lib()
    def run(which, trans, x, ny, alpha=1.0, beta=0.0, y0=None):
        dx = torch.from_numpy(x).cuda()
        dy = torch.from_numpy(y0 if y0 is not None else np.zeros(ny)).cuda()
        L.check(lib.mnk_sc_spmv(kh._h, which, trans, alpha, dx.data_ptr(), beta, dy.data_ptr()))
        ctx.synchronize()
        return dy.cpu().numpy()
    np.testing.assert_allclose(run(L.MNK_SC_JT, 0, xm, P.n), ko.jt_csc.matvec(xm), rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(run(L.MNK_SC_JT, 1, xn, P.m), ko.jt_csc.rmatvec(xn), rtol=1e-13, atol=1e-13)
    y0 = rng.standard_normal(P.n)
    np.testing.assert_allclose(run(L.MNK_SC_HESS, 0, xn, P.n, -1.0, 1.0, y0),
                               y0 - ko.hess_com.symmetric_lower_matvec(xn), rtol=1e-12, atol=1e-12)
    kh.close()


# --------------------------------------------------------------------------- end-to-end IPM parity
def _ipm_pair(kind, nlp, ctx, tol, device_kkt_ops=False):
    """Run the same IPM driver twice from identical inputs: CPU oracle back-end (LAPACK
    Bunch-Kaufman) and HIP back-end (static-pivot LDL^T)."""
    from madnlp_jl_amd.ipm import IPMOptions, MadNLPSolver
    from tests.test_ipm_oracle import oracle_factory
    sparse = kind == "sparse_condensed"

    def hip_factory(info):
        opt = mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN)
        if kind == "sparse_condensed":
            return mj.SparseCondensedKKTSystem(info["n"], info["m"], nlp.jac_I, nlp.jac_J, nlp.hess_I, nlp.hess_J,
                                               info["ind_ineq"], info["ind_lb"], info["ind_ub"], ctx=ctx,
                                               opt_linear_solver=opt, device_kkt_ops=device_kkt_ops)
        if kind == "dense_condensed":
            return mj.DenseCondensedKKTSystem(info["n"], info["m"], info["ind_ineq"], info["ind_eq"], info["ind_lb"],
                                              info["ind_ub"], ctx=ctx, opt_linear_solver=opt,
                                              device_kkt_ops=device_kkt_ops)
        return mj.DenseKKTSystem(info["n"], info["m"], info["ind_ineq"], info["ind_lb"], info["ind_ub"], ctx=ctx,
                                 opt_linear_solver=opt, device_kkt_ops=device_kkt_ops)

    out = []
    for fac in (oracle_factory(kind, nlp), hip_factory):
        opt = IPMOptions(tol=tol)
        if sparse:
            opt.relax_equality, opt.dual_initialization = True, "zero"
        s = MadNLPSolver(nlp, fac, opt, sparse=sparse)
        s.solve()
        out.append(s)
    return out


def _assert_ipm_parity(so, sh, n):
    """CPU == GPU acceptance of the reference (lib/MadNLPGPU/test/densekkt_rocm.jl:31-37): same
    iteration count, objective, solution and multipliers atol 1e-6 -- plus per-iteration residual
    parity (primal/dual infeasibility, complementarity, mu, delta_w), rtol 1e-5 above 1e-9."""
    assert sh.status == so.status == "SOLVE_SUCCEEDED", (so.status, sh.status)
    assert sh.cnt.k == so.cnt.k
    assert sh.cnt.factorization_cnt == so.cnt.factorization_cnt
    np.testing.assert_allclose(sh.x[:n], so.x[:n], atol=1e-6)
    np.testing.assert_allclose(sh.y, so.y, atol=1e-6)
    assert abs(sh.obj_val - so.obj_val) <= 1e-6 * max(1.0, abs(so.obj_val))
    for a, b in zip(so.history, sh.history):
        assert a.mu == b.mu and a.del_w == b.del_w
        for fld in ("inf_pr", "inf_du", "inf_compl"):
            va, vb = getattr(a, fld), getattr(b, fld)
            assert abs(va - vb) <= 1e-5 * max(abs(va), 1e-9) + 1e-12, (a.k, fld, va, vb)


@pytest.mark.parametrize("kind", ["dense", "dense_condensed", "sparse_condensed"])
def test_ipm_hs15_cpu_vs_hip(ctx, kind):
    from madnlp_jl_amd.problems import HS15Model
    so, sh = _ipm_pair(kind, HS15Model(), ctx, 1e-8 if kind != "sparse_condensed" else 1e-6)
    _assert_ipm_parity(so, sh, 2)
    sh.kkt.close()


@pytest.mark.parametrize("n,m,n_eq", [(10, 5, 0), (50, 10, 0), (20, 15, 2)])
@pytest.mark.parametrize("kind", ["dense", "dense_condensed"])
def test_ipm_dense_qp_cpu_vs_hip(ctx, kind, n, m, n_eq):
    """reference lib/MadNLPGPU/test/densekkt_rocm.jl:4-40 on the DenseDummyQP sizes of test/madnlp_dense.jl."""
    from madnlp_jl_amd.problems import DenseQPModel
    so, sh = _ipm_pair(kind, DenseQPModel(n, m, n_eq), ctx, 1e-8)
    _assert_ipm_parity(so, sh, n)
    sh.kkt.close()


@pytest.mark.parametrize("case", ["case30", "case118"])
def test_ipm_sparse_qp_cpu_vs_hip(ctx, case):
    from madnlp_jl_amd.problems import SparseQPModel
    nlp = SparseQPModel(case)
    so, sh = _ipm_pair("sparse_condensed", nlp, ctx, 1e-6)
    _assert_ipm_parity(so, sh, nlp.n)
    sh.kkt.close()


# --------------------------------------------------------------------------- orchestration stress
@pytest.mark.parametrize("N", [2500, 4672])
@pytest.mark.parametrize("alg", [mj.CHOLESKY, mj.LDL])
@pytest.mark.parametrize("nbo,la", [(128, False), (256, True), (512, True), (1024, False), (192, True)])
def test_factorization_schedules_agree(ctx, N, alg, nbo, la):
    """Every blocking / look-ahead schedule must produce the same factor (to rounding) and the same
    solve: several outer panels, a ragged last panel, outer widths that are not multiples of the
    256-column middle level, with and without the two-stream look-ahead."""
    rng = np.random.default_rng(N)
    R = rng.standard_normal((N, 96))
    A = np.asfortranarray(R @ R.T + np.diag(10.0 ** rng.uniform(-2, 3, N)))
    if alg == mj.LDL:  # make it quasi-definite: negate a trailing block (LDL^T without pivoting exists)
        k = N // 5
        A[N - k:, N - k:] *= -1.0
        A[N - k:, N - k:] -= np.eye(k) * 5.0
        A = np.asfortranarray((A + A.T) / 2)
    dA = torch.from_numpy(A).cuda()  # column-major view of a symmetric matrix
    M = mj.HipLinearSolver(dA, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=alg, outer_block=nbo, lookahead=la,
                                                                single_rows=0))  # force the multi-panel schedules
    M.factorize()
    ref_alg = CHOLESKY if alg == mj.CHOLESKY else BUNCHKAUFMAN
    ref = LapackCPUSolver(A, ref_alg).factorize()
    assert M.inertia() == ref.inertia()
    b = rng.standard_normal(N)
    x = M.solve_linear_system(b.copy())
    nrm = np.abs(A).sum(axis=1).max()
    res = np.abs(A @ x - b).max() / (nrm * np.abs(x).max() + np.abs(b).max())
    assert res <= 1e-13, res
    xr = ref.solve_linear_system(b.copy())
    assert np.abs(x - xr).max() <= 1e-8 * np.abs(xr).max()
    M.close()


@pytest.mark.parametrize("alg", [mj.CHOLESKY, mj.LDL])
@pytest.mark.parametrize("nbo", [256, 512])
@pytest.mark.parametrize("small", [0, 100000])
def test_shared_tile_queue_schedule_is_bit_identical(ctx, alg, nbo, small):
    """When the trailing update is drained through the tile queue by both look-ahead streams the factor
    must be bit-identical to the unshared schedule: each tile is computed by exactly one workgroup with
    the same arithmetic, whichever stream pulls it.  Also with the 64x64-tile variant of the next-panel
    update switched on/off (that one changes the tiling, not the sums)."""
    rng = np.random.default_rng(5)
    N = 3200
    R = rng.standard_normal((N, 80))
    A = np.asfortranarray(R @ R.T + np.diag(10.0 ** rng.uniform(-1, 3, N)))
    if alg == mj.LDL:
        k = N // 4
        A[N - k:, N - k:] *= -1.0
        A[N - k:, N - k:] -= np.eye(k) * 5.0
        A = np.asfortranarray((A + A.T) / 2)
    dA = torch.from_numpy(A).cuda()
    facs = []
    for share in (0, 1, 2, 2):  # off, adaptive, forced (every trailing update goes through the queue)
        M = mj.HipLinearSolver(dA, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=alg, outer_block=nbo,
                                                                   share=share, small_tiles=small, single_rows=0))
        M.factorize()
        Lf, d = M.get_factor()
        facs.append((np.tril(Lf).copy(), None if d is None else np.array(d)))
        inert = M.inertia()
        M.close()
    assert inert == ((N, 0, 0) if alg == mj.CHOLESKY else (N - N // 4, 0, N // 4))
    for Lf, d in facs[1:]:
        assert np.array_equal(facs[0][0], Lf)
        if d is not None:
            assert np.array_equal(facs[0][1], d)


def test_repeated_factorizations_are_deterministic(ctx):
    """Refactorizing the same matrix (the IPM does it every iteration) gives bit-identical factors:
    no race between the look-ahead streams, no stale state between calls."""
    rng = np.random.default_rng(11)
    N = 3000
    R = rng.standard_normal((N, 64))
    A = np.asfortranarray(R @ R.T + N * np.eye(N))
    M = mj.HipLinearSolver(A, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=mj.CHOLESKY, single_rows=0))
    M.factorize()
    L0, _ = M.get_factor()
    for _ in range(4):
        M.factorize()
        L1, _ = M.get_factor()
        assert np.array_equal(np.tril(L0), np.tril(L1))
    M.close()


# --------------------------------------------------------------------------- persistent solve
@pytest.mark.parametrize("alg", [mj.CHOLESKY, mj.LDL])
@pytest.mark.parametrize("N", [64, 200, 1000, 3333, 6000])
def test_persistent_solve_matches_stepwise_solve(ctx, alg, N):
    """The one-launch solve (workgroups exchanging block results through polled global values) and the
    one-launch-per-step solve apply the same operators (explicit 256x256 diagonal inverses, panel GEMVs) with
    different partial-sum groupings: solutions agree to rounding and both meet the backward-error bound."""
    rng = np.random.default_rng(N)
    R = rng.standard_normal((N, 48))
    A = np.asfortranarray(R @ R.T + np.diag(10.0 ** rng.uniform(-1, 2, N)))
    if alg == mj.LDL:
        k = max(1, N // 3)
        A[N - k:, N - k:] *= -1.0
        A[N - k:, N - k:] -= 3.0 * np.eye(k)
        A = np.asfortranarray((A + A.T) / 2)
    b = rng.standard_normal(N)
    xs = []
    for persistent in (True, False):
        M = mj.HipLinearSolver(A, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=alg, persistent_solve=persistent))
        M.factorize()
        x = M.solve_linear_system(b.copy())
        nrm = np.abs(A).sum(axis=1).max()
        res = np.abs(A @ x - b).max() / (nrm * np.abs(x).max() + np.abs(b).max())
        assert res <= 1e-13, (persistent, res)
        xs.append(x)
        M.close()
    assert np.abs(xs[0] - xs[1]).max() <= 1e-9 * np.abs(xs[1]).max()


def test_persistent_solve_is_deterministic_and_handles_matrix_rhs(ctx):
    rng = np.random.default_rng(3)
    N = 2500
    R = rng.standard_normal((N, 32))
    A = np.asfortranarray(R @ R.T + N * np.eye(N))
    M = mj.HipLinearSolver(A, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=mj.CHOLESKY))
    M.factorize()
    B = np.asfortranarray(rng.standard_normal((N, 3)))
    X1 = M.solve_linear_system(B.copy(order="F"))
    X2 = M.solve_linear_system(B.copy(order="F"))
    assert np.array_equal(X1, X2)  # fixed reduction orders: bitwise repeatable
    assert np.abs(A @ X1 - B).max() <= 1e-10 * np.abs(B).max() * N
    M.close()


def test_persistent_solve_multiblock_ownership(ctx):
    """N > 64 * #CUs: workgroups own several 64-row blocks (streamed updates).  Checked through the residual
    of a system with a known solution (a CPU factorization of this size would take minutes)."""
    N = 64 * 256 + 64 * 37 + 5  # 18757: 294 blocks over 256 workgroups, ragged last step
    g = torch.Generator(device="cuda").manual_seed(5)
    R = torch.randn(N, 40, dtype=torch.float64, device="cuda", generator=g)
    A = R @ R.T + torch.diag(10.0 ** (3 * torch.rand(N, dtype=torch.float64, device="cuda", generator=g)))
    xt = torch.randn(N, dtype=torch.float64, device="cuda", generator=g)
    b = A @ xt
    x = b.clone()
    torch.cuda.synchronize()  # A, b were produced on torch's stream; the solver runs on the context's stream
    M = mj.HipLinearSolver(A, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=mj.LDL))
    M.factorize()
    assert M.inertia() == (N, 0, 0)
    M.solve_linear_system(x)
    ctx.synchronize()
    res = (A @ x - b).abs().max() / (A.abs().sum(dim=1).max() * x.abs().max() + b.abs().max())
    assert float(res) <= 1e-13
    assert float((x - xt).abs().max()) <= 1e-7 * float(xt.abs().max())
    M.close()


def test_persistent_solves_from_concurrent_contexts(ctx):
    """Solves issued from several contexts/threads at once are chained on the device (a persistent kernel needs
    all its workgroups resident): no dead-lock, every result correct."""
    import threading
    rng = np.random.default_rng(9)
    N = 3000
    R = rng.standard_normal((N, 32))
    A = np.asfortranarray(R @ R.T + N * np.eye(N))
    results, errors = {}, []

    def work(tid):
        try:
            c = mj.HipContext(0)
            M = mj.HipLinearSolver(A, ctx=c, opt=mj.HipSolverOptions(lapack_algorithm=mj.CHOLESKY))
            M.factorize()
            b = np.random.default_rng(100 + tid).standard_normal(N)
            x = b.copy()
            for _ in range(6):
                x = M.solve_linear_system(b.copy())
            results[tid] = float(np.abs(A @ x - b).max())
            M.close()
            c.close()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    ths = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for th in ths:
        th.start()
    for th in ths:
        th.join(timeout=120)
    assert not errors, errors
    assert len(results) == 4 and all(v <= 1e-9 * N for v in results.values()), results


# --------------------------------------------------------------------------- device-side solve_kkt! / mul!
@pytest.mark.parametrize("case,du", [("case30", 0.0), ("case118", 1e-8), ("case1354pegase", 1e-8)])
def test_device_solve_kkt_and_mul_match_oracle(ctx, case, du):
    """`mnk_sc_solve_kkt` / `mnk_sc_mul` (the whole primal-dual vector stays on the device) against the CPU
    restatement of `solve_kkt!` / `mul!` (reference src/IPM/factorization.jl:143-167,289-308) on the same
    IPM-like diagonals and random vectors.  mul!: elementwise relative 1e-12 (same sums, different order);
    solve_kkt!: the two solutions agree through the KKT residual and to 1e-6 relative (cond-dependent)."""
    P = opf_shaped(case, du=du)
    ko, kh = _oracle_sc(P), _hip_sc(P, ctx)
    for k in (ko, kh):
        k.compress_jacobian(); k.compress_hessian()
        okern.set_aug_diagonal(k) if k is ko else k.set_aug_diagonal()
        k.build_kkt()
        k.linear_solver.factorize()
    kh.upload_barrier_terms()
    rng = np.random.default_rng(21)
    # mul!(w, kkt, x, alpha, beta) with non-trivial alpha/beta
    xo, wo = okern.UnreducedKKTVector.from_kkt(ko), okern.UnreducedKKTVector.from_kkt(ko)
    xh, wh = mj.UnreducedKKTVector.from_kkt(kh), mj.UnreducedKKTVector.from_kkt(kh)
    xv, wv = rng.standard_normal(len(xo.values)), rng.standard_normal(len(xo.values))
    for alpha, beta in ((1.0, 0.0), (-1.0, 1.0), (0.75, -0.5)):
        xo.values[:] = xv; wo.values[:] = wv; xh.values[:] = xv; wh.values[:] = wv
        ko.mul(wo, xo, alpha, beta)
        kh.mul_device(wh, xh, alpha, beta)
        scale = np.abs(wo.values).max()
        assert np.abs(wh.values - wo.values).max() <= 1e-12 * scale, (alpha, beta)
    # the host mirror and the device path of the HIP system agree too
    wh2 = mj.UnreducedKKTVector.from_kkt(kh); wh2.values[:] = wv
    xh.values[:] = xv
    kh.mul(wh2, xh, 0.75, -0.5)
    assert np.abs(wh2.values - wh.values).max() <= 1e-12 * np.abs(wh.values).max()
    # solve_kkt!
    bo, bh = okern.UnreducedKKTVector.from_kkt(ko), mj.UnreducedKKTVector.from_kkt(kh)
    bv = rng.standard_normal(len(bo.values))
    bo.values[:] = bv; bh.values[:] = bv
    ko.solve_kkt(bo)
    kh.solve_kkt_device(bh)
    # forward agreement is condition-dependent (Sigma_s spans 16 decades; two different factorizations):
    # 1e-6 relative here, the sharp statement is the residual below
    assert np.abs(bh.values - bo.values).max() <= 1e-6 * np.abs(bo.values).max()
    # residual of the device solution through the oracle's mul!: K x = b
    r = okern.UnreducedKKTVector.from_kkt(ko); r.values[:] = 0.0
    xs = okern.UnreducedKKTVector.from_kkt(ko); xs.values[:] = bh.values
    ko.mul(r, xs, 1.0, 0.0)
    ro = okern.UnreducedKKTVector.from_kkt(ko); ro.values[:] = 0.0
    ko.mul(ro, bo, 1.0, 0.0)
    res_h = np.abs(r.values - bv).max() / (np.abs(bv).max() + np.abs(bh.values).max())
    res_o = np.abs(ro.values - bv).max() / (np.abs(bv).max() + np.abs(bo.values).max())
    assert res_h <= max(1e-10, 100 * res_o), (res_h, res_o)
    # a device-resident vector is solved in place without touching the host
    bd = torch.from_numpy(bv.copy()).cuda()
    kh.solve_kkt_device(bd)
    ctx.synchronize()
    assert np.abs(bd.cpu().numpy() - bh.values).max() == 0.0
    kh.close()


@pytest.mark.parametrize("case", ["case30", "case118"])
def test_ipm_sparse_qp_with_device_kkt_ops(ctx, case):
    """The IPM driver with `solve_kkt!`/`mul!` on the device follows the CPU oracle run iteration by iteration
    (same acceptance as test_ipm_sparse_qp_cpu_vs_hip)."""
    from madnlp_jl_amd.problems import SparseQPModel
    nlp = SparseQPModel(case)
    so, sh = _ipm_pair("sparse_condensed", nlp, ctx, 1e-6, device_kkt_ops=True)
    assert sh.kkt.device_kkt_ops
    _assert_ipm_parity(so, sh, nlp.n)
    sh.kkt.close()


@pytest.mark.parametrize("n,m,n_eq", [(10, 5, 0), (50, 10, 0), (20, 15, 2)])
@pytest.mark.parametrize("kind", ["dense", "dense_condensed"])
def test_ipm_dense_qp_with_device_kkt_ops(ctx, kind, n, m, n_eq):
    """Dense twins of the device-side `solve_kkt!`/`mul!` (`mnk_dc_solve_kkt`, `mnk_dc_mul`: symv on the lower
    triangle of hess, gemv with jac, the condensed right-hand side and its expansion): the IPM driver with them
    switched on reproduces the CPU oracle's iteration history (DenseDummyQP sizes of test/madnlp_dense.jl)."""
    from madnlp_jl_amd.problems import DenseQPModel
    so, sh = _ipm_pair(kind, DenseQPModel(n, m, n_eq), ctx, 1e-8, device_kkt_ops=True)
    assert sh.kkt.device_kkt_ops
    _assert_ipm_parity(so, sh, n)
    sh.kkt.close()


@pytest.mark.parametrize("kind", ["dense", "dense_condensed"])
def test_dense_device_mul_matches_host_mirror(ctx, kind):
    """`mnk_dc_mul` against the host mirror of `mul!` (itself checked against the oracle) for several (alpha, beta)."""
    P = dense_dummy_qp(60, 25, 4 if kind == "dense_condensed" else 0)
    if kind == "dense_condensed":
        k = mj.DenseCondensedKKTSystem(P.n, P.m, P.ind_ineq, P.ind_eq, P.ind_lb, P.ind_ub, ctx=ctx)
    else:
        k = mj.DenseKKTSystem(P.n, P.m, P.ind_ineq, P.ind_lb, P.ind_ub, ctx=ctx)
    for f in ("reg", "l_diag", "u_diag", "l_lower", "u_lower", "du_diag"):
        getattr(k, f)[:] = getattr(P, f)
    k.hess[...] = P.hess
    k.jac[...] = P.jac
    k.set_aug_diagonal()
    k.build_kkt()
    k.upload_barrier_terms()
    rng = np.random.default_rng(4)
    x, w1, w2 = (mj.UnreducedKKTVector.from_kkt(k) for _ in range(3))
    xv, wv = rng.standard_normal(len(x.values)), rng.standard_normal(len(x.values))
    for alpha, beta in ((1.0, 0.0), (-1.0, 1.0), (0.3, -2.0)):
        x.values[:] = xv; w1.values[:] = wv; w2.values[:] = wv
        k.mul(w1, x, alpha, beta)
        k.mul_device(w2, x, alpha, beta)
        assert np.abs(w1.values - w2.values).max() <= 1e-12 * max(1.0, np.abs(w1.values).max()), (alpha, beta)
    k.close()


# --------------------------------------------------------------------------- edge cases of the domain
def _edge_system(ctx, n, m, jI, jJ, hI, hJ, seed):
    """Oracle and HIP sparse condensed systems on an arbitrary COO pattern (duplicates / upper entries allowed)."""
    rng = np.random.default_rng(seed)
    ind_ineq = np.arange(m)
    ind_lb = np.arange(0, n + m, 3)
    ind_ub = np.arange(1, n + m, 4)
    ko = osc.SparseCondensedKKTSystem(n, m, jI, jJ, hI, hJ, ind_ineq, ind_lb, ind_ub, lambda A: LapackCPUSolver(A, CHOLESKY))
    kh = mj.SparseCondensedKKTSystem(n, m, jI, jJ, hI, hJ, ind_ineq, ind_lb, ind_ub, ctx=ctx,
                                     opt_linear_solver=mj.HipSolverOptions(lapack_algorithm=mj.CHOLESKY))
    jac = rng.standard_normal(len(jI))
    hess = 0.1 * rng.standard_normal(len(hI))
    pr = np.concatenate((5.0 + rng.random(n), 10.0 ** rng.uniform(-3, 3, m)))
    for k in (ko, kh):
        k.jac[:] = jac
        k.hess[:] = hess
        k.pr_diag[:] = pr
        k.du_diag[:] = -1e-8
        k.compress_jacobian(); k.compress_hessian()
        k.build_kkt()
    return ko, kh


def test_duplicate_and_upper_triangular_coo_entries_bit_exact(ctx):
    """Collisions of the domain: repeated (i, j) entries in the callback's COO patterns share one CSC slot and
    are summed in COO order; Hessian entries reported in the upper triangle are folded to the lower one
    (reference src/matrixtools.jl:55-95,129-137).  Values of jt_csc, hess_com and aug_com equal the CPU
    restatement bit for bit."""
    n, m = 9, 6
    jI = np.array([0, 0, 0, 2, 2, 5, 5, 3, 1, 4, 4]); jJ = np.array([4, 1, 4, 0, 0, 8, 7, 3, 2, 6, 6])
    hI = np.array([0, 1, 3, 2, 2, 8, 7, 5]); hJ = np.array([2, 1, 0, 0, 0, 8, 8, 5])  # (0,2),(7,8) upper; (2,0) three times
    ko, kh = _edge_system(ctx, n, m, jI, jJ, hI, hJ, 1)
    np.testing.assert_array_equal(kh.jt_csc.data, ko.jt_csc.nzval)
    np.testing.assert_array_equal(kh.hess_com.data, ko.hess_com.nzval)
    np.testing.assert_array_equal(kh.aug_com.nzval_host(), ko.aug_com.nzval)
    kh.close()


@pytest.mark.parametrize("variant", ["no_constraints", "zero_hessian", "single_entry"])
def test_empty_patterns(ctx, variant):
    """Empty inputs of the domain: m = 0 (K = H + Sigma_x), an LP (no Hessian entries), a 1 x 1 system."""
    if variant == "no_constraints":
        n, m = 7, 0
        jI = jJ = np.zeros(0, int)
        hI = np.array([0, 3, 6, 6]); hJ = np.array([0, 1, 6, 2])
    elif variant == "zero_hessian":
        n, m = 6, 4
        jI = np.array([0, 1, 2, 3, 3]); jJ = np.array([0, 2, 4, 5, 1])
        hI = hJ = np.zeros(0, int)
    else:
        n, m = 1, 1
        jI = np.array([0]); jJ = np.array([0]); hI = np.array([0]); hJ = np.array([0])
    ko, kh = _edge_system(ctx, n, m, jI, jJ, hI, hJ, 2)
    np.testing.assert_array_equal(kh.aug_com.nzval_host(), ko.aug_com.nzval)
    for k in (ko, kh):
        k.linear_solver.factorize()
    assert kh.linear_solver.inertia() == ko.linear_solver.inertia() == (n, 0, 0)
    b = np.arange(1.0, n + 1.0)
    xo = ko.linear_solver.solve_linear_system(b.copy())
    xh = kh.linear_solver.solve_linear_system(b.copy())
    np.testing.assert_allclose(xh, xo, rtol=1e-12, atol=1e-14)
    kh.close()


def test_full_size_case9241_round_trip(ctx):
    """BASELINE's largest configuration (case9241pegase shape, N = 85 568, 58.6 GB dense factor): no oracle
    factorization at this size, so the size-independent property is checked -- assemble, factorize, solve, and
    the solution satisfies K x = b for the SPARSE K assembled by the bit-exact condensation (residual relative to
    |K| |x| + |b| below 1e-12), with the inertia the IPM expects."""
    free, _ = torch.cuda.mem_get_info()
    if free < 75e9:
        pytest.skip("needs ~62 GB of free HBM")
    P = opf_shaped("case9241pegase", du=1e-8)
    kh = _hip_sc(P, ctx, mj.BUNCHKAUFMAN)
    kh.compress_jacobian(); kh.compress_hessian()
    kh.set_aug_diagonal()
    kh.build_kkt()
    kh.linear_solver.factorize()
    assert kh.linear_solver.inertia() == (P.n, 0, 0)
    Kl = kh.aug_com.to_scipy()  # lower-triangular CSC from the device
    K = Kl + sp.tril(Kl, -1).T
    rng = np.random.default_rng(0)
    b = rng.standard_normal(P.n)
    x = kh.linear_solver.solve_linear_system(b.copy())
    res = np.abs(K @ x - b).max() / (np.abs(K).sum(axis=1).max() * np.abs(x).max() + np.abs(b).max())
    assert res <= 1e-12, res
    kh.close()


# --------------------------------------------------------------------------- interface contract corners
def test_solver_contract_errors_and_strided_inputs(ctx):
    """AbstractLinearSolver contract (reference src/LinearSolvers/linearsolvers.jl:13-137): a solve before a
    factorization and a short right-hand side raise SolveException, an algorithm the device does not implement
    raises at construction, a numerically failed factorization does NOT raise; the matrix may live in a larger
    device buffer (leading dimension > N) and a matrix right-hand side may be a strided device view."""
    rng = np.random.default_rng(12)
    N = 300
    R = rng.standard_normal((N, 20))
    A = np.asfortranarray(R @ R.T + N * np.eye(N))
    with pytest.raises(mj.SymbolicException):
        mj.HipLinearSolver(A, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm="LU"))
    M = mj.HipLinearSolver(A, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=mj.CHOLESKY))
    with pytest.raises(mj.SolveException):
        M.solve_linear_system(np.ones(N))
    M.factorize()
    with pytest.raises(mj.SolveException):
        M.solve_linear_system(np.ones(N - 1))
    M.close()
    # matrix inside a bigger device allocation: leading dimension 384 > N (column-major view via transpose)
    big = torch.zeros(384, 384, dtype=torch.float64, device="cuda")
    big[:N, :N] = torch.from_numpy(A).cuda()          # symmetric: row-major view == column-major view
    view = big[:N, :N]
    torch.cuda.synchronize()
    M = mj.HipLinearSolver(view, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=mj.CHOLESKY))
    M.factorize()
    assert M.inertia() == (N, 0, 0)
    B = torch.zeros(384, 4, dtype=torch.float64, device="cuda").t().contiguous().t()  # column-major 384 x 4
    Bh = rng.standard_normal((N, 4))
    B[:N, :] = torch.from_numpy(Bh).cuda()
    torch.cuda.synchronize()
    X = B[:N, :]                                       # strided view: ld 384
    M.solve_linear_system(X)
    ctx.synchronize()
    assert np.abs(A @ X.cpu().numpy() - Bh).max() <= 1e-10 * N
    M.close()
    # a factorization that fails numerically reports it through inertia, without raising
    Aneg = np.asfortranarray(-A)
    M = mj.HipLinearSolver(Aneg, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=mj.CHOLESKY))
    M.factorize()
    assert M.inertia() == (0, N, 0)
    M.close()


def test_ldl_pivot_tolerance_reports_tiny_pivots_as_zero(ctx):
    """`pivot_tol`: a pivot with |d| <= pivot_tol is counted as num_zero (and replaced by 1) so the IPM's inertia
    correction regularizes, instead of dividing by it."""
    N = 130
    d = np.ones(N)
    d[70] = 1e-13
    A = np.asfortranarray(np.diag(d))
    M = mj.HipLinearSolver(A, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=mj.LDL, pivot_tol=1e-10))
    M.factorize()
    assert M.inertia() == (N - 1, 1, 0)
    M.close()
    M = mj.HipLinearSolver(A, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=mj.LDL))
    M.factorize()
    assert M.inertia() == (N, 0, 0)  # pivot_tol = 0: only an exact zero counts, as in LAPACK
    M.close()
