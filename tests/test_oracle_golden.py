"""Pins the CPU oracle against every known-answer test the reference holds for
the KKT hot path.  CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import hs15
from oracle.dense import DenseCondensedKKTSystem, DenseKKTSystem
from oracle.kernels import UnreducedKKTVector, set_aug_diagonal
from oracle.lapack_cpu import BUNCHKAUFMAN, CHOLESKY, EVD, LU, LapackCPUSolver
from oracle.matrixtools import coo_to_csc
from oracle.sparse_condensed import SparseCondensedKKTSystem, sym_length

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def solcmp(x, sol, atol=1e-4, rtol=1e-4):
    """reference `solcmp` lib/MadNLPTests/src/MadNLPTests.jl:18-22."""
    aerr = np.linalg.norm(x - sol, np.inf)
    return aerr < atol or aerr / np.linalg.norm(sol, np.inf) < rtol


@pytest.mark.parametrize("alg", [BUNCHKAUFMAN, CHOLESKY, LU, EVD])
@pytest.mark.parametrize("as_csc", [False, True])
def test_linear_solver_known_answer(alg, as_csc):
    """reference test/matrix_test.jl:21-30 and MadNLPTests.test_linear_solver (:24-51)."""
    row, col, val = np.array([0, 1, 1]), np.array([0, 0, 1]), np.array([1.0, 0.1, 2.0])
    csc, mp = coo_to_csc(2, 2, row, col)
    csc.nzval[mp] = val
    A = csc if as_csc else csc.to_dense()
    M = LapackCPUSolver(A, alg)
    assert "Lapack-CPU" in M.introduce()
    assert M.improve() is False
    M.factorize()
    if alg != LU:
        assert M.inertia() == (2, 0, 0)
    x = M.solve_linear_system(np.array([1.0, 3.0]))
    assert solcmp(x, np.array([0.8542713567839195, 1.4572864321608041]))
    np.testing.assert_allclose(x, [0.8542713567839195, 1.4572864321608041], rtol=1e-14)


def _make(kind):
    fac = lambda A: LapackCPUSolver(A, BUNCHKAUFMAN)  # noqa: E731
    if kind == "sparse_condensed":
        return SparseCondensedKKTSystem(hs15.N, hs15.M, hs15.JAC_I, hs15.JAC_J, hs15.HESS_I,
                                        hs15.HESS_J, hs15.IND_INEQ, hs15.IND_LB, hs15.IND_UB, fac)
    if kind == "dense_condensed":
        return DenseCondensedKKTSystem(hs15.N, hs15.M, hs15.IND_INEQ, hs15.IND_EQ, hs15.IND_LB,
                                       hs15.IND_UB, fac)
    return DenseKKTSystem(hs15.N, hs15.M, hs15.IND_INEQ, hs15.IND_LB, hs15.IND_UB, fac)


def run_test_kkt_system(kkt, sparse):
    """reference `test_kkt_system` lib/MadNLPTests/src/MadNLPTests.jl:53-110."""
    m, p = kkt.size()
    assert m == p
    kkt.initialize()
    x0, y0 = np.zeros(2), np.zeros(2)
    if sparse:
        kkt.get_jacobian()[:] = hs15.jac_coord(x0)
        kkt.get_hessian()[:] = hs15.hess_coord(x0, y0)
    else:
        kkt.get_jacobian()[...] = hs15.jac_dense(x0)
        kkt.get_hessian()[...] = hs15.hess_dense(x0, y0)
    kkt.compress_jacobian()
    kkt.compress_hessian()
    kkt.l_lower[:] = 1e-3
    kkt.u_lower[:] = 1e-3
    set_aug_diagonal(kkt)
    kkt.build_kkt()
    kkt.linear_solver.factorize()
    x = UnreducedKKTVector.from_kkt(kkt)
    x.values[:] = 1.0
    out1 = kkt.solve_kkt(x)
    assert out1 is x
    y = x.copy()
    y.values[:] = 0.0
    kkt.mul(y, x)
    np.testing.assert_allclose(y.values, np.ones(len(x.values)), rtol=0, atol=1e-13)
    ni, mi, pi = kkt.linear_solver.inertia()
    assert kkt.is_inertia_correct(ni, mi, pi)
    sol = x.values.copy()
    kkt.regularize_diagonal(1.0, 1.0)
    return sol


@pytest.mark.parametrize("kind", ["sparse_condensed", "dense_condensed", "dense"])
def test_kkt_system_hs15(kind):
    """reference test/kkt_test.jl:27-48."""
    kkt = _make(kind)
    sol = run_test_kkt_system(kkt, sparse=(kind == "sparse_condensed"))
    gold = json.load(open(os.path.join(GOLDEN, "hs15_kkt.json")))
    # pr_diag and condensed K for HS15.
    np.testing.assert_allclose(kkt.pr_diag - 1.0, np.array(gold["pr_diag"]), rtol=1e-15)
    np.testing.assert_allclose(sol, np.array(gold["oracle_solve_kkt_ones"]), rtol=1e-12, atol=1e-14)
    if kind == "dense_condensed":
        np.testing.assert_allclose(kkt.aug_com, np.diag(gold["K_condensed_diag"]), rtol=1e-15)
    if kind == "sparse_condensed":
        np.testing.assert_allclose(kkt.aug_com.to_dense(), np.diag(gold["K_condensed_diag"]),
                                   rtol=1e-15)


def test_symbolic_against_dense_formula():
    """build_condensed_aug_symbolic + coord == tril(H + diag + J'DJ)."""
    rng = np.random.default_rng(7)
    n, m = 13, 17
    dens = rng.random((m, n)) < 0.3
    dens[np.arange(m), rng.integers(0, n, m)] = True
    jI, jJ = np.nonzero(dens)
    # duplicates in the COO pattern must share a CSC slot
    jI = np.concatenate((jI, jI[:5]))
    jJ = np.concatenate((jJ, jJ[:5]))
    hd = np.tril(rng.random((n, n)) < 0.25)
    hI, hJ = np.nonzero(hd)
    hI, hJ = np.concatenate((hI, hJ[:3])), np.concatenate((hJ, hI[:3]))  # some upper entries
    kkt = SparseCondensedKKTSystem(n, m, jI, jJ, hI, hJ, np.arange(m), np.arange(n, n + m),
                                   np.arange(3), lambda A: LapackCPUSolver(A, CHOLESKY))
    assert len(kkt.jptr[0]) == sym_length(kkt.jt_csc)
    kkt.jac[:] = rng.standard_normal(len(jI))
    kkt.hess[:] = rng.standard_normal(len(hI))
    kkt.pr_diag[:] = rng.random(n + m) + 0.5
    kkt.du_diag[:] = -rng.random(m) * 1e-2
    kkt.compress_jacobian()
    kkt.compress_hessian()
    kkt.build_kkt()
    J = np.zeros((m, n))
    np.add.at(J, (jI, jJ), kkt.jac)
    H = np.zeros((n, n))
    lo_i, lo_j = np.maximum(hI, hJ), np.minimum(hI, hJ)
    np.add.at(H, (lo_i, lo_j), kkt.hess)
    D = kkt.pr_diag[n:] / (1 - kkt.du_diag * kkt.pr_diag[n:])
    Kref = np.tril(H + np.diag(kkt.pr_diag[:n]) + J.T @ np.diag(D) @ J)
    np.testing.assert_allclose(kkt.aug_com.to_dense(), Kref, rtol=1e-13, atol=1e-13)
    # rows sorted within each column, lower triangular
    for c in range(n):
        r = kkt.aug_com.rowval[kkt.aug_com.colptr[c]:kkt.aug_com.colptr[c + 1]]
        assert np.all(np.diff(r) > 0) and np.all(r >= c)
