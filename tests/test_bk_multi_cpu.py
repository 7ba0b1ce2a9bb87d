"""The multi-workgroup Bunch-Kaufman panel's data flow (csrc/bk.hip, bkp_panel_mw_kernel) on the host: tools/bk_mw_model.py
runs one object per workgroup that only sees its own rows, the replicated table of panel positions and the hop messages,
and must take exactly the pivots of a plain dsytf2 restatement with physical interchanges (reference:
src/LinearSolvers/lapack.jl:164-167 calls dsytrf; its unblocked kernel's pivoting rule is what both restate)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import bk_mw_model as M  # noqa: E402


def _matrix(kind, N, rng):
    if kind == "random":
        S = rng.standard_normal((N, N))
        return (S + S.T) / 2
    if kind == "saddle":                     # [[H, J'], [J, 0]] with an indefinite H: 2x2 pivots
        n1 = 2 * N // 3
        H = rng.standard_normal((n1, n1)); H = (H + H.T) / 2
        J = rng.standard_normal((N - n1, n1))
        A = np.zeros((N, N)); A[:n1, :n1] = H; A[n1:, :n1] = J; A[:n1, n1:] = J.T
        return A
    A = rng.standard_normal((N, N)); A = A + A.T     # exact zeros on the diagonal and one zero row / column
    A[np.arange(0, N, 7), np.arange(0, N, 7)] = 0.0
    A[:, 5] = 0.0; A[5, :] = 0.0
    return A


@pytest.mark.parametrize("kind", ["random", "saddle", "zeros"])
@pytest.mark.parametrize("N,rows_per_wg", [(40, 8), (150, 16), (200, 256), (130, 4)])
def test_model_takes_the_pivots_of_the_plain_factorization(kind, N, rows_per_wg, monkeypatch):
    monkeypatch.setattr(M, "T", rows_per_wg)
    A = _matrix(kind, N, np.random.default_rng(N))
    L, d, off, pt, perm, info, hops = M.factor(A)
    L2, d2, off2, pt2, perm2, info2 = M.factor_plain(A)
    assert np.array_equal(perm, perm2) and np.array_equal(pt, pt2) and info == info2
    assert np.abs(d - d2).max() <= 1e-11 * np.abs(d2).max() and np.abs(L - L2).max() <= 1e-9 * max(1.0, np.abs(L2).max())
    R = M.reconstruct(L, d, off, pt)
    assert np.abs(R - A[np.ix_(perm, perm)]).max() <= 1e-12 * np.abs(A).max() * N
    assert hops <= 2 * N                      # one message round per column, a second one only when the 1x1 test fails
