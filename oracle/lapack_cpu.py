"""Oracle restatement of reference `src/LinearSolvers/lapack.jl` and
`src/LinearSolvers/lapack_common.jl` (TEST INFRASTRUCTURE ONLY).

The arithmetic is LAPACK dsytrf/dsytrs/dpotrf/dpotrs from scipy's bundled
OpenBLAS -- the same routine family the reference reaches through
libblastrampoline/OpenBLAS32_jll (`src/LinearSolvers/lapack.jl:56-138`).
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import lapack

from .matrixtools import CSC, tril_to_full

BUNCHKAUFMAN, LU, QR, CHOLESKY, EVD = "BUNCHKAUFMAN", "LU", "QR", "CHOLESKY", "EVD"


class SolveException(Exception):
    pass


def num_neg_ev(n, D, ipiv):
    """reference `num_neg_ev` `src/LinearSolvers/lapack.jl:247-268`.

    `ipiv` is LAPACK's (1-based, negative for 2x2 blocks) pivot vector."""
    numneg = 0
    t = 0.0
    for k in range(n):
        d = D[k, k]
        if ipiv[k] < 0:
            if t == 0:
                t = abs(D[k + 1, k])
                d = (d / t) * D[k + 1, k + 1] - t
            else:
                d = t
                t = 0.0
        if d < 0:
            numneg += 1
        if d == 0:
            numneg = -1
            break
    return numneg


def inertia_bk(fact, ipiv, info):
    """reference `inertia(fact, ipiv, info)` `src/LinearSolvers/lapack.jl:240-245`."""
    n = fact.shape[0]
    numneg = num_neg_ev(n, fact, ipiv)
    numzero = 1 if info > 0 else 0
    numpos = n - numneg - numzero
    return (numpos, numzero, numneg)


class LapackCPUSolver:
    """reference `LapackCPUSolver` (`src/LinearSolvers/lapack.jl:5-44`).

    Keeps a *reference* to A (dense column-major ndarray or oracle CSC) and a
    private dense `fact`; `factorize` first does `transfer_matrix!`
    (`lapack_common.jl:28`): dense copy or CSC -> dense zero-fill + scatter."""

    def __init__(self, A, algorithm=BUNCHKAUFMAN):
        self.A = A
        self.n = A.m if isinstance(A, CSC) else A.shape[0]
        self.algorithm = algorithm
        self.fact = np.zeros((self.n, self.n), order="F")
        self.ipiv = np.zeros(self.n, dtype=np.int32)
        self.info = 0
        self.lwork = None
        if algorithm == BUNCHKAUFMAN:  # setup_bunchkaufman! workspace query `:155-162`
            work, info = lapack.dsytrf_lwork(self.n, lower=1)
            self.lwork = int(work)

    def introduce(self):
        return f"Lapack-CPU ({self.algorithm})"

    def improve(self):
        return False

    @staticmethod
    def input_type():
        return "dense"

    def is_inertia(self):
        """`lapack_common.jl:91-94`, `lapack.jl:47`."""
        return self.algorithm in (CHOLESKY, EVD, BUNCHKAUFMAN)

    def transfer_matrix(self):
        if isinstance(self.A, CSC):
            self.fact = self.A.to_dense()
        else:
            self.fact = np.array(self.A, order="F", copy=True)

    def factorize(self):
        """`lapack_common.jl:54-66`."""
        self.transfer_matrix()
        if self.algorithm == BUNCHKAUFMAN:
            self.fact, self.ipiv_py, self.info = lapack.dsytrf(
                self.fact, lower=1, lwork=self.lwork, overwrite_a=1)
            # scipy returns LAPACK's ipiv (1-based, negative for 2x2 blocks) unchanged.
            self.ipiv = self.ipiv_py
        elif self.algorithm == CHOLESKY:
            self.fact, self.info = lapack.dpotrf(self.fact, lower=1, clean=0, overwrite_a=1)
        elif self.algorithm == LU:
            tril_to_full(self.fact)
            self.fact, self.ipiv, self.info = lapack.dgetrf(self.fact, overwrite_a=1)
        elif self.algorithm == EVD:
            self.Lam, self.fact, self.info = lapack.dsyevd(self.fact, compute_v=1, lower=1,
                                                          overwrite_a=1)
        else:
            raise NotImplementedError(self.algorithm)
        return self

    def inertia(self):
        """`lapack_common.jl:96-109`, `lapack.jl:48`."""
        if self.algorithm == BUNCHKAUFMAN:
            return inertia_bk(self.fact, self.ipiv, self.info)
        if self.algorithm == CHOLESKY:
            return (self.n, 0, 0) if self.info == 0 else (0, self.n, 0)
        if self.algorithm == EVD:
            numpos = int(np.sum(self.Lam > 0))
            numneg = int(np.sum(self.Lam < 0))
            return (numpos, self.n - numpos - numneg, numneg)
        raise NotImplementedError

    def solve_linear_system(self, x):
        """`lapack_common.jl:75-81`: in place on the caller's vector."""
        if self.algorithm == BUNCHKAUFMAN:
            sol, info = lapack.dsytrs(self.fact, self.ipiv, x, lower=1)
        elif self.algorithm == CHOLESKY:
            sol, info = lapack.dpotrs(self.fact, x, lower=1)
        elif self.algorithm == LU:
            sol, info = lapack.dgetrs(self.fact, self.ipiv, x)
        elif self.algorithm == EVD:
            sol = self.fact @ ((self.fact.T @ x) / self.Lam)
        else:
            raise NotImplementedError
        x[:] = sol.reshape(x.shape)
        return x
