"""Oracle restatement of reference `src/matrixtools.jl` (TEST INFRASTRUCTURE ONLY).

0-based indices throughout.  CSC matrices are plain tuples/objects of numpy
arrays so that the structures can be compared entry-wise with what the C-ABI
library exports.
"""
from __future__ import annotations

import numpy as np


class CSC:
    """Minimal stand-in for SparseMatrixCSC{Float64,Int32} (0-based)."""

    def __init__(self, m, n, colptr, rowval, nzval=None):
        self.m = int(m)
        self.n = int(n)
        self.colptr = np.asarray(colptr, dtype=np.int64)
        self.rowval = np.asarray(rowval, dtype=np.int64)
        self.nzval = (np.zeros(len(self.rowval)) if nzval is None
                      else np.asarray(nzval, dtype=np.float64))

    @property
    def nnz(self):
        return len(self.rowval)

    def colidx(self):
        """Column index of every stored entry."""
        return np.repeat(np.arange(self.n, dtype=np.int64), np.diff(self.colptr))

    def to_dense(self):
        """`copyto!(fact, A)` for a SparseMatrixCSC source: zero-fill + scatter
        (reference `src/LinearSolvers/lapack_common.jl:28`)."""
        d = np.zeros((self.m, self.n), order="F")
        d[self.rowval, self.colidx()] = self.nzval
        return d

    def matvec(self, x):
        y = np.zeros(self.m)
        np.add.at(y, self.rowval, self.nzval * x[self.colidx()])
        return y

    def rmatvec(self, x):
        """A' * x."""
        y = np.zeros(self.n)
        np.add.at(y, self.colidx(), self.nzval * x[self.rowval])
        return y

    def symmetric_lower_matvec(self, x):
        """Symmetric(A, :L) * x for a lower-triangular-stored A."""
        ci = self.colidx()
        y = np.zeros(self.m)
        np.add.at(y, self.rowval, self.nzval * x[ci])
        off = self.rowval != ci
        np.add.at(y, ci[off], self.nzval[off] * x[self.rowval[off]])
        return y


def force_lower_triangular(I, J):
    """reference `src/matrixtools.jl:129-137`: swap so that row >= col."""
    I = np.array(I, dtype=np.int64, copy=True)
    J = np.array(J, dtype=np.int64, copy=True)
    swap = J > I
    I[swap], J[swap] = J[swap], I[swap]
    return I, J


def coo_to_csc(m, n, I, J):
    """reference `src/matrixtools.jl:55-95` (`coo_to_csc` + `get_mapping`).

    Returns (csc, map) where csc has the *structure* of sparse(I, J, 1) (sorted
    rows within each column, duplicates merged) and zero values, and map[k] is
    the CSC nz slot of COO entry k (duplicates share one slot).
    """
    I = np.asarray(I, dtype=np.int64)
    J = np.asarray(J, dtype=np.int64)
    key = J * m + I
    ukey, inv = np.unique(key, return_inverse=True)
    rowval = ukey % m
    cols = ukey // m
    colptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(colptr, cols + 1, 1)
    colptr = np.cumsum(colptr)
    return CSC(m, n, colptr, rowval), inv.astype(np.int64)


def transfer(dest_nz, src_vals, map_):
    """reference `_transfer!` `src/matrixtools.jl:79-84`: zero-fill, then
    sequential scatter-add in COO order (np.add.at keeps that order)."""
    dest_nz[:] = 0.0
    np.add.at(dest_nz, map_, src_vals)
    return dest_nz


def diag(dest, src):
    """reference `diag!` `src/matrixtools.jl:34-39`."""
    dest[:] = np.diagonal(src)
    return dest


def tril_to_full(dense):
    """reference `tril_to_full!` `src/matrixtools.jl:47-53` (LU/QR only)."""
    il = np.tril_indices(dense.shape[0], -1)
    dense[il[1], il[0]] = dense[il]
    return dense
