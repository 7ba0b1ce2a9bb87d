"""Oracle restatement of reference `src/KKT/Sparse/condensed.jl` and the shared
sparse utilities `src/KKT/Sparse/utils.jl` (TEST INFRASTRUCTURE ONLY)."""
from __future__ import annotations

import numpy as np

from .matrixtools import CSC, coo_to_csc, force_lower_triangular, transfer
from . import kernels as K


def sym_length(Jt: CSC) -> int:
    """reference `_sym_length` `src/KKT/Sparse/condensed.jl:158-165`."""
    k = np.diff(Jt.colptr)
    return int(np.sum(k * (k + 1) // 2))


def _jt_pairs(Jt: CSC):
    """Enumerate (c, j, k) with colptr[c] <= j <= k < colptr[c+1] in the nested
    order of reference `_build_condensed_aug_symbolic_jt`
    (`src/KKT/Sparse/condensed.jl:177-190`): c outer, j middle, k inner."""
    cnt = np.diff(Jt.colptr)
    total = int(np.sum(cnt * (cnt + 1) // 2))
    cc = np.empty(total, dtype=np.int64)
    jj = np.empty(total, dtype=np.int64)
    kk = np.empty(total, dtype=np.int64)
    # start offset of every column's block of pairs
    per_col = cnt * (cnt + 1) // 2
    off = np.concatenate(([0], np.cumsum(per_col)))
    for kc in np.unique(cnt):
        if kc == 0:
            continue
        cols = np.nonzero(cnt == kc)[0]
        a, b = np.triu_indices(int(kc))  # row-major upper incl. diagonal: j outer, k inner
        base = Jt.colptr[cols][:, None]
        dst = (off[cols][:, None] + np.arange(len(a))[None, :]).ravel()
        cc[dst] = np.repeat(cols, len(a))
        jj[dst] = (base + a[None, :]).ravel()
        kk[dst] = (base + b[None, :]).ravel()
    return cc, jj, kk


def build_condensed_aug_symbolic(H: CSC, Jt: CSC):
    """reference `build_condensed_aug_symbolic` `src/KKT/Sparse/condensed.jl:201-301`.

    Returns (aug_com, dptr, hptr, jptr):
      dptr = (dst[n], src[n])           K.nz[dst] += pr_diag[src]
      hptr = (dst[nnzH], src[nnzH])     K.nz[dst] += H.nz[src]
      jptr = (dst[L], c[L], k[L], l[L]) K.nz[dst] += D[c] * Jt.nz[k] * Jt.nz[l]
    Each list is ordered as in the reference after its (stable) sort by
    (col, row) of the destination, i.e. non-decreasing in dst.
    """
    n = H.n
    nnzh = H.nnz
    cc, jj, kk = _jt_pairs(Jt)
    L = len(cc)
    # sym2 = (row, col) of destination; sym = source descriptor.
    kind = np.concatenate((np.full(n, -1), np.zeros(nnzh, dtype=np.int64), np.ones(L, dtype=np.int64)))
    row = np.concatenate((np.arange(n), H.rowval, Jt.rowval[kk]))
    col = np.concatenate((np.arange(n), H.colidx(), Jt.rowval[jj]))
    a = np.concatenate((np.arange(n), np.arange(nnzh), cc))
    b = np.concatenate((np.zeros(n + nnzh, dtype=np.int64), jj))
    c = np.concatenate((np.zeros(n + nnzh, dtype=np.int64), kk))

    p = np.argsort(col * n + row, kind="stable")
    kind, row, col, a, b, c = kind[p], row[p], col[p], a[p], b[p], c[p]
    key = col * n + row
    new = np.ones(len(key), dtype=bool)
    new[1:] = key[1:] != key[:-1]
    guide = np.cumsum(new) - 1  # 0-based destination slot

    ptr = np.nonzero(new)[0]
    rowval = row[ptr]
    colptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(colptr, col[ptr] + 1, 1)
    colptr = np.cumsum(colptr)
    aug = CSC(n, n, colptr, rowval)

    isd, ish, isj = kind == -1, kind == 0, kind == 1
    dptr = (guide[isd], a[isd])
    hptr = (guide[ish], a[ish])
    jptr = (guide[isj], a[isj], b[isj], c[isj])
    return aug, dptr, hptr, jptr


def build_condensed_aug_coord(aug_nz, pr_diag, H_nz, Jt_nz, diag_buffer, dptr, hptr, jptr):
    """reference `_build_condensed_aug_coord!` `src/KKT/Sparse/condensed.jl:328-345`.
    Accumulation order is kept: zero, all hptr terms, all dptr terms, all jptr terms."""
    aug_nz[:] = 0.0
    np.add.at(aug_nz, hptr[0], H_nz[hptr[1]])
    np.add.at(aug_nz, dptr[0], pr_diag[dptr[1]])
    np.add.at(aug_nz, jptr[0], (diag_buffer[jptr[1]] * Jt_nz[jptr[2]]) * Jt_nz[jptr[3]])
    return aug_nz


class SparseCondensedKKTSystem:
    """reference struct + ctor `src/KKT/Sparse/condensed.jl:8-133`.

    `jac_I/jac_J` (constraint row, variable col) and `hess_I/hess_J` are the
    0-based COO sparsity patterns the callback reports; all constraints must
    be inequalities (`:68-70`)."""

    def __init__(self, n, m, jac_I, jac_J, hess_I, hess_J, ind_ineq, ind_lb, ind_ub,
                 linear_solver_factory):
        if len(ind_ineq) != m:
            raise ValueError("SparseCondensedKKTSystem does not support equality constrained NLPs.")
        self.n, self.m = n, m
        hI, hJ = force_lower_triangular(hess_I, hess_J)
        self.hess = np.zeros(len(hI))
        self.jac = np.zeros(len(jac_I))
        self.hess_raw = (hI, hJ)
        # jt_coo = J' : rows = variable, cols = constraint (`:104-109`)
        self.jt_coo = (np.asarray(jac_J, dtype=np.int64), np.asarray(jac_I, dtype=np.int64))
        self.jt_csc, self.jt_csc_map = coo_to_csc(n, m, self.jt_coo[0], self.jt_coo[1])
        self.hess_com, self.hess_csc_map = coo_to_csc(n, n, hI, hJ)
        self.aug_com, self.dptr, self.hptr, self.jptr = build_condensed_aug_symbolic(
            self.hess_com, self.jt_csc)
        nlb, nub = len(ind_lb), len(ind_ub)
        self.reg = np.zeros(n + m)
        self.pr_diag = np.zeros(n + m)
        self.du_diag = np.zeros(m)
        self.l_diag = np.zeros(nlb)
        self.u_diag = np.zeros(nub)
        self.l_lower = np.zeros(nlb)
        self.u_lower = np.zeros(nub)
        self.buffer = np.zeros(m)
        self.buffer2 = np.zeros(m)
        self.diag_buffer = np.zeros(m)
        self.ind_ineq = np.asarray(ind_ineq, dtype=np.int64)
        self.ind_lb = np.asarray(ind_lb, dtype=np.int64)
        self.ind_ub = np.asarray(ind_ub, dtype=np.int64)
        self.linear_solver = linear_solver_factory(self.aug_com)

    # -- interface -----------------------------------------------------------------
    def num_variables(self):
        return len(self.pr_diag)

    def size(self):
        return (self.aug_com.m, self.aug_com.n)

    def initialize(self):
        """sparse `initialize!` reference `src/KKT/Sparse/utils.jl:52-62`."""
        K.initialize(self)
        self.l_lower[:] = 0.0
        self.u_lower[:] = 0.0
        self.l_diag[:] = 1.0
        self.u_diag[:] = 1.0
        self.hess_com.nzval[:] = 0.0

    def get_jacobian(self):
        return self.jac

    def get_hessian(self):
        return self.hess

    def compress_jacobian(self):
        """`:145-148` -> `transfer!`."""
        transfer(self.jt_csc.nzval, self.jac, self.jt_csc_map)

    def compress_hessian(self):
        """reference `src/KKT/Sparse/utils.jl:48-50`."""
        transfer(self.hess_com.nzval, self.hess, self.hess_csc_map)

    def build_kkt(self):
        """`:354-366`."""
        n, m = self.n, self.m
        Sx = self.pr_diag[:n]  # noqa: F841  (only the x block enters K through dptr)
        Ss = self.pr_diag[n:n + m]
        Sd = self.du_diag
        self.diag_buffer[:] = Ss / (1.0 - Sd * Ss)
        build_condensed_aug_coord(self.aug_com.nzval, self.pr_diag, self.hess_com.nzval,
                                  self.jt_csc.nzval, self.diag_buffer,
                                  self.dptr, self.hptr, self.jptr)

    def is_inertia_correct(self, num_pos, num_zero, num_neg):
        """`:138-140`."""
        return num_zero == 0 and num_pos == self.aug_com.m

    def should_regularize_dual(self, num_pos, num_zero, num_neg):
        """`:141`."""
        return True

    def regularize_diagonal(self, primal, dual):
        K.regularize_diagonal(self, primal, dual)

    def jtprod(self, y, x):
        """`:150-156`."""
        n = self.n
        y[:n] = self.jt_csc.matvec(x)
        y[n:] = -x
        return y

    def solve_kkt(self, w):
        """reference `src/IPM/factorization.jl:143-167`."""
        n, m = self.n, self.m
        full = w.values
        wx = full[:n]
        ws = full[n:n + m]
        wz = full[n + m:n + 2 * m]
        Ss = self.pr_diag[n:n + m]
        K.reduce_rhs(self, w)
        self.buffer[:] = self.diag_buffer * (wz + ws / Ss)
        wx += self.jt_csc.matvec(self.buffer)
        self.linear_solver.solve_linear_system(wx)
        self.buffer2[:] = self.jt_csc.rmatvec(wx)
        wz[:] = -self.buffer + self.diag_buffer * self.buffer2
        ws[:] = (ws + wz) / Ss
        K.finish_aug_solve(self, w)
        return w

    def mul(self, w, x, alpha=1.0, beta=0.0):
        """reference `src/IPM/factorization.jl:278-299`."""
        n, m = self.n, self.m
        xf, wf = x.values, w.values
        xx, xs, xz = xf[:n], xf[n:n + m], xf[n + m:n + 2 * m]
        wx, ws, wz = wf[:n], wf[n:n + m], wf[n + m:n + 2 * m]
        wx[:] = alpha * self.hess_com.symmetric_lower_matvec(xx) + beta * wx
        wx += alpha * self.jt_csc.matvec(xz)
        wz[:] = alpha * self.jt_csc.rmatvec(xx) + beta * wz
        wz -= alpha * xs
        ws[:] = beta * ws - alpha * xz
        K.kktmul(w, x, self.reg, self.du_diag, self.l_lower, self.u_lower,
                 self.l_diag, self.u_diag, alpha, beta)
        return w

    def mul_hess_blk(self, wx, t):
        """reference `src/IPM/factorization.jl:333-338`."""
        n = self.n
        wx[:n] = self.hess_com.symmetric_lower_matvec(t[:n])
        wx[n:] = 0.0
        wx += t * self.pr_diag
        return wx
