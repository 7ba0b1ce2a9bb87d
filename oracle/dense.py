"""Oracle restatement of reference `src/KKT/Dense/{augmented,condensed,utils}.jl`
and the dense `solve_kkt!`/`mul!` methods of `src/IPM/factorization.jl`
(TEST INFRASTRUCTURE ONLY).  Dense matrices are column-major (order="F")."""
from __future__ import annotations

import numpy as np

from . import kernels as K


def _symv_lower(A, x):
    """BLAS dsymv('L'): only the lower triangle of A is referenced."""
    L = np.tril(A)
    return L @ x + np.tril(A, -1).T @ x


class _DenseCommon:
    def get_hessian(self):
        return self.hess

    def initialize(self):
        K.initialize(self)

    def regularize_diagonal(self, primal, dual):
        K.regularize_diagonal(self, primal, dual)

    def compress_jacobian(self):
        """`src/KKT/Dense/utils.jl:25-27`: no-op."""
        return

    def jtprod(self, y, x):
        """`src/KKT/Dense/utils.jl:12-23`."""
        nx = self.hess.shape[0]
        ns = len(self.ind_ineq)
        y[:nx] = self.jac[:, :nx].T @ x
        y[nx:nx + ns] = -x[self.ind_ineq]
        return y

    def mul(self, w, x, alpha=1.0, beta=0.0):
        """reference `src/IPM/factorization.jl:301-324` (AbstractDenseKKTSystem)."""
        m = self.jac.shape[0]
        n = self.hess.shape[0]
        jac = self.jac[:, :n]
        wp, xp = w.primal(), x.primal()
        wx, ws = wp[:n], wp[n:]
        xx, xs = xp[:n], xp[n:]
        wy, xy = w.dual(), x.dual()
        wx[:] = alpha * _symv_lower(self.hess, xx) + beta * wx
        if m > 0:
            wx += alpha * (jac.T @ xy)
            wy[:] = alpha * (jac @ xx) + beta * wy
        ws[:] = beta * ws - alpha * xy[self.ind_ineq]
        wy[self.ind_ineq] -= alpha * xs
        K.kktmul(w, x, self.reg, self.du_diag, self.l_lower, self.u_lower,
                 self.l_diag, self.u_diag, alpha, beta)
        return w

    def mul_hess_blk(self, wx, t):
        """reference `src/IPM/factorization.jl:326-331`."""
        n = self.hess.shape[0]
        wx[:n] = _symv_lower(self.hess, t[:n])
        wx[n:] = 0.0
        wx += t * self.pr_diag
        return wx


class DenseKKTSystem(_DenseCommon):
    """reference `src/KKT/Dense/augmented.jl:10-94` (order N = n + ns + m)."""

    def __init__(self, n, m, ind_ineq, ind_lb, ind_ub, linear_solver_factory):
        ns = len(ind_ineq)
        self.n, self.m, self.ns = n, m, ns
        self.hess = np.zeros((n, n), order="F")
        self.jac = np.zeros((m, n), order="F")
        self.reg = np.zeros(n + ns)
        self.pr_diag = np.zeros(n + ns)
        self.du_diag = np.zeros(m)
        self.diag_hess = np.zeros(n)
        nlb, nub = len(ind_lb), len(ind_ub)
        self.l_diag = np.ones(nlb)
        self.u_diag = np.ones(nub)
        self.l_lower = np.zeros(nlb)
        self.u_lower = np.zeros(nub)
        self.ind_ineq = np.asarray(ind_ineq, dtype=np.int64)
        self.ind_lb = np.asarray(ind_lb, dtype=np.int64)
        self.ind_ub = np.asarray(ind_ub, dtype=np.int64)
        N = n + ns + m
        self.aug_com = np.zeros((N, N), order="F")
        self.linear_solver = linear_solver_factory(self.aug_com)

    def num_variables(self):
        return len(self.pr_diag)

    def size(self):
        return self.aug_com.shape

    def get_jacobian(self):
        return self.jac

    def compress_hessian(self):
        """`augmented.jl:158-161`: diag!(diag_hess, hess)."""
        self.diag_hess[:] = np.diagonal(self.hess)

    def build_kkt(self):
        """`_build_dense_kkt_system!` `augmented.jl:116-145`: writes both triangles;
        entries never touched stay as they were (zero from the ctor)."""
        n, m, ns = self.n, self.m, self.ns
        d = self.aug_com
        il = np.tril_indices(n, -1)
        d[il] = self.hess[il]
        d[il[1], il[0]] = self.hess[il[1], il[0]]
        ii = np.arange(n)
        d[ii, ii] = self.pr_diag[:n] + self.diag_hess
        si = np.arange(ns) + n
        d[si, si] = self.pr_diag[n:n + ns]
        d[n + ns:, :n] = self.jac
        d[:n, n + ns:] = self.jac.T
        ri = self.ind_ineq + n + ns
        d[ri, si] = -1.0
        d[si, ri] = -1.0
        yi = np.arange(m) + n + ns
        d[yi, yi] = self.du_diag

    def is_inertia_correct(self, num_pos, num_zero, num_neg):
        """generic `src/KKT/KKTsystem.jl:242-244`."""
        return num_zero == 0 and num_pos == self.num_variables()

    def should_regularize_dual(self, num_pos, num_zero, num_neg):
        return num_zero != 0

    def solve_kkt(self, w):
        """AbstractReducedKKTSystem `src/IPM/factorization.jl:41-46`."""
        K.reduce_rhs(self, w)
        self.linear_solver.solve_linear_system(w.primal_dual())
        K.finish_aug_solve(self, w)
        return w


class DenseCondensedKKTSystem(_DenseCommon):
    """reference `src/KKT/Dense/condensed.jl:10-111` (order N = n + n_eq)."""

    def __init__(self, n, m, ind_ineq, ind_eq, ind_lb, ind_ub, linear_solver_factory):
        ns = len(ind_ineq)
        n_eq = m - ns
        assert n_eq == len(ind_eq)
        self.n, self.m, self.n_ineq, self.n_eq = n, m, ns, n_eq
        self.hess = np.zeros((n, n), order="F")
        self.jac = np.zeros((m, n), order="F")
        self.jac_ineq = np.zeros((ns, n), order="F")
        self.reg = np.zeros(n + ns)
        self.pr_diag = np.zeros(n + ns)
        self.du_diag = np.zeros(m)
        nlb, nub = len(ind_lb), len(ind_ub)
        self.l_diag = np.ones(nlb)
        self.u_diag = np.ones(nub)
        self.l_lower = np.zeros(nlb)
        self.u_lower = np.zeros(nub)
        self.pd_buffer = np.zeros(n + n_eq)
        self.diag_buffer = np.zeros(ns)
        self.buffer = np.zeros(m)
        self.ind_ineq = np.asarray(ind_ineq, dtype=np.int64)
        self.ind_eq = np.asarray(ind_eq, dtype=np.int64)
        self.ind_lb = np.asarray(ind_lb, dtype=np.int64)
        self.ind_ub = np.asarray(ind_ub, dtype=np.int64)
        self.ind_eq_shifted = self.ind_eq + n + ns
        self.ind_ineq_shifted = self.ind_ineq + n + ns
        N = n + n_eq
        self.aug_com = np.zeros((N, N), order="F")
        self.linear_solver = linear_solver_factory(self.aug_com)

    def num_variables(self):
        return self.hess.shape[0]

    def size(self):
        return self.aug_com.shape

    def get_jacobian(self):
        return self.jac

    def compress_hessian(self):
        """no-op for DenseCondensed (`src/KKT/KKTsystem.jl:256`)."""
        return

    def build_kkt(self):
        """`build_kkt!` `condensed.jl:157-186`."""
        n, ns, n_eq = self.n, self.n_ineq, self.n_eq
        self.aug_com[...] = 0.0
        Ss = self.pr_diag[n:n + ns]
        Sd = self.du_diag[self.ind_ineq]
        self.diag_buffer[:] = Ss / (1.0 - Sd * Ss)
        # _build_ineq_jac! `:146-155`
        self.jac_ineq[...] = self.jac[self.ind_ineq, :] * np.sqrt(self.diag_buffer)[:, None]
        # mul!(W, jac_ineq', jac_ineq) `:178`
        self.aug_com[:n, :n] = self.jac_ineq.T @ self.jac_ineq
        # _build_condensed_kkt_system! `:120-144`
        d = self.aug_com
        il = np.tril_indices(n, -1)
        d[il] += self.hess[il]
        d[il[1], il[0]] += self.hess[il[1], il[0]]
        ii = np.arange(n)
        d[ii, ii] += self.pr_diag[:n] + np.diagonal(self.hess)
        if n_eq > 0:
            Je = self.jac[self.ind_eq, :]
            d[n:, :n] = Je
            d[:n, n:] = Je.T
            ei = np.arange(n_eq) + n
            d[ei, ei] = self.du_diag[self.ind_eq]

    def is_inertia_correct(self, num_pos, num_zero, num_neg):
        """`condensed.jl:189-191`."""
        return num_zero == 0 and num_neg == self.n_eq

    def should_regularize_dual(self, num_pos, num_zero, num_neg):
        return num_zero != 0

    def solve_kkt(self, w):
        """reference `src/IPM/factorization.jl:190-229`."""
        n, n_eq, ns = self.n, self.n_eq, self.n_ineq
        full = w.values
        wx = full[:n]
        ws = full[n:n + ns]
        x = self.pd_buffer
        xx, xy = x[:n], x[n:n + n_eq]
        Ss = self.pr_diag[n:n + ns]
        K.reduce_rhs(self, w)
        wz = full[self.ind_ineq_shifted]  # copy (fancy index); written back below
        wy = full[self.ind_eq_shifted]
        self.buffer[:] = 0.0
        self.buffer[self.ind_ineq] = self.diag_buffer * (wz + ws / Ss)
        xx[:] = self.jac.T @ self.buffer
        xx += wx
        xy[:] = wy
        self.linear_solver.solve_linear_system(x)
        wx[:] = xx
        dual = w.dual()
        dual[:] = self.jac @ wx
        full[self.ind_eq_shifted] = xy
        full[self.ind_ineq_shifted] *= self.diag_buffer
        dual -= self.buffer
        ws[:] = (ws + full[self.ind_ineq_shifted]) / Ss
        K.finish_aug_solve(self, w)
        return w
