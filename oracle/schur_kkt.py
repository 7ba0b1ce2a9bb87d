"""Oracle restatement of the reference's `SchurComplementKKTSystem` (`src/KKT/Schur/schur.jl`) -- TEST INFRASTRUCTURE ONLY
(only tests/, smoke() and bench.py's cpu_baseline leg may import this package; the product path never does).

Follows the reference's own formulation: every COO entry of the Hessian / Jacobian is classified once (scenario diagonal
block, coupling block, design block -- `_build_schur_symbolic`, schur.jl:140-600) and `build_kkt!` (:927-1001) is a sequence of
scatter-adds of single entries (`_scatter_add!`, `_scatter_quad_add!`); the per-scenario blocks are factorized by LAPACK's
Bunch-Kaufman (the reference's default scenario solver is MUMPS on a sparse block: any symmetric-indefinite solver), the
design block S by the oracle's `LapackCPUSolver`.  Dense blocks, plain loops: sizes of the reference's own tests.

Pinned by the reference's known answers (`test/schur_test.jl`): the analytic optimum of the coupled quadratic (:10-41), the
inactive-constraint solution (:110-139), agreement with the monolithic KKT system on the same QP (:43-80) -- tests/test_schur_kkt.py."""
from __future__ import annotations

import numpy as np

from . import kernels as K
from .lapack_cpu import BUNCHKAUFMAN, LapackCPUSolver


class SchurComplementKKTSystem:
    def __init__(self, n, m, jac_I, jac_J, hess_I, hess_J, ind_ineq, ind_eq, ind_lb, ind_ub, ns, nv, nd, nc,
                 linear_solver_factory=None):
        assert n == ns * nv + nd and m == ns * nc
        self.n, self.m, self.ns, self.nv, self.nd, self.nc = n, m, ns, nv, nd, nc
        self.jac_I, self.jac_J = np.asarray(jac_I, dtype=np.int64), np.asarray(jac_J, dtype=np.int64)
        hI, hJ = np.asarray(hess_I, dtype=np.int64), np.asarray(hess_J, dtype=np.int64)
        self.hess_I, self.hess_J = np.maximum(hI, hJ), np.minimum(hI, hJ)
        self.ind_ineq, self.ind_eq = np.asarray(ind_ineq, dtype=np.int64), np.asarray(ind_eq, dtype=np.int64)
        self.ind_lb, self.ind_ub = np.asarray(ind_lb, dtype=np.int64), np.asarray(ind_ub, dtype=np.int64)
        self.n_ineq, self.n_eq = len(self.ind_ineq), len(self.ind_eq)
        self.hess, self.jac = np.zeros(len(hI)), np.zeros(len(self.jac_I))
        nt = n + self.n_ineq
        self.reg, self.pr_diag, self.du_diag = np.zeros(nt), np.zeros(nt), np.zeros(m)
        self.l_diag, self.u_diag = np.ones(len(ind_lb)), np.ones(len(ind_ub))
        self.l_lower, self.u_lower = np.zeros(len(ind_lb)), np.zeros(len(ind_ub))
        # per-scenario equality rows in order of appearance (eq_global_indices, schur.jl:138-140) and local indices
        self.eq_rows = [[int(r) for r in self.ind_eq if r // nc == k] for k in range(ns)]
        self.nc_eq = len(self.eq_rows[0]) if ns else 0
        assert all(len(r) == self.nc_eq for r in self.eq_rows), "non-uniform equality counts"
        self.eq_local = {}
        for k in range(ns):
            for li, r in enumerate(self.eq_rows[k]):
                self.eq_local[r] = li
        self.ineq_bufidx = {int(r): i for i, r in enumerate(self.ind_ineq)}
        self.blk = nv + self.nc_eq
        self.A_kk = [np.zeros((self.blk, self.blk), order="F") for _ in range(ns)]
        self.C_dk = [np.zeros((nd, self.blk), order="F") for _ in range(ns)]
        self.tmp_blk_nd = [np.zeros((self.blk, nd), order="F") for _ in range(ns)]
        self.aug_com = np.zeros((nd, nd), order="F")
        self.diag_buffer = np.zeros(self.n_ineq)
        self.buffer = np.zeros(m)
        self.scenario_solvers = [None] * ns
        factory = linear_solver_factory or (lambda A: LapackCPUSolver(A, BUNCHKAUFMAN))
        self.linear_solver = factory(self.aug_com)
        # Jacobian entries of every constraint row (for the quadratic condensation terms)
        self.row_entries = {}
        for e, (r, c) in enumerate(zip(self.jac_I, self.jac_J)):
            self.row_entries.setdefault(int(r), []).append((e, int(c)))

    # ---- generic pieces
    def num_variables(self): return self.n
    def get_hessian(self): return self.hess
    def get_jacobian(self): return self.jac
    def get_kkt(self): return self.aug_com
    def size(self): return (self.nd, self.nd)
    def initialize(self): K.initialize(self)
    def regularize_diagonal(self, primal, dual): K.regularize_diagonal(self, primal, dual)
    def compress_jacobian(self): return
    def compress_hessian(self): return
    def is_inertia_correct(self, num_pos, num_zero, num_neg): return num_zero == 0 and num_pos == self.nd   # :901-903
    def should_regularize_dual(self, num_pos, num_zero, num_neg): return True                                 # :905

    def _scen(self, var):
        return var // self.nv if var < self.ns * self.nv else -1

    def jtprod(self, y, x):
        """schur.jl:907-915."""
        y[:self.n + self.n_ineq] = 0.0
        for e, (r, c) in enumerate(zip(self.jac_I, self.jac_J)):
            y[c] += self.jac[e] * x[r]
        y[self.n:self.n + self.n_ineq] = -x[self.ind_ineq]
        return y

    # ---- build_kkt! (schur.jl:927-1001)
    def build_kkt(self):
        n, ns, nv, nd, off = self.n, self.ns, self.nv, self.nd, self.ns * self.nv
        if self.n_ineq > 0:
            Ss, Sd = self.pr_diag[n:n + self.n_ineq], self.du_diag[self.ind_ineq]
            self.diag_buffer[:] = Ss / (1.0 - Sd * Ss)
        S = self.aug_com
        S[...] = 0.0
        for A in self.A_kk:
            A[...] = 0.0
        for C in self.C_dk:
            C[...] = 0.0
        # Hessian entries (lower triangle): scenario block, coupling block (row = design var), design block
        for e, (i, j) in enumerate(zip(self.hess_I, self.hess_J)):
            si, sj, v = self._scen(i), self._scen(j), self.hess[e]
            if si >= 0 and sj >= 0:
                assert si == sj, "Hessian entry couples two scenarios"
                li, lj = i - si * nv, j - sj * nv
                self.A_kk[si][li, lj] += v
                if li != lj:
                    self.A_kk[si][lj, li] += v
            elif si < 0 and sj < 0:
                S[i - off, j - off] += v
                if i != j:
                    S[j - off, i - off] += v
            else:                       # (design, scenario) -- lower triangle: the design variable is the row
                d, sv, k = (i, j, sj) if si < 0 else (j, i, si)
                self.C_dk[k][d - off, sv - k * nv] += v
        for k in range(ns):
            for j in range(nv):
                self.A_kk[k][j, j] += self.pr_diag[k * nv + j]
            for li, r in enumerate(self.eq_rows[k]):
                self.A_kk[k][nv + li, nv + li] += self.du_diag[r]
        for j in range(nd):
            S[j, j] += self.pr_diag[off + j]
        # Jacobian entries of equality rows
        for e, (r, c) in enumerate(zip(self.jac_I, self.jac_J)):
            r, c = int(r), int(c)
            if r not in self.eq_local:
                continue
            k, li, v = r // self.nc, self.eq_local[r], self.jac[e]
            if c < off:
                assert c // nv == k, "constraint reaches another scenario"
                self.A_kk[k][nv + li, c - k * nv] += v
                self.A_kk[k][c - k * nv, nv + li] += v
            else:
                self.C_dk[k][c - off, nv + li] += v
        # inequality rows: J' D J spread over A_kk, C_dk and S, one pair of entries at a time (_scatter_quad_add!)
        for r, b in self.ineq_bufidx.items():
            k, D = r // self.nc, self.diag_buffer[b]
            ents = self.row_entries.get(r, [])
            for (e1, c1) in ents:
                for (e2, c2) in ents:
                    v = self.jac[e1] * D * self.jac[e2]
                    if c1 < off and c2 < off:
                        self.A_kk[k][c1 - k * nv, c2 - k * nv] += v
                    elif c1 >= off and c2 < off:
                        self.C_dk[k][c1 - off, c2 - k * nv] += v
                    elif c1 >= off and c2 >= off:
                        S[c1 - off, c2 - off] += v
        # phase 1: factorize the blocks, T_k = A_k^-1 C_dk'; phase 2: S -= C_dk T_k
        for k in range(ns):
            self.scenario_solvers[k] = LapackCPUSolver(self.A_kk[k], BUNCHKAUFMAN).factorize()
            for j in range(nd):
                col = self.C_dk[k][j, :].copy()
                self.scenario_solvers[k].solve_linear_system(col)
                self.tmp_blk_nd[k][:, j] = col
        for k in range(ns):
            S -= self.C_dk[k] @ self.tmp_blk_nd[k]

    def factorize_kkt(self):
        return self.linear_solver.factorize()

    # ---- solve_kkt! (schur.jl:1040-1110)
    def solve_kkt(self, w):
        n, ns, nv, nd, ni, off = self.n, self.ns, self.nv, self.nd, self.n_ineq, self.ns * self.nv
        full = w.values
        wx, ws, wy = full[:n], full[n:n + ni], w.dual()
        Ss = self.pr_diag[n:n + ni]
        K.reduce_rhs(self, w)
        self.buffer[:] = 0.0
        if ni > 0:
            self.buffer[self.ind_ineq] = self.diag_buffer * (wy[self.ind_ineq] + ws / Ss)
            for e, (r, c) in enumerate(zip(self.jac_I, self.jac_J)):
                wx[c] += self.jac[e] * self.buffer[r]
        rhs_k = []
        for k in range(ns):
            rhs = np.concatenate((wx[k * nv:(k + 1) * nv], wy[self.eq_rows[k]])) if self.nc_eq else wx[k * nv:(k + 1) * nv].copy()
            rhs_k.append(rhs)
        rhs_d = wx[off:].copy()
        for k in range(ns):
            self.scenario_solvers[k].solve_linear_system(rhs_k[k])
        for k in range(ns):
            rhs_d -= self.C_dk[k] @ rhs_k[k]
        self.linear_solver.solve_linear_system(rhs_d)
        for k in range(ns):
            rhs_k[k] -= self.tmp_blk_nd[k] @ rhs_d
        for k in range(ns):
            wx[k * nv:(k + 1) * nv] = rhs_k[k][:nv]
            if self.nc_eq:
                wy[self.eq_rows[k]] = rhs_k[k][nv:]
        wx[off:] = rhs_d
        if ni > 0:
            wy_eq = wy[self.ind_eq].copy()
            Jx = np.zeros(self.m)
            for e, (r, c) in enumerate(zip(self.jac_I, self.jac_J)):
                Jx[r] += self.jac[e] * wx[c]
            wy[:] = Jx
            wy[self.ind_eq] = wy_eq
            wy[self.ind_ineq] = self.diag_buffer * wy[self.ind_ineq] - self.buffer[self.ind_ineq]
            ws[:] = (ws + wy[self.ind_ineq]) / Ss
        K.finish_aug_solve(self, w)
        return w

    # ---- mul! / mul_hess_blk! (schur.jl:1113-1146)
    def _hmul(self, x):
        y = np.zeros(self.n)
        for e, (i, j) in enumerate(zip(self.hess_I, self.hess_J)):
            y[i] += self.hess[e] * x[j]
            if i != j:
                y[j] += self.hess[e] * x[i]
        return y

    def mul(self, w, x, alpha=1.0, beta=0.0):
        n = self.n
        wp, xp = w.primal(), x.primal()
        wx, ws, xx, xs = wp[:n], wp[n:], xp[:n], xp[n:]
        wy, xy = w.dual(), x.dual()
        wx[:] = beta * wx + alpha * self._hmul(xx)
        wy[:] = beta * wy
        for e, (r, c) in enumerate(zip(self.jac_I, self.jac_J)):
            wx[c] += alpha * self.jac[e] * xy[r]
            wy[r] += alpha * self.jac[e] * xx[c]
        ws[:] = beta * ws - alpha * xy[self.ind_ineq]
        wy[self.ind_ineq] -= alpha * xs
        K.kktmul(w, x, self.reg, self.du_diag, self.l_lower, self.u_lower, self.l_diag, self.u_diag, alpha, beta)
        return w

    def mul_hess_blk(self, wx, t):
        wx[:self.n] = self._hmul(t[:self.n])
        wx[self.n:] = 0.0
        wx += t * self.pr_diag
        return wx
