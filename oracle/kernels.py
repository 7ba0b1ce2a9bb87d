"""Oracle restatement of the KKT-facing pieces of reference `src/IPM/kernels.jl`,
`src/KKT/KKTsystem.jl` and `src/KKT/rhs.jl` (TEST INFRASTRUCTURE ONLY)."""
from __future__ import annotations

import numpy as np


class UnreducedKKTVector:
    """reference `src/KKT/rhs.jl:90-150`: layout [x, s ; y ; zl ; zu].

    n = number of primal entries (x and slacks), m = number of constraints."""

    def __init__(self, n, m, nlb, nub, ind_lb, ind_ub):
        self.n, self.m, self.nlb, self.nub = n, m, nlb, nub
        self.values = np.zeros(n + m + nlb + nub)
        self.ind_lb = np.asarray(ind_lb, dtype=np.int64)
        self.ind_ub = np.asarray(ind_ub, dtype=np.int64)

    @classmethod
    def from_kkt(cls, kkt):
        return cls(len(kkt.pr_diag), len(kkt.du_diag), len(kkt.l_diag), len(kkt.u_diag),
                   kkt.ind_lb, kkt.ind_ub)

    def copy(self):
        c = UnreducedKKTVector(self.n, self.m, self.nlb, self.nub, self.ind_lb, self.ind_ub)
        c.values[:] = self.values
        return c

    # views (numpy basic slices share memory, like the reference's unsafe wraps)
    def full(self):
        return self.values

    def primal(self):
        return self.values[:self.n]

    def dual(self):
        return self.values[self.n:self.n + self.m]

    def primal_dual(self):
        return self.values[:self.n + self.m]

    def dual_lb(self):
        return self.values[self.n + self.m:self.n + self.m + self.nlb]

    def dual_ub(self):
        return self.values[self.n + self.m + self.nlb:]


def initialize(kkt):
    """reference `initialize!` `src/KKT/KKTsystem.jl:210-216`."""
    kkt.reg[:] = 1.0
    kkt.pr_diag[:] = 1.0
    kkt.du_diag[:] = 0.0
    kkt.hess[...] = 0.0


def regularize_diagonal(kkt, primal, dual):
    """reference `regularize_diagonal!` `src/KKT/KKTsystem.jl:222-226`."""
    kkt.reg += primal
    kkt.pr_diag += primal
    kkt.du_diag -= dual


def set_aug_diagonal(kkt):
    """reference `_set_aug_diagonal!` `src/IPM/kernels.jl:22-27`."""
    kkt.pr_diag[:] = kkt.reg
    kkt.pr_diag[kkt.ind_lb] -= kkt.l_lower / kkt.l_diag
    kkt.pr_diag[kkt.ind_ub] -= kkt.u_lower / kkt.u_diag


def reduce_rhs(kkt, d):
    """reference `reduce_rhs!` `src/IPM/kernels.jl:182-195`."""
    v = d.values
    v[d.ind_lb] -= d.dual_lb() / kkt.l_diag
    v[d.ind_ub] -= d.dual_ub() / kkt.u_diag


def finish_aug_solve(kkt, d):
    """reference `finish_aug_solve!` `src/IPM/kernels.jl:198-204`."""
    v = d.values
    dlb, dub = d.dual_lb(), d.dual_ub()
    dlb[:] = (-dlb + kkt.l_lower * v[d.ind_lb]) / kkt.l_diag
    dub[:] = (dub - kkt.u_lower * v[d.ind_ub]) / kkt.u_diag


def kktmul(w, x, reg, du_diag, l_lower, u_lower, l_diag, u_diag, alpha, beta):
    """reference `_kktmul!` `src/IPM/kernels.jl:161-180`."""
    wv, xv = w.values, x.values
    w.primal()[:] += alpha * reg * x.primal()
    w.dual()[:] += alpha * du_diag * x.dual()
    wv[w.ind_lb] -= alpha * x.dual_lb()
    wv[w.ind_ub] += alpha * x.dual_ub()
    w.dual_lb()[:] = beta * w.dual_lb() + alpha * (xv[x.ind_lb] * l_lower - x.dual_lb() * l_diag)
    w.dual_ub()[:] = beta * w.dual_ub() + alpha * (xv[x.ind_ub] * u_lower + x.dual_ub() * u_diag)
