"""Richardson iterative refinement around `solve_kkt!` -- host-side mirror of reference
`src/LinearSolvers/backsolve.jl:27-76` (the caller of `solve_linear_system!`)."""
from __future__ import annotations

import numpy as np


class RichardsonIterator:
    def __init__(self, kkt, tol=1e-8, max_iter=10):
        """defaults of reference `backsolve.jl:1-5,25`."""
        self.kkt = kkt
        self.richardson_tol = tol ** (5 / 4)
        self.richardson_acceptable_tol = tol ** (5 / 8)
        self.richardson_max_iter = max_iter
        self.ir = 0
        self.residual_ratio = 0.0

    def solve_refine(self, x, b, w):
        norm_b = np.linalg.norm(b.values, np.inf)
        residual_ratio = 0.0
        x.values[:] = 0.0
        if norm_b != 0:
            w.values[:] = b.values
            self.ir = 0
            while True:
                self.kkt.solve_kkt(w)
                x.values += w.values
                w.values[:] = b.values
                self.kkt.mul(w, x, -1.0, 1.0)
                norm_w = np.linalg.norm(w.values, np.inf)
                norm_x = np.linalg.norm(x.values, np.inf)
                residual_ratio = norm_w / (min(norm_x, 1e6 * norm_b) + norm_b)
                self.ir += 1
                if self.ir >= self.richardson_max_iter or residual_ratio < self.richardson_tol:
                    break
        self.residual_ratio = residual_ratio
        return residual_ratio < self.richardson_acceptable_tol
