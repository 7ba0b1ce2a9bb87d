"""Oracle restatement (TEST INFRASTRUCTURE ONLY) of the dense `S` stage of the reference's
`SchurComplementKKTSystem` (`src/KKT/Schur/schur.jl`): `build_kkt!` phases 1-2 (:955-999), `factorize_kkt!`
(:1003-1005) and steps 3-5 of `solve_kkt!` (:1040-1058), on dense scenario blocks.

The reference factors each scenario block `A_kk` with a sparse symmetric-indefinite solver (MUMPS); the arithmetic
restated here is the same block algebra with LAPACK `dsytrf/dsytrs` (Bunch-Kaufman) per block, so that it is the CPU
twin of what libmadnlp_hip does with dense blocks."""
from __future__ import annotations

import numpy as np

from .lapack_cpu import BUNCHKAUFMAN, LapackCPUSolver


class SchurDenseStage:
    """`A[k]`: (blk, blk) symmetric (lower read); `C[k]`: (nd, blk); `S0`: (nd, nd)."""

    def __init__(self, A, C, S0, algorithm=BUNCHKAUFMAN):
        self.ns = len(A)
        self.blk = A[0].shape[0] if self.ns else 0
        self.nd = S0.shape[0]
        self.A = [np.asfortranarray(a) for a in A]
        self.C_dk = [np.asfortranarray(c) for c in C]
        self.S0 = np.asfortranarray(S0)
        self.algorithm = algorithm
        self.scenario_solvers = [LapackCPUSolver(a, algorithm) for a in self.A]
        self.tmp_blk_nd = [np.zeros((self.blk, self.nd), order="F") for _ in range(self.ns)]
        self.aug_com = np.zeros((self.nd, self.nd), order="F")
        self.linear_solver = LapackCPUSolver(self.aug_com, algorithm)

    def build_local(self, with_s0=True):
        """This rank's contribution to S (reference :941-999; S0 only on the rank that owns it)."""
        S = self.S0.copy(order="F") if with_s0 else np.zeros((self.nd, self.nd), order="F")
        for k in range(self.ns):
            self.scenario_solvers[k].factorize()                       # :972
            for j in range(self.nd):                                   # :975-985: column by column
                buf = self.C_dk[k][j, :].copy()
                self.scenario_solvers[k].solve_linear_system(buf)
                self.tmp_blk_nd[k][:, j] = buf
        for k in range(self.ns):                                       # :993-999
            S -= self.C_dk[k] @ self.tmp_blk_nd[k]
        return S

    def factorize(self, S):
        self.aug_com[:] = S
        self.linear_solver.factorize()                                 # :1003-1005
        return self.linear_solver.inertia()

    def forward(self, rhs_k):
        """Step 3 (:1040-1049): rhs_k (ns, blk) solved in place; returns -sum_k C_dk rhs_k."""
        contrib = np.zeros(self.nd)
        for k in range(self.ns):
            self.scenario_solvers[k].solve_linear_system(rhs_k[k])
        for k in range(self.ns):
            contrib -= self.C_dk[k] @ rhs_k[k]
        return contrib

    def solve_s(self, rhs_d):
        return self.linear_solver.solve_linear_system(rhs_d)          # :1052

    def backward(self, rhs_k, x_d):
        for k in range(self.ns):                                       # :1055-1058
            rhs_k[k] -= self.tmp_blk_nd[k] @ x_d
        return rhs_k
