"""`SchurComplementKKTSystem` on the MI355X: the reference's KKT system for two-stage stochastic programs
(`src/KKT/Schur/schur.jl:72-1146`) around the HIP `S` stage of `schur.py` / `csrc/schur.hip`.

Variable layout `[v_1 .. v_ns (nv each), d (nd)]`, constraint layout `[c_1 .. c_ns (nc each)]`.  Per iteration
(`build_kkt!`, reference :927-1001) every scenario contributes a block

    A_k  = [ H_kk + Sigma_k + J_I,k' D_I J_I,k    J_E,k' ]        C_dk = [ H_dk + J_I,d' D_I J_I,k    J_E,d' ]
           [ J_E,k                                 du_diag_E ]               (nd x blk)

of order blk = nv + (equality rows per scenario) -- the scenario's INEQUALITY rows are condensed (D_I = Sigma_s /
(1 - Sigma_d Sigma_s), the `diag_buffer` of the condensed systems), its equality rows stay -- and the design block starts as
`S = H_dd + Sigma_d + sum_k J_I,d' D_I J_I,d`.  The device then factors the ns blocks as ONE batch, forms
`S -= sum_k C_dk A_k^-1 C_dk'` on the matrix cores and factors S (`mnk_schur_build_local`, `mnk_schur_factorize_s`);
`solve_kkt!` (:1040-1110) runs the forward / design / backward steps of the stage between the host-side condensation and
recovery of the inequality rows.  Inertia is judged on S alone, as the reference does (:901-903).

What is host-side here: the vector algebra around the three device steps.  The scatter of the COO values into the dense
blocks (the reference: precomputed index maps on the CPU, nine @atomic kernels in its CUDA extension) runs on the device since
round 6 (`mnk_schur_set_structure` once, `mnk_schur_assemble` per iteration).  The product path has no CPU fallback for the factorizations and solves: they are the HIP stage's."""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from .kkt import _KKTCommon
from .linear_solver import HipContext, _LIVE_OBJECTS
from .schur import SchurDenseStage


def build_schur_symbolic(n, m, ns, nv, nd, nc, hess_I, hess_J, jac_I, jac_J, ind_eq, ind_ineq):
    """The layout checks of the reference's `_build_schur_symbolic` (schur.jl:140-600; error cases of test/schur_test.jl:176-236)
    and the per-scenario row sets.  0-based indices.  Raises ValueError where the reference throws an ErrorException:
    a Hessian entry that couples two scenarios, a constraint that reaches another scenario's variables, equality / inequality
    counts that differ between scenarios, a sparsity pattern that differs between scenarios."""
    if n != ns * nv + nd or m != ns * nc:
        raise ValueError(f"SchurComplementKKTSystem: n = {n}, m = {m} do not match ns * nv + nd = {ns * nv + nd}, ns * nc = {ns * nc}")
    hess_I, hess_J, jac_I, jac_J = (np.asarray(a, dtype=np.int64) for a in (hess_I, hess_J, jac_I, jac_J))
    off = ns * nv
    scen_of_var = np.where(np.arange(n) < off, np.arange(n) // max(nv, 1), -1)   # -1: design
    hi, hj = scen_of_var[hess_I], scen_of_var[hess_J]
    if ((hi >= 0) & (hj >= 0) & (hi != hj)).any():
        raise ValueError("SchurComplementKKTSystem: a Hessian entry couples two scenarios")
    con_s = jac_I // max(nc, 1)
    vs = scen_of_var[jac_J]
    if ((vs >= 0) & (vs != con_s)).any():
        raise ValueError("SchurComplementKKTSystem: a constraint reaches the variables of another scenario")
    ind_eq, ind_ineq = np.asarray(ind_eq, dtype=np.int64), np.asarray(ind_ineq, dtype=np.int64)
    eq_rows = [np.sort(ind_eq[ind_eq // max(nc, 1) == k]) for k in range(ns)]
    ineq_pos = [np.nonzero(ind_ineq // max(nc, 1) == k)[0] for k in range(ns)]      # positions in ind_ineq (= slack index)
    if len({len(r) for r in eq_rows}) > 1 or len({len(r) for r in ineq_pos}) > 1:
        raise ValueError("SchurComplementKKTSystem: the scenarios have different numbers of equality / inequality constraints")
    # the same local pattern in every scenario (the reference builds ONE symbolic block and reuses it)
    def local_pattern(k):
        lo_i, lo_j = np.maximum(hess_I, hess_J), np.minimum(hess_I, hess_J)
        sel = (scen_of_var[lo_i] == k) & (scen_of_var[lo_j] == k)
        hp = set(zip((lo_i[sel] - k * nv).tolist(), (lo_j[sel] - k * nv).tolist()))
        selj = (con_s == k) & (vs == k)
        jp = set(zip((jac_I[selj] - k * nc).tolist(), (jac_J[selj] - k * nv).tolist()))
        return hp, jp
    if ns > 0:
        p0 = local_pattern(0)
        for k in range(1, ns):
            if local_pattern(k) != p0:
                raise ValueError("SchurComplementKKTSystem: the sparsity pattern differs between scenarios")
    return dict(eq_rows=eq_rows, ineq_pos=ineq_pos, nc_eq=len(eq_rows[0]) if ns else 0, nc_ineq=len(ineq_pos[0]) if ns else 0)


class _DesignSolver:
    """What the interior-point loop sees as `kkt.linear_solver`: the solver of the design block S (reference :868-870,
    `factorize_kkt!` :1003-1005).  The scenario blocks are factored inside `build_kkt!`."""

    def __init__(self, stage):
        self.stage = stage
        self.info = 0

    def factorize(self):
        self.info = self.stage.factorize_kkt()
        return self

    def inertia(self):
        return self.stage.inertia()

    def is_inertia(self):
        return True

    def improve(self):
        return False

    def introduce(self):
        return "HIP-MI355X Schur stage (scenario blocks: batched LDL', design block: dense LDL')"


class SchurComplementKKTSystem(_KKTCommon):
    """reference `src/KKT/Schur/schur.jl:72-140` (fields), `:927-1001` (`build_kkt!`), `:1040-1110` (`solve_kkt!`),
    `:1113-1146` (`mul!`, `mul_hess_blk!`)."""

    device_assembly = True     # False: the host assembly of rounds 4-5 (diagnostic A/B runs)

    def __init__(self, n, m, jac_I, jac_J, hess_I, hess_J, ind_ineq, ind_eq, ind_lb, ind_ub, ns, nv, nd, nc, ctx=None):
        import torch
        self.torch = torch
        sym = build_schur_symbolic(n, m, ns, nv, nd, nc, hess_I, hess_J, jac_I, jac_J, ind_eq, ind_ineq)
        self.n, self.m, self.ns, self.nv, self.nd, self.nc = n, m, ns, nv, nd, nc
        self.n_ineq, self.n_eq = len(ind_ineq), len(ind_eq)
        self.nc_eq, self.nc_ineq = sym["nc_eq"], sym["nc_ineq"]
        self.blk = nv + self.nc_eq
        self.eq_rows, self.ineq_pos = sym["eq_rows"], sym["ineq_pos"]
        self.ind_ineq = np.asarray(ind_ineq, dtype=np.int64)
        self.ind_eq = np.asarray(ind_eq, dtype=np.int64)
        self.ind_lb = np.asarray(ind_lb, dtype=np.int64)
        self.ind_ub = np.asarray(ind_ub, dtype=np.int64)
        self.jac_I, self.jac_J = np.asarray(jac_I, dtype=np.int64), np.asarray(jac_J, dtype=np.int64)
        hI, hJ = np.asarray(hess_I, dtype=np.int64), np.asarray(hess_J, dtype=np.int64)
        self.hess_I, self.hess_J = np.maximum(hI, hJ), np.minimum(hI, hJ)       # force_lower_triangular!
        self.hess = np.zeros(len(hI))
        self.jac = np.zeros(len(self.jac_I))
        nt = n + self.n_ineq
        self.reg, self.pr_diag, self.du_diag = np.zeros(nt), np.zeros(nt), np.zeros(m)
        nlb, nub = len(ind_lb), len(ind_ub)
        self.l_diag, self.u_diag = np.ones(nlb), np.ones(nub)
        self.l_lower, self.u_lower = np.zeros(nlb), np.zeros(nub)
        self.diag_buffer = np.zeros(self.n_ineq)
        self.buffer = np.zeros(m)
        self.J = sp.csr_matrix((m, n))
        self.H = sp.csr_matrix((n, n))
        self.ctx = ctx or HipContext()
        zA = [np.eye(self.blk) for _ in range(ns)]
        zC = [np.zeros((nd, self.blk)) for _ in range(ns)]
        self.stage = SchurDenseStage(zA, zC, np.eye(nd), nd, self.blk, ctx=self.ctx)
        # the index maps of `_build_schur_symbolic` (reference :460-700) live in the library: one source list per touched entry of
        # A_k / C_dk / S0, built from the COO patterns once
        self.stage.set_structure(n, m, nv, nc, self.hess_I, self.hess_J, self.jac_I, self.jac_J, self.ind_ineq, self.ind_eq)
        self.aug_com = self.stage.S
        self.linear_solver = _DesignSolver(self.stage)
        self._order = nd
        _LIVE_OBJECTS.add(self)

    # ---- interface pieces
    def num_variables(self):
        return self.n

    def is_inertia_correct(self, num_pos, num_zero, num_neg):
        """reference :901-903."""
        return num_zero == 0 and num_pos == self.nd

    def should_regularize_dual(self, num_pos, num_zero, num_neg):
        """reference :905."""
        return True

    def compress_jacobian(self):
        """reference :917-919 (COO -> CSC transfer): here the CSR matrix the products use."""
        self.J = sp.csr_matrix((self.jac, (self.jac_I, self.jac_J)), shape=(self.m, self.n))

    def compress_hessian(self):
        """reference :921-923."""
        Lo = sp.csr_matrix((self.hess, (self.hess_I, self.hess_J)), shape=(self.n, self.n))
        self.H = Lo + sp.tril(Lo, -1).T

    def jtprod(self, y, x):
        """reference :907-915."""
        y[:self.n] = self.J.T @ x
        y[self.n:self.n + self.n_ineq] = -x[self.ind_ineq]
        return y

    # ---- build_kkt! (reference :927-1001)
    def assemble_blocks(self):
        """CHECKER ONLY since round 6 (tests; `build_kkt` assembles on the device): the scenario blocks A_k, the coupling blocks
        C_dk and the design block before the Schur products (S0) as dense host arrays by matrix slicing -- what the reference
        scatters into `A_kk.nzval`, `C_dk` and `aug_com` (:935-972, :993-996)."""
        n, ns, nv, nd = self.n, self.ns, self.nv, self.nd
        off = ns * nv
        if self.n_ineq > 0:
            Ss = self.pr_diag[n:n + self.n_ineq]
            Sd = self.du_diag[self.ind_ineq]
            self.diag_buffer[:] = Ss / (1.0 - Sd * Ss)
        J, H = self.J.tocsr(), self.H.tocsr()
        Jd_all = J[:, off:].toarray()                     # m x nd
        S0 = H[off:, off:].toarray() + np.diag(self.pr_diag[off:n])
        A, Cd = [], []
        for k in range(ns):
            v = slice(k * nv, (k + 1) * nv)
            E, Ipos = self.eq_rows[k], self.ineq_pos[k]
            Irows = self.ind_ineq[Ipos]
            Hvv = H[v, v].toarray() + np.diag(self.pr_diag[v])
            Hdv = H[off:, v].toarray()
            Jv = J[:, v]
            JvE, JvI = Jv[E].toarray(), Jv[Irows].toarray()
            JdE, JdI = Jd_all[E], Jd_all[Irows]
            D = self.diag_buffer[Ipos]
            Ak = np.zeros((self.blk, self.blk))
            Ak[:nv, :nv] = Hvv + JvI.T @ (D[:, None] * JvI)
            Ak[nv:, :nv] = JvE
            Ak[:nv, nv:] = JvE.T
            Ak[nv:, nv:] = np.diag(self.du_diag[E])
            Ck = np.zeros((nd, self.blk))
            Ck[:, :nv] = Hdv + JdI.T @ (D[:, None] * JvI)
            Ck[:, nv:] = JdE.T
            S0 += JdI.T @ (D[:, None] * JdI)
            A.append(Ak)
            Cd.append(Ck)
        return A, Cd, S0

    def build_kkt(self):
        """reference :927-1001.  The scatter of the callback values into A_k / C_dk / S0 runs on the device (`mnk_schur_assemble`:
        four uploads -- hess, jac, pr_diag, du_diag -- and one launch instead of ns (blk^2 + nd blk) doubles assembled with numpy
        and uploaded every iteration); the host keeps its copy of `diag_buffer` for the vector algebra of `solve_kkt!`."""
        if not self.device_assembly:     # (rounds 4-5, kept for the before / after record of tools/bench_schur_kkt.py only)
            A, Cd, S0 = self.assemble_blocks()
            self.stage.set_blocks(A, Cd, S0)
            self.stage.build_kkt()
            return
        if self.n_ineq > 0:
            Ss = self.pr_diag[self.n:self.n + self.n_ineq]
            Sd = self.du_diag[self.ind_ineq]
            self.diag_buffer[:] = Ss / (1.0 - Sd * Ss)
        self.stage.assemble(self.hess, self.jac, self.pr_diag, self.du_diag)
        self.stage.build_kkt()          # device: the ns blocks as one batch, S -= sum_k C_dk A_k^-1 C_dk'

    def factorize_kkt(self):
        return self.linear_solver.factorize()

    # ---- solve_kkt! (reference :1040-1110)
    def solve_kkt(self, w):
        n, ns, nv, nd, ni = self.n, self.ns, self.nv, self.nd, self.n_ineq
        off = ns * nv
        full = w.values
        wx, ws, wy = full[:n], full[n:n + ni], w.dual()
        Ss = self.pr_diag[n:n + ni]
        self._reduce_rhs(w)
        self.buffer[:] = 0.0
        if ni > 0:                                             # step 1: condense the inequality rows
            self.buffer[self.ind_ineq] = self.diag_buffer * (wy[self.ind_ineq] + ws / Ss)
            wx += self.J.T @ self.buffer
        rk = np.empty((ns, self.blk))                          # step 2: per-scenario right-hand sides
        for k in range(ns):
            rk[k, :nv] = wx[k * nv:(k + 1) * nv]
            rk[k, nv:] = wy[self.eq_rows[k]]
        rd = np.ascontiguousarray(wx[off:])
        self.stage.solve_host(rk, rd)                          # steps 3-5 on the device (mnk_schur_solve)
        for k in range(ns):                                    # step 6: write back
            wx[k * nv:(k + 1) * nv] = rk[k, :nv]
            wy[self.eq_rows[k]] = rk[k, nv:]
        wx[off:] = rd
        if ni > 0:                                             # step 7: inequality duals and slacks
            wy_eq = wy[self.ind_eq].copy()
            wy[:] = self.J @ wx
            wy[self.ind_eq] = wy_eq
            wy[self.ind_ineq] = self.diag_buffer * wy[self.ind_ineq] - self.buffer[self.ind_ineq]
            ws[:] = (ws + wy[self.ind_ineq]) / Ss
        self._finish_aug_solve(w)
        return w

    # ---- mul! / mul_hess_blk! (reference :1113-1146)
    def mul(self, w, x, alpha=1.0, beta=0.0):
        n = self.n
        wp, xp = w.primal(), x.primal()
        wx, ws = wp[:n], wp[n:]
        xx, xs = xp[:n], xp[n:]
        wy, xy = w.dual(), x.dual()
        wx[:] = beta * wx + alpha * (self.H @ xx)
        if self.m > 0:
            wx += alpha * (self.J.T @ xy)
            wy[:] = alpha * (self.J @ xx) + beta * wy
        else:
            wy[:] = beta * wy
        ws[:] = beta * ws - alpha * xy[self.ind_ineq]
        wy[self.ind_ineq] -= alpha * xs
        self._kktmul(w, x, alpha, beta)
        return w

    def mul_hess_blk(self, wx, t):
        n = self.n
        wx[:n] = self.H @ t[:n]
        wx[n:] = 0.0
        wx += t * self.pr_diag
        return wx

    def close(self):
        if getattr(self, "stage", None) is not None:
            self.stage.close()
            self.stage = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
