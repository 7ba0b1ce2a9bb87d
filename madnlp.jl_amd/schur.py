"""Host-side mirror of the dense `S` stage of MadNLP's `SchurComplementKKTSystem` on the MI355X
(reference `src/KKT/Schur/schur.jl:927-1058`; C ABI `mnk_schur_*`, `csrc/schur.hip`).

One object per rank holds that rank's scenario blocks (scenario k of the global problem lives on rank
`k % world`, the same instance-index partition as BASELINE config C5).  The only communication is one all-reduce
of the nd x nd Schur complement per `build_kkt!` and one of the nd-vector per solve, both through
`torch.distributed` (backend nccl = RCCL over xGMI on the GPU node, gloo in the CPU tests)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .linear_solver import BUNCHKAUFMAN, _ALGO, _LIVE_OBJECTS, FactorizationException, HipContext, SolveException


def shard(ns: int, rank: int, world: int):
    """Global scenario indices owned by `rank` (round-robin, as the C5 instance partition)."""
    return list(range(rank, ns, world))


class SchurDenseStage:
    """`A`: this rank's (blk, blk) scenario blocks; `C`: its (nd, blk) coupling blocks; `S0`: (nd, nd) design block
    on the rank that owns it (rank 0), else None.  `dist`: an initialized `torch.distributed` module or None."""

    def __init__(self, A, C_dk, S0, nd, blk, ctx: HipContext | None = None, algorithm=BUNCHKAUFMAN, dist=None):
        import torch
        self.torch = torch
        self.ctx = ctx or HipContext()
        self.ns, self.nd, self.blk = len(A), int(nd), int(blk)
        self.dist = dist
        self._h = C.c_void_p()
        L.check(L.lib().mnk_schur_create(self.ctx.handle, self.ns, self.blk, self.nd, _ALGO[algorithm], C.byref(self._h)),
                "mnk_schur_create")
        self.S0 = None
        self.set_blocks(A, C_dk, S0)
        dev = torch.device("cuda", self.ctx.device)
        self.S = torch.zeros(self.nd * self.nd, dtype=torch.float64, device=dev)   # column-major nd x nd
        self._contrib = torch.zeros(self.nd, dtype=torch.float64, device=dev)
        _LIVE_OBJECTS.add(self)

    def set_blocks(self, A, C_dk, S0):
        """New values of the scenario blocks (`build_kkt!` assembles them every iteration: reference :955-972) and of the design
        block; entries of `A` / `C_dk` that are None keep their block."""
        assert len(A) == self.ns and len(C_dk) == self.ns
        for k in range(self.ns):
            if A[k] is None:
                continue
            a = np.asfortranarray(A[k], dtype=np.float64)
            c = np.asfortranarray(C_dk[k], dtype=np.float64)
            assert a.shape == (self.blk, self.blk) and c.shape == (self.nd, self.blk)
            L.check(L.lib().mnk_schur_set_block(self._h, k, a.ctypes.data, self.blk, c.ctypes.data, self.nd, L.MNK_HOST),
                    "mnk_schur_set_block")
        self.S0 = None if S0 is None else np.asfortranarray(S0, dtype=np.float64)
        self._assembled = False

    # ---- device-side assembly (mnk_schur_set_structure / mnk_schur_assemble)
    def set_structure(self, n, m, nv, nc, hess_I, hess_J, jac_I, jac_J, ind_ineq, ind_eq, ns_global=None, local_scen=None, own_design=True):
        """Once: the COO patterns (0-based) from which the library builds, per touched entry of A_k / C_dk / S0, the list of its
        sources in the reference's scatter order (schur.jl:935-972).  Raises ValueError where the reference's
        `_build_schur_symbolic` throws (:140-236)."""
        hI, hJ = (np.ascontiguousarray(a, dtype=np.int32) for a in (hess_I, hess_J))
        jI, jJ = (np.ascontiguousarray(a, dtype=np.int32) for a in (jac_I, jac_J))
        ii, ie = (np.ascontiguousarray(a, dtype=np.int64) for a in (ind_ineq, ind_eq))
        ls = None if local_scen is None else np.ascontiguousarray(local_scen, dtype=np.int64)
        rc = L.lib().mnk_schur_set_structure(self._h, n, m, nv, nc, len(hI), hI.ctypes.data, hJ.ctypes.data, len(jI), jI.ctypes.data,
                                             jJ.ctypes.data, len(ii), ii.ctypes.data, len(ie), ie.ctypes.data, 0,
                                             self.ns if ns_global is None else ns_global, None if ls is None else ls.ctypes.data,
                                             1 if own_design else 0)
        if rc:
            raise ValueError(L.lib().mnk_last_error_string().decode())
        self._assembled = False

    def assemble(self, hess, jac, pr_diag, du_diag):
        """`build_kkt!`'s scatter (reference :935-972) on the device: A_k, C_dk of every local scenario and S0 from the callbacks'
        COO values and the diagonals (host arrays or device tensors)."""
        from .linear_solver import _ptr
        keep = [np.ascontiguousarray(a, dtype=np.float64) if isinstance(a, np.ndarray) else a for a in (hess, jac, pr_diag, du_diag)]
        (ph, l0), (pj, l1), (pp, l2), (pd, l3) = (_ptr(a) for a in keep)
        assert l0 == l1 == l2 == l3
        L.check(L.lib().mnk_schur_assemble(self._h, ph, pj, pp, pd, l0), "mnk_schur_assemble")
        self._assembled = True

    def get_block(self, k):
        """Host copies (blk x blk, nd x blk) of scenario k's assembled blocks -- tests."""
        a = np.zeros((self.blk, self.blk), order="F")
        c = np.zeros((self.nd, self.blk), order="F")
        L.check(L.lib().mnk_schur_get_block(self._h, k, a.ctypes.data, c.ctypes.data, None), "mnk_schur_get_block")
        return a, c

    def get_s0(self):
        """Host copy of the assembled design block S0 (before the Schur products) -- tests."""
        s0 = np.zeros((self.nd, self.nd), order="F")
        L.check(L.lib().mnk_schur_get_block(self._h, 0, None, None, s0.ctypes.data), "mnk_schur_get_block")
        return s0

    def _allreduce(self, t):
        if self.dist is not None:
            self.ctx.synchronize()          # the library works on its own stream
            self.dist.all_reduce(t)         # the ONE collective of this step (RCCL over xGMI)
            self.torch.cuda.synchronize()

    def build_kkt(self):
        """`build_kkt!` (:927-1001): local phases 1-2, then the all-reduce of S; returns S (device, flat column-major)."""
        if getattr(self, "_assembled", False):   # S0 was assembled on the device together with the blocks
            L.check(L.lib().mnk_schur_build_local(self._h, L.lib().mnk_schur_s0_buffer(self._h), self.nd, L.MNK_DEVICE, self.S.data_ptr(),
                                                  self.nd), "mnk_schur_build_local")
        else:
            s0 = None if self.S0 is None else self.S0.ctypes.data
            L.check(L.lib().mnk_schur_build_local(self._h, s0, self.nd, L.MNK_HOST, self.S.data_ptr(), self.nd),
                    "mnk_schur_build_local")
        self._allreduce(self.S)
        self.ctx.synchronize()   # S is handed to the caller (a torch tensor on the caller's stream)
        return self.S

    def factorize_kkt(self):
        info = C.c_int(0)
        rc = L.lib().mnk_schur_factorize_s(self._h, self.S.data_ptr(), self.nd, L.MNK_DEVICE, C.byref(info))
        if rc:
            raise FactorizationException(L.lib().mnk_last_error_string().decode())
        return info.value

    def inertia(self):
        p, z, n = C.c_int64(), C.c_int64(), C.c_int64()
        L.check(L.lib().mnk_schur_inertia_s(self._h, C.byref(p), C.byref(z), C.byref(n)), "mnk_schur_inertia_s")
        return (p.value, z.value, n.value)

    def scenario_inertia(self, k):
        p, z, n = C.c_int64(), C.c_int64(), C.c_int64()
        L.check(L.lib().mnk_schur_scenario_inertia(self._h, k, C.byref(p), C.byref(z), C.byref(n)), "scenario inertia")
        return (p.value, z.value, n.value)

    def is_inertia_correct(self, num_pos, num_zero, num_neg):
        """reference :901-903."""
        return num_zero == 0 and num_pos == self.nd

    def solve(self, rhs_k, rhs_d):
        """Steps 3-5 of `solve_kkt!` (:1040-1058).  rhs_k: (ns_local, blk) device tensor (row = scenario), rhs_d: (nd)
        device tensor, the SAME on every rank; both overwritten with the solution."""
        rk = rhs_k.data_ptr() if self.ns else None
        rc = L.lib().mnk_schur_forward(self._h, rk, self._contrib.data_ptr())
        if rc:
            raise SolveException(L.lib().mnk_last_error_string().decode())
        self._allreduce(self._contrib)
        self.ctx.synchronize()
        rhs_d += self._contrib
        self.torch.cuda.synchronize()
        if L.lib().mnk_schur_solve_s(self._h, rhs_d.data_ptr()) or L.lib().mnk_schur_backward(self._h, rk, rhs_d.data_ptr()):
            raise SolveException(L.lib().mnk_last_error_string().decode())
        self.ctx.synchronize()
        return rhs_k, rhs_d

    def solve_host(self, rhs_k, rhs_d):
        """Steps 3-5 on ONE rank with host vectors (`mnk_schur_solve`): rhs_k (ns, blk) C-contiguous, rhs_d (nd), in place."""
        assert self.dist is None and rhs_k.flags.c_contiguous and rhs_d.flags.c_contiguous
        rc = L.lib().mnk_schur_solve(self._h, rhs_k.ctypes.data if self.ns else None, rhs_d.ctypes.data, L.MNK_HOST)
        if rc:
            raise SolveException(L.lib().mnk_last_error_string().decode())
        return rhs_k, rhs_d

    def close(self):
        if self._h:
            L.lib().mnk_schur_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
