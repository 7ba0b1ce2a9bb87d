"""Host-side mirror of MadNLP's AbstractLinearSolver interface for the HIP solver.

Mirrors reference `src/LinearSolvers/linearsolvers.jl:13-137` (contract, exceptions)
and `src/LinearSolvers/lapack_common.jl` / `lapack.jl` (LapackCPUSolver behaviour):
same method names minus the `!`, same argument meaning, same error behaviour.  The
Julia glue a maintainer would add is `julia/MadNLPHIP.jl`; this
Python mirror exists because the build image has no Julia toolchain and is what the
parity tests drive.
"""
from __future__ import annotations

import atexit
import ctypes as C
import weakref
from dataclasses import dataclass

import numpy as np

from . import _lib as L

BUNCHKAUFMAN, LU, QR, CHOLESKY, LDL, EVD = "BUNCHKAUFMAN", "LU", "QR", "CHOLESKY", "LDL", "EVD"
_ALGO = {BUNCHKAUFMAN: L.MNK_BUNCHKAUFMAN, CHOLESKY: L.MNK_CHOLESKY, LDL: L.MNK_LDL}


class LinearSolverException(Exception):
    """reference `src/LinearSolvers/linearsolvers.jl:133`."""


class SymbolicException(LinearSolverException):
    pass


class FactorizationException(LinearSolverException):
    pass


class SolveException(LinearSolverException):
    pass


class InertiaException(LinearSolverException):
    pass


# Every live handle is closed explicitly before interpreter teardown (device memory must be
# released while the HIP runtime is still alive): solvers and KKT systems first, contexts last.
_LIVE_OBJECTS = weakref.WeakSet()
_LIVE_CONTEXTS = weakref.WeakSet()


@atexit.register
def _close_all():
    for obj in list(_LIVE_OBJECTS):
        try:
            obj.close()
        except Exception:
            pass
    for ctx in list(_LIVE_CONTEXTS):
        try:
            ctx.close()
        except Exception:
            pass


class HipContext:
    """One device + stream; wraps `mnk_ctx_*`.  `stream` may be a raw hipStream_t
    (e.g. `torch.cuda.current_stream().cuda_stream`) so that callers can time the
    library's work with events on their own stream."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self._h = C.c_void_p()
        L.check(L.lib().mnk_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(self._h)),
                "mnk_ctx_create")
        self.device = device
        _LIVE_CONTEXTS.add(self)

    @property
    def handle(self):
        return self._h

    def synchronize(self):
        L.check(L.lib().mnk_ctx_synchronize(self._h), "mnk_ctx_synchronize")

    def stream(self) -> int:
        return L.lib().mnk_ctx_stream(self._h) or 0

    def close(self):
        if self._h:
            L.lib().mnk_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class factorize_batch:
    """`with factorize_batch(): ...` -- the factorize! calls of this thread inside the block transfer their matrices and are
    queued; leaving the block launches them together (`mnk_factorize_batch_begin / _end`): instances of one order share one
    persistent launch, each one's chain-bound ends filled with its neighbours' trailing updates.  Independent instances only
    (scenario batches, BASELINE config C5); every factor is bit-identical to a lone factorize!."""

    def __enter__(self):
        L.check(L.lib().mnk_factorize_batch_begin(), "mnk_factorize_batch_begin")
        return self

    def __exit__(self, et, ev, tb):
        rc = L.lib().mnk_factorize_batch_end()
        if rc and et is None:
            raise FactorizationException(L.lib().mnk_last_error_string().decode())
        return False


def release_idle_streams(device: int = 0):
    """`mnk_release_idle_streams`: give the device's hardware queues back while this process has nothing to factorize (the
    CU-masked streams of the persistent schedules are made again on demand)."""
    L.check(L.lib().mnk_release_idle_streams(int(device)), "mnk_release_idle_streams")


class solve_batch:
    """`with solve_batch(): ...` -- the single-right-hand-side solves of this thread on DEVICE vectors inside the block are queued
    and run together when the block is left, up to four independent systems per launch (`mnk_solve_batch_begin / _end`).
    The vectors hold the solutions after the block (asynchronously: `check_solve` as usual)."""

    def __enter__(self):
        L.check(L.lib().mnk_solve_batch_begin(), "mnk_solve_batch_begin")
        return self

    def __exit__(self, et, ev, tb):
        rc = L.lib().mnk_solve_batch_end()
        if rc and et is None:
            raise SolveException(L.lib().mnk_last_error_string().decode())
        return False


def _ptr(a):
    """Raw pointer + location of a numpy array (host) or a torch tensor (host/device)."""
    if a is None:
        return None, L.MNK_HOST
    if isinstance(a, np.ndarray):
        return C.c_void_p(a.ctypes.data), L.MNK_HOST
    # torch tensor
    return C.c_void_p(a.data_ptr()), (L.MNK_DEVICE if a.is_cuda else L.MNK_HOST)


@dataclass
class HipSolverOptions:
    """Analogue of `LapackOptions` (reference `src/LinearSolvers/lapack.jl:1-3`).
    BUNCHKAUFMAN (the reference default) maps to the static-pivot LDL^T."""
    lapack_algorithm: str = BUNCHKAUFMAN
    pivot_tol: float = 0.0
    outer_block: int = 0          # 0: by size (512; 1024 from 32 768 rows on)
    lookahead: bool = True
    share: int = 1          # 0 off, 1 adaptive, 2 always: panel-stream CUs join the trailing update
    small_tiles: int = 400  # (a)-updates with fewer 128x128 tiles use 64x64 workgroup tiles
    persistent_solve: bool = True  # both triangular sweeps in one launch (False: one launch per 256-column step)
    single_rows: int = 2560  # systems up to this order are factored as one outer panel on the whole chip (0: never)
    panel_algo: int = 5      # 5: task-DAG schedule (persistent pivot chain + persistent bulk kernel); 4: persistent panel
                             # kernel per 256 columns + one trailing update per outer panel; 1: one launch per piece.
                             # 4 and 5 keep waiting workgroups resident: set 1 when several PROCESSES share the GPU


class HipLinearSolver:
    """`HipLinearSolver(A; opt)`: dense inertia-revealing solver on the MI355X.

    `A` is the matrix the solver keeps a *reference* to (as the reference does,
    `docs/src/man/linear_solvers.md:99-105`): a `DeviceCSC` / `DeviceDense` owned by
    one of our KKT systems, a dense numpy array (column-major, lower triangle
    read) or a `(colptr, rowval, nzval)` lower-triangular CSC triple (0-based)."""

    def __init__(self, A, ctx: HipContext | None = None, opt: HipSolverOptions | None = None):
        self.opt = opt or HipSolverOptions()
        if self.opt.lapack_algorithm not in _ALGO:
            raise SymbolicException(
                f"algorithm {self.opt.lapack_algorithm} is not implemented on device (CHOLESKY, LDL/BUNCHKAUFMAN)")
        self.A = A
        self.ctx = ctx or getattr(A, "ctx", None) or HipContext()
        self.n = _order_of(A)
        self._h = C.c_void_p()
        rc = L.lib().mnk_ls_create(self.ctx.handle, self.n, _ALGO[self.opt.lapack_algorithm], C.byref(self._h))
        if rc:
            raise SymbolicException(L.lib().mnk_last_error_string().decode())
        # (options fixed by the environment -- MNK_OPTIONS="key=value,..." -- are kept by the library)
        settings = [("pivot_tol", self.opt.pivot_tol), ("outer_block", self.opt.outer_block),
                    ("lookahead", float(self.opt.lookahead)), ("share", float(self.opt.share)),
                    ("small_tiles", float(self.opt.small_tiles)), ("persistent_solve", float(self.opt.persistent_solve)),
                    ("single_rows", float(self.opt.single_rows)), ("panel_algo", float(self.opt.panel_algo))]
        for key, val in settings:
            L.check(L.lib().mnk_ls_set_option(self._h, key.encode(), float(val)), "mnk_ls_set_option")
        self.info = 0
        _LIVE_OBJECTS.add(self)

    # -- AbstractLinearSolver interface (reference linearsolvers.jl:13-110) ----------
    def introduce(self) -> str:
        return f"HIP-MI355X ({self.opt.lapack_algorithm})"

    def improve(self) -> bool:
        return False

    @staticmethod
    def input_type() -> str:
        return "dense"

    @staticmethod
    def default_options():
        return HipSolverOptions()

    @staticmethod
    def is_supported(dtype) -> bool:
        return np.dtype(dtype) == np.float64

    def is_inertia(self) -> bool:
        return True

    def factorize(self):
        """`factorize!(M)`: transfer_matrix! + factorization; never raises on a numerical
        failure (reported through `inertia`)."""
        lib = L.lib()
        info = C.c_int(0)
        A = self.A
        if isinstance(A, DeviceCSC):
            rc = lib.mnk_ls_factorize_sc(self._h, A.owner_handle, C.byref(info))
        elif isinstance(A, DeviceDense):
            rc = lib.mnk_ls_factorize_dc(self._h, A.owner_handle, C.byref(info))
        elif isinstance(A, tuple):
            colptr, rowval, nzval = (np.ascontiguousarray(A[0], dtype=np.int32),
                                     np.ascontiguousarray(A[1], dtype=np.int32),
                                     np.ascontiguousarray(A[2], dtype=np.float64))
            rc = lib.mnk_ls_factorize_csc(self._h, colptr.ctypes.data, rowval.ctypes.data, nzval.ctypes.data, 0,
                                          C.byref(info))
        else:
            p, loc = _ptr(A)
            lda = A.shape[0] if isinstance(A, np.ndarray) else max(A.stride())  # torch: symmetric, either layout
            if isinstance(A, np.ndarray) and not A.flags.f_contiguous:
                raise FactorizationException("dense matrices must be column-major (order='F')")
            rc = lib.mnk_ls_factorize_dense(self._h, p, lda, loc, C.byref(info))
        if rc:
            raise FactorizationException(lib.mnk_last_error_string().decode())
        self.info = info.value
        return self

    def factorize_async(self):
        lib = L.lib()
        A = self.A
        if isinstance(A, DeviceCSC):
            rc = lib.mnk_ls_factorize_sc_async(self._h, A.owner_handle)
        elif isinstance(A, DeviceDense):
            rc = lib.mnk_ls_factorize_dc_async(self._h, A.owner_handle)
        else:
            raise FactorizationException("factorize_async needs a device-resident matrix")
        if rc:
            raise FactorizationException(lib.mnk_last_error_string().decode())
        return self

    def inertia(self):
        p, z, n = C.c_int64(), C.c_int64(), C.c_int64()
        rc = L.lib().mnk_ls_inertia(self._h, C.byref(p), C.byref(z), C.byref(n))
        if rc:
            raise InertiaException(L.lib().mnk_last_error_string().decode())
        return (p.value, z.value, n.value)

    def solve_linear_system(self, x):
        """`solve_linear_system!(M, x)`: in place; x is a numpy vector/matrix (host) or a
        torch tensor (host or device).  Matrix right-hand sides are column-major."""
        p, loc = _ptr(x)
        if isinstance(x, np.ndarray):
            if x.ndim == 2:
                if not x.flags.f_contiguous:
                    raise SolveException("matrix right-hand sides must be column-major")
                nrhs, ldx = x.shape[1], x.shape[0]
            else:
                if not x.flags.c_contiguous:
                    raise SolveException("x must be contiguous")
                nrhs, ldx = 1, x.shape[0]
        else:
            nrhs, ldx = (1, x.shape[0]) if x.dim() == 1 else (x.shape[1], x.stride(1))
        if ldx < self.n:
            raise SolveException("right-hand side is shorter than the system order")
        rc = L.lib().mnk_ls_solve(self._h, p, nrhs, ldx, loc)
        if rc:
            raise SolveException(L.lib().mnk_last_error_string().decode())
        return x

    def check_solve(self):
        """Device-resident callers: synchronize and raise SolveException if a one-launch solve gave up
        (`mnk_ls_check_solve`); the solver has then switched to the stepwise solve."""
        rc = L.lib().mnk_ls_check_solve(self._h)
        if rc:
            raise SolveException(L.lib().mnk_last_error_string().decode())

    def bk_info(self):
        """Which tier produced the current factor (`mnk_ls_bk_info`): returns (active, count, perm, doff) --
        `active`: the factor is P A P' = L D L' from the pivoted Bunch-Kaufman tier (2x2 blocks in D);
        `count`: factorizations of this solver that took that tier; perm/doff are None unless active."""
        act, cnt = C.c_int(0), C.c_int(0)
        L.check(L.lib().mnk_ls_bk_info(self._h, C.byref(act), C.byref(cnt), None, None), "mnk_ls_bk_info")
        if not act.value:
            return False, cnt.value, None, None
        perm = np.zeros(self.n, dtype=np.int32)
        doff = np.zeros(self.n)
        L.check(L.lib().mnk_ls_bk_info(self._h, C.byref(act), C.byref(cnt), perm.ctypes.data, doff.ctypes.data),
                "mnk_ls_bk_info")
        return True, cnt.value, perm, doff

    def get_stat(self, key: str) -> float:
        """`mnk_ls_get_stat`: "panel_algo" (algorithm that produced the current factor), "pp_fallbacks"."""
        v = C.c_double(0.0)
        L.check(L.lib().mnk_ls_get_stat(self._h, key.encode(), C.byref(v)), "mnk_ls_get_stat")
        return v.value

    def set_option(self, key: str, value: float):
        L.check(L.lib().mnk_ls_set_option(self._h, key.encode(), float(value)), "mnk_ls_set_option")

    def get_factor(self):
        """(L, D) on the host, for tests."""
        Lm = np.zeros((self.n, self.n), order="F")
        D = np.zeros(self.n)
        L.check(L.lib().mnk_ls_get_factor(self._h, Lm.ctypes.data, D.ctypes.data, L.MNK_HOST), "mnk_ls_get_factor")
        return Lm, D

    def get_factor_device(self):
        """(L, D) as torch tensors on the solver's device (column-major image in a row-major tensor is transposed back), for
        tests at sizes whose factor should not cross PCIe."""
        import torch
        dev = torch.device("cuda", self.ctx.device)
        Lt = torch.empty((self.n, self.n), dtype=torch.float64, device=dev)   # receives the column-major factor: Lt = L'
        D = torch.empty(self.n, dtype=torch.float64, device=dev)
        L.check(L.lib().mnk_ls_get_factor(self._h, C.c_void_p(Lt.data_ptr()), C.c_void_p(D.data_ptr()), L.MNK_DEVICE),
                "mnk_ls_get_factor")
        return Lt.T, D

    def close(self):
        if self._h:
            rc = L.lib().mnk_ls_destroy(self._h)
            if rc:
                #  the solver is queued in ANOTHER thread's open factorization batch: the library refused to free it and
                # it stays alive -- and ours: the handle is kept, so that a later close() (after that batch has ended) frees it
                # instead of leaking a solver the other thread's batch_end would launch on
                raise RuntimeError("HipLinearSolver.close: " + L.lib().mnk_last_error_string().decode())
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceCSC:
    """`aug_com` of a sparse condensed system: lower-triangular CSC living in HBM.
    Structure on the host (0-based), values on the device."""

    def __init__(self, owner, n, colptr, rowval):
        self.owner = owner
        self.ctx = owner.ctx
        self.m = self.n = n
        self.colptr, self.rowval = colptr, rowval

    @property
    def owner_handle(self):
        return self.owner._h

    @property
    def nzval(self):
        return self.owner._values(L.MNK_SC_AUG, len(self.rowval))

    def to_dense(self):
        d = np.zeros((self.n, self.n), order="F")
        ci = np.repeat(np.arange(self.n), np.diff(self.colptr))
        d[self.rowval, ci] = self.nzval
        return d

    def nzval_host(self):
        """Values of the stored (lower-triangular) entries, copied from the device."""
        return self.nzval

    def to_scipy(self):
        """Lower-triangular scipy CSC with the current device values."""
        import scipy.sparse as sp
        return sp.csc_matrix((self.nzval, np.asarray(self.rowval), np.asarray(self.colptr)), shape=(self.n, self.n))


class DeviceDense:
    """`aug_com` of a dense KKT system living in HBM."""

    def __init__(self, owner, order):
        self.owner = owner
        self.ctx = owner.ctx
        self.order = order
        self.shape = (order, order)

    @property
    def owner_handle(self):
        return self.owner._h

    def to_host(self):
        out = np.zeros((self.order, self.order), order="F")
        L.check(L.lib().mnk_dc_get_aug(self.owner._h, out.ctypes.data, L.MNK_HOST), "mnk_dc_get_aug")
        return out


class LeadingBlock(DeviceCSC):
    """The leading principal block of order `m` of a sparse condensed system's `aug_com`: what a probe solver factors
    (`mnk_ls_factorize_sc_async` with a solver of smaller order)."""

    def __init__(self, full: DeviceCSC, m: int):
        keep = np.repeat(np.arange(full.n), np.diff(full.colptr)) < m
        keep &= np.asarray(full.rowval) < m
        colptr = np.concatenate(([0], np.cumsum(np.bincount(np.repeat(np.arange(full.n), np.diff(full.colptr))[keep], minlength=full.n)[:m])))
        super().__init__(full.owner, m, colptr.astype(np.int64), np.asarray(full.rowval)[keep])
        self._keep = keep
        self._full = full

    @property
    def nzval(self):
        return self._full.nzval[self._keep]


def _order_of(A):
    if isinstance(A, DeviceCSC):
        return A.n
    if isinstance(A, DeviceDense):
        return A.order
    if isinstance(A, tuple):
        return len(A[0]) - 1
    if A.shape[0] != A.shape[1]:
        raise SymbolicException("matrix must be square")
    return A.shape[0]
