"""madnlp.jl_amd -- MI355X-native KKT hot path for MadNLP-style interior-point solvers.

Only what the path needs: `csrc/` (HIP kernels + the C ABI of include/madnlp_hip.h) and the
host-side mirror of the reference's KKT-system / linear-solver interfaces.  Importing the
package does not load the HIP library; the first use of a KKT system or solver does, and
fails loudly if it has not been built (no CPU fallback).
"""
from ._lib import build, lib, LIBPATH, HipError  # noqa: F401
from .linear_solver import (  # noqa: F401
    BUNCHKAUFMAN, CHOLESKY, LDL, HipContext, HipLinearSolver, HipSolverOptions, factorize_batch, solve_batch, release_idle_streams,
    LinearSolverException, SymbolicException, FactorizationException, SolveException, InertiaException,
)
from .kkt import (  # noqa: F401
    UnreducedKKTVector, SparseCondensedKKTSystem, DenseCondensedKKTSystem, DenseKKTSystem, ScenarioBatch,
)
from .backsolve import RichardsonIterator  # noqa: F401

__version__ = "0.1.0"
from .schur import SchurDenseStage  # noqa: F401,E402
from .ipm_device import IPMDeviceKernels  # noqa: F401,E402
