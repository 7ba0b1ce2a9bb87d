"""CPU oracle for the MadNLP KKT hot path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy + scipy's OpenBLAS LAPACK) of the
reference algorithm for the per-iteration KKT path of MadNLP.jl v0.10.1:

    compress_hessian!/compress_jacobian! -> build_kkt! -> factorize! ->
    solve_linear_system! (+ solve_kkt!/mul!/Richardson refinement around it)

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it -- and there only as the *checker* / reported CPU baseline, never as
the thing measured or shipped.  The product path (`madnlp.jl_amd/`, the C-ABI
library) never imports or links anything from here and fails loudly when the
HIP library is missing.

Parity pinning.  The reference is Julia and cannot be run
in the build container (no Julia toolchain), and the factorization arithmetic
lives in a third-party dependency that is not vendored in the reference tree:
OpenBLAS32_jll "0.3" through libblastrampoline (reference `Project.toml:11,26`,
call sites `src/LinearSolvers/lapack.jl:56-138`).  scipy 1.15.3 exposes the same
LAPACK routine family (dsytrf/dsytrs/dpotrf/dpotrs, OpenBLAS-backed), which is
what `oracle.lapack_cpu` calls.  The oracle is pinned against every golden
vector / known-answer test the reference holds for this path:

  * `test/matrix_test.jl:21-30` + `lib/MadNLPTests/src/MadNLPTests.jl:24-51`
    (2x2 solve x = [0.8542713567839195, 1.4572864321608041], inertia (2,0,0));
  * `lib/MadNLPTests/src/MadNLPTests.jl:53-110` (`test_kkt_system` on HS15:
    mul!(y, kkt, solve_kkt!(kkt, ones)) == ones, inertia correct) for
    DenseKKTSystem, DenseCondensedKKTSystem and SparseCondensedKKTSystem;
  * `test/madnlp_dense.jl:8-53` (dense == sparse formulations on DenseDummyQP,
    restated with our own seeded RNG -- Julia's `Random.seed!(1)` stream is not
    reproducible without Julia).

Parity with MadNLP's *numerical iterates* beyond those fixtures is unpinned by
any golden file in the reference; see DESIGN.md "Oracle".

All indices are 0-based here; the reference is 1-based.
"""
