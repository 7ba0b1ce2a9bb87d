# MadNLPHIP.jl -- Julia glue binding libmadnlp_hip.so into MadNLP.jl v0.10.1.
#
# NOT executed in this repository's CI: the build image has no Julia toolchain.  What IS checked
# here (tests/test_julia_glue.py): every `ccall` in this file names a symbol that include/madnlp_hip.h
# declares, with the same arity and C types; every option key passed to mnk_ls_set_option is one the
# library accepts; the option defaults equal the library's.  The same call sequence is exercised on the
# GPU through the ctypes mirror in madnlp.jl_amd/ (tests/test_hip_parity.py).
#
# It adds three types that plug into MadNLP's own option seam
#     MadNLPHIP.madnlp_hip(nlp)      # = madnlp(nlp; MadNLPHIP.hip_sparse_condensed_options()...)
#     madnlp(nlp; kkt_system = MadNLPHIP.HipDenseCondensedKKTSystem,  linear_solver = MadNLPHIP.HipLinearSolver)
# (reference src/IPM/options.jl:121-122, consumed at src/IPM/IPM.jl:157-165); nothing else in the IPM
# loop changes.  The reference keys four option presets on `kkt_system <: MadNLP.SparseCondensedKKTSystem`
# (src/IPM/options.jl:146-147,160,226); HipSparseCondensedKKTSystem is a sibling type, not a subtype, so
# `hip_sparse_condensed_options` passes the same four values explicitly (see "option presets" below).  Pattern: ccall + finalizer as in src/LinearSolvers/mumps.jl:148-165,211 and
# src/LinearSolvers/lapack.jl:50-139; KKT-system contract as in docs/src/tutorials/diag_kkt.jl:6-215.
module MadNLPHIP

import MadNLP
import MadNLP: AbstractLinearSolver, AbstractCondensedKKTSystem, AbstractKKTVector, MadNLPLogger,
    LinearFactorization, SymbolicException, FactorizationException, SolveException, InertiaException,
    BUNCHKAUFMAN, CHOLESKY, LDL, SparseCallback, AbstractCallback, SparseMatrixCOO,
    ExactHessian, QuasiNewtonOptions, create_quasi_newton, build_hessian_structure, create_array,
    _jac_sparsity_wrapper!, force_lower_triangular!, transfer!, default_options,
    full, primal, dual, dual_lb, dual_ub
import LinearAlgebra: mul!, Symmetric
import SparseArrays: SparseMatrixCSC, nnz

const libmadnlp_hip = get(ENV, "MADNLP_HIP_LIB", "libmadnlp_hip.so")
const MNK_HOST = Cint(0)
const MNK_DEVICE = Cint(1)
const MNK_ALGO = Dict(BUNCHKAUFMAN => Cint(1), CHOLESKY => Cint(4), LDL => Cint(5))
const MNK_SC_JT, MNK_SC_HESS, MNK_SC_AUG = Cint(0), Cint(1), Cint(2)

lasterr() = unsafe_string(ccall((:mnk_last_error_string, libmadnlp_hip), Cstring, ()))
check(rc, E) = rc == 0 ? nothing : (@error("libmadnlp_hip: " * lasterr()); throw(E()))

# ------------------------------------------------------------------ handles
# Julia does not order finalizers.  The library therefore reference-counts the context: mnk_ctx_destroy
# on a context that still has live children (solver / KKT handles) only marks it released, and the last
# child's destroy frees it -- any finalizer order is safe.  Every wrapper below is additionally
# idempotent (handle reset to C_NULL).
mutable struct HipContext
    handle::Ptr{Cvoid}
    function HipContext(device::Integer = 0)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:mnk_ctx_create, libmadnlp_hip), Cint, (Cint, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), device, C_NULL, h)
        check(rc, SymbolicException)
        ctx = new(h[])
        finalizer(ctx) do c
            c.handle == C_NULL || ccall((:mnk_ctx_destroy, libmadnlp_hip), Cint, (Ptr{Cvoid},), c.handle)
            c.handle = C_NULL
        end
        return ctx
    end
end

mutable struct HipSC            # mnk_sc*: device state of the sparse condensed system
    handle::Ptr{Cvoid}
    ctx::HipContext
end
mutable struct HipDC            # mnk_dc*: device state of the dense (condensed) system
    handle::Ptr{Cvoid}
    ctx::HipContext
end
function release!(h::HipSC)
    h.handle == C_NULL || ccall((:mnk_sc_destroy, libmadnlp_hip), Cint, (Ptr{Cvoid},), h.handle)
    h.handle = C_NULL
end
function release!(h::HipDC)
    h.handle == C_NULL || ccall((:mnk_dc_destroy, libmadnlp_hip), Cint, (Ptr{Cvoid},), h.handle)
    h.handle = C_NULL
end

# ------------------------------------------------------------------ linear solver
# Defaults = the library's (csrc/ls.h); tests/test_julia_glue.py compares them.
@kwdef mutable struct HipSolverOptions <: MadNLP.AbstractOptions
    lapack_algorithm::LinearFactorization = BUNCHKAUFMAN   # served by the device LDL^T
    pivot_tol::Float64 = 0.0
    outer_block::Int = 0             # 0: by size (512 columns; 1024 from 32 768 rows on)
    lookahead::Bool = true
    share::Int = 1                  # 0 off / 1 adaptive / 2 always: panel-stream CUs join the trailing update
    persistent_solve::Bool = true   # both triangular sweeps in one launch
    single_rows::Int = 2560         # systems up to this order: one outer panel on the whole chip
    panel_algo::Int = 5             # 5: task-DAG schedule (persistent pivot chain + persistent bulk kernel); 4: persistent panel
                                    # kernel per 256 columns + one trailing update per outer panel; 1: one launch per piece (set 1
                                    # when several processes share the GPU: 4 and 5 keep waiting workgroups resident)
    bk_fallback::Bool = true        # BUNCHKAUFMAN: refactor with 1x1/2x2 Bunch-Kaufman pivoting when the static-pivot
                                    # LDL^T breaks down (false: report the breakdown as num_zero, the IPM regularizes)
end

"aug_com of `HipSparseCondensedKKTSystem`: lower-triangular CSC whose VALUES live in HBM."
struct HipAugCSC{T} <: AbstractMatrix{T}
    sc::HipSC
    structure::SparseMatrixCSC{T, Int32}       # host copy of the pattern (nzval unused)
end
Base.size(A::HipAugCSC) = size(A.structure)
nnz(A::HipAugCSC) = nnz(A.structure)
Base.getindex(A::HipAugCSC, i::Int, j::Int) = error("HipAugCSC values live on the device; use MadNLPHIP.values(A)")
"Copy nzval of the device aug_com to the host (diagnostics, tests)."
function values(A::HipAugCSC{T}) where T
    out = Vector{T}(undef, nnz(A.structure))
    rc = ccall((:mnk_sc_get_values, libmadnlp_hip), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Cint), A.sc.handle, MNK_SC_AUG, out, MNK_HOST)
    check(rc, SymbolicException)
    return out
end

"aug_com of `HipDenseCondensedKKTSystem`: the (n + n_eq)^2 dense matrix assembled in HBM."
struct HipAugDense{T} <: AbstractMatrix{T}
    dc::HipDC
    order::Int
end
Base.size(A::HipAugDense) = (A.order, A.order)
Base.getindex(A::HipAugDense, i::Int, j::Int) = Matrix(A)[i, j]
function Base.Matrix(A::HipAugDense{T}) where T
    out = Matrix{T}(undef, A.order, A.order)
    rc = ccall((:mnk_dc_get_aug, libmadnlp_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint), A.dc.handle, out, MNK_HOST)
    check(rc, SymbolicException)
    return out
end

"""
    HipLinearSolver(A; opt, logger)

`A` is the `aug_com` MadNLP hands to the solver: a device-resident `HipAugCSC` / `HipAugDense` (nothing
crosses PCIe in `factorize!`), or a host `Matrix{Float64}` / `SparseMatrixCSC{Float64,Int32}` (lower
triangle) -- the reference's own KKT systems -- to which the solver keeps a reference exactly like
`LapackCPUSolver` does (src/LinearSolvers/lapack.jl:5-44).
"""
mutable struct HipLinearSolver{T, MT} <: AbstractLinearSolver{T}
    A::MT
    handle::Ptr{Cvoid}
    ctx::HipContext
    n::Int
    info::Base.RefValue{Cint}
    opt::HipSolverOptions
    logger::MadNLPLogger
end

context_of(A::HipAugCSC) = A.sc.ctx
context_of(A::HipAugDense) = A.dc.ctx
context_of(::AbstractMatrix) = HipContext()

function set_option!(h::Ptr{Cvoid}, key::String, val::Real)
    rc = ccall((:mnk_ls_set_option, libmadnlp_hip), Cint, (Ptr{Cvoid}, Cstring, Cdouble), h, key, Float64(val))
    check(rc, SymbolicException)
end

function HipLinearSolver(A::MT; opt = HipSolverOptions(), logger = MadNLPLogger(),
                         ctx::HipContext = context_of(A)) where {MT <: AbstractMatrix}
    n = size(A, 1)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:mnk_ls_create, libmadnlp_hip), Cint, (Ptr{Cvoid}, Int64, Cint, Ptr{Ptr{Cvoid}}),
               ctx.handle, n, MNK_ALGO[opt.lapack_algorithm], h)
    check(rc, SymbolicException)
    set_option!(h[], "pivot_tol", opt.pivot_tol)
    set_option!(h[], "outer_block", opt.outer_block)
    set_option!(h[], "lookahead", opt.lookahead)
    set_option!(h[], "share", opt.share)
    set_option!(h[], "persistent_solve", opt.persistent_solve)
    set_option!(h[], "single_rows", opt.single_rows)
    set_option!(h[], "panel_algo", opt.panel_algo)
    set_option!(h[], "bk_fallback", opt.bk_fallback)
    M = HipLinearSolver{Float64, MT}(A, h[], ctx, n, Ref{Cint}(0), opt, logger)
    finalizer(M) do m
        m.handle == C_NULL || ccall((:mnk_ls_destroy, libmadnlp_hip), Cint, (Ptr{Cvoid},), m.handle)
        m.handle = C_NULL
    end
    return M
end

# factorize!: transfer_matrix! + dsytrf/dpotrf replacement (src/LinearSolvers/lapack_common.jl:54-66).
# rc != 0 is a HIP/runtime error only; a bad pivot is NOT an error, it surfaces through `inertia`.
function MadNLP.factorize!(M::HipLinearSolver{T, <:HipAugCSC}) where T
    rc = ccall((:mnk_ls_factorize_sc, libmadnlp_hip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cint}), M.handle, M.A.sc.handle, M.info)
    check(rc, FactorizationException)
    return M
end
function MadNLP.factorize!(M::HipLinearSolver{T, <:HipAugDense}) where T
    rc = ccall((:mnk_ls_factorize_dc, libmadnlp_hip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cint}), M.handle, M.A.dc.handle, M.info)
    check(rc, FactorizationException)
    return M
end
function MadNLP.factorize!(M::HipLinearSolver{T, <:Matrix}) where T
    rc = ccall((:mnk_ls_factorize_dense, libmadnlp_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Int64, Cint, Ptr{Cint}),
               M.handle, M.A, size(M.A, 1), MNK_HOST, M.info)
    check(rc, FactorizationException)
    return M
end
function MadNLP.factorize!(M::HipLinearSolver{T, <:SparseMatrixCSC}) where T
    A = M.A
    rc = ccall((:mnk_ls_factorize_csc, libmadnlp_hip), Cint,
               (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}, Ptr{Cdouble}, Cint, Ptr{Cint}),
               M.handle, A.colptr, A.rowval, A.nzval, 1, M.info)
    check(rc, FactorizationException)
    return M
end

# Batches of INDEPENDENT solver instances (scenario batches; the reference threads over them,
# src/KKT/Schur/schur.jl:927-1001 `@blas_safe_threads for k in 1:ns`): the factorizations are queued and launched
# together -- instances of the same order share one persistent launch and fill each other's chain-bound ends
# (include/madnlp_hip.h: mnk_factorize_batch_begin / _end).  `inertia` / `solve_linear_system!` of each solver
# afterwards as usual; every factor is bit-identical to a lone factorize!.
function factorize_async!(M::HipLinearSolver{T, <:HipAugCSC}) where T
    rc = ccall((:mnk_ls_factorize_sc_async, libmadnlp_hip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), M.handle, M.A.sc.handle)
    check(rc, FactorizationException)
    return M
end
function factorize_async!(M::HipLinearSolver{T, <:HipAugDense}) where T
    rc = ccall((:mnk_ls_factorize_dc_async, libmadnlp_hip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), M.handle, M.A.dc.handle)
    check(rc, FactorizationException)
    return M
end
# (The batch is THREAD-LOCAL state of the library.  A Julia task may migrate to another OS thread at any yield point inside
# f() -- I/O, a lock, a GC safepoint with a task switch -- and the `_end` call would then find no batch open on its thread while
# the first thread keeps one open for good.  The task is therefore made sticky for the duration of the block.)
function factorize_batch(f)
    t = current_task()
    was_sticky = t.sticky
    t.sticky = true
    check(ccall((:mnk_factorize_batch_begin, libmadnlp_hip), Cint, ()), FactorizationException)
    try
        f()
    finally
        rc = ccall((:mnk_factorize_batch_end, libmadnlp_hip), Cint, ())
        t.sticky = was_sticky
        check(rc, FactorizationException)
    end
    return nothing
end
factorize_batch!(Ms) = (factorize_batch(() -> foreach(factorize_async!, Ms)); Ms)
# ... and of independent solves on device vectors (mnk_solve_batch_begin / _end): up to four systems per launch
function solve_batch(f)
    t = current_task()
    was_sticky = t.sticky
    t.sticky = true          # (thread-local batch: see factorize_batch)
    check(ccall((:mnk_solve_batch_begin, libmadnlp_hip), Cint, ()), SolveException)
    try
        f()
    finally
        rc = ccall((:mnk_solve_batch_end, libmadnlp_hip), Cint, ())
        t.sticky = was_sticky
        check(rc, SolveException)
    end
    return nothing
end

"""
    release_idle_streams(device = 0)

Give the device's hardware queues back while this process has nothing to factorize (`mnk_release_idle_streams`): the CU-masked
streams of the persistent schedules are destroyed and made again by the first operation that needs them.  For a host process
that stays alive next to other GPU processes (INTEGRATION.md section 0).
"""
release_idle_streams(device::Integer = 0) =
    (check(ccall((:mnk_release_idle_streams, libmadnlp_hip), Cint, (Cint,), device), SymbolicException); nothing)

function MadNLP.solve_linear_system!(M::HipLinearSolver, x::Vector{Float64})
    rc = ccall((:mnk_ls_solve, libmadnlp_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Int64, Int64, Cint),
               M.handle, x, 1, length(x), MNK_HOST)
    check(rc, SolveException)
    return x
end

"""
    check_solve(M)

For callers that keep their vectors on the device: synchronize and throw `SolveException` if a one-launch solve gave up
(the solver has then switched to the stepwise solve; repeat the solve).  Host-vector callers never need it.
"""
function check_solve(M::HipLinearSolver)
    rc = ccall((:mnk_ls_check_solve, libmadnlp_hip), Cint, (Ptr{Cvoid},), M.handle)
    check(rc, SolveException)
    return nothing
end

"Which tier produced the current factor: (pivoted::Bool, count::Int) -- `mnk_ls_bk_info`."
function bk_info(M::HipLinearSolver)
    active = Ref{Cint}(0); count = Ref{Cint}(0)
    rc = ccall((:mnk_ls_bk_info, libmadnlp_hip), Cint, (Ptr{Cvoid}, Ptr{Cint}, Ptr{Cint}, Ptr{Int32}, Ptr{Cdouble}),
               M.handle, active, count, C_NULL, C_NULL)
    check(rc, InertiaException)
    return (active[] != 0, Int(count[]))
end

MadNLP.is_inertia(::HipLinearSolver) = true
function MadNLP.inertia(M::HipLinearSolver)
    p = Ref{Int64}(0); z = Ref{Int64}(0); n = Ref{Int64}(0)
    rc = ccall((:mnk_ls_inertia, libmadnlp_hip), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}), M.handle, p, z, n)
    check(rc, InertiaException)
    return (p[], z[], n[])
end
MadNLP.improve!(::HipLinearSolver) = false
MadNLP.introduce(M::HipLinearSolver) = "HIP-MI355X ($(M.opt.lapack_algorithm))"
MadNLP.input_type(::Type{<:HipLinearSolver}) = :dense
MadNLP.default_options(::Type{<:HipLinearSolver}) = HipSolverOptions()
MadNLP.is_supported(::Type{<:HipLinearSolver}, ::Type{Float64}) = true
MadNLP.is_supported(::Type{<:HipLinearSolver}, ::Type{Float32}) = false

# ------------------------------------------------------------------ sparse condensed KKT system
# Field-for-field the reference's SparseCondensedKKTSystem (src/KKT/Sparse/condensed.jl:8-52): every field
# the generic code touches (src/KKT/KKTsystem.jl:210-234, src/IPM/kernels.jl:4-27, src/KKT/rhs.jl:119-129)
# stays a host vector; `hess_com` / `jt_csc` are host mirrors (for jtprod! and mul_hess_blk!); the
# condensation, aug_com, the factorization, solve_kkt! and mul! live on the device behind `sc`.
struct HipSparseCondensedKKTSystem{T, VT, MT, QN, LS, VI, VI32} <: AbstractCondensedKKTSystem{T, VT, MT, QN}
    hess::VT
    hess_raw::SparseMatrixCOO{T, Int32, VT, VI32}
    hess_com::MT
    hess_csc_map::VI
    jac::VT
    jt_coo::SparseMatrixCOO{T, Int32, VT, VI32}
    jt_csc::MT
    jt_csc_map::VI
    quasi_newton::QN
    reg::VT
    pr_diag::VT
    du_diag::VT
    l_diag::VT
    u_diag::VT
    l_lower::VT
    u_lower::VT
    buffer::VT
    buffer2::VT
    aug_com::HipAugCSC{T}
    diag_buffer::VT
    linear_solver::LS
    ind_ineq::VI
    ind_lb::VI
    ind_ub::VI
    sc::HipSC
end

"0-based (colptr, rowval) of a derived structure -> 1-based SparseMatrixCSC with zero values."
function fetch_structure(sc::HipSC, which::Cint, nrow::Int, ncol::Int, nz::Int)
    colptr = Vector{Int32}(undef, ncol + 1)
    rowval = Vector{Int32}(undef, nz)
    rc = ccall((:mnk_sc_get_structure, libmadnlp_hip), Cint, (Ptr{Cvoid}, Cint, Ptr{Int32}, Ptr{Int32}), sc.handle, which, colptr, rowval)
    check(rc, SymbolicException)
    colptr .+= Int32(1)
    rowval .+= Int32(1)
    return SparseMatrixCSC{Float64, Int32}(nrow, ncol, colptr, rowval, zeros(Float64, nz))
end
function fetch_map(sc::HipSC, which::Cint, len::Int)
    map = Vector{Int64}(undef, len)
    rc = ccall((:mnk_sc_get_map, libmadnlp_hip), Cint, (Ptr{Cvoid}, Cint, Ptr{Int64}), sc.handle, which, map)
    check(rc, SymbolicException)
    return Vector{Int}(map .+ 1)
end

# ------------------------------------------------------------------ option presets
# The reference selects the sparse-condensed presets by type, `kkt_system <: MadNLP.SparseCondensedKKTSystem`
# (src/IPM/options.jl:146-147: fixed_variable_treatment = RelaxBound, equality_treatment = RelaxEquality;
# :160: dual_initialization_method = DualInitializeSetZero; :215,226: tol = get_tolerance(T, kkt_system) = 1e-4 for
# Float64).  HipSparseCondensedKKTSystem cannot be a subtype of that concrete struct, so without these four values the
# options would default to EnforceEquality / MakeParameter / DualInitializeLeastSquares / tol = 1e-8 and
# create_kkt_system below would (rightly) refuse any NLP with equality constraints -- AC-OPF included.
# `tol` needs no keyword: MadNLPOptions{T}(nlp; kkt_system, ...) calls get_tolerance(T, kkt_system) (options.jl:215).
MadNLP.get_tolerance(::Type{T}, ::Type{HipSparseCondensedKKTSystem}) where T =
    MadNLP.get_tolerance(T, MadNLP.SparseCondensedKKTSystem)

"""
    hip_sparse_condensed_options(T = Float64)

Keyword arguments that make `madnlp` / `MadNLPSolver` take the device path with exactly the presets the reference
applies to `MadNLP.SparseCondensedKKTSystem` (src/IPM/options.jl:146-147,160,226).  Later keywords override them:
`madnlp(nlp; hip_sparse_condensed_options()..., tol = 1e-6)`.
"""
hip_sparse_condensed_options(::Type{T} = Float64) where T = (
    kkt_system = HipSparseCondensedKKTSystem,
    linear_solver = HipLinearSolver,
    fixed_variable_treatment = MadNLP.RelaxBound,
    equality_treatment = MadNLP.RelaxEquality,
    dual_initialization_method = MadNLP.DualInitializeSetZero,
    tol = MadNLP.get_tolerance(T, HipSparseCondensedKKTSystem),
)

"`madnlp(nlp; ...)` on the device path: the presets above first, the caller's keywords after (they win)."
madnlp_hip(nlp::MadNLP.AbstractNLPModel{T}; kwargs...) where T =
    MadNLP.madnlp(nlp; hip_sparse_condensed_options(T)..., kwargs...)

# create_kkt_system: reference src/KKT/Sparse/condensed.jl:55-133.  coo_to_csc (x2) and
# build_condensed_aug_symbolic are replaced by ONE call, mnk_sc_create (host C++), whose results come back
# through mnk_sc_sizes / mnk_sc_get_structure / mnk_sc_get_map.
function MadNLP.create_kkt_system(
    ::Type{HipSparseCondensedKKTSystem},
    cb::SparseCallback{T, VT},
    linear_solver::Type;
    opt_linear_solver = default_options(linear_solver),
    hessian_approximation = ExactHessian,
    qn_options = QuasiNewtonOptions(),
    device::Integer = 0,
) where {T, VT}
    T === Float64 || error("HipSparseCondensedKKTSystem supports Float64 only.")
    ind_ineq = cb.ind_ineq
    n = cb.nvar
    m = cb.ncon
    length(ind_ineq) == m || error("HipSparseCondensedKKTSystem does not support equality constrained NLPs: pass " *
        "equality_treatment = MadNLP.RelaxEquality (MadNLPHIP.hip_sparse_condensed_options() does), as the reference's " *
        "own preset for SparseCondensedKKTSystem does (src/IPM/options.jl:147).")

    jac_sparsity_I = create_array(cb, Int32, cb.nnzj)
    jac_sparsity_J = create_array(cb, Int32, cb.nnzj)
    _jac_sparsity_wrapper!(cb, jac_sparsity_I, jac_sparsity_J)
    quasi_newton = create_quasi_newton(hessian_approximation, cb, n; options = qn_options)
    hess_sparsity_I, hess_sparsity_J = build_hessian_structure(cb, hessian_approximation)
    force_lower_triangular!(hess_sparsity_I, hess_sparsity_J)
    n_jac = length(jac_sparsity_I)
    n_hess = length(hess_sparsity_I)
    nlb = length(cb.ind_lb)
    nub = length(cb.ind_ub)

    reg = VT(undef, n + m); pr_diag = VT(undef, n + m); du_diag = VT(undef, m)
    l_diag = VT(undef, nlb); u_diag = VT(undef, nub); l_lower = VT(undef, nlb); u_lower = VT(undef, nub)
    buffer = VT(undef, m); buffer2 = VT(undef, m); diag_buffer = VT(undef, m)
    hess = VT(undef, n_hess)
    jac = fill!(VT(undef, n_jac), zero(T))
    hess_raw = SparseMatrixCOO(n, n, hess_sparsity_I, hess_sparsity_J, hess)
    jt_coo = SparseMatrixCOO(n, m, jac_sparsity_J, jac_sparsity_I, jac)

    ctx = HipContext(device)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:mnk_sc_create, libmadnlp_hip), Cint,
               (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Int32}, Ptr{Int32}, Int64, Ptr{Int32}, Ptr{Int32}, Cint, Ptr{Ptr{Cvoid}}),
               ctx.handle, n, m, n_jac, jac_sparsity_I, jac_sparsity_J, n_hess, hess_sparsity_I, hess_sparsity_J, 1, h)
    check(rc, SymbolicException)
    sc = HipSC(h[], ctx)
    finalizer(release!, sc)

    nz_jt = Ref{Int64}(0); nz_h = Ref{Int64}(0); nz_aug = Ref{Int64}(0); len_jptr = Ref{Int64}(0)
    rc = ccall((:mnk_sc_sizes, libmadnlp_hip), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}),
               sc.handle, nz_jt, nz_h, nz_aug, len_jptr)
    check(rc, SymbolicException)
    jt_csc = fetch_structure(sc, MNK_SC_JT, n, m, Int(nz_jt[]))
    hess_com = fetch_structure(sc, MNK_SC_HESS, n, n, Int(nz_h[]))
    jt_csc_map = fetch_map(sc, MNK_SC_JT, n_jac)
    hess_csc_map = fetch_map(sc, MNK_SC_HESS, n_hess)
    aug_com = HipAugCSC{T}(sc, fetch_structure(sc, MNK_SC_AUG, n, n, Int(nz_aug[])))

    # device-side solve_kkt!/mul!: the bound index sets go up once
    ind_lb64 = Vector{Int64}(cb.ind_lb); ind_ub64 = Vector{Int64}(cb.ind_ub)
    rc = ccall((:mnk_sc_set_bounds, libmadnlp_hip), Cint, (Ptr{Cvoid}, Int64, Ptr{Int64}, Int64, Ptr{Int64}, Cint),
               sc.handle, nlb, ind_lb64, nub, ind_ub64, 1)
    check(rc, SymbolicException)

    _linear_solver = linear_solver(aug_com; opt = opt_linear_solver)
    # is_inertia_correct accepts (n, 0, 0) only: "not positive definite" from the static-pivot tier is final, the pivoted
    # tier could only confirm the rejection
    _linear_solver isa HipLinearSolver && set_option!(_linear_solver.handle, "accept_only_pd", 1)
    # ... and `should_regularize_dual` is `true` whatever the counts (:141): only "positive definite or not" is read from this
    # system's inertia, so the factorization of a matrix that is not may stop at its first non-positive pivot (as dpotrf does)
    _linear_solver isa HipLinearSolver && set_option!(_linear_solver.handle, "early_reject", 1)
    return HipSparseCondensedKKTSystem(
        hess, hess_raw, hess_com, hess_csc_map,
        jac, jt_coo, jt_csc, jt_csc_map,
        quasi_newton,
        reg, pr_diag, du_diag, l_diag, u_diag, l_lower, u_lower,
        buffer, buffer2, aug_com, diag_buffer,
        _linear_solver, ind_ineq, cb.ind_lb, cb.ind_ub, sc,
    )
end

MadNLP.num_variables(kkt::HipSparseCondensedKKTSystem) = length(kkt.pr_diag)
MadNLP.is_inertia_correct(kkt::HipSparseCondensedKKTSystem, num_pos, num_zero, num_neg) =
    (num_zero == 0) && (num_pos == size(kkt.aug_com, 1))                      # condensed.jl:138-140
MadNLP.should_regularize_dual(::HipSparseCondensedKKTSystem, num_pos, num_zero, num_neg) = true   # :141
MadNLP.get_jacobian(kkt::HipSparseCondensedKKTSystem) = kkt.jac               # :368
MadNLP.nnz_jacobian(kkt::HipSparseCondensedKKTSystem) = nnz(kkt.jt_coo)
Base.size(kkt::HipSparseCondensedKKTSystem, n::Int) = size(kkt.aug_com, n)

function MadNLP.initialize!(kkt::HipSparseCondensedKKTSystem{T}) where T       # src/KKT/Sparse/utils.jl:52-62
    fill!(kkt.reg, one(T)); fill!(kkt.pr_diag, one(T)); fill!(kkt.du_diag, zero(T)); fill!(kkt.hess, zero(T))
    fill!(kkt.l_lower, zero(T)); fill!(kkt.u_lower, zero(T)); fill!(kkt.l_diag, one(T)); fill!(kkt.u_diag, one(T))
    fill!(nonzeros_of(kkt.hess_com), zero(T))
    return
end
nonzeros_of(A::SparseMatrixCSC) = A.nzval

# compress_jacobian! (condensed.jl:145-148): the device does the segmented COO -> CSC sum (mnk_sc_compress_jacobian);
# the host mirror is kept current for jtprod! (one O(nnz) scatter, as the reference does).
function MadNLP.compress_jacobian!(kkt::HipSparseCondensedKKTSystem)
    transfer!(kkt.jt_csc, kkt.jt_coo, kkt.jt_csc_map)
    rc = ccall((:mnk_sc_compress_jacobian, libmadnlp_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint), kkt.sc.handle, kkt.jac, MNK_HOST)
    check(rc, SymbolicException)
    return
end
# compress_hessian! (src/KKT/Sparse/utils.jl:48-50)
function MadNLP.compress_hessian!(kkt::HipSparseCondensedKKTSystem)
    transfer!(kkt.hess_com, kkt.hess_raw, kkt.hess_csc_map)
    rc = ccall((:mnk_sc_compress_hessian, libmadnlp_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint), kkt.sc.handle, kkt.hess, MNK_HOST)
    check(rc, SymbolicException)
    return
end

# build_kkt! (condensed.jl:354-366): D = Sigma_s ./ (1 - Sigma_d Sigma_s) and the condensation run on the device;
# the barrier terms of the iterate follow for the device-side solve_kkt!/mul!.
function MadNLP.build_kkt!(kkt::HipSparseCondensedKKTSystem)
    rc = ccall((:mnk_sc_build, libmadnlp_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Cint),
               kkt.sc.handle, kkt.pr_diag, kkt.du_diag, MNK_HOST)
    check(rc, SymbolicException)
    rc = ccall((:mnk_sc_set_barrier_terms, libmadnlp_hip), Cint,
               (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint),
               kkt.sc.handle, kkt.reg, kkt.l_diag, kkt.u_diag, kkt.l_lower, kkt.u_lower, MNK_HOST)
    check(rc, SymbolicException)
    return
end

function MadNLP.jtprod!(y::AbstractVector, kkt::HipSparseCondensedKKTSystem, x::AbstractVector)   # condensed.jl:150-156
    n = size(kkt.hess_com, 1)
    mul!(view(y, 1:n), kkt.jt_csc, x)
    y[n+1:end] .= .-x
    return y
end

# solve_kkt! (src/IPM/factorization.jl:143-167): reduce_rhs!, condensation of the right-hand side, the
# triangular solves, the expansion and finish_aug_solve! all run on the device; full(w) makes one round trip.
function MadNLP.solve_kkt!(kkt::HipSparseCondensedKKTSystem, w::AbstractKKTVector)
    rc = ccall((:mnk_sc_solve_kkt, libmadnlp_hip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Cint),
               kkt.sc.handle, kkt.linear_solver.handle, full(w), MNK_HOST)
    check(rc, SolveException)
    return w
end
# mul!(w, kkt, x, alpha, beta) (src/IPM/factorization.jl:278-299 + _kktmul! src/IPM/kernels.jl:161-180)
function mul!(w::AbstractKKTVector{T}, kkt::HipSparseCondensedKKTSystem, x::AbstractKKTVector, alpha = one(T), beta = zero(T)) where T
    rc = ccall((:mnk_sc_mul, libmadnlp_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Cdouble, Cint),
               kkt.sc.handle, full(w), full(x), Float64(alpha), Float64(beta), MNK_HOST)
    check(rc, SolveException)
    return w
end
function MadNLP.mul_hess_blk!(wx, kkt::HipSparseCondensedKKTSystem, t)            # factorization.jl:333-338
    n = size(kkt.hess_com, 1)
    mul!(@view(wx[1:n]), Symmetric(kkt.hess_com, :L), @view(t[1:n]))
    fill!(@view(wx[n+1:end]), 0)
    wx .+= t .* kkt.pr_diag
end

# ------------------------------------------------------------------ dense condensed KKT system
# reference src/KKT/Dense/condensed.jl:10-111.  The callbacks write kkt.hess / kkt.jac on the host (get_hessian /
# get_jacobian return them, src/KKT/KKTsystem.jl:238-239); build_kkt! uploads both and assembles aug_com on the
# device (scale + fp64-MFMA Gram product + scatter).
struct HipDenseCondensedKKTSystem{T, VT <: AbstractVector{T}, MT <: AbstractMatrix{T}, QN, LS, VI <: AbstractVector{Int}} <:
       AbstractCondensedKKTSystem{T, VT, MT, QN}
    hess::MT
    jac::MT
    quasi_newton::QN
    reg::VT
    pr_diag::VT
    du_diag::VT
    l_diag::VT
    u_diag::VT
    l_lower::VT
    u_lower::VT
    aug_com::HipAugDense{T}
    n_eq::Int
    ind_eq::VI
    n_ineq::Int
    ind_ineq::VI
    ind_lb::VI
    ind_ub::VI
    linear_solver::LS
    dc::HipDC
end

function MadNLP.create_kkt_system(
    ::Type{HipDenseCondensedKKTSystem},
    cb::AbstractCallback{T, VT},
    linear_solver::Type;
    opt_linear_solver = default_options(linear_solver),
    hessian_approximation = ExactHessian,
    qn_options = QuasiNewtonOptions(),
    device::Integer = 0,
) where {T, VT}
    T === Float64 || error("HipDenseCondensedKKTSystem supports Float64 only.")
    n = cb.nvar
    m = cb.ncon
    ns = length(cb.ind_ineq)
    n_eq = m - ns
    nlb = length(cb.ind_lb)
    nub = length(cb.ind_ub)
    hess = fill!(create_array(cb, n, n), zero(T))
    jac = fill!(create_array(cb, m, n), zero(T))
    reg = VT(undef, n + ns)
    pr_diag = fill!(VT(undef, n + ns), zero(T))
    du_diag = fill!(VT(undef, m), zero(T))
    l_diag = fill!(VT(undef, nlb), one(T)); u_diag = fill!(VT(undef, nub), one(T))
    l_lower = fill!(VT(undef, nlb), zero(T)); u_lower = fill!(VT(undef, nub), zero(T))

    ctx = HipContext(device)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    ind_ineq64 = Vector{Int64}(cb.ind_ineq); ind_eq64 = Vector{Int64}(cb.ind_eq)
    rc = ccall((:mnk_dc_create, libmadnlp_hip), Cint,
               (Ptr{Cvoid}, Cint, Int64, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Cint, Ptr{Ptr{Cvoid}}),
               ctx.handle, 1, n, m, ns, ind_ineq64, ind_eq64, 1, h)
    check(rc, SymbolicException)
    dc = HipDC(h[], ctx)
    finalizer(release!, dc)
    ind_lb64 = Vector{Int64}(cb.ind_lb); ind_ub64 = Vector{Int64}(cb.ind_ub)
    rc = ccall((:mnk_dc_set_bounds, libmadnlp_hip), Cint, (Ptr{Cvoid}, Int64, Ptr{Int64}, Int64, Ptr{Int64}, Cint),
               dc.handle, nlb, ind_lb64, nub, ind_ub64, 1)
    check(rc, SymbolicException)
    order = Int(ccall((:mnk_dc_order, libmadnlp_hip), Int64, (Ptr{Cvoid},), dc.handle))
    aug_com = HipAugDense{T}(dc, order)

    quasi_newton = create_quasi_newton(hessian_approximation, cb, n; options = qn_options)
    _linear_solver = linear_solver(aug_com; opt = opt_linear_solver)
    (n_eq == 0 && _linear_solver isa HipLinearSolver) && set_option!(_linear_solver.handle, "accept_only_pd", 1)
    return HipDenseCondensedKKTSystem(
        hess, jac, quasi_newton,
        reg, pr_diag, du_diag, l_diag, u_diag, l_lower, u_lower,
        aug_com, n_eq, cb.ind_eq, ns, cb.ind_ineq, cb.ind_lb, cb.ind_ub,
        _linear_solver, dc,
    )
end

MadNLP.num_variables(kkt::HipDenseCondensedKKTSystem) = size(kkt.hess, 1)
MadNLP.get_slack_regularization(kkt::HipDenseCondensedKKTSystem) =
    view(kkt.pr_diag, size(kkt.hess, 1)+1:size(kkt.hess, 1)+kkt.n_ineq)
MadNLP.is_inertia_correct(kkt::HipDenseCondensedKKTSystem, num_pos, num_zero, num_neg) =
    (num_zero == 0 && num_neg == kkt.n_eq)                                    # Dense/condensed.jl:189-191
MadNLP.compress_jacobian!(::HipDenseCondensedKKTSystem) = nothing              # Dense/utils.jl:25-27
MadNLP.nnz_jacobian(kkt::HipDenseCondensedKKTSystem) = length(kkt.jac)
function MadNLP.jtprod!(y::AbstractVector, kkt::HipDenseCondensedKKTSystem, x::AbstractVector)   # Dense/utils.jl:4-18
    n = size(kkt.hess, 1)
    ns = kkt.n_ineq
    mul!(view(y, 1:n), kkt.jac', x)
    fill!(view(y, n+1:n+ns), zero(eltype(y)))
    view(y, n+1:n+ns) .-= view(x, kkt.ind_ineq)
    return y
end

# build_kkt! (Dense/condensed.jl:157-186)
function MadNLP.build_kkt!(kkt::HipDenseCondensedKKTSystem)
    n = size(kkt.hess, 1); m = size(kkt.jac, 1)
    rc = ccall((:mnk_dc_set_hess, libmadnlp_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Int64, Cint), kkt.dc.handle, kkt.hess, n, MNK_HOST)
    check(rc, SymbolicException)
    rc = ccall((:mnk_dc_set_jac, libmadnlp_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Int64, Cint), kkt.dc.handle, kkt.jac, m, MNK_HOST)
    check(rc, SymbolicException)
    rc = ccall((:mnk_dc_build, libmadnlp_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Cint),
               kkt.dc.handle, kkt.pr_diag, kkt.du_diag, MNK_HOST)
    check(rc, SymbolicException)
    rc = ccall((:mnk_dc_set_barrier_terms, libmadnlp_hip), Cint,
               (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint),
               kkt.dc.handle, kkt.reg, kkt.l_diag, kkt.u_diag, kkt.l_lower, kkt.u_lower, MNK_HOST)
    check(rc, SymbolicException)
    return
end

# solve_kkt! (src/IPM/factorization.jl:190-229) and mul! (:310-330) on the device
function MadNLP.solve_kkt!(kkt::HipDenseCondensedKKTSystem, w::AbstractKKTVector)
    rc = ccall((:mnk_dc_solve_kkt, libmadnlp_hip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Cint),
               kkt.dc.handle, kkt.linear_solver.handle, full(w), MNK_HOST)
    check(rc, SolveException)
    return w
end
function mul!(w::AbstractKKTVector{T}, kkt::HipDenseCondensedKKTSystem, x::AbstractKKTVector, alpha = one(T), beta = zero(T)) where T
    rc = ccall((:mnk_dc_mul, libmadnlp_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Cdouble, Cint),
               kkt.dc.handle, full(w), full(x), Float64(alpha), Float64(beta), MNK_HOST)
    check(rc, SolveException)
    return w
end
function MadNLP.mul_hess_blk!(wx, kkt::HipDenseCondensedKKTSystem, t)              # factorization.jl:326-331
    n = size(kkt.hess, 1)
    mul!(@view(wx[1:n]), Symmetric(kkt.hess, :L), @view(t[1:n]))
    fill!(@view(wx[n+1:end]), 0)
    wx .+= t .* kkt.pr_diag
end

# ---- device-side feeders (SURVEY 8(a)11 on the device; first slice of 8(f).4) ---------------------------------------
# set_aug_diagonal!(kkt, solver) (src/IPM/kernels.jl:4-27) and regularize_diagonal! (src/KKT/KKTsystem.jl:222-226)
# evaluated INSIDE the handle from the iterate's full primal-length vectors, then build_kkt! from the handle's own
# diagonals: for callers that keep x, xl, xu, zl, zu in host arrays but do not want to form the diagonals on the host
# (the same entry points take device pointers with MNK_DEVICE for a device-resident IPM).
function set_aug_diagonal_device!(kkt::HipSparseCondensedKKTSystem, x::Vector{Float64}, xl::Vector{Float64},
                                  xu::Vector{Float64}, zl::Vector{Float64}, zu::Vector{Float64};
                                  primal_reg::Float64 = 0.0, dual_reg::Float64 = 0.0)
    rc = ccall((:mnk_sc_set_aug_diagonal, libmadnlp_hip), Cint,
               (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Cdouble, Cint),
               kkt.sc.handle, x, xl, xu, zl, zu, primal_reg, dual_reg, MNK_HOST)
    check(rc, SymbolicException)
    return
end
function regularize_diagonal_device!(kkt::HipSparseCondensedKKTSystem, primal::Float64, dual::Float64)
    rc = ccall((:mnk_sc_regularize_diagonal, libmadnlp_hip), Cint, (Ptr{Cvoid}, Cdouble, Cdouble), kkt.sc.handle, primal, dual)
    check(rc, SymbolicException)
    return
end
function build_kkt_device!(kkt::HipSparseCondensedKKTSystem)
    rc = ccall((:mnk_sc_build, libmadnlp_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Cint),
               kkt.sc.handle, C_NULL, C_NULL, MNK_DEVICE)
    check(rc, SymbolicException)
    return
end
function set_aug_diagonal_device!(kkt::HipDenseCondensedKKTSystem, x::Vector{Float64}, xl::Vector{Float64},
                                  xu::Vector{Float64}, zl::Vector{Float64}, zu::Vector{Float64};
                                  primal_reg::Float64 = 0.0, dual_reg::Float64 = 0.0)
    rc = ccall((:mnk_dc_set_aug_diagonal, libmadnlp_hip), Cint,
               (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Cdouble, Cint),
               kkt.dc.handle, x, xl, xu, zl, zu, primal_reg, dual_reg, MNK_HOST)
    check(rc, SymbolicException)
    return
end
function regularize_diagonal_device!(kkt::HipDenseCondensedKKTSystem, primal::Float64, dual::Float64)
    rc = ccall((:mnk_dc_regularize_diagonal, libmadnlp_hip), Cint, (Ptr{Cvoid}, Cdouble, Cdouble), kkt.dc.handle, primal, dual)
    check(rc, SymbolicException)
    return
end

"What ran: `get_stat(M, \"panel_algo\")`, `\"pp_fallbacks\"`, `\"growth\"`, ... (the keys of `mnk_ls_get_stat`, INTEGRATION.md)."
function get_stat(M::HipLinearSolver, key::AbstractString)
    v = Ref{Cdouble}(0.0)
    rc = ccall((:mnk_ls_get_stat, libmadnlp_hip), Cint, (Ptr{Cvoid}, Cstring, Ptr{Cdouble}), M.handle, key, v)
    check(rc, SymbolicException)
    return v[]
end

# ------------------------------------------------------------------ dense S stage of the Schur-complement KKT system
# Reference: `SchurComplementKKTSystem`, src/KKT/Schur/schur.jl -- `build_kkt!` :927-1001 (factor every scenario block,
# S = S0 - sum_k C_dk A_k^-1 C_dk'), `factorize_kkt!` :1003-1005, steps 3-5 of `solve_kkt!` :1040-1058, `is_inertia_correct`
# :901-903.  The scenario blocks arrive dense (the reference factors them with a sparse solver per scenario, outside this
# path).  One handle holds the scenarios of ONE rank; the two sums over ranks (S, and the nd-vector of the forward sweep)
# are the caller's all-reduces (RCCL through the MPI / AMDGPU layer of the application).  `S`, `rhs_k` (blk x ns_local,
# column k = scenario k), `rhs_d` and `contrib_d` are DEVICE buffers, passed as raw pointers (`pointer(::ROCArray)`):
# the glue does not depend on AMDGPU.jl.
mutable struct HipSchurStage
    handle::Ptr{Cvoid}
    ctx::HipContext
    ns_local::Int
    blk::Int
    nd::Int
end
function HipSchurStage(ns_local::Integer, blk::Integer, nd::Integer; ctx::HipContext = HipContext(),
                       lapack_algorithm::LinearFactorization = BUNCHKAUFMAN)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:mnk_schur_create, libmadnlp_hip), Cint, (Ptr{Cvoid}, Int64, Int64, Int64, Cint, Ptr{Ptr{Cvoid}}),
               ctx.handle, ns_local, blk, nd, MNK_ALGO[lapack_algorithm], h)
    check(rc, SymbolicException)
    st = HipSchurStage(h[], ctx, ns_local, blk, nd)
    finalizer(st) do x
        x.handle == C_NULL || ccall((:mnk_schur_destroy, libmadnlp_hip), Cint, (Ptr{Cvoid},), x.handle)
        x.handle = C_NULL
    end
    return st
end
"Scenario block k (0-based): `A_kk` blk x blk (lower triangle read), `C_dk` nd x blk, host matrices."
function set_block!(st::HipSchurStage, k::Integer, A_kk::Matrix{Float64}, C_dk::Matrix{Float64})
    rc = ccall((:mnk_schur_set_block, libmadnlp_hip), Cint, (Ptr{Cvoid}, Int64, Ptr{Cdouble}, Int64, Ptr{Cdouble}, Int64, Cint),
               st.handle, k, A_kk, size(A_kk, 1), C_dk, size(C_dk, 1), MNK_HOST)
    check(rc, SymbolicException)
    return st
end
"`build_kkt!`: S_out (device, nd x nd) = S0 - sum over the local scenarios; `S0` (host) on the rank that owns it, `nothing` elsewhere."
function build_local!(st::HipSchurStage, S0::Union{Nothing, Matrix{Float64}}, S_out::Ptr{Cdouble})
    rc = ccall((:mnk_schur_build_local, libmadnlp_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Int64, Cint, Ptr{Cdouble}, Int64),
               st.handle, S0 === nothing ? C_NULL : S0, st.nd, MNK_HOST, S_out, st.nd)
    check(rc, FactorizationException)
    return st
end
"""
Once per system: the COO patterns (1-based, as the callbacks give them) of the Lagrangian Hessian and of the Jacobian (row =
constraint), the inequality rows in slack order and the equality rows.  The library checks the two-stage layout (what
`_build_schur_symbolic` checks, schur.jl:140-236) and builds the source list of every touched entry of `A_k` / `C_dk` / `S0`.
"""
function set_structure!(st::HipSchurStage, n::Integer, m::Integer, nv::Integer, nc::Integer, hess_I::Vector{Int32},
                        hess_J::Vector{Int32}, jac_I::Vector{Int32}, jac_J::Vector{Int32}, ind_ineq::Vector{Int64},
                        ind_eq::Vector{Int64}; ns_global::Integer = st.ns_local, local_scen::Union{Nothing, Vector{Int64}} = nothing,
                        own_design::Bool = true)
    rc = ccall((:mnk_schur_set_structure, libmadnlp_hip), Cint,
               (Ptr{Cvoid}, Int64, Int64, Int64, Int64, Int64, Ptr{Int32}, Ptr{Int32}, Int64, Ptr{Int32}, Ptr{Int32}, Int64, Ptr{Int64},
                Int64, Ptr{Int64}, Cint, Int64, Ptr{Int64}, Cint),
               st.handle, n, m, nv, nc, length(hess_I), hess_I, hess_J, length(jac_I), jac_I, jac_J, length(ind_ineq), ind_ineq,
               length(ind_eq), ind_eq, 1, ns_global, local_scen === nothing ? C_NULL : local_scen, own_design ? 1 : 0)
    check(rc, SymbolicException)
    return st
end
"`build_kkt!`'s scatter (schur.jl:935-972) on the device: `A_k`, `C_dk` of every local scenario and `S0` from the callbacks' values."
function assemble!(st::HipSchurStage, hess::Vector{Float64}, jac::Vector{Float64}, pr_diag::Vector{Float64}, du_diag::Vector{Float64})
    rc = ccall((:mnk_schur_assemble, libmadnlp_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint),
               st.handle, hess, jac, pr_diag, du_diag, MNK_HOST)
    check(rc, SymbolicException)
    return st
end
"`build_kkt!` behind `assemble!`: S_out = (the assembled S0, a device buffer of the stage) - sum over the local scenarios."
function build_local_assembled!(st::HipSchurStage, S_out::Ptr{Cdouble})
    s0 = ccall((:mnk_schur_s0_buffer, libmadnlp_hip), Ptr{Cvoid}, (Ptr{Cvoid},), st.handle)
    s0 == C_NULL && throw(SymbolicException())
    rc = ccall((:mnk_schur_build_local, libmadnlp_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Int64, Cint, Ptr{Cdouble}, Int64),
               st.handle, Ptr{Cdouble}(s0), st.nd, MNK_DEVICE, S_out, st.nd)
    check(rc, FactorizationException)
    return st
end
"`factorize_kkt!`: factor the (all-reduced) S, a device buffer."
function factorize_s!(st::HipSchurStage, S::Ptr{Cdouble})
    info = Ref{Cint}(0)
    rc = ccall((:mnk_schur_factorize_s, libmadnlp_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Int64, Cint, Ptr{Cint}),
               st.handle, S, st.nd, MNK_DEVICE, info)
    check(rc, FactorizationException)
    return st
end
function inertia_s(st::HipSchurStage)
    p = Ref{Int64}(0); z = Ref{Int64}(0); n = Ref{Int64}(0)
    rc = ccall((:mnk_schur_inertia_s, libmadnlp_hip), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}), st.handle, p, z, n)
    check(rc, InertiaException)
    return (Int(p[]), Int(z[]), Int(n[]))
end
function scenario_inertia(st::HipSchurStage, k::Integer)
    p = Ref{Int64}(0); z = Ref{Int64}(0); n = Ref{Int64}(0)
    rc = ccall((:mnk_schur_scenario_inertia, libmadnlp_hip), Cint, (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}),
               st.handle, k, p, z, n)
    check(rc, InertiaException)
    return (Int(p[]), Int(z[]), Int(n[]))
end
"The reference's test on S (schur.jl:901-903): positive definite of order nd."
is_inertia_correct(st::HipSchurStage) = inertia_s(st) == (st.nd, 0, 0)
"Step 3: r_k <- A_k^-1 r_k and contrib_d = -sum_k C_dk r_k (the caller adds r_d and all-reduces the nd doubles)."
function forward!(st::HipSchurStage, rhs_k::Ptr{Cdouble}, contrib_d::Ptr{Cdouble})
    rc = ccall((:mnk_schur_forward, libmadnlp_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}), st.handle, rhs_k, contrib_d)
    check(rc, SolveException)
    return st
end
"Step 4: S x_d = r_d, in place."
function solve_s!(st::HipSchurStage, rhs_d::Ptr{Cdouble})
    rc = ccall((:mnk_schur_solve_s, libmadnlp_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), st.handle, rhs_d)
    check(rc, SolveException)
    return st
end
"Step 5: x_k = r_k - (A_k^-1 C_dk') x_d, in place in `rhs_k`."
function backward!(st::HipSchurStage, rhs_k::Ptr{Cdouble}, x_d::Ptr{Cdouble})
    rc = ccall((:mnk_schur_backward, libmadnlp_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}), st.handle, rhs_k, x_d)
    check(rc, SolveException)
    return st
end

# ------------------------------------------------------------------ SchurComplementKKTSystem on the HIP stage
# Reference: `SchurComplementKKTSystem` (src/KKT/Schur/schur.jl:72-1146).  The host side of that type -- COO / CSC copies of the
# Hessian and the Jacobian, the per-scenario index maps of `_build_schur_symbolic` (:140-700), the scatter of the callback
# values into the scenario blocks `A_kk` / `C_dk` and into S (:935-972, :993-996), the vector algebra of `solve_kkt!`
# (:1040-1110), `mul!` / `jtprod!` -- is exactly the reference's: this type WRAPS a reference system (built with a scenario
# solver that does nothing) and replaces what the reference does with one sparse solver per scenario and a host LAPACK call:
# the ns block factorizations (one batch on the device), `S -= sum_k C_dk A_k^-1 C_dk'` (matrix cores), the factorization of S,
# and steps 3-5 of the solve (`mnk_schur_build_local`, `_factorize_s`, `_solve`).  `madnlp(nlp; kkt_system =
# HipSchurComplementKKTSystem, linear_solver = HipLinearSolver, kkt_options = schur_opts(; ns, nv, nd, nc))`.
"A scenario solver that is never asked to factorize (the device stage owns the blocks)."
struct HipNoScenarioSolver{T} <: AbstractLinearSolver{T}
    A::SparseMatrixCSC{T, Int32}
end
HipNoScenarioSolver(A::SparseMatrixCSC{T, Int32}; kwargs...) where T = HipNoScenarioSolver{T}(A)
MadNLP.factorize!(M::HipNoScenarioSolver) = M
MadNLP.introduce(::HipNoScenarioSolver) = "none (HIP Schur stage)"

"The solver of the design block S as the interior-point loop sees it: `factorize!` / `inertia` of the stage's S."
mutable struct HipSchurDesignSolver{T} <: AbstractLinearSolver{T}
    stage::HipSchurStage
    S::Ptr{Cdouble}            # device buffer of the stage (mnk_schur_s_buffer)
end
HipSchurDesignSolver(A::AbstractMatrix; kwargs...) = error("HipSchurDesignSolver is created by HipSchurComplementKKTSystem")
MadNLP.factorize!(M::HipSchurDesignSolver) = (factorize_s!(M.stage, M.S); M)
MadNLP.is_inertia(::HipSchurDesignSolver) = true
MadNLP.inertia(M::HipSchurDesignSolver) = inertia_s(M.stage)
MadNLP.improve!(::HipSchurDesignSolver) = false
MadNLP.introduce(::HipSchurDesignSolver) = "HIP-MI355X Schur stage (batched scenario blocks + dense S)"

struct HipSchurComplementKKTSystem{T, VT, MT, QN, K <: MadNLP.SchurComplementKKTSystem{T, VT, MT, QN}} <:
       AbstractCondensedKKTSystem{T, VT, MT, QN}
    inner::K                       # the reference system: value buffers, diagonals, host-side algebra of solve_kkt! / mul!
    stage::HipSchurStage
    linear_solver::HipSchurDesignSolver{T}
    rhs_all::Matrix{T}             # blk x ns: the scenarios' right-hand sides, column k = scenario k
end
# every field the generic IPM code reads by name (KKTsystem.jl:210-234, kernels.jl) lives in the reference system
function Base.getproperty(kkt::HipSchurComplementKKTSystem, f::Symbol)
    f in (:inner, :stage, :linear_solver, :rhs_all) && return getfield(kkt, f)
    return getproperty(getfield(kkt, :inner), f)
end

function MadNLP.create_kkt_system(
    ::Type{HipSchurComplementKKTSystem}, cb::SparseCallback{T, VT}, linear_solver::Type;
    opt_linear_solver = default_options(linear_solver), hessian_approximation = ExactHessian, qn_options = QuasiNewtonOptions(),
    schur_ns::Int = 0, schur_nv::Int = 0, schur_nd::Int = 0, schur_nc::Int = 0, device::Integer = 0,
) where {T <: Float64, VT <: Vector{T}}
    inner = MadNLP.create_kkt_system(MadNLP.SchurComplementKKTSystem, cb, MadNLP.LapackCPUSolver;
                                     hessian_approximation = hessian_approximation, qn_options = qn_options,
                                     schur_ns = schur_ns, schur_nv = schur_nv, schur_nd = schur_nd, schur_nc = schur_nc,
                                     schur_scenario_linear_solver = HipNoScenarioSolver)
    ctx = HipContext(device)
    alg = hasproperty(opt_linear_solver, :lapack_algorithm) ? opt_linear_solver.lapack_algorithm : BUNCHKAUFMAN
    stage = HipSchurStage(inner.ns, inner.blk_size, inner.nd; ctx = ctx, lapack_algorithm = alg)
    Sdev = Ptr{Cdouble}(ccall((:mnk_schur_s_buffer, libmadnlp_hip), Ptr{Cvoid}, (Ptr{Cvoid},), stage.handle))
    Sdev == C_NULL && throw(SymbolicException())
    # the index maps of the scatter live in the library (built from the COO patterns the callbacks gave; jt_coo is J': I = variable)
    set_structure!(stage, MadNLP.num_variables(inner), length(inner.du_diag), inner.nv, inner.nc,
                   Vector{Int32}(inner.hess_raw.I), Vector{Int32}(inner.hess_raw.J), Vector{Int32}(inner.jt_coo.J),
                   Vector{Int32}(inner.jt_coo.I), Vector{Int64}(inner.ind_ineq), Vector{Int64}(inner.ind_eq))
    return HipSchurComplementKKTSystem(inner, stage, HipSchurDesignSolver{T}(stage, Sdev), zeros(T, inner.blk_size, inner.ns))
end

MadNLP.num_variables(kkt::HipSchurComplementKKTSystem) = MadNLP.num_variables(kkt.inner)
MadNLP.get_slack_regularization(kkt::HipSchurComplementKKTSystem) = MadNLP.get_slack_regularization(kkt.inner)
MadNLP.is_inertia_correct(kkt::HipSchurComplementKKTSystem, num_pos, num_zero, num_neg) =
    (num_zero == 0) && (num_pos == kkt.inner.nd)                                          # schur.jl:901-903
MadNLP.should_regularize_dual(::HipSchurComplementKKTSystem, num_pos, num_zero, num_neg) = true   # :905
MadNLP.jtprod!(y::AbstractVector, kkt::HipSchurComplementKKTSystem, x::AbstractVector) = MadNLP.jtprod!(y, kkt.inner, x)
MadNLP.compress_jacobian!(kkt::HipSchurComplementKKTSystem) = MadNLP.compress_jacobian!(kkt.inner)
MadNLP.compress_hessian!(kkt::HipSchurComplementKKTSystem) = MadNLP.compress_hessian!(kkt.inner)
MadNLP.nnz_jacobian(kkt::HipSchurComplementKKTSystem) = MadNLP.nnz_jacobian(kkt.inner)
MadNLP.get_jacobian(kkt::HipSchurComplementKKTSystem) = kkt.inner.jac
MadNLP.get_hessian(kkt::HipSchurComplementKKTSystem) = kkt.inner.hess
MadNLP.initialize!(kkt::HipSchurComplementKKTSystem) = MadNLP.initialize!(kkt.inner)
Base.size(kkt::HipSchurComplementKKTSystem, n::Int) = kkt.inner.nd
mul!(w::AbstractKKTVector{T}, kkt::HipSchurComplementKKTSystem, x::AbstractKKTVector, alpha = one(T), beta = zero(T)) where T =
    mul!(w, kkt.inner, x, alpha, beta)
MadNLP.mul_hess_blk!(wx, kkt::HipSchurComplementKKTSystem, t) = MadNLP.mul_hess_blk!(wx, kkt.inner, t)

# build_kkt! (schur.jl:927-1001).  Round 6: the scatter of the callback values into A_kk / C_dk / S0 (:935-972) is ONE call -- the
# library sums every touched entry of the dense blocks on the device, in the reference's own order, from the four value vectors;
# rounds 4-5 ran the reference's `_scatter_add!` / `_scatter_quad_add!` on the host, densified every sparse A_kk and uploaded
# ns (blk^2 + nd blk) doubles per iteration.  The host keeps `diag_buffer` (the vector algebra of solve_kkt! reads it).
function MadNLP.build_kkt!(kkt::HipSchurComplementKKTSystem{T}) where T
    k0 = kkt.inner
    n = MadNLP.num_variables(k0)
    if k0.n_ineq > 0
        Sigma_s = view(k0.pr_diag, n+1:n+k0.n_ineq)
        Sigma_d = @view(k0.du_diag[k0.ind_ineq])
        k0.diag_buffer .= Sigma_s ./ (one(T) .- Sigma_d .* Sigma_s)
    end
    assemble!(kkt.stage, k0.hess, k0.jac, k0.pr_diag, k0.du_diag)
    build_local_assembled!(kkt.stage, kkt.linear_solver.S)   # blocks factored as one batch, S = S0 - sum_k C_dk A_k^-1 C_dk'
    return
end

MadNLP.factorize_kkt!(kkt::HipSchurComplementKKTSystem) = MadNLP.factorize!(kkt.linear_solver)

# solve_kkt! (schur.jl:1040-1110): the reference's steps 1-2 and 6-7 on the host, steps 3-5 in one device call
function MadNLP.solve_kkt!(kkt::HipSchurComplementKKTSystem, w::AbstractKKTVector{T}) where T
    k0 = kkt.inner
    ns, nv, nd, n, blk, nc_eq = k0.ns, k0.nv, k0.nd, MadNLP.num_variables(k0), k0.blk_size, k0.nc_eq_per_s
    wx = view(full(w), 1:n)
    ws = view(full(w), n+1:n+k0.n_ineq)
    wy = dual(w)
    Sigma_s = MadNLP.get_slack_regularization(k0)
    MadNLP.reduce_rhs!(k0, w)
    fill!(k0.buffer, zero(T))
    if k0.n_ineq > 0
        k0.buffer[k0.ind_ineq] .= k0.diag_buffer .* (wy[k0.ind_ineq] .+ ws ./ Sigma_s)
        mul!(wx, k0.jt_csc, k0.buffer, one(T), one(T))
    end
    R = kkt.rhs_all
    @inbounds for k in 1:ns
        for i in 1:nv
            R[i, k] = wx[(k-1)*nv + i]
        end
        for ci in 1:nc_eq
            R[nv+ci, k] = wy[k0.eq_global_indices[(k-1)*nc_eq + ci]]
        end
    end
    @inbounds for i in 1:nd
        k0.rhs_d[i] = wx[ns*nv+i]
    end
    rc = ccall((:mnk_schur_solve, libmadnlp_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Cint),
               kkt.stage.handle, R, k0.rhs_d, MNK_HOST)
    check(rc, SolveException)
    @inbounds for k in 1:ns
        for i in 1:nv
            wx[(k-1)*nv + i] = R[i, k]
        end
        for ci in 1:nc_eq
            wy[k0.eq_global_indices[(k-1)*nc_eq + ci]] = R[nv+ci, k]
        end
    end
    @inbounds for i in 1:nd
        wx[ns*nv+i] = k0.rhs_d[i]
    end
    if k0.n_ineq > 0
        copyto!(k0.wy_eq_buf, view(wy, k0.ind_eq))
        mul!(wy, k0.jt_csc', wx)
        view(wy, k0.ind_eq) .= k0.wy_eq_buf
        @inbounds for idx in 1:length(k0.ind_ineq)
            gi = k0.ind_ineq[idx]
            wy[gi] = k0.diag_buffer[idx] * wy[gi] - k0.buffer[gi]
        end
        ws .= (ws .+ view(wy, k0.ind_ineq)) ./ Sigma_s
    end
    MadNLP.finish_aug_solve!(k0, w)
    return w
end

end # module
