# MadNLPHIP.jl -- Julia glue binding libmadnlp_hip.so into MadNLP (NOT executed in this
# repository's CI: the build image has no Julia toolchain; the same call sequence is
# exercised through the ctypes mirror in madnlp.jl_amd/ and tests/).
#
# It adds two types that plug into MadNLP's own option seam
#     madnlp(nlp; kkt_system = HipSparseCondensedKKTSystem, linear_solver = HipLinearSolver)
# (reference src/IPM/options.jl:121-122, consumed at src/IPM/IPM.jl:157-165); nothing else in
# the IPM loop changes.  Pattern: ccall + finalizer exactly as src/LinearSolvers/mumps.jl:148-165,211
# and src/LinearSolvers/lapack.jl:50-139.
module MadNLPHIP

import MadNLP
import MadNLP: AbstractLinearSolver, AbstractCondensedKKTSystem, MadNLPLogger, LinearFactorization,
    SymbolicException, FactorizationException, SolveException, InertiaException,
    BUNCHKAUFMAN, CHOLESKY, LDL
import SparseArrays: SparseMatrixCSC

const libmadnlp_hip = get(ENV, "MADNLP_HIP_LIB", "libmadnlp_hip.so")
const MNK_HOST = Cint(0)
const MNK_ALGO = Dict(BUNCHKAUFMAN => Cint(1), CHOLESKY => Cint(4), LDL => Cint(5))

lasterr() = unsafe_string(ccall((:mnk_last_error_string, libmadnlp_hip), Cstring, ()))

# ------------------------------------------------------------------ context (one per solver)
mutable struct HipContext
    handle::Ptr{Cvoid}
    function HipContext(device::Integer = 0)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:mnk_ctx_create, libmadnlp_hip), Cint, (Cint, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), device, C_NULL, h)
        rc == 0 || throw(SymbolicException())
        ctx = new(h[])
        finalizer(c -> ccall((:mnk_ctx_destroy, libmadnlp_hip), Cint, (Ptr{Cvoid},), c.handle), ctx)
        return ctx
    end
end

# ------------------------------------------------------------------ linear solver
@kwdef mutable struct HipSolverOptions <: MadNLP.AbstractOptions
    lapack_algorithm::LinearFactorization = BUNCHKAUFMAN   # served by the static-pivot LDL^T
    pivot_tol::Float64 = 0.0
    outer_block::Int = 256
    lookahead::Bool = true
end

"""
    HipLinearSolver(A; opt, logger)

`A` is either the `HipAugCSC` handle owned by a `HipSparseCondensedKKTSystem` (device
resident; nothing crosses PCIe in `factorize!`), or a host `Matrix{Float64}` /
`SparseMatrixCSC{Float64,Int32}` (lower triangle), to which the solver keeps a reference
exactly like `LapackCPUSolver` does (src/LinearSolvers/lapack.jl:5-44).
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

struct HipAugCSC{T} <: AbstractMatrix{T}      # aug_com of the sparse condensed system, lives in HBM
    sc::Ptr{Cvoid}                             # mnk_sc*
    ctx::HipContext
    structure::SparseMatrixCSC{T, Int32}       # host copy of the pattern (values are on the device)
end
Base.size(A::HipAugCSC) = size(A.structure)

function HipLinearSolver(A::MT; opt = HipSolverOptions(), logger = MadNLPLogger(),
                         ctx = (A isa HipAugCSC ? A.ctx : HipContext())) where {MT <: AbstractMatrix}
    n = size(A, 1)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:mnk_ls_create, libmadnlp_hip), Cint, (Ptr{Cvoid}, Int64, Cint, Ptr{Ptr{Cvoid}}),
               ctx.handle, n, MNK_ALGO[opt.lapack_algorithm], h)
    rc == 0 || throw(SymbolicException())
    for (k, v) in (("pivot_tol", opt.pivot_tol), ("outer_block", Float64(opt.outer_block)), ("lookahead", Float64(opt.lookahead)))
        ccall((:mnk_ls_set_option, libmadnlp_hip), Cint, (Ptr{Cvoid}, Cstring, Cdouble), h[], k, v)
    end
    M = HipLinearSolver{Float64, MT}(A, h[], ctx, n, Ref{Cint}(0), opt, logger)
    finalizer(m -> ccall((:mnk_ls_destroy, libmadnlp_hip), Cint, (Ptr{Cvoid},), m.handle), M)
    return M
end

# factorize!: transfer_matrix! + dsytrf/dpotrf replacement (src/LinearSolvers/lapack_common.jl:54-66)
function MadNLP.factorize!(M::HipLinearSolver{T, <:HipAugCSC}) where T
    rc = ccall((:mnk_ls_factorize_sc, libmadnlp_hip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cint}), M.handle, M.A.sc, M.info)
    rc == 0 || throw(FactorizationException())      # HIP/runtime error only; a bad pivot is NOT an error
    return M
end
function MadNLP.factorize!(M::HipLinearSolver{T, <:Matrix}) where T
    rc = ccall((:mnk_ls_factorize_dense, libmadnlp_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Int64, Cint, Ptr{Cint}),
               M.handle, M.A, size(M.A, 1), MNK_HOST, M.info)
    rc == 0 || throw(FactorizationException())
    return M
end
function MadNLP.factorize!(M::HipLinearSolver{T, <:SparseMatrixCSC}) where T
    A = M.A
    rc = ccall((:mnk_ls_factorize_csc, libmadnlp_hip), Cint,
               (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}, Ptr{Cdouble}, Cint, Ptr{Cint}),
               M.handle, A.colptr, A.rowval, A.nzval, 1, M.info)
    rc == 0 || throw(FactorizationException())
    return M
end

function MadNLP.solve_linear_system!(M::HipLinearSolver, x::Vector{Float64})
    rc = ccall((:mnk_ls_solve, libmadnlp_hip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Int64, Int64, Cint),
               M.handle, x, 1, length(x), MNK_HOST)
    rc == 0 || throw(SolveException())
    return x
end

MadNLP.is_inertia(::HipLinearSolver) = true
function MadNLP.inertia(M::HipLinearSolver)
    p = Ref{Int64}(0); z = Ref{Int64}(0); n = Ref{Int64}(0)
    rc = ccall((:mnk_ls_inertia, libmadnlp_hip), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}), M.handle, p, z, n)
    rc == 0 || throw(InertiaException())
    return (p[], z[], n[])
end
MadNLP.improve!(::HipLinearSolver) = false
MadNLP.introduce(M::HipLinearSolver) = "HIP-MI355X ($(M.opt.lapack_algorithm))"
MadNLP.input_type(::Type{<:HipLinearSolver}) = :dense
MadNLP.default_options(::Type{<:HipLinearSolver}) = HipSolverOptions()
MadNLP.is_supported(::Type{<:HipLinearSolver}, ::Type{Float64}) = true
MadNLP.is_supported(::Type{<:HipLinearSolver}, ::Type{Float32}) = false

# ------------------------------------------------------------------ sparse condensed KKT system
# The struct keeps every field the generic code touches (src/KKT/KKTsystem.jl:210-234,
# src/IPM/kernels.jl:4-27, src/KKT/rhs.jl:119-129) on the host; only the assembly
# (compress_*, build_kkt!) and the factorization/solve go to the device.  `jt_csc` and
# `hess_com` are host mirrors rebuilt from the structures the library exports
# (mnk_sc_get_structure / mnk_sc_get_map), so jtprod!/mul!/solve_kkt! of
# src/KKT/Sparse/condensed.jl:150-156 and src/IPM/factorization.jl:143-167,278-299 run unchanged.
#
# create_kkt_system(::Type{HipSparseCondensedKKTSystem}, cb, linear_solver; ...) mirrors
# src/KKT/Sparse/condensed.jl:55-133 with these substitutions:
#
#   coo_to_csc(jt_coo), coo_to_csc(hess_raw), build_condensed_aug_symbolic(...)
#       -> ccall(:mnk_sc_create, ..., n, m, nnzj, jac_I, jac_J, nnzh, hess_I, hess_J, 1, sc)
#          + mnk_sc_sizes / mnk_sc_get_structure / mnk_sc_get_map (1-based conversion on the Julia side)
#   compress_jacobian!(kkt) -> ccall(:mnk_sc_compress_jacobian, ..., kkt.sc, kkt.jac, MNK_HOST)
#                              (+ transfer!(kkt.jt_csc, kkt.jt_coo, kkt.jt_csc_map) on the host mirror)
#   compress_hessian!(kkt)  -> ccall(:mnk_sc_compress_hessian, ..., kkt.sc, kkt.hess, MNK_HOST)
#   build_kkt!(kkt)         -> kkt.diag_buffer .= Ss ./ (1 .- Sd .* Ss)        (host copy for solve_kkt!)
#                              ccall(:mnk_sc_build, ..., kkt.sc, kkt.pr_diag, kkt.du_diag, MNK_HOST)
#   kkt.aug_com             -> HipAugCSC(sc, ctx, pattern)   (what `linear_solver(aug_com; opt)` receives)
#
#   solve_kkt!(kkt, w)      -> optional device version (keeps [x; s; z; zl; zu] on the device, one round trip):
#                              constructor: ccall(:mnk_sc_set_bounds, ..., kkt.sc, nlb, ind_lb, nub, ind_ub, 1)
#                              build_kkt!:  ccall(:mnk_sc_set_barrier_terms, ..., kkt.sc, kkt.reg, kkt.l_diag, kkt.u_diag,
#                                                 kkt.l_lower, kkt.u_lower, MNK_HOST)
#                              solve_kkt!:  ccall(:mnk_sc_solve_kkt, ..., kkt.sc, kkt.linear_solver.handle, full(w), MNK_HOST)
#   mul!(w, kkt, x, a, b)   -> ccall(:mnk_sc_mul, ..., kkt.sc, full(w), full(x), a, b, MNK_HOST)
#
# See INTEGRATION.md for the full listing and the dense (DenseCondensedKKTSystem) twin.

end # module
