# AdaProxHIP.jl — Julia host shim over libadaprox_hip.so (include/adaprox_hip.h).
#
# UNVERIFIED IN THIS REPOSITORY'S PIPELINE: neither the build container nor the GPU test
# boxes have a Julia runtime, so this file has never been executed.  It is the binding a
# maintainer of pylat/adaptive-proximal-algorithms would add; the same C entry points are
# exercised end-to-end by the Python ctypes mirror (adaptive-proximal-algorithms_amd/api.py)
# and `tests/test_cabi_symbols.py` checks the struct layouts used below against the header.
#
# Usage (drop-in for the three hot-path solvers; everything else stays AdaProx):
#
#     using AdaProx, ProximalOperators
#     include("AdaProxHIP.jl"); using .AdaProxHIP
#     ctx = AdaProxHIP.Context(0)
#     A_d = AdaProxHIP.HipMatrix(ctx, A)                       # one upload; A stays in HBM
#     f   = AdaProxHIP.HipLeastSquares(A_d, b)                  # replaces LinearLeastSquares(A, b)
#     sol, numit = AdaProxHIP.adaptive_proxgrad(zeros(n); f = f, g = NormL1(lam),
#                                               rule = AdaProx.OurRule(gamma = gam_init), tol, maxit, name)
#
# Records are replayed as `@logmsg AdaProx.Record` with the schema of src/AdaProx.jl:351 / :56,
# so experiments/logging.jl and the plotting code work unchanged.
module AdaProxHIP

using Logging
using LinearAlgebra
using SparseArrays
import AdaProx
using ProximalCore: Zero, IndZero
using ProximalOperators: NormL1, NormL2, IndBox, Translate

const LIB = get(ENV, "ADAPROX_HIP_LIB", joinpath(@__DIR__, "..", "adaptive-proximal-algorithms_amd", "csrc", "libadaprox_hip.so"))

# ---- enums (include/adaprox_hip.h) -------------------------------------------------------
const APX_F64, APX_F32 = Int32(0), Int32(1)
const F_ZERO, F_LS, F_LOGISTIC, F_CALLBACK = Int32(0), Int32(1), Int32(2), Int32(3)
const G_ZERO, G_L1, G_BOX, G_BOXV, G_INDZERO, G_L2 = Int32(0), Int32(1), Int32(2), Int32(3), Int32(4), Int32(5)
const RULE_FIXED, RULE_MM, RULE_OURS, RULE_OURS_PLUS = Int32(0), Int32(1), Int32(2), Int32(3)

struct ProxDesc            # apx_prox_desc
    kind::Int32; _pad::Int32
    lambda::Cdouble; lb::Cdouble; ub::Cdouble
    lb_vec::Ptr{Cvoid}; ub_vec::Ptr{Cvoid}; shift::Ptr{Cvoid}
end

struct ProblemDesc         # apx_problem_desc
    f_kind::Int32; dtype::Int32
    n::Int64; m::Int64
    A_f::Ptr{Cvoid}; b::Ptr{Cvoid}; n_total_rows::Int64
    eval_cb::Ptr{Cvoid}; pullback_cb::Ptr{Cvoid}; cb_user::Ptr{Cvoid}
    g::ProxDesc; h::ProxDesc
    A_lin::Ptr{Cvoid}
    allreduce::Ptr{Cvoid}; allreduce_user::Ptr{Cvoid}
    shard_mask::Int32; _pad2::Int32
    comm::Ptr{Cvoid}
end

struct Rule                # apx_rule
    kind::Int32; _pad::Int32
    gamma::Cdouble; t::Cdouble; norm_A::Cdouble; delta::Cdouble; Theta::Cdouble
    xi::Cdouble; nu::Cdouble; r::Cdouble
end

struct Record              # apx_record
    it::Int64; gamma::Cdouble; sigma::Cdouble; norm_res::Cdouble; objective::Cdouble; f_x::Cdouble
    f_evals::Int64; grad_evals::Int64
end

mutable struct Result      # apx_result
    numit::Int64; converged::Int32; flags::Int32
    gamma::Cdouble; sigma::Cdouble; norm_res::Cdouble; objective::Cdouble
    f_evals::Int64; grad_f_evals::Int64; prox_g_evals::Int64; prox_h_evals::Int64; A_evals::Int64; At_evals::Int64
    n_records::Int64
    Result() = new(0, 0, 0, 0.0, 0.0, 0.0, 0.0, 0, 0, 0, 0, 0, 0, 0)
end

struct SolveOpts           # apx_solve_opts
    records::Ptr{Record}; record_cap::Int64; use_graph::Int32; poll_every::Int32
end

function check(code::Integer)
    code == 0 && return
    msg = unsafe_string(ccall((:apx_last_error, LIB), Cstring, ()))
    code == 4 && throw(DomainError(NaN, msg))      # APX_ERR_DOMAIN: sqrt of a negative inside a rule
    error("adaprox_hip error $code: $msg")
end

# ---- handles -----------------------------------------------------------------------------
mutable struct Context
    h::Ptr{Cvoid}
    function Context(device::Integer = 0)
        out = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:apx_ctx_create, LIB), Int32, (Int32, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), device, C_NULL, out))
        c = new(out[])
        finalizer(c -> ccall((:apx_ctx_destroy, LIB), Int32, (Ptr{Cvoid},), c.h), c)
        return c
    end
end

"Execution policy: the two-kernel iteration for small dense problems (default on) and the opt-in sparse-iterate product."
fused_small!(c::Context, on::Bool = true) = check(ccall((:apx_ctx_fused_small, LIB), Int32, (Ptr{Cvoid}, Int32), c.h, on))
"Execution policy for large dense matrices: 0 = the two GEMV passes, 1 = one pass over A per iteration (default), 2 = one pass
 without the team kernels (a GPU shared with other work)."
one_pass!(c::Context, mode::Integer = 1) = check(ccall((:apx_ctx_one_pass, LIB), Int32, (Ptr{Cvoid}, Int32), c.h, mode))
"Options by key (include/adaprox_hip.h APX_OPT_*): A/B switches and test hooks; the library reads no environment variables."
set_option!(c::Context, key::Integer, value::Integer) = check(ccall((:apx_ctx_set_option, LIB), Int32, (Ptr{Cvoid}, Int32, Int64), c.h, key, value))
"What the last failure of the one-pass team kernels on this thread looked like (\"\" if none) — also after a call that recovered
 by itself on the two GEMV passes (bit 8 of `Result.flags`, APX_FLAG_TEAM_FALLBACK)."
last_incident() = unsafe_string(ccall((:apx_last_incident, LIB), Cstring, ()))
"warns when a one-shot call recovered from a failed team hand-off by running again on the two GEMV passes"
warn_team_fallback(res::Result) = (res.flags & 8) != 0 && @warn "adaprox_hip: a hand-off of the one-pass team kernels failed; the call was run again on the two GEMV passes (its result is valid)" incident = last_incident()
one_pass_threshold!(c::Context, min_bytes::Integer) = check(ccall((:apx_ctx_one_pass_threshold, LIB), Int32, (Ptr{Cvoid}, Int64), c.h, min_bytes))
exploit_sparsity!(c::Context, on::Bool = true) = check(ccall((:apx_ctx_exploit_sparsity, LIB), Int32, (Ptr{Cvoid}, Int32), c.h, on))

auto_transpose!(c::Context, on::Bool = true) = check(ccall((:apx_ctx_auto_transpose, LIB), Int32, (Ptr{Cvoid}, Int32), c.h, on))

# ---- RCCL communicators: the row-sharded multi-GPU path (include/adaprox_hip.h, apx_comm_*) --------------
# A = [A_1; ...; A_P] by rows.  One Julia process drives all GPUs of the node:
#
#     ctxs  = [AdaProxHIP.Context(d) for d in 0:P-1]
#     comms = AdaProxHIP.init_all(ctxs)                                   # ncclCommInitAll
#     fs    = [AdaProxHIP.HipLeastSquares(AdaProxHIP.HipMatrix(ctxs[k], A[rows[k], :]), b[rows[k]]; comm = comms[k]) for k in 1:P]
#     sol, numit = AdaProxHIP.adaptive_proxgrad_sharded(zeros(n); f = fs, g = NormL1(lam), rule = AdaProx.OurRule(gamma = gam), tol, maxit)
#
# (keyword arguments do not dispatch in Julia, hence the separate name; it calls apx_adaptive_primal_dual_group: every device's loop on its own host thread
# inside the call, ONE ncclAllReduce of [A_k' r_k ; partial f] per gradient on each loop's stream).  With one process per
# GPU (MPI.jl, Distributed): rank 0 calls `unique_id()`, broadcasts the 128 bytes, every rank `Comm(ctx, nranks, rank, id)`.
const COMM_ID_BYTES = 128
mutable struct Comm
    h::Ptr{Cvoid}
    ctx::Context
    function Comm(h::Ptr{Cvoid}, ctx::Context)
        c = new(h, ctx)
        finalizer(c -> ccall((:apx_comm_destroy, LIB), Int32, (Ptr{Cvoid},), c.h), c)
        return c
    end
end
function unique_id()
    id = Vector{UInt8}(undef, COMM_ID_BYTES)
    check(ccall((:apx_comm_unique_id, LIB), Int32, (Ptr{Cvoid},), id)); id
end
function Comm(ctx::Context, nranks::Integer, rank::Integer, id::Vector{UInt8})
    @assert length(id) == COMM_ID_BYTES
    out = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:apx_comm_init_rank, LIB), Int32, (Ptr{Cvoid}, Int32, Int32, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), ctx.h, nranks, rank, id, out))
    Comm(out[], ctx)
end
function init_all(ctxs::Vector{Context})
    hs = Ptr{Cvoid}[c.h for c in ctxs]
    out = fill(C_NULL, length(ctxs))
    check(ccall((:apx_comm_init_all, LIB), Int32, (Ptr{Ptr{Cvoid}}, Int32, Ptr{Ptr{Cvoid}}), hs, length(ctxs), out))
    [Comm(out[k], ctxs[k]) for k in eachindex(ctxs)]
end
function comm_info(c::Comm)
    n = Ref{Int32}(0); r = Ref{Int32}(0)
    check(ccall((:apx_comm_info, LIB), Int32, (Ptr{Cvoid}, Ref{Int32}, Ref{Int32}), c.h, n, r)); (Int(n[]), Int(r[]))
end
"sum of a Float64 vector over the ranks (set-up code: Lf by power iteration over the shards)"
function allreduce!(c::Comm, v::Vector{Float64})
    check(ccall((:apx_comm_allreduce_host, LIB), Int32, (Ptr{Cvoid}, Ptr{Cdouble}, Int64), c.h, v, length(v))); v
end

dtype_code(::Type{Float64}) = APX_F64
dtype_code(::Type{Float32}) = APX_F32

const APX_LAYOUT_COLS = Int32(0)
const APX_LAYOUT_ROWS = Int32(1)

"""Dense matrix resident in HBM — plays the role of the Julia `Matrix` A (which stays column-major on the host).
`layout = :rows` keeps ONE copy on the device, in the row-contiguous storage the one-pass kernels read (40 GB instead of
80 at 50 000 x 100 000 Float64) ; `:cols` (default) is column-major, and a solver builds the row copy next to it when it needs it."""
mutable struct HipMatrix{T}
    ctx::Context
    h::Ptr{Cvoid}
    m::Int; n::Int
    function HipMatrix(ctx::Context, A::StridedMatrix{T}; layout::Symbol = :cols) where {T<:Union{Float32,Float64}}
        out = Ref{Ptr{Cvoid}}(C_NULL)
        m, n = size(A)
        GC.@preserve A check(ccall((:apx_matrix_upload_layout, LIB), Int32,
            (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Int32, Int32, Ref{Ptr{Cvoid}}),
            ctx.h, pointer(A), m, n, stride(A, 2), dtype_code(T), layout === :rows ? APX_LAYOUT_ROWS : APX_LAYOUT_COLS, out))
        M = new{T}(ctx, out[], m, n)
        finalizer(M -> ccall((:apx_matrix_free, LIB), Int32, (Ptr{Cvoid},), M.h), M)
        return M
    end
    # the `SparseMatrixCSC` the experiments load from LIBSVM files (experiments/libsvm.jl:3-61 ; `f.X * w`, `f.X' * r` at
    # experiments/sparse_logreg/runme.jl:24,31): Julia's colptr / rowval, 0-based, as the CSC layout of apx_matrix_upload_sparse —
    # the library keeps both orientations as compressed rows and picks its kernels from them (csrc/sparse.hip)
    function HipMatrix(ctx::Context, A::SparseMatrixCSC{T,<:Integer}) where {T<:Union{Float32,Float64}}
        out = Ref{Ptr{Cvoid}}(C_NULL)
        m, n = size(A)
        colptr = Int64.(A.colptr .- 1)
        rowval = Int32.(A.rowval .- 1)
        nz = A.nzval
        GC.@preserve colptr rowval nz check(ccall((:apx_matrix_upload_sparse, LIB), Int32,
            (Ptr{Cvoid}, Int64, Int64, Int64, Int32, Ptr{Int64}, Ptr{Int32}, Ptr{Cvoid}, Int32, Ref{Ptr{Cvoid}}),
            ctx.h, m, n, length(nz), Int32(1), colptr, rowval, nz, dtype_code(T), out))
        M = new{T}(ctx, out[], m, n)
        finalizer(M -> ccall((:apx_matrix_free, LIB), Int32, (Ptr{Cvoid},), M.h), M)
        return M
    end
end
"keeps only the row-contiguous storage of a matrix uploaded column-major (one resident copy from here on)"
drop_primary!(A::HipMatrix) = (check(ccall((:apx_matrix_drop_primary, LIB), Int32, (Ptr{Cvoid},), A.h)); A)
Base.size(A::HipMatrix) = (A.m, A.n)
Base.eltype(::HipMatrix{T}) where {T} = T

# `A*x`, `A'*y`, `norm(A)` (src/counting.jl:65-82)
function Base.:*(A::HipMatrix{T}, x::Vector{T}) where {T}
    y = Vector{T}(undef, A.m)
    check(ccall((:apx_gemv_n, LIB), Int32, (Ptr{Cvoid}, Ptr{T}, Ptr{T}), A.h, x, y)); y
end
struct HipAdjoint{T}; A::HipMatrix{T}; end
LinearAlgebra.adjoint(A::HipMatrix) = HipAdjoint(A)
function Base.:*(At::HipAdjoint{T}, r::Vector{T}) where {T}
    g = Vector{T}(undef, At.A.n)
    check(ccall((:apx_gemv_t, LIB), Int32, (Ptr{Cvoid}, Ptr{T}, Ptr{T}), At.A.h, r, g)); g
end
function LinearAlgebra.norm(A::HipMatrix)          # Frobenius, like norm(::Matrix)
    s = Ref{Cdouble}(0)
    check(ccall((:apx_matrix_sumsq, LIB), Int32, (Ptr{Cvoid}, Ref{Cdouble}), A.h, s)); sqrt(s[])
end

# ---- smooth terms --------------------------------------------------------------------------
"LinearLeastSquares of experiments/lasso/runme.jl:16-27 with A resident on the GPU."
struct HipLeastSquares{T}; A::HipMatrix{T}; b::Vector{T}; comm::Union{Nothing,Comm}; end
HipLeastSquares(A::HipMatrix{T}, b::Vector{T}; comm = nothing) where {T} = HipLeastSquares{T}(A, b, comm)
"LogisticLoss of experiments/sparse_logreg/runme.jl:18-39 (bias = last coordinate).  Row-sharded: n_total_rows = the N of the mean."
struct HipLogistic{T}; X::HipMatrix{T}; y::Vector{T}; comm::Union{Nothing,Comm}; n_total_rows::Int; end
HipLogistic(X::HipMatrix{T}, y::Vector{T}; comm = nothing, n_total_rows = 0) where {T} = HipLogistic{T}(X, y, comm, n_total_rows)

f_fields(f::HipLeastSquares) = (F_LS, f.A, f.b, f.A.n)
f_fields(f::HipLogistic) = (F_LOGISTIC, f.X, f.y, f.X.n + 1)

# ---- prox terms -> descriptors (keep the arrays alive in `keep`) -----------------------------
proxdesc(::Zero, T, keep) = ProxDesc(G_ZERO, 0, 0.0, -Inf, Inf, C_NULL, C_NULL, C_NULL)
proxdesc(::IndZero, T, keep) = ProxDesc(G_INDZERO, 0, 0.0, -Inf, Inf, C_NULL, C_NULL, C_NULL)
proxdesc(g::NormL1{<:Real}, T, keep) = ProxDesc(G_L1, 0, Float64(g.lambda), -Inf, Inf, C_NULL, C_NULL, C_NULL)
proxdesc(g::NormL2, T, keep) = ProxDesc(G_L2, 0, Float64(g.lambda), -Inf, Inf, C_NULL, C_NULL, C_NULL)
function proxdesc(g::IndBox, T, keep)
    if g.lb isa Real && g.ub isa Real
        return ProxDesc(G_BOX, 0, 0.0, Float64(g.lb), Float64(g.ub), C_NULL, C_NULL, C_NULL)
    end
    lb, ub = collect(T, g.lb), collect(T, g.ub); push!(keep, lb, ub)
    ProxDesc(G_BOXV, 0, 0.0, -Inf, Inf, pointer(lb), pointer(ub), C_NULL)
end
function proxdesc(g::Translate, T, keep)
    d = proxdesc(g.f, T, keep)
    b = collect(T, g.b); push!(keep, b)
    ProxDesc(d.kind, 0, d.lambda, d.lb, d.ub, d.lb_vec, d.ub_vec, pointer(b))
end

# ---- rules ---------------------------------------------------------------------------------
ruledesc(r::AdaProx.FixedStepsize) = Rule(RULE_FIXED, 0, r.gamma, r.t, 0, 0, 0, 0, 0, 0)
ruledesc(r::AdaProx.MalitskyMishchenkoRule) = Rule(RULE_MM, 0, r.gamma, r.t, 0, 0, 0, 0, 0, 0)
ruledesc(r::AdaProx.OurRule) = Rule(RULE_OURS, 0, r.gamma, r.t, r.norm_A, r.delta, r.Theta, 0, 0, 0)
ruledesc(r::AdaProx.OurRulePlus) = Rule(RULE_OURS_PLUS, 0, r.gamma, 1, 0, 0, 0, r.xi, r.nu, r.r)

unwrap(c::AdaProx.Counting) = c.f
unwrap(x) = x
cnt(c::AdaProx.Counting, f::Symbol, inc) = getfield(c, f) + inc
cnt(_, ::Symbol, _) = nothing

function with_problem(body, f, g, h, A, ::Type{T}) where {T}
    fu, gu, hu, Au = unwrap(f), unwrap(g), unwrap(h), unwrap(A)
    keep = Any[]
    cm = fu isa Union{HipLeastSquares,HipLogistic} ? fu.comm : nothing
    ntot = fu isa HipLogistic ? fu.n_total_rows : 0
    A_lin = Au isa HipMatrix ? Au.h : C_NULL
    m = Au isa HipMatrix ? Au.m : 0
    if fu isa Zero                       # experiments/least_absolute_deviation/runme.jl:18-21
        fk, Af_h, b, n, ctx = F_ZERO, C_NULL, T[], Au.n, Au.ctx
    else
        fk, Af, b, n = f_fields(fu)
        Af_h, ctx = Af.h, Af.ctx
    end
    desc = ProblemDesc(fk, dtype_code(T), n, m, Af_h, isempty(b) ? C_NULL : pointer(b), ntot, C_NULL, C_NULL, C_NULL,
                       proxdesc(gu, T, keep), proxdesc(hu, T, keep), A_lin, C_NULL, C_NULL, cm === nothing ? 0 : 1, 0,
                       cm === nothing ? C_NULL : cm.h)
    out = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve b keep cm begin
        check(ccall((:apx_problem_create, LIB), Int32, (Ptr{Cvoid}, Ref{ProblemDesc}, Ref{Ptr{Cvoid}}), ctx.h, Ref(desc), out))
        try
            return body(out[])
        finally
            ccall((:apx_problem_destroy, LIB), Int32, (Ptr{Cvoid},), out[])
        end
    end
end

function bump!(f, g, h, A, res::Result)
    warn_team_fallback(res)      # (every one-shot wrapper comes through here with its apx_result)
    f isa AdaProx.Counting && (f.eval_count += res.f_evals; f.grad_count += res.grad_f_evals)
    g isa AdaProx.Counting && (g.prox_count += res.prox_g_evals)
    h isa AdaProx.Counting && res.prox_h_evals >= 0 && (h.prox_count += res.prox_h_evals)
    A isa AdaProx.Counting && res.A_evals >= 0 && (A.mul_count += res.A_evals; A.amul_count += res.At_evals)
end

"adaptive_primal_dual — src/AdaProx.jl:312-364"
function adaptive_primal_dual(x::Vector{T}, y::Vector{T}; f, g, h, A, rule, tol = 1e-5, maxit = 10_000, name = "AdaPDM") where {T}
    maxit = floor(Int, maxit)
    with_problem(f, g, h, A, T) do prob
        recs = Vector{Record}(undef, max(maxit, 1))
        opts = SolveOpts(pointer(recs), maxit, 0, 0)
        res = Result(); x_out = similar(x); y_out = similar(y)
        lin = unwrap(A) isa HipMatrix
        GC.@preserve recs check(ccall((:apx_adaptive_primal_dual, LIB), Int32,
            (Ptr{Cvoid}, Ref{Rule}, Ptr{T}, Ptr{T}, Cdouble, Int64, Ref{SolveOpts}, Ptr{T}, Ptr{T}, Ref{Result}),
            prob, Ref(ruledesc(rule)), x, lin ? y : C_NULL, tol, maxit, Ref(opts), x_out, lin ? y_out : C_NULL, res))
        for i in 1:res.n_records                      # replay `@logmsg Record` (:351)
            r = recs[i]; k = r.it
            @logmsg AdaProx.Record "" method=name it=k gamma=r.gamma sigma=r.sigma norm_res=r.norm_res objective=r.objective grad_f_evals=cnt(f, :grad_count, k + 1) prox_g_evals=cnt(g, :prox_count, k) prox_h_evals=cnt(h, :prox_count, k) A_evals=cnt(A, :mul_count, k + 1) At_evals=cnt(A, :amul_count, k) f_evals=cnt(f, :eval_count, k + 1)
        end
        bump!(f, g, h, A, res)
        return x_out, (lin ? y_out : zero(x_out)), Int(res.numit)
    end
end

# several problems at once (one per device), innermost first: body receives the vector of handles
with_problems(body, fs, g, ::Type{T}, acc = Ptr{Cvoid}[]) where {T} =
    length(acc) == length(fs) ? body(acc) :
    with_problem(p -> with_problems(body, fs, g, T, vcat(acc, p)), fs[length(acc) + 1], g, Zero(), 0, T)

"adaptive_proxgrad on row shards held by ONE process: `f` is a vector of HipLeastSquares / HipLogistic, one per device, each with its `comm`"
function adaptive_proxgrad_sharded(x::Vector{T}; f::AbstractVector, g, rule, tol = 1e-5, maxit = 100_000, name = "AdaPGM") where {T}
    maxit = floor(Int, maxit)
    with_problems(f, g, T) do probs
        recs = Vector{Record}(undef, max(maxit, 1))
        opts = SolveOpts(pointer(recs), maxit, 0, 0)
        res = Result(); x_out = similar(x)
        GC.@preserve recs probs check(ccall((:apx_adaptive_primal_dual_group, LIB), Int32,
            (Int32, Ptr{Ptr{Cvoid}}, Ref{Rule}, Ptr{T}, Ptr{Ptr{Cvoid}}, Cdouble, Int64, Ref{SolveOpts}, Ptr{T}, Ptr{Ptr{Cvoid}}, Ref{Result}),
            length(probs), probs, Ref(ruledesc(rule)), x, C_NULL, tol, maxit, Ref(opts), x_out, C_NULL, res))
        for i in 1:res.n_records                      # replay `@logmsg Record` (:351)
            r = recs[i]
            @logmsg AdaProx.Record "" method=name it=r.it gamma=r.gamma sigma=r.sigma norm_res=r.norm_res objective=r.objective grad_f_evals=nothing prox_g_evals=cnt(g, :prox_count, r.it) prox_h_evals=nothing A_evals=nothing At_evals=nothing f_evals=nothing
        end
        g isa AdaProx.Counting && (g.prox_count += res.prox_g_evals)
        return x_out, Int(res.numit)
    end
end

"adaptive_proxgrad — src/AdaProx.jl:418-421"
function adaptive_proxgrad(x; f, g, rule, tol = 1e-5, maxit = 100_000, name = "AdaPGM")
    x, _, numit = adaptive_primal_dual(x, zero(x); f, g, h = Zero(), A = 0, rule, tol, maxit, name)
    return x, numit
end

"fixed_proxgrad — src/AdaProx.jl:457-459"
fixed_proxgrad(x; f, g, gamma, tol = 1e-5, maxit = 100_000, name = "Fixed stepsize PGM") =
    adaptive_proxgrad(x; f, g, rule = AdaProx.FixedStepsize(gamma, one(gamma)), tol, maxit, name)

"backtracking_proxgrad — src/AdaProx.jl:50-64"
function backtracking_proxgrad(x0::Vector{T}; f, g, gamma0, xi = 1.0, shrink = 0.5, tol = 1e-5, maxit = 100_000, name = "Backtracking PG") where {T}
    maxit = floor(Int, maxit)
    with_problem(f, g, Zero(), 0, T) do prob
        recs = Vector{Record}(undef, max(maxit, 1))
        opts = SolveOpts(pointer(recs), maxit, 0, 0)
        res = Result(); z = similar(x0)
        GC.@preserve recs check(ccall((:apx_backtracking_proxgrad, LIB), Int32,
            (Ptr{Cvoid}, Ptr{T}, Cdouble, Cdouble, Cdouble, Cdouble, Int64, Ref{SolveOpts}, Ptr{T}, Ref{Result}),
            prob, x0, gamma0, xi, shrink, tol, maxit, Ref(opts), z, res))
        (res.flags & 1) != 0 && @error "step size became too small ($(res.gamma))"     # :40-42
        for i in 1:res.n_records                      # replay `@logmsg Record` (:56)
            r = recs[i]
            @logmsg AdaProx.Record "" method=name it=r.it gamma=r.gamma norm_res=r.norm_res objective=r.objective grad_f_evals=cnt(f, :grad_count, r.grad_evals) prox_g_evals=cnt(g, :prox_count, r.f_evals - 1) f_evals=cnt(f, :eval_count, r.f_evals)
        end
        bump!(f, g, nothing, nothing, res)
        return z, Int(res.numit)
    end
end

struct LinesearchParams    # apx_linesearch_params
    gamma::Cdouble; eta::Cdouble; t::Cdouble; delta::Cdouble; Theta::Cdouble; r::Cdouble; R::Cdouble
    max_linesearch::Int64
end

# shared tail of the primal-dual style entry points: replay records with the :351 schema
function replay_pd(name, recs, res, f, g, h, A, at_of, proxh_of)
    for i in 1:res.n_records
        r = recs[i]; k = r.it
        @logmsg AdaProx.Record "" method=name it=k gamma=r.gamma sigma=r.sigma norm_res=r.norm_res objective=r.objective grad_f_evals=cnt(f, :grad_count, k + 1) prox_g_evals=cnt(g, :prox_count, k) prox_h_evals=cnt(h, :prox_count, proxh_of(r)) A_evals=cnt(A, :mul_count, k + 1) At_evals=cnt(A, :amul_count, at_of(r)) f_evals=cnt(f, :eval_count, k + 1)
    end
end

"adaptive_linesearch_primal_dual (AdaPDM+) — src/AdaProx.jl:463-550"
function adaptive_linesearch_primal_dual(x::Vector{T}, y::Vector{T}; f, g, h, A, gamma = nothing, eta = 1.0, t = 1.0,
        delta = 1e-8, Theta = 1.2, r = 2, R = 0.95, tol = 1e-5, maxit = 10_000, name = "AdaPDM+") where {T}
    @assert eta > 0 "eta must be positive"
    @assert Theta > (delta + 1) "must be Theta > (delta + 1)"
    maxit = floor(Int, maxit)
    with_problem(f, g, h, A, T) do prob
        recs = Vector{Record}(undef, max(maxit, 1))
        opts = SolveOpts(pointer(recs), maxit, 0, 0)
        prm = LinesearchParams(gamma === nothing ? 0.0 : gamma, eta, t, delta, Theta, r, R, 0)
        res = Result(); x_out = similar(x); y_out = similar(y)
        GC.@preserve recs check(ccall((:apx_adaptive_linesearch_primal_dual, LIB), Int32,
            (Ptr{Cvoid}, Ref{LinesearchParams}, Ptr{T}, Ptr{T}, Cdouble, Int64, Ref{SolveOpts}, Ptr{T}, Ptr{T}, Ref{Result}),
            prob, Ref(prm), x, y, tol, maxit, Ref(opts), x_out, y_out, res))
        replay_pd(name, recs, res, f, g, h, A, r -> r.grad_evals, r -> r.grad_evals - 1)
        bump!(f, g, h, A, res)
        return x_out, y_out, Int(res.numit)
    end
end

"malitsky_pock — src/AdaProx.jl:581-629"
function malitsky_pock(x::Vector{T}, y::Vector{T}; f, g, h, A, sigma, t = 1.0, tol = 1e-5, maxit = 10_000, name = "MP-ls") where {T}
    maxit = floor(Int, maxit)
    with_problem(f, g, h, A, T) do prob
        recs = Vector{Record}(undef, max(maxit, 1))
        opts = SolveOpts(pointer(recs), maxit, 0, 0)
        res = Result(); x_out = similar(x); y_out = similar(y)
        GC.@preserve recs check(ccall((:apx_malitsky_pock, LIB), Int32,
            (Ptr{Cvoid}, Ptr{T}, Ptr{T}, Cdouble, Cdouble, Cdouble, Int64, Ref{SolveOpts}, Ptr{T}, Ptr{T}, Ref{Result}),
            prob, x, y, sigma, t, tol, maxit, Ref(opts), x_out, y_out, res))
        for i in 1:res.n_records                   # record schema :621; trials = f_evals - it
            r = recs[i]; k = r.it; tr = r.f_evals - k
            @logmsg AdaProx.Record "" method=name it=k gamma=r.gamma sigma=r.sigma norm_res=r.norm_res objective=r.objective grad_f_evals=cnt(f, :grad_count, 2k) prox_g_evals=cnt(g, :prox_count, tr) prox_h_evals=cnt(h, :prox_count, k) A_evals=cnt(A, :mul_count, 1 + tr) At_evals=cnt(A, :amul_count, 1 + k) f_evals=cnt(f, :eval_count, r.f_evals)
        end
        bump!(f, g, h, A, res)
        return x_out, y_out, Int(res.numit)
    end
end

# proximal-gradient style baselines share one driver: `call` performs the ccall
function pg_baseline(call, name, x0::Vector{T}, f, g, maxit, counts) where {T}
    maxit = floor(Int, maxit)
    with_problem(f, g, Zero(), 0, T) do prob
        recs = Vector{Record}(undef, max(maxit, 1))
        opts = SolveOpts(pointer(recs), maxit, 0, 0)
        res = Result(); x_out = similar(x0)
        GC.@preserve recs check(call(prob, Ref(opts), x_out, res))
        for i in 1:res.n_records                   # record schema :74 / :135 / :184
            r = recs[i]; (ge, pe, fe) = counts(r)
            @logmsg AdaProx.Record "" method=name it=r.it gamma=r.gamma norm_res=r.norm_res objective=r.objective grad_f_evals=cnt(f, :grad_count, ge) prox_g_evals=cnt(g, :prox_count, pe) f_evals=cnt(f, :eval_count, fe)
        end
        bump!(f, g, nothing, nothing, res)
        return x_out, Int(res.numit)
    end
end

"backtracking_nesterov — src/AdaProx.jl:66-84"
backtracking_nesterov(x0::Vector{T}; f, g, gamma0, shrink = 0.5, tol = 1e-5, maxit = 100_000, name = "Backtracking Nesterov") where {T} =
    pg_baseline(name, x0, f, g, maxit, r -> (r.grad_evals, r.f_evals - r.it, r.f_evals)) do prob, opts, x_out, res
        ccall((:apx_backtracking_nesterov, LIB), Int32,
              (Ptr{Cvoid}, Ptr{T}, Cdouble, Cdouble, Cdouble, Int64, Ref{SolveOpts}, Ptr{T}, Ref{Result}),
              prob, x0, gamma0, shrink, tol, floor(Int, maxit), opts, x_out, res)
    end

"fixed_nesterov — src/AdaProx.jl:91-142"
function fixed_nesterov(x0::Vector{T}; f, g, Lf = nothing, muf = 0, mug = 0, gamma = nothing, theta = nothing, tol = 1e-5,
                        maxit = 100_000, name = "Fixed Nesterov") where {T}
    @assert (gamma === nothing) != (Lf === nothing)
    gamma === nothing && (gamma = 1 / Lf)
    pg_baseline(name, x0, f, g, maxit, r -> (r.it, r.it, r.it)) do prob, opts, x_out, res
        ccall((:apx_fixed_nesterov, LIB), Int32,
              (Ptr{Cvoid}, Ptr{T}, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Int64, Ref{SolveOpts}, Ptr{T}, Ref{Result}),
              prob, x0, gamma, muf, mug, theta === nothing ? -1.0 : theta, tol, floor(Int, maxit), opts, x_out, res)
    end
end

"agraal — src/AdaProx.jl:150-192"
function agraal(x1::Vector{T}; f, g, x0 = nothing, gamma0 = nothing, gamma_max = 1e6, phi = 1.5, tol = 1e-5,
                maxit = 100_000, name = "aGRAAL") where {T}
    x0 === nothing && (x0 = x1 + randn(size(x1)))
    pg_baseline(name, x1, f, g, maxit, r -> (r.it + 1, r.it, r.it + 1)) do prob, opts, x_out, res
        ccall((:apx_agraal, LIB), Int32,
              (Ptr{Cvoid}, Ptr{T}, Ptr{T}, Cdouble, Cdouble, Cdouble, Cdouble, Int64, Ref{SolveOpts}, Ptr{T}, Ref{Result}),
              prob, x1, x0, gamma0 === nothing ? 0.0 : gamma0, gamma_max, phi, tol, floor(Int, maxit), opts, x_out, res)
    end
end

end # module
