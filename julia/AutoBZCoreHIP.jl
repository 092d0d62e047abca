# AutoBZCoreHIP.jl -- reference-side binding of libabzhip.so (include/abzhip.h).
#
# UNTESTED SOURCE: there is no Julia toolchain in the build pipeline (SURVEY.md section 8c), so this
# file shows the shim a maintainer of AutoBZCore.jl v0.3.8 would add; the same C ABI is exercised
# by the Python ctypes mirror in autobzcore.jl_amd/ and by tests/.
#
# What it plugs into (file:line in lxvm/AutoBZCore.jl v0.3.8):
#   * the dispatch pair init_cacheval / do_solve          src/interfaces.jl:59-62,116-118
#   * FourierIntegrand / FourierValue                     src/fourier.jl:22-58,111-118
#   * PTR / AutoPTR / IAI on a SymmetricBZ                src/brillouin.jl:337-444, src/fourier.jl:338-389,488-510
#   * BatchIntegrand f!(y, x, p)                          src/batch.jl:1-38
#   * batchsolve                                          src/interfaces.jl:199-243
#   * DOSProblem + GGR                                    src/dos_ggr.jl:1-104
module AutoBZCoreHIP

using AutoBZCore
using AutoBZCore: FourierIntegrand, FourierValue, ParameterIntegrand, MixedParameters, SymmetricBZ,
    IntegralSolution, IntegralSolver, nsyms, PTR, AutoPTR, IAI, CubicLimits, TetrahedralLimits
using FourierSeriesEvaluators: FourierSeries, period
using LinearAlgebra, StaticArrays

const libabz = "libabzhip"   # autobzcore.jl_amd/libabzhip.so on the loader path

# ---------------------------------------------------------------- constants of abzhip.h
const WANT_H, WANT_EIG, WANT_VEL = Cint(1), Cint(2), Cint(4)
# rules of a Hermitian series (n <= 4) keep H(k) as its upper triangle: n^2 value planes instead of 2 n^2; ignored otherwise.
# Device integrands read those planes, abz_rule_export still returns full matrices (abzhip.h).
const WANT_H_COMPACT = Cint(8)
const WANT_H_ROW_MAJOR = Cint(16)  # abz_eval_nodes: row-major matrices (not for Julia arrays; listed for completeness)
const WANT_HC = WANT_H | WANT_H_COMPACT
const F_ONE, F_LINEAR, F_LINEAR_X, F_DOS, F_TRGLOC, F_GLOC, F_DOS_EIG = Cint.(0:6)
const LIMS_CUBIC, LIMS_TETRAHEDRAL, LIMS_POLYHEDRAL, LIMS_POLYGON = Cint(0), Cint(1), Cint(2), Cint(3)

# (kind, lim_a, lim_b) of the reference's limits types for abz_iai_solve(_many); the SymmetryReduceBZ
# extension's Polyhedron3 travels as packed faces [nv, x y z ...] with lim_b = [length(lim_a)]
function pack_limits(l)
    l isa CubicLimits && return (LIMS_CUBIC, Float64[l.a...], Float64[l.b...])
    hasproperty(l, :face_coord) || return (LIMS_TETRAHEDRAL, Float64[l.a...], Float64[])
    a = Float64[]
    for f in l.face_coord            # nv x 3 matrices, vertices in order around the face
        push!(a, size(f, 1)); append!(a, vec(permutedims(f)))
    end
    return (LIMS_POLYHEDRAL, a, Float64[length(a)])
end

function check(rc::Cint)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:abz_last_error, libabz), Cstring, ()))
    rc == -1 && throw(ArgumentError(msg))        # ABZ_ERR_ARG <-> the reference's ArgumentError sites
    rc == -5 && throw(OutOfMemoryError())          # ABZ_ERR_NOMEM (device allocation, or std::bad_alloc caught at the boundary)
    error("libabzhip error $rc: $msg")             # -2 HIP, -3 no GPU, -4 unsupported, -6 a C++ exception caught at the boundary
end

# ---------------------------------------------------------------- handles
# Lifetimes: the library reference-counts context <- series <- rule, so these finalizers may run in any order
# (abz_*_destroy only marks a handle closed while dependants still use the object).
mutable struct HIPContext
    h::Ptr{Cvoid}
    HIPContext(h::Ptr{Cvoid}, ::Val{:adopt}) = new(h)
    function HIPContext(device::Integer=0)
        ref = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:abz_ctx_create, libabz), Cint, (Cint, Ptr{Ptr{Cvoid}}), device, ref))
        ctx = new(ref[])
        finalizer(c -> ccall((:abz_ctx_destroy, libabz), Cint, (Ptr{Cvoid},), c.h), ctx)
    end
end
const CTX = Ref{HIPContext}()           # one context per Julia thread in a threaded build
context() = isassigned(CTX) ? CTX[] : (CTX[] = HIPContext())

"Device-resident coefficients: replaces workspace_allocate_vec (src/fourier.jl:61-86)."
mutable struct HIPSeries{N}
    h::Ptr{Cvoid}
    n::Int
    rules::Dict{Tuple{Int,UInt,Cint},Any}
    exchange::Any                             # the @cfunction of shard_iai! (kept alive with the series)
end
function HIPSeries(s::FourierSeries{S,N}) where {S,N}
    c = s.c                                   # Array{SMatrix{n,n,ComplexF64}} or Array{<:Number}
    n = eltype(c) <: Number ? 1 : size(eltype(c), 1)
    coef = reinterpret(Float64, vec(c))       # Julia memory order == the ABI's order
    dims = Cint[size(c)...]
    first = Cint[(1 .+ s.o)...]               # frequency of c[1]: offset + 1 (OffsetArray axes fold into s.o)
    per = Float64[period(s)...]
    ref = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve coef dims first per begin
        check(ccall((:abz_series_create, libabz), Cint,
            (Ptr{Cvoid}, Ptr{Float64}, Cint, Ptr{Cint}, Ptr{Cint}, Ptr{Float64}, Cint, Ptr{Ptr{Cvoid}}),
            context().h, coef, N, dims, first, per, n, ref))
    end
    hs = HIPSeries{N}(ref[], n, Dict(), nothing)
    finalizer(x -> ccall((:abz_series_destroy, libabz), Cint, (Ptr{Cvoid},), x.h), hs)
end

"Cached PTR rule: replaces FourierPTR / FourierMonkhorstPack (src/fourier.jl:127-174,210-277)."
mutable struct HIPRule
    h::Ptr{Cvoid}
    nk::Int
    npt::Int
    nsyms::Int
end
function symptr_rule(npt, ::Val{d}, syms) where {d}   # replaces AutoSymPTR.symptr_rule (src/fourier.jl:271)
    S = Cint[round(Int, M[a, b]) for M in syms for a in 1:d for b in 1:d]   # row-major per matrix
    n = Ref{Int64}(0)
    check(ccall((:abz_symptr_rule, libabz), Cint, (Cint, Cint, Ptr{Cint}, Cint, Ptr{Int64}, Ptr{Cint}, Ptr{Int64}),
        npt, d, S, length(syms), n, C_NULL, C_NULL))
    idx = Matrix{Cint}(undef, d, n[]); w = Vector{Int64}(undef, n[])
    check(ccall((:abz_symptr_rule, libabz), Cint, (Cint, Cint, Ptr{Cint}, Cint, Ptr{Int64}, Ptr{Cint}, Ptr{Int64}),
        npt, d, S, length(syms), n, idx, w))
    return idx, w
end
function rule!(hs::HIPSeries{d}, npt::Integer, syms, want::Cint) where {d}
    key = (Int(npt), hash(syms), want)
    get!(hs.rules, key) do
        ref = Ref{Ptr{Cvoid}}(C_NULL)
        if syms === nothing
            check(ccall((:abz_ptr_rule_build, libabz), Cint,
                (Ptr{Cvoid}, Cint, Int64, Ptr{Cint}, Ptr{Int64}, Cint, Ptr{Ptr{Cvoid}}),
                hs.h, npt, 0, C_NULL, C_NULL, want, ref))
            nk, ns = npt^d, 1
        elseif length(syms) <= 48
            # orbit tables, contraction plan and values on the device: the node list never visits the host
            S = Cint[round(Int, M[a, b]) for M in syms for a in 1:d for b in 1:d]   # row-major per matrix
            check(ccall((:abz_ptr_rule_build_sym, libabz), Cint,
                (Ptr{Cvoid}, Cint, Ptr{Cint}, Cint, Cint, Ptr{Ptr{Cvoid}}), hs.h, npt, S, length(syms), want, ref))
            n = Ref{Int64}(0)
            check(ccall((:abz_rule_info, libabz), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Cint}, Ptr{Cint}, Ptr{Cint}, Ptr{Cint}),
                ref[], n, C_NULL, C_NULL, C_NULL, C_NULL))
            nk, ns = Int(n[]), length(syms)
        else
            idx, w = symptr_rule(npt, Val(d), syms)
            check(ccall((:abz_ptr_rule_build, libabz), Cint,
                (Ptr{Cvoid}, Cint, Int64, Ptr{Cint}, Ptr{Int64}, Cint, Ptr{Ptr{Cvoid}}),
                hs.h, npt, length(w), idx, w, want, ref))
            nk, ns = length(w), length(syms)
        end
        r = HIPRule(ref[], nk, npt, ns)
        finalizer(x -> ccall((:abz_rule_destroy, libabz), Cint, (Ptr{Cvoid},), x.h), r)
    end
end

# ---------------------------------------------------------------- device integrands
"Integrands evaluated on the GPU; everything else goes through `FourierValue` batches."
abstract type HIPIntegrand end
struct DOSIntegrand <: HIPIntegrand end       # (h_k, eta, omega) -> -imag(tr(inv((omega+im*eta)I - h_k.s)))/pi
struct TrGlocIntegrand <: HIPIntegrand end    # (h_k; eta, omega) -> tr(inv(complex(omega,eta)I - h_k.s))
struct GlocIntegrand <: HIPIntegrand end      # (h_k; eta, omega) -> inv(complex(omega,eta)I - h_k.s)
struct LinearIntegrand <: HIPIntegrand end    # (x, a; b) -> a*x.s + b
struct UnitIntegrand <: HIPIntegrand end
fid(::DOSIntegrand) = F_DOS; fid(::TrGlocIntegrand) = F_TRGLOC; fid(::GlocIntegrand) = F_GLOC
fid(::LinearIntegrand) = F_LINEAR; fid(::UnitIntegrand) = F_ONE
# host fall-backs keep the objects usable with every other AutoBZCore algorithm
(::DOSIntegrand)(h::FourierValue, eta, omega) = -imag(tr(inv((omega + im * eta) * I - h.s))) / pi
(::LinearIntegrand)(x::FourierValue, a; b) = a * x.s + b
bind(::Union{DOSIntegrand,TrGlocIntegrand,GlocIntegrand}, p::MixedParameters) =
    (Float64[get(getfield(p, :kwargs), :eta, p[1])], Float64(get(getfield(p, :kwargs), :omega, p[end])))
bind(::LinearIntegrand, p::MixedParameters) = (Float64[p[1], getfield(p, :kwargs).b], 0.0)
bind(::UnitIntegrand, p) = (Float64[], 0.0)

"Components of a device integrand's value (the library's integrand_ncomp): the buffers below are sized with it."
ncomp(::GlocIntegrand, n::Int, d::Int) = n * n
ncomp(::HIPIntegrand, n::Int, d::Int) = 1
"complex [ncomp] from the device -> the value the reference integrand returns (real only where the integrand is real)."
shape_value(::Union{DOSIntegrand,UnitIntegrand}, v::AbstractVector{ComplexF64}, n::Int) = real(v[1])
shape_value(::GlocIntegrand, v::AbstractVector{ComplexF64}, n::Int) = n == 1 ? v[1] : SMatrix{n,n,ComplexF64,n * n}(v)  # column-major block
shape_value(::HIPIntegrand, v::AbstractVector{ComplexF64}, n::Int) = v[1]          # TrGloc, Linear: complex scalars

const HIPFourierIntegrand = FourierIntegrand{<:HIPIntegrand}

"rule(f, B) = quadsum(...) (src/fourier.jl:204-207,289-292) for all sweep values in one pass."
function reduce_rule(r::HIPRule, f::HIPIntegrand, params::Vector{Float64}, sweep::Vector{Float64}, ncomp::Int)
    out = Vector{ComplexF64}(undef, ncomp * length(sweep))
    GC.@preserve params sweep out begin
        check(ccall((:abz_rule_reduce, libabz), Cint,
            (Ptr{Cvoid}, Cint, Ptr{Float64}, Cint, Ptr{Float64}, Cint, Cint, Ptr{ComplexF64}),
            r.h, fid(f), params, length(params), sweep, length(sweep), r.nsyms, out))
    end
    return reshape(out, ncomp, :)
end

# ---------------------------------------------------------------- the dispatch pair
# cacheval = the device series; rules are cached inside it by (npt, syms, want), so unlike
# src/interfaces.jl:174-179 nothing is rebuilt per solver call.
AutoBZCore.init_cacheval(f::HIPFourierIntegrand, bz::SymmetricBZ, p, ::Union{PTR,AutoPTR,IAI}) = HIPSeries(f.w.series)

function AutoBZCore.do_solve(f::HIPFourierIntegrand, bz::SymmetricBZ, p, alg::PTR, hs::HIPSeries{d};
    abstol=nothing, reltol=nothing, maxiters=typemax(Int)) where {d}
    j = abs(det(bz.B))                                   # src/brillouin.jl:340
    r = rule!(hs, alg.npt, bz.syms, WANT_HC)
    params, omega = bind(f.f.f, merge(f.f.p, p))
    u = reduce_rule(r, f.f.f, params, [omega], ncomp(f.f.f, hs.n, d))
    return IntegralSolution(j * nsyms(bz) * shape_value(f.f.f, view(u, :, 1), hs.n), nothing, true, r.nk)   # TrivialRep: src/brillouin.jl:107
end

"""
The p-adaptive loop of `AutoPTR` runs inside the library (`abz_autoptr_solve_many`: grid sequence, rules kept by the series,
store-free sums for grids used once, error test, `numevals`) -- the ONE implementation the Python mirror and its tests
drive as well (round 3 had a second, untested copy of the loop here).  `omegas`: one solve per swept value, in lock-step.
"""
function autoptr_solve(hs::HIPSeries{d}, f::HIPIntegrand, params::Vector{Float64}, omegas::Vector{Float64}, bz::SymmetricBZ,
    alg::AutoPTR; abstol=nothing, reltol=nothing, maxiters=typemax(Int)) where {d}
    j = abs(det(bz.B))
    n0 = clamp(round(Int, alg.n₀ / alg.a), alg.nmin, alg.nmax)       # MonkhorstPackRule's integers, src/fourier.jl:301-321
    dn = clamp(round(Int, alg.Δn / alg.a), alg.nmin, alg.nmax)
    syms = bz.syms === nothing ? Cint[] : Cint[round(Int, M[a, b]) for M in bz.syms for a in 1:d for b in 1:d]
    ns = bz.syms === nothing ? 0 : length(bz.syms)
    nsolve = length(omegas); nc = ncomp(f, hs.n, d)
    out = Vector{ComplexF64}(undef, nc * nsolve); err = Vector{Float64}(undef, nsolve)   # the library writes nc values per solve
    nev = Vector{Int64}(undef, nsolve); npt = Vector{Cint}(undef, nsolve)
    GC.@preserve syms params omegas out err nev npt begin
        check(ccall((:abz_autoptr_solve_many, libabz), Cint,
            (Ptr{Cvoid}, Ptr{Cint}, Cint, Cint, Ptr{Float64}, Cint, Ptr{Float64}, Cint, Cint, Cint, Float64, Float64, Int64, Cint,
             Float64, Ptr{ComplexF64}, Ptr{Float64}, Ptr{Int64}, Ptr{Cint}),
            hs.h, ns == 0 ? C_NULL : pointer(syms), ns, fid(f), params, length(params), omegas, nsolve, n0, dn,
            abstol === nothing ? -1.0 : abstol / j,                  # src/brillouin.jl:433
            reltol === nothing ? -1.0 : reltol, min(maxiters, typemax(Int64) >> 1), alg.keepmost,
            Float64(nsyms(bz)),                                      # TrivialRep inside every rule: src/brillouin.jl:127-130
            out, err, nev, npt))
    end
    vals = reshape(out, nc, nsolve)
    return [IntegralSolution(shape_value(f, view(vals, :, i), hs.n) * j, err[i] * j, true, Int(nev[i])) for i in 1:nsolve]
end

"One AutoPTR solve through the single-value entry point (abz_autoptr_solve): what `do_solve(::AutoPTR)` calls."
function autoptr_solve(hs::HIPSeries{d}, f::HIPIntegrand, params::Vector{Float64}, omega::Float64, bz::SymmetricBZ,
    alg::AutoPTR; abstol=nothing, reltol=nothing, maxiters=typemax(Int)) where {d}
    j = abs(det(bz.B))
    n0 = clamp(round(Int, alg.n₀ / alg.a), alg.nmin, alg.nmax)
    dn = clamp(round(Int, alg.Δn / alg.a), alg.nmin, alg.nmax)
    syms = bz.syms === nothing ? Cint[] : Cint[round(Int, M[a, b]) for M in bz.syms for a in 1:d for b in 1:d]
    ns = bz.syms === nothing ? 0 : length(bz.syms)
    out = Vector{ComplexF64}(undef, ncomp(f, hs.n, d)); err = Ref{Float64}(0); nev = Ref{Int64}(0); npt = Ref{Cint}(0)
    GC.@preserve syms params out begin
        check(ccall((:abz_autoptr_solve, libabz), Cint,
            (Ptr{Cvoid}, Ptr{Cint}, Cint, Cint, Ptr{Float64}, Cint, Float64, Cint, Cint, Float64, Float64, Int64, Cint,
             Float64, Ptr{ComplexF64}, Ptr{Float64}, Ptr{Int64}, Ptr{Cint}),
            hs.h, ns == 0 ? C_NULL : pointer(syms), ns, fid(f), params, length(params), omega, n0, dn,
            abstol === nothing ? -1.0 : abstol / j, reltol === nothing ? -1.0 : reltol,
            min(maxiters, typemax(Int64) >> 1), alg.keepmost, Float64(nsyms(bz)), out, err, nev, npt))
    end
    return IntegralSolution(shape_value(f, out, hs.n) * j, err[] * j, true, Int(nev[]))
end

function AutoBZCore.do_solve(f::HIPFourierIntegrand, bz::SymmetricBZ, p, alg::AutoPTR, hs::HIPSeries;
    abstol=nothing, reltol=nothing, maxiters=typemax(Int))
    params, omega = bind(f.f.f, merge(f.f.p, p))
    return autoptr_solve(hs, f.f.f, params, omega, bz, alg; abstol, reltol, maxiters)
end

function AutoBZCore.do_solve(f::HIPFourierIntegrand, bz::SymmetricBZ, p, alg::IAI, hs::HIPSeries{d};
    abstol=nothing, reltol=nothing, maxiters=typemax(Int)) where {d}
    j = abs(det(bz.B)); ns = nsyms(bz)
    params, omega = bind(f.f.f, merge(f.f.p, p))
    kind, a, b = pack_limits(bz.lims)
    out = Vector{ComplexF64}(undef, ncomp(f.f.f, hs.n, d)); err = Ref{Float64}(0); nev = Ref{Int64}(0); npan = Ref{Int64}(0)
    GC.@preserve params a b out begin
        check(ccall((:abz_iai_solve, libabz), Cint,
            (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}, Cint, Ptr{Float64}, Cint, Float64, Float64, Float64, Int64, Int64,
             Ptr{ComplexF64}, Ptr{Float64}, Ptr{Int64}, Ptr{Float64}, Int64, Ptr{Int64}),
            hs.h, kind, a, isempty(b) ? C_NULL : pointer(b), fid(f.f.f), params, length(params), omega,
            abstol === nothing ? -1.0 : abstol / (j * ns),        # src/brillouin.jl:342
            reltol === nothing ? -1.0 : Float64(reltol), min(maxiters, typemax(Int64) >> 1), 0,
            out, err, nev, C_NULL, 0, npan))
    end
    return IntegralSolution(j * ns * shape_value(f.f.f, out, hs.n), j * ns * err[], true, nev[])
end

# ---------------------------------------------------------------- BatchIntegrand body for user closures
"""
    fourier_batch(g, s::FourierSeries)

`BatchIntegrand` whose body evaluates H(k) for all nodes of a batch on the GPU and applies the user
closure `g(FourierValue(k, H_k), p)` on the host (src/batch.jl:4-6 names this as the GPU hook).
"""
function fourier_batch(g, s::FourierSeries{S,N}, ::Type{Y}=ComplexF64) where {S,N,Y}
    hs = HIPSeries(s); n = hs.n
    return AutoBZCore.BatchIntegrand(Y, SVector{N,Float64}) do y, x, p
        k = reinterpret(Float64, x); H = Vector{ComplexF64}(undef, n * n * length(x))
        GC.@preserve k H check(ccall((:abz_eval_nodes, libabz), Cint,
            (Ptr{Cvoid}, Ptr{Float64}, Int64, Cint, Ptr{ComplexF64}, Ptr{Float64}),
            hs.h, k, length(x), WANT_H, H, C_NULL))
        Hs = reinterpret(SMatrix{n,n,ComplexF64,n * n}, H)
        @inbounds for i in eachindex(x)
            y[i] = g(FourierValue(x[i], Hs[i]), p)
        end
        return nothing
    end
end

# ---------------------------------------------------------------- fused parameter sweep
"batchsolve for a HIP integrand under PTR: one pass over the cached rule for all parameters."
function AutoBZCore.batchsolve(s::IntegralSolver{<:HIPFourierIntegrand,<:SymmetricBZ,<:PTR}, omegas::AbstractVector{<:Real})
    f, bz = s.f, s.dom
    hs = HIPSeries(f.w.series)
    r = rule!(hs, s.alg.npt, bz.syms, WANT_HC)
    params, _ = bind(f.f.f, merge(f.f.p, MixedParameters(first(omegas))))
    u = reduce_rule(r, f.f.f, params, Float64.(omegas), ncomp(f.f.f, hs.n, ndims(f.w.series.c)))
    return [abs(det(bz.B)) * nsyms(bz) * shape_value(f.f.f, view(u, :, i), hs.n) for i in axes(u, 2)]
end

"batchsolve for a HIP integrand under AutoPTR: the solves refine in lock-step, every grid is visited once for all that are still active."
function AutoBZCore.batchsolve(s::IntegralSolver{<:HIPFourierIntegrand,<:SymmetricBZ,<:AutoPTR}, omegas::AbstractVector{<:Real})
    f, bz = s.f, s.dom
    hs = HIPSeries(f.w.series)
    params, _ = bind(f.f.f, merge(f.f.p, MixedParameters(first(omegas))))
    sols = autoptr_solve(hs, f.f.f, params, Float64.(omegas), bz, s.alg; abstol=get(s.kwargs, :abstol, nothing),
        reltol=get(s.kwargs, :reltol, nothing), maxiters=get(s.kwargs, :maxiters, typemax(Int)))
    return [sol.u for sol in sols]
end

"batchsolve for a HIP integrand under IAI: all solves advance in lock-step and share their launches."
function AutoBZCore.batchsolve(s::IntegralSolver{<:HIPFourierIntegrand,<:SymmetricBZ,<:IAI}, omegas::AbstractVector{<:Real})
    f, bz = s.f, s.dom
    hs = HIPSeries(f.w.series)
    j = abs(det(bz.B)); ns = nsyms(bz); m = length(omegas)
    params, _ = bind(f.f.f, merge(f.f.p, MixedParameters(first(omegas))))
    kind, a, b = pack_limits(bz.lims)
    abstol = get(s.kwargs, :abstol, nothing); reltol = get(s.kwargs, :reltol, nothing)
    nc = ncomp(f.f.f, hs.n, ndims(f.w.series.c))
    sw = Float64.(omegas); out = Vector{ComplexF64}(undef, nc * m); err = Vector{Float64}(undef, m); nev = Vector{Int64}(undef, m)
    npan = Ref{Int64}(0)
    GC.@preserve params a b sw out err nev begin
        check(ccall((:abz_iai_solve_many, libabz), Cint,
            (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}, Cint, Ptr{Float64}, Cint, Ptr{Float64}, Cint, Float64, Float64,
             Int64, Int64, Ptr{ComplexF64}, Ptr{Float64}, Ptr{Int64}, Ptr{Float64}, Int64, Ptr{Int64}),
            hs.h, kind, a, isempty(b) ? C_NULL : pointer(b), fid(f.f.f), params, length(params), sw, m,
            abstol === nothing ? -1.0 : abstol / (j * ns), reltol === nothing ? -1.0 : Float64(reltol),
            typemax(Int64) >> 1, 0, out, err, nev, C_NULL, 0, npan))
    end
    vals = reshape(out, nc, m)
    return [j * ns * shape_value(f.f.f, view(vals, :, i), hs.n) for i in 1:m]
end

# ---------------------------------------------------------------- store-free rule value
"rule(f, B) on the full npt^d grid without materialising FourierPTR (abz_ptr_sum): grids used once or beyond HBM."
function ptr_sum(hs::HIPSeries{d}, npt::Integer, f::HIPIntegrand, params::Vector{Float64}, omegas::Vector{Float64};
    z0=0, z1=npt, nsyms::Integer=1) where {d}
    nc = ncomp(f, hs.n, d)
    out = Vector{ComplexF64}(undef, nc * length(omegas))                 # [ncomp, n_sweep] like reduce_rule
    GC.@preserve params omegas out check(ccall((:abz_ptr_sum, libabz), Cint,
        (Ptr{Cvoid}, Cint, Cint, Cint, Cint, Ptr{Float64}, Cint, Ptr{Float64}, Cint, Cint, Ptr{ComplexF64}),
        hs.h, npt, z0, z1, fid(f), params, length(params), omegas, length(omegas), nsyms, out))
    return nc == 1 ? out : reshape(out, nc, :)
end

# ---------------------------------------------------------------- k-sharded rules (one solve on several GPUs)
"""
    slab_rule(hs, npt, rank, world, want)

This rank's slab of the full PTR grid (outermost index in `[z0, z1)`), for a solve sharded over k: sum
the `reduce_rule` results of all ranks (e.g. `MPI.Allreduce` / RCCL) to obtain the rule value.
"""
function slab_rule(hs::HIPSeries{N}, npt::Integer, rank::Integer, world::Integer, want::Integer=WANT_H) where {N}
    z0 = (npt * rank) ÷ world; z1 = (npt * (rank + 1)) ÷ world
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:abz_ptr_rule_build_slab, libabz), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Cint, Ptr{Ptr{Cvoid}}),
        hs.h, npt, z0, z1, want, h))
    return HIPRule(h[], (z1 - z0) * npt^(N - 1), npt, 1)
end

# ---------------------------------------------------------------- GGR
"get_ggr_data + sum_ggr on the GPU (src/dos_ggr.jl:14-65)."
function ggr(h::FourierSeries{S,N}, bz::SymmetricBZ, Es::Vector{Float64}; npt=50) where {S,N}
    hs = HIPSeries(h)
    r = rule!(hs, npt, bz.syms, WANT_EIG | WANT_VEL)
    out = similar(Es)
    check(ccall((:abz_rule_ggr, libabz), Cint, (Ptr{Cvoid}, Ptr{Float64}, Cint, Ptr{Float64}), r.h, Es, length(Es), out))
    return out
end

# ---------------------------------------------------------------- cached rule -> the reference's own containers
"""
    export_rule(r, hs; H=true, eig=false, vel=false)

Host copies of a cached rule in the reference's layout -- `(x, w, H, eig, vel)` with `x :: Vector{SVector{d}}`, integer-valued
weights `w`, `H :: Vector{SMatrix{n,n,ComplexF64}}` -- i.e. what iterating a `FourierPTR` / `FourierMonkhorstPack` yields
(src/fourier.jl:177-207,279-292).  A user closure under `PTR` is fed from the cached rule through this:
`sum(w[i] * f(FourierValue(x[i], H[i]), p...) for i in eachindex(x)) / (npt^d * nsyms)`.
"""
function export_rule(r::HIPRule, hs::HIPSeries{d}; H::Bool=true, eig::Bool=false, vel::Bool=false) where {d}
    n = hs.n
    x = Matrix{Float64}(undef, d, r.nk); w = Vector{Float64}(undef, r.nk)
    Hb = H ? Array{ComplexF64}(undef, n, n, r.nk) : ComplexF64[]
    Eb = eig ? Matrix{Float64}(undef, n, r.nk) : Float64[]
    Vb = vel ? Array{Float64}(undef, n, d, r.nk) : Float64[]
    GC.@preserve x w Hb Eb Vb begin
        check(ccall((:abz_rule_export, libabz), Cint,
            (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{ComplexF64}, Ptr{Float64}, Ptr{Float64}),
            r.h, x, w, H ? pointer(Hb) : C_NULL, eig ? pointer(Eb) : C_NULL, vel ? pointer(Vb) : C_NULL))
    end
    xs = [SVector{d,Float64}(view(x, :, k)) for k in 1:r.nk]
    Hs = H ? [SMatrix{n,n,ComplexF64}(view(Hb, :, :, k)) for k in 1:r.nk] : nothing
    return (x=xs, w=w, H=Hs, eig=eig ? Eb : nothing, vel=vel ? Vb : nothing)
end

"A plain-function integrand under `PTR` fed from the cached rule (src/fourier.jl:204-207): `rule(f, B)`."
function rule_sum(f, r::HIPRule, hs::HIPSeries{d}, args...; kws...) where {d}
    ex = export_rule(r, hs)
    acc = sum(ex.w[i] * f(FourierValue(ex.x[i], ex.H[i]), args...; kws...) for i in eachindex(ex.x))
    return acc / (r.npt^d * r.nsyms)
end

"Drop the rules the library keeps with the series for `abz_autoptr_solve*`."
drop_rules!(hs::HIPSeries) = check(ccall((:abz_series_drop_rules, libabz), Cint, (Ptr{Cvoid},), hs.h))

# ---------------------------------------------------------------- IAI building blocks (a host-language adaptive loop)
# What `abz_iai_solve*` is built from, for a client that keeps its own heaps (e.g. a custom error norm or termination):
# contract the outermost remaining variable at a batch of nodes, evaluate the integrand on innermost lines, apply the
# GK(7,15) rule to batches of panels.  Replaces workspace_contract! / workspace_evaluate! (src/fourier.jl:445-478) and
# QuadGK.evalrule on a batch.
"15 Kronrod nodes of the panel `(a, b)` in the library's (and the reference's) order."
function gk15_nodes(a::Real, b::Real)
    x = Vector{Float64}(undef, 15)
    check(ccall((:abz_gk15_nodes, libabz), Cint, (Cdouble, Cdouble, Ptr{Float64}), a, b, x))
    return x
end

"GK(7,15) on `npanels` panels: `ab` is 2 x npanels, `values` ncomp x 15 x npanels -> `(I :: ncomp x npanels, E)`."
function gk15_batch(ab::Matrix{Float64}, values::Array{ComplexF64,3})
    ncomp, np = size(values, 1), size(values, 3)
    (size(ab) == (2, np) && size(values, 2) == 15) || throw(ArgumentError("gk15_batch: ab is 2 x npanels, values ncomp x 15 x npanels"))
    I = Matrix{ComplexF64}(undef, ncomp, np); E = Vector{Float64}(undef, np)
    GC.@preserve ab values I E check(ccall((:abz_gk15_batch, libabz), Cint,
        (Ptr{Float64}, Ptr{ComplexF64}, Int64, Cint, Ptr{ComplexF64}, Ptr{Float64}), ab, values, np, ncomp, I, E))
    return I, E
end

"Contract the outermost remaining variable of the level-`src_level` sets `parents` (slot 0 at level d = the series) at `x`: slots of the new sets."
function contract_nodes(hs::HIPSeries, src_level::Integer, parents::Vector{Int64}, x::Vector{Float64})
    length(parents) == length(x) || throw(ArgumentError("contract_nodes: one parent per node"))
    slots = Vector{Int64}(undef, length(x))
    GC.@preserve parents x slots check(ccall((:abz_contract_nodes, libabz), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Int64}, Ptr{Float64}, Int64, Ptr{Int64}), hs.h, src_level, parents, x, length(x), slots))
    return slots
end

"Integrand values at innermost nodes `x` of the level-1 sets `parents` (`tail`: (d-1) x nnodes outer coordinates, for `F_LINEAR_X` only)."
function eval_line_nodes(hs::HIPSeries, parents::Vector{Int64}, x::Vector{Float64}, f::HIPIntegrand, params::Vector{Float64}, sweep::Real,
                         ncomp::Integer; tail::Union{Nothing,Matrix{Float64}}=nothing)
    vals = Matrix{ComplexF64}(undef, ncomp, length(x))
    tp = tail === nothing ? Ptr{Float64}(C_NULL) : pointer(tail)
    GC.@preserve parents x params vals tail check(ccall((:abz_eval_line_nodes, libabz), Cint,
        (Ptr{Cvoid}, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}, Int64, Cint, Ptr{Float64}, Cint, Cdouble, Ptr{ComplexF64}),
        hs.h, parents, x, tp, length(x), fid(f), params, length(params), sweep, vals))
    return vals
end

"Drop every contracted set below `level` (the client's loop is done with them)."
release_level!(hs::HIPSeries, level::Integer) = check(ccall((:abz_release_level, libabz), Cint, (Ptr{Cvoid}, Cint), hs.h, level))

# ---------------------------------------------------------------- the rest of abzhip.h (housekeeping)
"""(live device bytes, cached device bytes, this context's scratch bytes, its pinned host bytes, live blocks)"""
function mem_info(; ctx::HIPContext=context())
    info = zeros(Int64, 5)
    check(ccall((:abz_mem_info, libabz), Cint, (Ptr{Cvoid}, Ptr{Int64}), ctx.h, info))
    return Tuple(info)
end

version() = Int(ccall((:abz_version, libabz), Cint, ()))
function device_count()
    n = Ref{Cint}(0)
    check(ccall((:abz_device_count, libabz), Cint, (Ptr{Cint},), n))
    return Int(n[])
end
synchronize(ctx::HIPContext=context()) = check(ccall((:abz_ctx_sync, libabz), Cint, (Ptr{Cvoid},), ctx.h))

"New coefficients of the same shape (mutating `h.c`, test/dos.jl:122-129); cached rules go stale until `rebuild!`."
function update!(hs::HIPSeries, s::FourierSeries)
    coef = reinterpret(Float64, vec(s.c))
    GC.@preserve coef check(ccall((:abz_series_update, libabz), Cint, (Ptr{Cvoid}, Ptr{Float64}), hs.h, coef))
    try
        foreach(rebuild!, values(hs.rules))   # cached rule values follow the coefficients (the reference rebuilds its rule per solve)
    catch err
        # an upper-triangle (WANT_H_COMPACT) rule cannot be refilled from a series that stopped being Hermitian
        # (ABZ_ERR_ARG -> ArgumentError): drop the cache, the next solve builds full-layout rules
        err isa ArgumentError || rethrow()
        empty!(hs.rules)
    end
    return hs
end
"Context on a stream the caller owns (e.g. AMDGPU.jl's task-local HIP stream): launches are ordered with the caller's work."
function HIPContext(device::Integer, stream::Ptr{Cvoid})
    ref = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:abz_ctx_create_on_stream, libabz), Cint, (Cint, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), device, stream, ref))
    ctx = HIPContext(ref[], Val(:adopt))
    finalizer(c -> ccall((:abz_ctx_destroy, libabz), Cint, (Ptr{Cvoid},), c.h), ctx)
end
"abz_rule_reduce with device-resident sweep values and sums (both `Ptr{Cvoid}` device addresses): no host round trip."
function reduce_rule_device!(out_dev::Ptr{Cvoid}, r::HIPRule, f::HIPIntegrand, params::Vector{Float64}, sweep_dev::Ptr{Cvoid}, nsweep::Integer, nsyms::Integer=r.nsyms)
    GC.@preserve params check(ccall((:abz_rule_reduce_device, libabz), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Float64}, Cint, Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}),
        r.h, fid(f), params, length(params), sweep_dev, nsweep, nsyms, out_dev))
    return out_dev
end
"""
    shard_iai!(hs, allgather!, rank, world)

One IAI solve on several GPUs (`abz_iai_set_exchange`): `allgather!(buf::Vector{Float64}, per_rank::Int)` must fill every
rank's segment of `buf` (MPI.Allgather!(MPI.IN_PLACE, UBuffer(buf, per_rank), comm) does).  All ranks then call the solver
with the same arguments and get the single-GPU result bit for bit.  Pass `nothing` to switch it off.
"""
function shard_iai!(hs::HIPSeries, allgather!, rank::Integer, world::Integer)
    if allgather! === nothing
        check(ccall((:abz_iai_set_exchange, libabz), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint), hs.h, C_NULL, C_NULL, 0, 1))
        return hs
    end
    cb = (user::Ptr{Cvoid}, buf::Ptr{Float64}, per::Int64) -> begin
        try
            allgather!(unsafe_wrap(Array, buf, Int(per) * Int(world)), Int(per)); Cint(0)
        catch
            Cint(1)
        end
    end
    fptr = @cfunction($cb, Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64))
    hs.exchange = fptr                                   # keep the closure alive with the series
    check(ccall((:abz_iai_set_exchange, libabz), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint),
                hs.h, Base.unsafe_convert(Ptr{Cvoid}, fptr), C_NULL, rank, world))
    return hs
end
"Device address and size of the rule's value block (tiled planar layout, DESIGN.md section 3)."
function values_ptr(r::HIPRule)
    base = Ref{Ptr{Cvoid}}(C_NULL); nb = Ref{Int64}(0)
    check(ccall((:abz_rule_values_ptr, libabz), Cint, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Ptr{Int64}), r.h, base, nb))
    return base[], nb[]
end
rebuild!(r::HIPRule) = (check(ccall((:abz_rule_rebuild, libabz), Cint, (Ptr{Cvoid},), r.h)); r)

function rule_info(r::HIPRule)
    nk = Ref{Int64}(0); n = Ref{Cint}(0); d = Ref{Cint}(0); npt = Ref{Cint}(0); want = Ref{Cint}(0)
    check(ccall((:abz_rule_info, libabz), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Cint}, Ptr{Cint}, Ptr{Cint}, Ptr{Cint}),
                r.h, nk, n, d, npt, want))
    return (nk=nk[], n=Int(n[]), d=Int(d[]), npt=Int(npt[]), want=want[])
end

"`symptr_rule` with the orbit tables computed on the GPU (bit-identical to the host version, ~50x faster on large grids)."
function symptr_rule_device(npt, ::Val{d}, syms; ctx::HIPContext=context()) where {d}
    S = Cint[round(Int, M[a, b]) for M in syms for a in 1:d for b in 1:d]   # row-major per matrix
    nirr = Ref{Int64}(0)
    check(ccall((:abz_symptr_rule_device, libabz), Cint,
                (Ptr{Cvoid}, Cint, Cint, Ptr{Cint}, Cint, Ptr{Int64}, Ptr{Cint}, Ptr{Int64}),
                ctx.h, npt, d, S, length(syms), nirr, C_NULL, C_NULL))   # first call: count
    idx = Matrix{Cint}(undef, d, nirr[]); w = Vector{Int64}(undef, nirr[])
    check(ccall((:abz_symptr_rule_device, libabz), Cint,
                (Ptr{Cvoid}, Cint, Cint, Ptr{Cint}, Cint, Ptr{Int64}, Ptr{Cint}, Ptr{Int64}),
                ctx.h, npt, d, S, length(syms), nirr, idx, w))
    return idx, w
end

"HIP-event timing of the library's own launches: `prof_enable(true)`, run, `prof_read(K_EVAL)` -> (ms, launches)."
const K_CONTRACT, K_EVAL, K_REDUCE, K_GGR, K_EIG, K_GGRBUILD = Cint.(0:5)
prof_enable(on::Bool=true; ctx::HIPContext=context()) =
    check(ccall((:abz_prof_enable, libabz), Cint, (Ptr{Cvoid}, Cint), ctx.h, on ? 1 : 0))
prof_reset(; ctx::HIPContext=context()) = check(ccall((:abz_prof_reset, libabz), Cint, (Ptr{Cvoid},), ctx.h))
function prof_read(kernel::Cint; ctx::HIPContext=context())
    ms = Ref{Float64}(0.0); n = Ref{Int64}(0)
    check(ccall((:abz_prof_read, libabz), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Int64}), ctx.h, kernel, ms, n))
    return ms[], n[]
end

end # module
