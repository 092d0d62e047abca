# make_golden.jl -- emit bit-level golden vectors of the REFERENCE (AutoBZCore.jl v0.3.8 and its pinned dependencies)
# in the layout tests/golden/ uses, so that the oracle and the HIP path can be pinned to the reference itself instead
# of to a restatement of its un-vendored dependencies (AutoSymPTR, IteratedIntegration/QuadGK, FourierSeriesEvaluators).
#
# This pipeline has no Julia toolchain: the script is for a maintainer who has one.
#
#   julia --project=<env with AutoBZCore@0.3.8> julia/make_golden.jl tests/golden/reference_v038.json
#
# then `python -m pytest tests/test_reference_golden.py` (CPU: oracle vs reference; `-m gpu`: HIP path vs reference).
# While the file is absent those tests are skipped with that reason.
#
# Contents (all arrays flattened, numbers printed with `repr` = shortest round-trip decimal):
#   symptr[i]    = {kind, d, npt, idx (nirr*d, 0-based grid indices, point-major, column-major point order), w (nirr), nsym}
#                  from AutoSymPTR.symptr_rule as called at src/fourier.jl:271
#   panels[i]    = {kind, d, abstol, batch, coef (re, im interleaved, Julia memory order), dims, first, n, eta, omega,
#                   segs (outermost final segments a, b sorted), u, resid, numevals}
#                  IAI() on load_bz(kind, I) for the 2-band 5x5 series of tests/test_gpu_parity.py::test_iai_panel_tree_bit_exact
#   npt_sequence = {a: [npt of the rules AutoPTR(a=a) visits]} for a in (1, 0.1, 0.01)   (src/fourier.jl:301-321)
#   svo          = {omega[10], u[10]}: IntegralSolver(FourierIntegrand(dos, h, eta=0.1), CubicSymIBZ, AutoPTR(); abstol=1e-3)
#                  on aps_example/svo_hr.dat (aps_example/aps_example.jl:7-34)
using AutoBZCore, LinearAlgebra, StaticArrays, Random
using AutoBZCore: AutoSymPTR, IteratedIntegration, FourierSeriesEvaluators
using OffsetArrays

out = length(ARGS) >= 1 ? ARGS[1] : "reference_v038.json"

# ---------------------------------------------------------------- tiny JSON writer (no package needed)
jnum(x::Integer) = string(x)
jnum(x::AbstractFloat) = isfinite(x) ? repr(Float64(x)) : "null"
jval(x::Real) = jnum(x)
jval(x::AbstractString) = "\"" * x * "\""
jval(x::Bool) = x ? "true" : "false"
jval(x::AbstractArray) = "[" * join((jval(v) for v in x), ",") * "]"
jval(x::AbstractDict) = "{" * join(("\"$(k)\":" * jval(v) for (k, v) in x), ",") * "}"
jval(x::NamedTuple) = jval(Dict(String(k) => v for (k, v) in pairs(x)))

# ---------------------------------------------------------------- (i) symptr_rule tables
function symptr_table(kind, d, npt)
    bz = load_bz(kind == "CubicSymIBZ" ? CubicSymIBZ() : InversionSymIBZ(), Matrix{Float64}(I, d, d))
    wsym, flags, nsym = AutoSymPTR.symptr_rule(npt, Val(d), bz.syms)
    idx = Int[]; w = Int[]
    for ci in CartesianIndices(wsym)            # column-major: i_1 fastest, the order of the rule's values
        wsym[ci] == 0 && continue
        append!(idx, Tuple(ci) .- 1); push!(w, wsym[ci])
    end
    return (kind=kind, d=d, npt=npt, idx=idx, w=w, nsym=nsym)
end
symptr = [symptr_table(k, d, n) for (k, d, n) in (("InversionSymIBZ", 1, 9), ("InversionSymIBZ", 2, 8), ("InversionSymIBZ", 3, 7),
                                                 ("CubicSymIBZ", 2, 9), ("CubicSymIBZ", 3, 8), ("CubicSymIBZ", 3, 50))]

# ---------------------------------------------------------------- (ii) panel trees of IAI
# A fixed Hermitian 2-band series with 5 x 5 coefficients (splitmix64 stream: language-neutral, SURVEY 8d)
mutable struct SplitMix; s::UInt64; end
function next!(r::SplitMix)
    r.s += 0x9e3779b97f4a7c15
    z = r.s
    z = (z ⊻ (z >> 30)) * 0xbf58476d1ce4e5b9
    z = (z ⊻ (z >> 27)) * 0x94d049bb133111eb
    z = z ⊻ (z >> 31)
    return 2 * (Float64(z >> 11) * 2.0^-53) - 1
end
function hermitian_series(seed, dims, n)
    rng = SplitMix(seed)
    c = [SMatrix{n,n,ComplexF64}([complex(next!(rng), next!(rng)) for a in 1:n, b in 1:n]) for _ in CartesianIndices(dims)]
    c = reshape(c, dims)
    h = similar(c)
    for ci in CartesianIndices(dims)                # c(-R) = c(R)'
        mi = CartesianIndex((dims .+ 1) .- Tuple(ci))
        h[ci] = (c[ci] + c[mi]') / 2
    end
    first = .-(dims .÷ 2)
    return h, first
end
function panel_case(kind, abstol; batch=false)
    d, n, eta, omega = 2, 2, 0.25, 0.3
    h, first = hermitian_series(0x1234, (5, 5), n)
    hs = FourierSeries(OffsetArray(h, first[1]:first[1]+4, first[2]:first[2]+4), period=1.0)
    bz = load_bz(kind == "FBZ" ? FBZ() : kind == "CubicSymIBZ" ? CubicSymIBZ() : InversionSymIBZ(), Matrix{Float64}(I, d, d))
    dos(hk, eta, omega) = -imag(tr(inv((omega + im * eta) * I - hk.s))) / pi
    f = FourierIntegrand(dos, hs, eta)
    prob = IntegralProblem(f, bz, (omega,))
    cache = AutoBZCore.init(prob, EvalCounter(IAI()); abstol=abstol)
    sol = AutoBZCore.solve!(cache)
    segbuf = cache.cacheval[2]                      # the outermost auxquadgk's segment buffer (src/fourier.jl:394-431,488-491)
    segs = sort([(s.a, s.b) for s in segbuf])
    coef = collect(reinterpret(Float64, vec(h)))
    return (kind=kind, d=d, abstol=abstol, batch=batch, coef=coef, dims=collect(size(h)), first=collect(first), n=n, eta=eta,
            omega=omega, segs=collect(Iterators.flatten(segs)), u=sol.u, resid=sol.resid, numevals=sol.numevals)
end
panels = [panel_case(k, 1e-4) for k in ("FBZ", "InversionSymIBZ", "CubicSymIBZ")]

# ---------------------------------------------------------------- (iii) AutoPTR npt sequence
function npt_sequence(a)
    rule = AutoSymPTR.MonkhorstPackRule(nothing, a, 50, 1000, 6.0, log(10))
    return [rule.n₀ + k * rule.Δn for k in 0:5]
end
npts = Dict(string(a) => npt_sequence(a) for a in (1.0, 0.1, 0.01))

# ---------------------------------------------------------------- (iv) ten solver values on SVO
function svo_values()
    file = joinpath(@__DIR__, "..", "..", "reference", "aps_example", "svo_hr.dat")   # adjust to where svo_hr.dat lives
    isfile(file) || (file = get(ENV, "SVO_HR_DAT", file))
    lines = readlines(file)
    nw = parse(Int, lines[2]); nr = parse(Int, lines[3])
    ndeg = cld(nr, 15)
    deg = reduce(vcat, [parse.(Int, split(l)) for l in lines[4:3+ndeg]])
    H = OffsetArray(zeros(SMatrix{nw,nw,ComplexF64,nw*nw}, 11, 11, 11), -5:5, -5:5, -5:5)
    blocks = Dict{NTuple{3,Int},Matrix{ComplexF64}}()
    for l in lines[4+ndeg:end]
        t = split(l); R = Tuple(parse.(Int, t[1:3])); m, n = parse.(Int, t[4:5])
        get!(blocks, R, zeros(ComplexF64, nw, nw))[m, n] = complex(parse(Float64, t[6]), parse(Float64, t[7]))
    end
    Rs = sort(collect(keys(blocks)))               # file order: R3 fastest among R
    for (i, R) in enumerate(Rs)
        H[R...] = SMatrix{nw,nw,ComplexF64}(blocks[R] / deg[i])
    end
    h = FourierSeries(H, period=1.0)
    bz = load_bz(CubicSymIBZ(), 3.85856 * Matrix{Float64}(I, 3, 3))
    dos(hk, eta, omega) = -imag(tr(inv((omega + im * eta) * I - hk.s))) / pi
    solver = IntegralSolver(FourierIntegrand(dos, h, 0.1), bz, AutoPTR(); abstol=1e-3)
    omega = collect(range(10.5, 14.5, length=10))
    return (omega=omega, u=[solver(w) for w in omega])
end
svo = try svo_values() catch err; @warn "SVO values skipped" err; (omega=Float64[], u=Float64[]) end

open(out, "w") do io
    write(io, "{\"reference\":\"AutoBZCore.jl v0.3.8\",\"symptr\":", jval(symptr), ",\"panels\":", jval(panels),
          ",\"npt_sequence\":", jval(npts), ",\"svo\":", jval(svo), "}\n")
end
println("wrote ", out)
