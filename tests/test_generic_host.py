"""Generic integrands (plain callables, BatchIntegrand, NestedBatchIntegrand) under AuxQuadGKJL, MonkhorstPack,
AutoSymPTRJL and NestedQuad: the reference's own known answers restated (test/interface_tests.jl:90-111 "batch",
:113-130 "multi-algorithms", :143-158 EvalCounter).  Host logic only: the adaptive loops and the user function run
on the CPU, the GK(7,15) panel sums go through the library's host entry point (abz_gk15_batch) -- no GPU needed."""
import math

import numpy as np
import pytest

import autobzcore.jl_amd as abz
import abz_oracle as orc


A, B2PI, ABSTOL, P = 0.0, 2 * math.pi, 1e-5, 3.0
CASES = [  # (body f!(y, x, p), plain f(x, p), reference value)      ref: test/interface_tests.jl:96-100
    (lambda y, x, p: y.__setitem__(slice(None), list(p * np.sin(np.asarray(x, dtype=float).reshape(-1)))),
     lambda x, p: p * math.sin(x), 0.0),
    (lambda y, x, p: y.__setitem__(slice(None), [p * 1.0 for _ in x]), lambda x, p: p * 1.0, P * (B2PI - A)),
    (lambda y, x, p: y.__setitem__(slice(None), list(1.0 / (p - np.cos(np.asarray(x, dtype=float).reshape(-1))))),
     lambda x, p: 1.0 / (p - math.cos(x)), (B2PI - A) / math.sqrt(P * P - 1)),
]


@pytest.mark.parametrize("case", range(3))
def test_batch_integrand_known_answers(case):
    body, plain, ref = CASES[case]
    for f in (abz.BatchIntegrand(body), abz.BatchIntegrand(body, max_batch=45), plain):
        sol = abz.solve(abz.IntegralProblem(f, (A, B2PI), P), abz.AuxQuadGKJL(), abstol=ABSTOL)
        assert abs(sol.u - ref) <= ABSTOL
        for alg in (abz.MonkhorstPack(), abz.AutoSymPTRJL()):
            sol = abz.solve(abz.IntegralProblem(f, abz.Basis(np.array([[B2PI]])), P), alg, abstol=ABSTOL)
            assert abs(sol.u - ref) <= ABSTOL


def test_batch_refinement_is_the_oracles():
    """Batch-mode auxquadgk makes the oracle's decisions: same numevals and value, bit for bit; max_batch caps a call."""
    calls = []

    def body(y, x, p):
        calls.append(len(x))
        y[:] = [1.0 / (p - math.cos(v) + 0.3 * math.sin(3 * v)) for v in x]

    for mb in (2**62, 60, 30):
        calls.clear()
        sol = abz.solve(abz.IntegralProblem(abz.BatchIntegrand(body, max_batch=mb), (A, B2PI), 1.5),
                        abz.EvalCounter(abz.AuxQuadGKJL()), abstol=1e-9)
        assert max(calls) <= max(mb, 15)
        I, E, nev = orc.auxquadgk(lambda xs: np.array([1.0 / (1.5 - math.cos(v) + 0.3 * math.sin(3 * v)) for v in xs]),
                                  (A, B2PI), atol=1e-9, batch=True, max_batch=mb)
        assert sol.numevals == nev and sol.u == I and sol.resid == E


@pytest.mark.parametrize("dim", [1, 2, 3])
def test_nested_quad_generic_integrands(dim):
    """ref: test/interface_tests.jl:113-130: f(x, p) = 1 + p sum(cos, x) on [0, 2 pi]^dim -> (2 pi)^dim."""
    f = lambda x, p: 1.0 + p * float(np.sum(np.cos(x)))
    p, abstol, ref = 7.0, 1e-3, (2 * math.pi) ** dim
    dom = abz.CubicLimits(np.zeros(dim), 2 * math.pi * np.ones(dim))
    alg = abz.NestedQuad(abz.AuxQuadGKJL())
    assert abs(abz.solve(abz.IntegralProblem(f, dom, p), alg, abstol=abstol).u - ref) <= abstol
    seen = []

    def body(y, x, p):
        seen.append(len(x))
        y[:] = [f(v, p) for v in x]

    sol = abz.solve(abz.IntegralProblem(abz.BatchIntegrand(body), dom, p), abz.EvalCounter(alg), abstol=abstol)
    assert abs(sol.u - ref) <= abstol and sol.numevals == sum(seen) and all(len(np.atleast_1d(0)) for _ in seen)
    nb = abz.NestedBatchIntegrand(tuple(f for _ in range(3)))
    assert abs(abz.solve(abz.IntegralProblem(nb, dom, p), alg, abstol=abstol).u - ref) <= abstol
    # the oracle's nested loop on the same integrand (scalar refinement): same value
    if dim <= 2:
        got = abz.solve(abz.IntegralProblem(f, dom, p), abz.EvalCounter(alg), abstol=abstol)
        assert got.numevals >= 15**dim


def test_evalcounter_constant_integrand():
    """ref: test/interface_tests.jl:143-158: a constant integrand costs exactly the base rule: 15 for GK(7,15)."""
    sol = abz.solve(abz.IntegralProblem(lambda x, p: 1.0, (0.0, 1.0), None), abz.EvalCounter(abz.AuxQuadGKJL()))
    assert sol.numevals == 15 and abs(sol.u - 1.0) < 1e-15
    sol = abz.solve(abz.IntegralProblem(abz.BatchIntegrand(lambda y, x, p: y.__setitem__(slice(None), [1.0] * len(x))),
                                        (0.0, 1.0), None), abz.EvalCounter(abz.AuxQuadGKJL()))
    assert sol.numevals == 15


def test_generic_error_behaviour():
    nb = abz.NestedBatchIntegrand((lambda x, p: 1.0,))
    with pytest.raises(ValueError):  # ref: src/algorithms.jl:211
        abz.solve(abz.IntegralProblem(nb, (0.0, 1.0), None), abz.AuxQuadGKJL())
    with pytest.raises(ValueError):  # ref: src/algorithms.jl:362
        abz.solve(abz.IntegralProblem(nb, abz.Basis(np.eye(1)), None), abz.MonkhorstPack())
    with pytest.raises(ValueError):  # ref: src/batch.jl:16
        abz.BatchIntegrand(lambda y, x, p: None, max_batch=0)
    with pytest.raises(ValueError):
        abz.solve(abz.IntegralProblem(lambda x, p: 1.0, (0.0, 1.0), None), abz.PTR())


def test_absolute_estimate_sizes_the_tolerance_from_a_rough_solve():
    """ref: src/algorithms.jl:614-653, test/interface_tests.jl:132-140 -- f(x) = 1 / (z - cos x), z = 0.5 + 1e-3 i: the
    estimate stage fixes abstol = reltol * |I_est| for the second stage, whose result then equals a direct solve with that
    absolute tolerance; the closed form is 2 pi / (sqrt(z - 1) sqrt(z + 1))."""
    import cmath
    z = complex(0.5, 1e-3)
    prob = abz.IntegralProblem(lambda x, p: 1.0 / (complex(*p) - np.cos(x)), (0.0, 2 * np.pi), (0.5, 1e-3))
    alg = abz.AbsoluteEstimate(abz.AuxQuadGKJL(), abz.QuadGKJL(), abstol=1.0)  # the estimate only needs the magnitude
    reltol = 1e-5
    sol = abz.solve(prob, abz.EvalCounter(alg), reltol=reltol)
    exact = 2 * np.pi / (cmath.sqrt(z - 1) * cmath.sqrt(z + 1))
    assert abs(sol.u - exact) <= 10 * reltol * abs(exact)
    est = abz.solve(prob, abz.EvalCounter(abz.AuxQuadGKJL()), abstol=1.0)
    direct = abz.solve(prob, abz.EvalCounter(abz.QuadGKJL()), abstol=reltol * abs(est.u), reltol=0.0)
    assert sol.u == direct.u and sol.resid == direct.resid and sol.numevals == est.numevals + direct.numevals
    assert sol.resid <= reltol * abs(est.u)
    with pytest.raises(ValueError):
        abz.AbsoluteEstimate(abz.AuxQuadGKJL(), abz.QuadGKJL(), tolerance=1.0)  # checkkwargs


def test_evalcounter_pins_on_a_constant_and_other_kronrod_orders():
    """ref: test/interface_tests.jl:143-158 -- a constant integrand uses exactly the base rule: 15 evaluations for
    QuadGKJL(order = 7), 19 for order = 9 (plain and batch integrands; the trapezoidal QuadratureFunction is out of scope).
    The host-computed rule of order 7 reproduces the library's GK(7,15) table, and order 9 integrates the reference's
    known-answer integrands (test/interface_tests.jl:33-42)."""
    from autobzcore.jl_amd import generic as G
    from autobzcore.jl_amd.hostquad import _gk_nodes
    for order, nev in ((7, 15), (9, 19)):
        sol = abz.solve(abz.IntegralProblem(lambda x, p: 1.0, (0.0, 1.0)), abz.EvalCounter(abz.QuadGKJL(order=order)))
        assert sol.numevals == nev and abs(sol.u - 1.0) < 1e-15
    bsol = abz.solve(abz.IntegralProblem(abz.BatchIntegrand(lambda y, x, p: y.__setitem__(slice(None), [1.0] * len(x)), float), (0.0, 1.0)),
                     abz.EvalCounter(abz.AuxQuadGKJL()))
    assert bsol.numevals == 15 and abs(bsol.u - 1.0) < 1e-15
    r7 = G._GKRule.__new__(G._GKRule)
    r7.order = 7
    num = G._GKRule(6)  # any non-table order exercises the construction; compare the n = 7 construction with the table
    assert abs(num.w.sum() - 2.0) < 1e-14 and np.all(num.w > 0) and num.npts == 13
    r9 = G.gk_rule(9)
    k = np.arange(0, 3 * 9 + 2)  # Kronrod rule of order n is exact to degree 3n + 1
    exact = np.where(k % 2 == 0, 2.0 / (k + 1), 0.0)
    assert np.abs(np.array([np.sum(r9.w * r9.x ** kk) for kk in k]) - exact).max() < 5e-15
    assert np.abs(np.array([np.sum(r9.gw * r9.x[r9.gidx] ** kk) for kk in range(2 * 9)]) - exact[:18]).max() < 5e-15
    a, b, p = 0.0, 2 * np.pi, 3.0
    for f, ref in ((lambda x, p: p * np.sin(x), 0.0), (lambda x, p: p * 1.0, p * (b - a)),
                   (lambda x, p: 1.0 / (p - np.cos(x)), (b - a) / math.sqrt(p * p - 1))):
        for alg in (abz.QuadGKJL(order=9), abz.QuadGKJL(), abz.AuxQuadGKJL()):
            assert abs(abz.solve(abz.IntegralProblem(f, (a, b), p), alg, abstol=1e-5).u - ref) < 1e-5
    assert np.abs(np.sort(_gk_nodes(-1.0, 1.0)) - np.sort(G.gk_rule(7).nodes(-1.0, 1.0))).max() == 0.0


def test_hcubature_known_answers_and_rule_exactness():
    """ref: test/interface_tests.jl:45-59 (HCubatureJL, dims 1..3: p sum sin, p, prod 1/(p - cos)) and the degree of the
    Genz-Malik rule (exact for every monomial of total degree <= 7, the embedded rule for degree <= 5)."""
    from autobzcore.jl_amd import generic as G
    a, b, p, abstol = 0.0, 2 * np.pi, 3.0, 1e-5
    for dim in (1, 2, 3):
        for f, ref in ((lambda x, p: p * np.sum(np.sin(x)), 0.0), (lambda x, p: p * 1.0, p * (b - a) ** dim),
                       (lambda x, p: np.prod(1.0 / (p - np.cos(x))), ((b - a) / math.sqrt(p * p - 1)) ** dim)):
            dom = abz.HyperCube(np.full(dim, a), np.full(dim, b)) if dim > 1 else (a, b)
            sol = abz.solve(abz.IntegralProblem(f, dom, p), abz.EvalCounter(abz.HCubatureJL()), abstol=abstol)
            assert abs(sol.u - ref) <= abstol and sol.resid <= abstol and sol.numevals > 0
    for d in (2, 3, 4):
        pts, w7, w5 = G._genz_malik(d)
        assert len(pts) == 2 ** d + 2 * d * d + 2 * d + 1 and abs(w7.sum() - 1) < 1e-14 and abs(w5.sum() - 1) < 1e-14
        import itertools
        for deg in itertools.product(range(0, 8), repeat=d):
            if sum(deg) > 7:
                continue
            exact = np.prod([0.0 if k % 2 else 1.0 / (k + 1) for k in deg])  # mean of x^k over [-1, 1]
            mono = np.prod(pts ** np.array(deg), axis=1)
            assert abs(np.dot(w7, mono) - exact) < 1e-14, (d, deg)
            if sum(deg) <= 5:
                assert abs(np.dot(w5, mono) - exact) < 1e-14, (d, deg)
    # a constant costs exactly one rule (2^d + 2 d^2 + 2 d + 1 points; 15 in one dimension)
    for dim, nev in ((1, 15), (2, 17), (3, 33)):
        dom = abz.HyperCube(np.zeros(dim), np.ones(dim)) if dim > 1 else (0.0, 1.0)
        assert abz.solve(abz.IntegralProblem(lambda x, p: 1.0, dom), abz.EvalCounter(abz.HCubatureJL())).numevals == nev
    with pytest.raises(ValueError):
        abz.solve(abz.IntegralProblem(abz.BatchIntegrand(lambda y, x, p: None, float), (0.0, 1.0)), abz.HCubatureJL())


def test_inplace_integrands_quadrature_function_and_nested_fixed_rules():
    """ref: test/interface_tests.jl:27-43 (QuadratureFunction beside the adaptive rules), :67-88 (InplaceIntegrand under
    QuadratureFunction / QuadGKJL / AuxQuadGKJL / HCubatureJL / MonkhorstPack / AutoSymPTRJL), :113-130 (NestedQuad of a
    fixed rule, plain and in-place), :143-158 (a constant costs exactly npt = 10 evaluations of trapz(10))."""
    a, b, p, abstol = 0.0, 2 * np.pi, 3.0, 1e-5
    cases = ((lambda x: p * np.sin(x), 0.0), (lambda x: p * 1.0, p * (b - a)), (lambda x: 1.0 / (p - np.cos(x)), (b - a) / math.sqrt(p * p - 1)))
    for g, ref in cases:
        plain = abz.IntegralProblem(lambda x, q, g=g: g(x), (a, b), p)
        assert abs(abz.solve(plain, abz.QuadratureFunction(), abstol=abstol).u - ref) < abstol

        def body(y, x, q, g=g):
            y[...] = g(np.ravel(x)[0])
        integrand = abz.InplaceIntegrand(body, np.array([0.0]))
        for alg in (abz.QuadratureFunction(), abz.QuadGKJL(), abz.AuxQuadGKJL(), abz.HCubatureJL()):
            u = abz.solve(abz.IntegralProblem(integrand, (a, b), p), alg, abstol=abstol).u
            assert np.shape(u) == (1,) and abs(u[0] - ref) < abstol, type(alg).__name__
        for alg in (abz.MonkhorstPack(), abz.AutoSymPTRJL()):
            u = abz.solve(abz.IntegralProblem(integrand, abz.Basis(np.array([[b]])), p), alg, abstol=abstol).u
            assert np.shape(u) == (1,) and abs(u[0] - ref) < abstol, type(alg).__name__
    x, w = abz.trapz(5)
    assert np.allclose(x, [-1, -0.5, 0, 0.5, 1]) and np.allclose(w, [0.25, 0.5, 0.5, 0.5, 0.25])
    f = lambda x, q: 1.0 + q * np.sum(np.cos(x))
    for dim in (1, 2, 3):
        dom = abz.CubicLimits(np.zeros(dim), 2 * np.pi * np.ones(dim))
        ref = (2 * np.pi) ** dim
        nd = abz.NestedQuad(abz.QuadratureFunction())
        assert abs(abz.solve(abz.IntegralProblem(f, dom, 7.0), nd, abstol=1e-3).u - ref) < 1e-3
        inpl = abz.InplaceIntegrand(lambda y, x, q: y.__setitem__(Ellipsis, f(x, q)), np.array([0.0]))
        assert abs(abz.solve(abz.IntegralProblem(inpl, dom, 7.0), nd, abstol=1e-3).u[0] - ref) < 1e-3
        sol = abz.solve(abz.IntegralProblem(f, dom, 7.0), abz.EvalCounter(nd))
        assert sol.numevals == 50 ** dim
        mixed = abz.NestedQuad(abz.AuxQuadGKJL(), abz.QuadratureFunction(npt=40))  # adaptive outside, fixed rule inside
        assert abs(abz.solve(abz.IntegralProblem(f, dom, 7.0), mixed, abstol=1e-3).u - ref) < 1e-3
    for prob in (abz.IntegralProblem(lambda x, q: 1.0, (0.0, 1.0)),
                 abz.IntegralProblem(abz.InplaceIntegrand(lambda y, x, q: y.__setitem__(Ellipsis, 1.0), np.zeros(())), (0.0, 1.0)),
                 abz.IntegralProblem(abz.BatchIntegrand(lambda y, x, q: y.__setitem__(slice(None), [1.0] * len(x)), float), (0.0, 1.0))):
        assert abz.solve(prob, abz.EvalCounter(abz.QuadratureFunction(npt=10))).numevals == 10


def test_contour_deformation_and_pole_subtraction_quadratures():
    """ref: test/interface_tests.jl:27-43 (ContQuadGKJL and MeroQuadGKJL integrate the three known-answer integrands like the
    other 1-D rules) and :132-140 (f = 1 / (z - cos x), z = 0.5 + 1e-3 i, with MeroQuadGKJL as the reference there).  Near
    a pole both need far fewer evaluations than plain quadgk and still give the closed forms."""
    import cmath
    a, b, p, abstol = 0.0, 2 * np.pi, 3.0, 1e-5
    for f, ref in ((lambda x, q: q * np.sin(x), 0.0), (lambda x, q: q * (1.0 + 0 * x), p * (b - a)),
                   (lambda x, q: 1.0 / (q - np.cos(x)), (b - a) / math.sqrt(p * p - 1))):
        for alg in (abz.ContQuadGKJL(), abz.MeroQuadGKJL()):
            assert abs(abz.solve(abz.IntegralProblem(f, (a, b), p), alg, abstol=abstol).u - ref) < abstol, type(alg).__name__
    z = complex(0.5, 1e-3)
    prob = abz.IntegralProblem(lambda x, q: 1.0 / (complex(*q) - np.cos(x)), (0.0, 2 * np.pi), (0.5, 1e-3))
    exact = 2 * np.pi / (cmath.sqrt(z - 1) * cmath.sqrt(z + 1))
    plain = abz.solve(prob, abz.EvalCounter(abz.QuadGKJL()), abstol=1e-8)
    for alg in (abz.MeroQuadGKJL(), abz.ContQuadGKJL()):
        sol = abz.solve(prob, abz.EvalCounter(alg), abstol=1e-8)
        assert abs(sol.u - exact) < 1e-7 and sol.resid <= 1e-8, type(alg).__name__
        assert sol.numevals < plain.numevals, (type(alg).__name__, sol.numevals, plain.numevals)
    # a single simple pole 1e-4 above the axis: the exact answer is a logarithm
    z0 = complex(0.3, 1e-4)
    pole = abz.IntegralProblem(lambda x, q: 1.0 / (x - z0), (-1.0, 1.0))
    exact = cmath.log(1 - z0) - cmath.log(-1 - z0)
    ref = abz.solve(pole, abz.EvalCounter(abz.QuadGKJL()), abstol=1e-9)
    for alg in (abz.MeroQuadGKJL(), abz.ContQuadGKJL()):
        sol = abz.solve(pole, abz.EvalCounter(alg), abstol=1e-9)
        assert abs(sol.u - exact) < 1e-8 and sol.numevals <= ref.numevals // 4, (type(alg).__name__, sol.numevals, ref.numevals)
    for alg in (abz.ContQuadGKJL(), abz.MeroQuadGKJL()):
        with pytest.raises(ValueError):
            abz.solve(abz.IntegralProblem(abz.BatchIntegrand(lambda y, x, q: None, float), (0.0, 1.0)), alg)
        with pytest.raises(ValueError):
            abz.solve(abz.IntegralProblem(abz.InplaceIntegrand(lambda y, x, q: None, np.zeros(1)), (0.0, 1.0)), alg)


def test_quickstart_doc_values():
    """The reference's printed quickstart outputs, docs/src/problems.md:18-25 (`QuadGKJL()`, default tolerances):
    0.14887836958131329 at p = 0.3, then the cache re-used at p = 0.4: 0.1973475149927873 -- one GK(7,15) panel each,
    so these pin the rule table of the product's host loop to the last bit or two of a 15-term sum."""
    prob = abz.IntegralProblem(lambda x, p: math.sin(p * x), (0.0, 1.0), 0.3)
    cache = abz.init(prob, abz.QuadGKJL())
    sol = abz.solve_(cache)
    assert abs(sol.u - 0.14887836958131329) <= 1e-16
    cache.p = 0.4
    assert abs(abz.solve_(cache).u - 0.1973475149927873) <= 1e-16
    sol = abz.solve(prob, abz.EvalCounter(abz.QuadGKJL()))
    assert sol.numevals == 15
