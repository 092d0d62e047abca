"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Floating-point tolerance (BASELINE north_star: "integrals within the solver's own abstol/reltol"):
series values / eigenvalues agree to 1e-12 * ||H||; integrals to the solver tolerance (and, where
both sides run the same deterministic rule, to 1e-11 relative).  Integer data (irreducible nodes,
weights, panel trees, evaluation counts) must match bit for bit.
"""
import json
import math
import os

import numpy as np
import pytest

import abz_oracle as orc

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def abz():
    import autobzcore.jl_amd as m
    return m


def rand_series(rng, dims, n, first=None, period=1.0, hermitian=False):
    d = len(dims)
    c = rng.standard_normal(dims + (n, n)) + 1j * rng.standard_normal(dims + (n, n))
    first = first if first is not None else tuple(-(m // 2) for m in dims)
    if hermitian:
        assert all(m % 2 == 1 for m in dims)
        flip = c[tuple(slice(None, None, -1) for _ in dims)]
        c = 0.5 * (c + np.conj(np.swapaxes(flip, -1, -2)))
    return c, first


def both(abz, c, first, period=1.0, ndim=None):
    return (abz.FourierSeries(c, period=period, first=first, ndim=ndim),
            orc.FourierSeries(c, period=period, first=first, ndim=ndim))


# ------------------------------------------------------------------ series evaluation
@pytest.mark.parametrize("d,n", [(1, 1), (1, 3), (2, 2), (3, 1), (3, 3), (3, 4)])
def test_eval_nodes_matches_oracle(abz, d, n):
    rng = np.random.default_rng(10 * d + n)
    dims = (3, 4, 5)[:d]
    c, first = rand_series(rng, dims, n, first=(-1, -2, 0)[:d])
    period = (1.0, 2.0, 0.5)[:d]
    s, so = both(abz, c, first, period, ndim=d)
    k = rng.uniform(-1, 2, size=(257, d))
    k[5] = k[4]  # shared outer coordinates exercise the run logic
    if d > 1:
        k[7, 1:] = k[6, 1:]
    H, E = s.device().eval_nodes(k, want=3)
    Ho = orc.evaluate_many(so, k)
    scale = np.abs(Ho).max()
    assert np.abs(H - Ho).max() <= 1e-12 * scale
    Eo = np.linalg.eigvalsh(Ho, UPLO="U")
    assert np.abs(E - Eo).max() <= 1e-12 * scale


def test_eig3_degenerate_and_clustered(abz):
    """3x3 Hermitian eigenvalues at LAPACK accuracy also for exactly / nearly degenerate spectra
    (the closed form alone would lose half the digits there).  Constant series: H(k) = c[0]."""
    rng = np.random.default_rng(42)
    mats = []
    for gap in (0.0, 1e-14, 1e-10, 1e-7, 1e-4, 1e-2, 1.0):
        for trip in range(6):
            q, _ = np.linalg.qr(rng.standard_normal((3, 3)) + 1j * rng.standard_normal((3, 3)))
            base = rng.uniform(-5, 5)
            for ev in ([base, base + gap, base + 3.0], [base - 2.0, base, base + gap], [base, base + gap, base + 2 * gap],
                       [base, base, base]):
                mats.append((q * np.array(ev)) @ q.conj().T)
    mats.append(np.diag([1.0, 1.0, 2.0]).astype(complex))
    mats.append(np.diag([3.0, -1.0, -1.0]).astype(complex))
    mats.append(np.zeros((3, 3), dtype=complex))
    for A in mats:
        A = 0.5 * (A + A.conj().T)
        s = abz.FourierSeries(A.reshape(1, 3, 3), first=0, ndim=1)
        H, E = s.device().eval_nodes(np.array([[0.3]]), want=3)
        ref = np.linalg.eigvalsh(A)
        assert np.abs(E[0] - ref).max() <= 1e-12 * max(1.0, np.abs(A).max()), (E[0], ref)


def test_eig4_closed_form_degenerate_and_clustered(abz):
    """4x4 Hermitian eigenvalues: characteristic-polynomial route (Ferrari + Newton) for separated spectra, Jacobi for
    clustered ones -- LAPACK accuracy over gaps 0 ... 1, every degeneracy pattern, and 4000 random matrices."""
    rng = np.random.default_rng(44)
    mats = []
    pats = ([0, 1, 2, 3], [0, 0, 1, 2], [0, 1, 1, 2], [0, 1, 2, 2], [0, 0, 1, 1], [0, 0, 0, 1], [0, 1, 1, 1], [0, 0, 0, 0])
    for gap in (0.0, 1e-14, 1e-10, 1e-7, 1e-5, 1e-4, 1e-3, 3e-3, 1e-2, 0.1, 1.0):
        for pat in pats:
            q, _ = np.linalg.qr(rng.standard_normal((4, 4)) + 1j * rng.standard_normal((4, 4)))
            base = rng.uniform(-5, 5, size=4)
            base.sort()
            lv = np.array([base[pat[i]] + gap * i for i in range(4)])
            mats.append((q * lv) @ q.conj().T)
    mats += [np.diag([1.0, 1.0, 2.0, 2.0]).astype(complex), np.diag([3.0, -1.0, -1.0, 0.5]).astype(complex), np.zeros((4, 4), dtype=complex)]
    for A in mats:
        A = 0.5 * (A + A.conj().T)
        s = abz.FourierSeries(A.reshape(1, 4, 4), first=0, ndim=1)
        H, E = s.device().eval_nodes(np.array([[0.3]]), want=3)
        ref = np.linalg.eigvalsh(A)
        assert np.abs(E[0] - ref).max() <= 1e-12 * max(1.0, np.abs(A).max()), (E[0], ref)
    # random spectra through a 1-D series: H(x) = A0 + 2 Re(A1 e^{2 pi i x}) at 4000 points, and a PTR rule (grid kernel)
    c, first = rand_series(rng, (3,), 4, hermitian=True)
    s, so = both(abz, c, first)
    x = rng.uniform(0, 1, size=(4000, 1))
    H, E = s.device().eval_nodes(x, want=3)
    ref = np.linalg.eigvalsh(H)
    assert np.abs(E - ref).max() <= 1e-12 * np.abs(H).max()
    out = s.device().rule(2000, None, want=3).export(H=True, eig=True)
    assert np.abs(out["eig"] - np.linalg.eigvalsh(out["H"])).max() <= 1e-12 * np.abs(out["H"]).max()


def test_eval_nodes_edge_cases(abz):
    rng = np.random.default_rng(0)
    c, first = rand_series(rng, (3, 3), 2)
    s, so = both(abz, c, first)
    assert s.device().eval_nodes(np.zeros((0, 2))).shape[0] == 0  # empty batch
    one = s.device().eval_nodes(np.array([[0.25, 0.75]]))
    assert np.allclose(one[0], orc.evaluate(so, [0.25, 0.75]), atol=1e-13)
    assert np.allclose(s([0.25, 0.75]), orc.evaluate(so, [0.25, 0.75]), atol=1e-13)


# ------------------------------------------------------------------ PTR rules
@pytest.mark.parametrize("d,n,npt", [(1, 1, 50), (2, 2, 17), (3, 1, 12), (3, 3, 10), (3, 4, 6), (3, 3, 64)])
def test_full_grid_rule_matches_oracle(abz, d, n, npt):
    rng = np.random.default_rng(100 + d + n)
    dims = (5, 3, 7)[:d]
    c, first = rand_series(rng, dims, n, hermitian=True)
    s, so = both(abz, c, first)
    rule = s.device().rule(npt, None, want=1 | 2)
    out = rule.export(H=True, eig=True)
    vals = orc.fourier_ptr(so, npt)  # [i1..id, n, n]
    perm = tuple(range(d - 1, -1, -1))
    ref = np.transpose(vals, perm + (d, d + 1)).reshape(-1, n, n)  # column-major node order
    scale = np.abs(ref).max()
    assert out["H"].shape == ref.shape
    assert np.abs(out["H"] - ref).max() <= 1e-12 * scale
    assert np.abs(out["eig"] - np.linalg.eigvalsh(ref, UPLO="U")).max() <= 1e-12 * scale
    # nodes in column-major order, unit weights
    x = orc.ptrpoints(npt)
    grids = np.meshgrid(*([x] * d), indexing="ij")
    xl = np.stack([np.transpose(g, perm).reshape(-1) for g in grids], axis=1)
    assert np.array_equal(out["x"], xl) and np.all(out["w"] == 1.0)


@pytest.mark.parametrize("kind,d,npt", [("InversionSymIBZ", 1, 9), ("InversionSymIBZ", 2, 8), ("InversionSymIBZ", 3, 7),
                                        ("CubicSymIBZ", 2, 9), ("CubicSymIBZ", 3, 8), ("CubicSymIBZ", 3, 50)])
def test_symmetric_rule_integers_and_values(abz, kind, d, npt):
    so = orc.tb_integer(d)
    s = abz.FourierSeries(so.c, period=1.0, first=so.first, ndim=d)
    bzo = orc.load_bz(kind, np.eye(d))
    idx, w = abz.symptr_rule(npt, d, bzo.syms)
    wo, xo, valo, idxo = orc.fourier_symptr(so, npt, bzo.syms)
    assert np.array_equal(idx, idxo) and np.array_equal(w, wo)  # integer parity, bit exact
    rule = s.device().rule(npt, bzo.syms, want=1)
    out = rule.export(H=True)
    assert np.array_equal(out["w"], wo.astype(float)) and np.array_equal(out["x"], xo)
    assert np.abs(out["H"] - valo).max() <= 1e-12 * np.abs(valo).max()


def test_symptr_rule_device_bit_exact(abz):
    """the orbit tables computed on the GPU equal the host routine bit for bit."""
    ctx = abz.Context.default()
    for kind, d, npt in (("InversionSymIBZ", 1, 9), ("InversionSymIBZ", 2, 10), ("CubicSymIBZ", 2, 9), ("CubicSymIBZ", 3, 8),
                         ("CubicSymIBZ", 3, 50), ("InversionSymIBZ", 3, 33), ("CubicSymIBZ", 3, 150)):
        syms = orc.load_bz(kind, np.eye(d)).syms
        ih, wh = abz.symptr_rule(npt, d, syms)
        ig, wg = abz.symptr_rule(npt, d, syms, ctx=ctx)
        assert np.array_equal(ih, ig) and np.array_equal(wh, wg) and wg.sum() == npt**d
    # a smaller point group (C4 about z with the mirror z -> -z): 8 elements, closed under products
    r = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1]])
    m = np.diag([1, 1, -1])
    syms = [np.linalg.matrix_power(r, k) @ mm for k in range(4) for mm in (np.eye(3, dtype=int), m)]
    ih, wh = abz.symptr_rule(9, 3, syms)
    ig, wg = abz.symptr_rule(9, 3, syms, ctx=ctx)
    assert np.array_equal(ih, ig) and np.array_equal(wh, wg) and wg.sum() == 9**3


def test_rule_reduce_matches_oracle_dos(abz):
    rng = np.random.default_rng(5)
    c, first = rand_series(rng, (3, 3, 3), 3, hermitian=True)
    s, so = both(abz, c, first)
    omegas = np.linspace(-3, 3, 7)
    eta = 0.2
    for syms, name in ((None, "FBZ"),):
        rule = s.device().rule(12, syms, want=3)
        got = rule.reduce(abz._lib.F_DOS, [eta], omegas)[:, 0].real
        got_e = rule.reduce(abz._lib.F_DOS_EIG, [eta], omegas)[:, 0].real
        for i, om in enumerate(omegas):
            ref, _ = orc._ptr_rule_sum(so, 12, syms, orc.f_dos(eta, om))
            assert abs(got[i] - ref) <= 1e-11 * abs(ref)
            assert abs(got_e[i] - ref) <= 1e-10 * abs(ref)
        g = rule.reduce(abz._lib.F_GLOC, [eta], omegas[:2])
        for i in range(2):
            ref, _ = orc._ptr_rule_sum(so, 12, syms, orc.f_gloc(eta, omegas[i]))
            assert np.abs(g[i].reshape(3, 3).T - ref).max() <= 1e-11 * np.abs(ref).max()
        t = rule.reduce(abz._lib.F_TRGLOC, [eta], omegas[:2])[:, 0]
        for i in range(2):
            ref, _ = orc._ptr_rule_sum(so, 12, syms, orc.f_gloc(eta, omegas[i]))
            assert abs(t[i] - np.trace(ref)) <= 1e-11 * abs(np.trace(ref))


def test_dos3_sweep_kernel_against_oracle_and_generic_scan(abz, monkeypatch):
    """The 3-band DOS sweep kernel (dos3_scan_kernel: real cubic / one-reciprocal eigenvalue form, transposed LDS
    reduction, one reciprocal per pair of nodes) against the oracle's quadsum (src/fourier.jl:204-207 with
    the DOS integrand) and against the generic scan (ABZ_DOS3_SCAN=0): full grid and symmetric rule (weights), sweep
    lengths around the 16-value tiles and beyond one 256-value pass, forced sweep rows,
    small eta (poles close to the axis)."""
    from autobzcore.jl_amd import _lib as L
    rng = np.random.default_rng(77)
    c, first = rand_series(rng, (3, 3, 3), 3, hermitian=True)
    s, so = both(abz, c, first)
    cubic = abz.load_bz(abz.CubicSymIBZ(), np.eye(3)).syms
    for syms, npt in ((None, 13), (cubic, 14), (None, 1), (None, 2), (cubic, 3)):  # incl. rules of 1 ... 8 nodes
        rule = s.device().rule(npt, syms, want=3)
        for eta in (0.2, 1e-3):
            for nw in (1, 15, 16, 17, 33, 300):
                omegas = np.sort(rng.uniform(-3.5, 3.5, size=nw))
                for fid, tol in ((L.F_DOS, 2e-12), (L.F_DOS_EIG, 1e-10)):
                    a = rule.reduce(fid, [eta], omegas)[:, 0]
                    monkeypatch.setenv("ABZ_DOS3_SCAN", "0")
                    b = rule.reduce(fid, [eta], omegas)[:, 0]
                    monkeypatch.delenv("ABZ_DOS3_SCAN")
                    assert np.all(a.imag == 0) and np.isfinite(a.real).all()
                    assert np.abs(a - b).max() <= tol * np.abs(b).max(), (npt, eta, nw, fid)
        omegas = np.linspace(-3, 3, 37)
        ref = np.array([orc._ptr_rule_sum(so, npt, syms, orc.f_dos(0.2, om))[0] for om in omegas[::9]])
        base = rule.reduce(L.F_DOS, [0.2], omegas)[:, 0].real
        assert np.abs(base[::9] - ref).max() <= 1e-11 * np.abs(ref).max()
        for rows in (2, 3):
            monkeypatch.setenv("ABZ_REDUCE_ROWS", str(rows))
            for fid in (L.F_DOS, L.F_DOS_EIG):
                got = rule.reduce(fid, [0.2], omegas)[:, 0].real
                assert np.abs(got - base).max() <= (1e-12 if fid == L.F_DOS else 1e-10) * np.abs(base).max(), (rows, fid)
            monkeypatch.delenv("ABZ_REDUCE_ROWS")
        rule.close()


@pytest.mark.parametrize("n", [2, 3, 6, 12, 20, 32])
def test_rule_reduce_non_hermitian_series(abz, n):
    """A series that is NOT Hermitian (e.g. H + a k-dependent self-energy) takes the general paths:
    full-matrix Fourier evaluation and the complex characteristic polynomial in the scan.  The decay of
    the anti-Hermitian part keeps (omega + i eta) - H invertible."""
    rng = np.random.default_rng(50 + n)
    c, first = rand_series(rng, (3, 3, 3), n, hermitian=True)
    extra, _ = rand_series(rng, (3, 3, 3), n, hermitian=False)
    c = c + 0.05 * extra
    s, so = both(abz, c, first)
    omegas = np.array([-1.5, 0.2, 2.0])
    eta = 0.6
    rule = s.device().rule(9, None, want=1)
    got = rule.reduce(abz._lib.F_DOS, [eta], omegas)[:, 0].real
    tr = rule.reduce(abz._lib.F_TRGLOC, [eta], omegas)[:, 0]
    g = rule.reduce(abz._lib.F_GLOC, [eta], omegas[:1])
    if True:  # store-free sums through the inverse of every node (kernels_big.hip: 8 / 4 / 2 nodes per wave up to 8 / 16 / 32 bands)
        gsf = s.device().ptr_sum(9, abz._lib.F_GLOC, [eta], omegas[:1])
        dsf = s.device().ptr_sum(9, abz._lib.F_DOS, [eta], omegas)[:, 0].real
        assert np.abs(gsf - g).max() <= 1e-11 * np.abs(g).max() and np.abs(dsf - got).max() <= 1e-11 * np.abs(got).max()
        sh, soh = both(abz, c - 0.05 * extra, first)  # the Hermitian part alone: G store-free against the scan of its rule
        rh = sh.device().rule(9, None, want=1)
        gh = rh.reduce(abz._lib.F_GLOC, [eta], omegas[:2])
        rh.close()
        ghs = sh.device().ptr_sum(9, abz._lib.F_GLOC, [eta], omegas[:2])
        assert np.abs(ghs - gh).max() <= 1e-11 * np.abs(gh).max()
    for i, om in enumerate(omegas):
        ref, _ = orc._ptr_rule_sum(so, 9, None, orc.f_gloc(eta, om))
        assert abs(tr[i] - np.trace(ref)) <= 1e-11 * abs(np.trace(ref))
        assert abs(got[i] + np.trace(ref).imag / np.pi) <= 1e-11 * abs(np.trace(ref))
        if i == 0:
            assert np.abs(g[0].reshape(n, n).T - ref).max() <= 1e-11 * np.abs(ref).max()
    # and the same through IAI (device-side inner loops on non-Hermitian values); 2-D keeps the pure-Python
    # oracle at a fraction of a second
    c2, first2 = rand_series(rng, (3, 3), n, hermitian=True)
    extra2, _ = rand_series(rng, (3, 3), n, hermitian=False)
    s2, so2 = both(abz, (c2 + 0.05 * extra2) / 2, first2)
    bz = abz.load_bz(abz.FBZ(), np.eye(2))
    sol = abz.do_solve(abz.FourierIntegrand(abz.TrGlocIntegrand(), s2, eta), bz, abz.MixedParameters(0.2),
                       abz.EvalCounter(abz.IAI()), abstol=1e-3)
    f_tr = lambda x, h: np.trace(orc.f_gloc(eta, 0.2)(x, h), axis1=-2, axis2=-1)
    ref = orc.solve_iai(so2, orc.load_bz("FBZ", np.eye(2)), f_tr, abstol=1e-3)
    assert sol.numevals == ref.numevals and abs(sol.u - ref.u) <= 1e-9 * abs(ref.u)


# ------------------------------------------------------------------ reference's own hot-path tests
@pytest.mark.parametrize("d", [1, 2, 3])
@pytest.mark.parametrize("kind", ["FBZ", "InversionSymIBZ"])
def test_bz_algorithms_linear_integrand(abz, d, kind):
    """ref: test/fourier.jl:40-56 -- f = 1.3 s + 1 integrates to (2 pi)^d for IAI / PTR / AutoPTR,
    with and without EvalCounter, abstol 1e-6, reltol 0; plus agreement with the oracle."""
    so = orc.integer_lattice(d)
    s = abz.FourierSeries(so.c[..., 0, 0], period=1.0, first=so.first, ndim=d)
    bz = abz.load_bz({"FBZ": abz.FBZ(), "InversionSymIBZ": abz.InversionSymIBZ()}[kind], np.eye(d))
    bzo = orc.load_bz(kind, np.eye(d))
    vol = (2 * np.pi) ** d
    integrand = abz.FourierIntegrand(abz.LinearIntegrand(), s, 1.3, b=1.0)
    prob = abz.IntegralProblem(integrand, bz)
    fo = orc.f_linear(1.3, 1.0)
    refs = {"IAI": orc.solve_iai(so, bzo, fo, abstol=1e-6, reltol=0.0),
            "PTR": orc.solve_ptr(so, bzo, fo, npt=50),
            "AutoPTR": orc.solve_autoptr(so, bzo, fo, abstol=1e-6, reltol=0.0)}
    for name, alg in (("IAI", abz.IAI()), ("PTR", abz.PTR()), ("AutoPTR", abz.AutoPTR())):
        for counter in (False, True):
            solver = abz.IntegralSolver(prob, abz.EvalCounter(alg) if counter else alg, reltol=0, abstol=1e-6)
            u = solver()
            assert abs(u - vol) < 1e-6
            assert abs(u - refs[name].u) < 1e-9
        sol = abz.solve(prob, abz.EvalCounter(alg), reltol=0, abstol=1e-6)
        assert sol.numevals == refs[name].numevals  # integer parity


def test_parameter_passing_equivalence(abz):
    """ref: test/fourier.jl:9-22 -- three ways of passing parameters give identical results."""
    for d in (1, 2, 3):
        so = orc.integer_lattice(d)
        s = abz.FourierSeries(so.c[..., 0, 0], period=1.0, first=so.first, ndim=d)
        dom = abz.CubicLimits(np.zeros(d), np.ones(d))
        f = abz.LinearXIntegrand()
        alg = abz.NestedQuad(abz.AuxQuadGKJL())
        u = abz.IntegralSolver(abz.IntegralProblem(abz.FourierIntegrand(f, s, 1.3, b=4.2), dom), alg)()
        v = abz.IntegralSolver(abz.FourierIntegrand(f, s), dom, alg)(1.3, b=4.2)
        w = abz.IntegralSolver(abz.FourierIntegrand(f, s, b=4.2), dom, alg)(1.3)
        assert np.array_equal(u, v) and np.array_equal(v, w)
        assert np.allclose(u, 4.2, atol=1e-7)
        # MonkhorstPack on Basis(I) == oracle PTR rule
        m = abz.solve(abz.IntegralProblem(abz.FourierIntegrand(f, s, 1.3, b=4.2), abz.Basis(np.eye(d))), abz.MonkhorstPack()).u
        ref, _ = orc._ptr_rule_sum(so, 50, None, orc.f_linear_x(1.3, 4.2))
        assert np.abs(m - ref).max() < 1e-12


def test_tai_and_hcubature_of_fourier_integrands(abz, svo):
    """The rest of the reference's own hot-path tests: TAI beside IAI / PTR / AutoPTR (test/fourier.jl:40-56: f = 1.3 s + 1
    integrates to (2 pi)^d on the FBZ and the inversion IBZ, with and without EvalCounter) and the three ways of passing
    parameters under HCubatureJL (test/fourier.jl:9-22; the analytic value of a s x + b is b per component).  The tree is
    on the host, every box's series values come from one abz_eval_nodes call (the fallback evaluator, src/fourier.jl:120-122);
    a device integrand and the same Python closure give the same number, and TAI agrees with IAI on the SVO DOS."""
    for d in (1, 2, 3):
        so = orc.integer_lattice(d)
        s = abz.FourierSeries(so.c[..., 0, 0], period=1.0, first=so.first, ndim=d)
        vol = (2 * np.pi) ** d
        for bzk in (abz.FBZ(), abz.InversionSymIBZ()):
            bz = abz.load_bz(bzk, np.eye(d))
            prob = abz.IntegralProblem(abz.FourierIntegrand(abz.LinearIntegrand(), s, 1.3, b=1.0), bz)
            for counter in (False, True):
                alg = abz.EvalCounter(abz.TAI()) if counter else abz.TAI()
                sol = abz.solve(prob, alg, reltol=0, abstol=1e-6)
                assert abs(sol.u - vol) < 1e-6 and sol.resid <= 1e-6
                assert (sol.numevals > 0) if counter else (sol.numevals == -1)
            closure = abz.IntegralProblem(abz.FourierIntegrand(lambda x, a, b: a * x.s + b, s, 1.3, b=1.0), bz)
            assert abs(abz.solve(closure, abz.TAI(), reltol=0, abstol=1e-6).u - vol) < 1e-6
        cube = abz.HyperCube(np.zeros(d), np.ones(d))
        f = abz.LinearXIntegrand()
        alg = abz.HCubatureJL()
        u = abz.IntegralSolver(abz.IntegralProblem(abz.FourierIntegrand(f, s, 1.3, b=4.2), cube), alg)()
        v = abz.IntegralSolver(abz.FourierIntegrand(f, s), cube, alg)(1.3, b=4.2)
        w = abz.IntegralSolver(abz.FourierIntegrand(f, s, b=4.2), cube, alg)(1.3)
        assert np.array_equal(u, v) and np.array_equal(v, w) and np.allclose(u, 4.2, atol=1e-7)
    s3, _ = svo
    bz = abz.load_bz(abz.InversionSymIBZ(), 3.85856 * np.eye(3))
    prob = abz.IntegralProblem(abz.FourierIntegrand(abz.DOSIntegrand(), s3, 0.5), bz, abz.MixedParameters(12.5))
    tai = abz.solve(prob, abz.EvalCounter(abz.TAI()), abstol=1e-3)
    iai = abz.solve(prob, abz.IAI(), abstol=1e-4)
    assert abs(tai.u - iai.u) < 2e-3 and tai.resid <= 1e-3 and tai.numevals % 33 == 0


def test_user_closure_matches_device_integrand(abz):
    """A Python closure (host path, H(k) exported) and the fused device integrand agree."""
    rng = np.random.default_rng(3)
    c, first = rand_series(rng, (3, 3), 2, hermitian=True)
    s = abz.FourierSeries(c, first=first, ndim=2)
    bz = abz.load_bz(abz.FBZ(), np.eye(2))

    def dos(h_k, eta, omega):
        return -np.imag(np.trace(np.linalg.inv((omega + 1j * eta) * np.eye(2) - h_k.s))) / np.pi

    for alg, tol in ((abz.PTR(npt=20), 1e-11), (abz.IAI(), 1e-6)):
        a = abz.IntegralSolver(abz.FourierIntegrand(dos, s, 0.3), bz, alg, abstol=1e-5)(0.4)
        b = abz.IntegralSolver(abz.FourierIntegrand(abz.DOSIntegrand(), s, 0.3), bz, alg, abstol=1e-5)(0.4)
        assert abs(a - b) <= tol * max(1.0, abs(b))


def test_greens_function_doc_values(abz):
    """The reference's own printed outputs (docs/src/examples.md:57-60, :101-106), results converged to 1e-3 only, matched
    to 1e-13 with the evaluation counts the oracle's restatement needs for them (285 / 21285): the HIP path walks the same
    panel tree as the run that printed them (GK(7,15) table, error norm, heap order, termination, `abstol/(|det B| nsyms)`
    and the nested `abstol/len`)."""
    s1 = abz.FourierSeries([0.5, 0.0, 0.5], period=1, offset=-2)
    # the doc's 1-D example as written: QuadGKJL over [0, 1] of a ParameterIntegrand that evaluates h(k) itself
    calls = [0]

    def gloc_integrand(k, h, eta, omega):
        calls[0] += 1
        return 1.0 / (complex(omega, eta) - np.asarray(h(k)).reshape(-1)[0])
    prob = abz.IntegralProblem(abz.ParameterIntegrand(gloc_integrand, s1, eta=0.1), (0.0, 1.0))
    g = abz.IntegralSolver(prob, abz.QuadGKJL(), abstol=1e-3)
    u = g(omega=0.0)
    assert abs(u.imag - (-0.9950375451895513)) <= 1e-13 and abs(u.real) <= 1e-13
    assert calls[0] == 285
    # ... and as a FourierIntegrand on the 1-D BZ with B = 1 (device path, counted evaluations)
    bz1 = abz.load_bz(abz.FBZ(1), [[2 * np.pi]])
    f1 = abz.FourierIntegrand(abz.GlocIntegrand(), s1, eta=0.1)
    sol = abz.do_solve(f1, bz1, abz.MixedParameters(omega=0.0), abz.EvalCounter(abz.IAI()), abstol=1e-3)
    u = np.asarray(sol.u).reshape(-1)[0]
    assert abs(u.imag - (-0.9950375451895513)) <= 1e-13 and abs(u.real) <= 1e-13
    assert sol.numevals == 285
    c2 = np.array([[0.0, 0.5, 0.0], [0.5, 0.0, 0.5], [0.0, 0.5, 0.0]])
    s2 = abz.FourierSeries(c2, period=1, offset=-2)
    bz2 = abz.load_bz(abz.FBZ(2), 2 * np.pi * np.eye(2))
    f2 = abz.FourierIntegrand(abz.GlocIntegrand(), s2, eta=0.1)
    sol = abz.do_solve(f2, bz2, abz.MixedParameters(omega=0.0), abz.EvalCounter(abz.IAI()), abstol=1e-3)
    u = np.asarray(sol.u).reshape(-1)[0]
    assert abs(u.imag - (-1.3941704019631334)) <= 1e-13 and abs(u.real) <= 1e-13
    assert sol.numevals == 21285
    g = abz.IntegralSolver(abz.IntegralProblem(f2, bz2), abz.IAI(), abstol=1e-3)  # the doc's call, verbatim
    u = np.asarray(g(omega=0.0)).reshape(-1)[0]
    assert abs(u.imag - (-1.3941704019631334)) <= 1e-13


def test_iai_panel_tree_bit_exact(abz):
    """North star: 'integer panel indices bit-exact'.  The outermost panel tree of the GPU-batched
    IAI equals the oracle's depth-first tree; all panels are dyadic."""
    rng = np.random.default_rng(11)
    c, first = rand_series(rng, (5, 5), 2, hermitian=True)
    s, so = both(abz, c, first)
    for kind in ("FBZ", "InversionSymIBZ", "CubicSymIBZ"):
        bz = abz.load_bz({"FBZ": abz.FBZ(), "InversionSymIBZ": abz.InversionSymIBZ(), "CubicSymIBZ": abz.CubicSymIBZ()}[kind], np.eye(2))
        bzo = orc.load_bz(kind, np.eye(2))
        f = abz.FourierIntegrand(abz.DOSIntegrand(), s, 0.25)
        sol = abz.do_solve(f, bz, abz.MixedParameters(0.3), abz.EvalCounter(abz.IAI()), abstol=1e-4, _panels=True)
        rec = []
        ref = orc.solve_iai(so, bzo, orc.f_dos(0.25, 0.3), abstol=1e-4, record=rec)
        assert np.array_equal(sol.extra["panels"], np.array(rec))
        assert sol.numevals == ref.numevals
        assert abs(sol.u - ref.u) <= 1e-10 * abs(ref.u)


def test_nested_batch_integrand(abz):
    """ref: test/fourier.jl:25-37 -- the nested-batch path agrees with the serial path for
    NestedQuad(AuxQuadGKJL()) and MonkhorstPack(); and its BatchIntegrand refinement (several panels
    popped per round) reproduces the oracle's batch-mode panel tree bit for bit."""
    for d in (1, 2, 3):
        so = orc.integer_lattice(d)
        s = abz.FourierSeries(so.c[..., 0, 0], period=1.0, first=so.first, ndim=d)
        p = abz.ParameterIntegrand(abz.LinearXIntegrand(), 1.3, b=4.2)
        nest = abz.NestedBatchIntegrand(tuple(p for _ in range(3)))
        for alg, dom in ((abz.NestedQuad(abz.AuxQuadGKJL()), abz.CubicLimits(np.zeros(d), np.ones(d))),
                         (abz.MonkhorstPack(), abz.Basis(np.eye(d)))):
            u1 = abz.solve(abz.IntegralProblem(abz.FourierIntegrand(p, s), dom), alg).u
            u2 = abz.solve(abz.IntegralProblem(abz.FourierIntegrand(p, s, nest), dom), alg).u
            assert np.allclose(u1, u2, rtol=1e-7, atol=1e-9)
    rng = np.random.default_rng(12)
    c, first = rand_series(rng, (5, 5), 2, hermitian=True)
    s, so = both(abz, c, first)
    bz = abz.load_bz(abz.FBZ(), np.eye(2))
    pi = abz.ParameterIntegrand(abz.DOSIntegrand(), 0.25)
    f = abz.FourierIntegrand(pi, s, abz.NestedBatchIntegrand((pi,)))
    sol = abz.do_solve(f, bz, abz.MixedParameters(0.3), abz.EvalCounter(abz.IAI()), abstol=1e-4, _panels=True)
    rec = []
    ref = orc.solve_iai(so, orc.load_bz("FBZ", np.eye(2)), orc.f_dos(0.25, 0.3), abstol=1e-4, record=rec, batch=True)
    assert np.array_equal(sol.extra["panels"], np.array(rec)) and sol.numevals == ref.numevals
    assert abs(sol.u - ref.u) <= 1e-10 * abs(ref.u)
    with pytest.raises(ValueError):
        abz.NestedBatchIntegrand((pi,), max_batch=0)  # ref: src/batch.jl:16


def test_iai_batchsolve_lockstep_is_bit_identical(abz):
    """batchsolve over omega with IAI (src/interfaces.jl:234-243) runs all solves in lock-step
    (abz_iai_solve_many): each must make exactly the decisions it makes alone -- same value bits,
    same error, same numevals -- for every depth, domain kind and for n <= 4 and the generic-n path."""
    rng = np.random.default_rng(21)
    om = np.array([-0.7, 0.05, 0.3, 0.31, 1.4, 0.3])  # includes a repeated and an off-band value
    for d, n, shape in ((1, 3, (7,)), (2, 2, (5, 5)), (3, 3, (3, 3, 3)), (2, 6, (3, 3))):
        c, first = rand_series(rng, shape, n, hermitian=True)
        s, _ = both(abz, c, first)
        for bzk in (abz.FBZ(), abz.CubicSymIBZ()):
            bz = abz.load_bz(bzk, np.eye(d))
            for integ in (abz.DOSIntegrand(), abz.TrGlocIntegrand()):
                f = abz.FourierIntegrand(integ, s, 0.2)
                solver = abz.IntegralSolver(f, bz, abz.EvalCounter(abz.IAI()), abstol=1e-3)
                got, meta = [], []
                many = abz.batchsolve(solver, om, callback=lambda sv, i, k, p, sol, t: meta.append((sol.resid, sol.numevals)))
                for k, w in enumerate(om):
                    one = solver.solve_p(abz.MixedParameters(w))
                    assert many[k] == one.u, (d, n, type(bzk).__name__, type(integ).__name__, w)
                    assert meta[k] == (one.resid, one.numevals)
    # NestedQuad on plain limits, vector-valued integrand with fixed parameters in the group key
    so = orc.integer_lattice(2)
    s = abz.FourierSeries(so.c[..., 0, 0], period=1.0, first=so.first, ndim=2)
    f = abz.FourierIntegrand(abz.LinearXIntegrand(), s, 1.3)
    solver = abz.IntegralSolver(f, abz.CubicLimits(np.zeros(2), np.ones(2)), abz.NestedQuad(abz.AuxQuadGKJL()), abstol=1e-8)
    bs = [4.2, -1.0, 0.0]
    many = abz.batchsolve(solver, [abz.MixedParameters(b=b) for b in bs])
    for k, b in enumerate(bs):
        assert np.array_equal(np.asarray(many[k]), np.asarray(solver(b=b)))


def test_iai_sweep_lanes_change_nothing(abz, svo, monkeypatch):
    """Sweeps of >= 64 independent IAI solves run on two lanes (two host threads, two device contexts with their own
    copies of the series): values, error estimates and evaluation counts must be those of the one-lane sweep, bit for bit,
    for the 3-band SVO model on the cubic IBZ and a 6-band generic-n series; a coefficient update reaches both lanes."""
    s, _ = svo
    om = np.linspace(11.0, 14.0, 96)
    bz = abz.load_bz(abz.CubicSymIBZ(), 3.85856 * np.eye(3))
    solver = abz.IntegralSolver(abz.FourierIntegrand(abz.DOSIntegrand(), s, 0.1), bz, abz.EvalCounter(abz.IAI()), abstol=1e-2)

    def sweep(sv, ps):
        meta = []
        vals = abz.batchsolve(sv, ps, callback=lambda s_, i, k, p, sol, t: meta.append((i, sol.resid, sol.numevals)))
        return np.asarray(vals), sorted(meta)
    monkeypatch.setenv("ABZ_IAI_LANES", "1")
    v1, m1 = sweep(solver, om)
    monkeypatch.setenv("ABZ_IAI_LANES", "2")
    v2, m2 = sweep(solver, om)
    monkeypatch.setenv("ABZ_IAI_LANES", "3")
    v3, m3 = sweep(solver, om)
    assert np.array_equal(v1, v2) and m1 == m2 and np.array_equal(v1, v3) and m1 == m3
    rng = np.random.default_rng(77)
    c, first = rand_series(rng, (3, 3), 6, hermitian=True)
    s6, _ = both(abz, c / np.sqrt(6), first)
    sol6 = abz.IntegralSolver(abz.FourierIntegrand(abz.TrGlocIntegrand(), s6, 0.2), abz.load_bz(abz.FBZ(), np.eye(2)),
                              abz.EvalCounter(abz.IAI()), abstol=1e-2)
    om6 = np.linspace(-1.0, 1.0, 70)
    monkeypatch.setenv("ABZ_IAI_LANES", "1")
    a1, n1 = sweep(sol6, om6)
    monkeypatch.setenv("ABZ_IAI_LANES", "2")
    a2, n2 = sweep(sol6, om6)
    assert np.array_equal(a1, a2) and n1 == n2
    # new coefficients must reach the second lane's view as well
    s6.c[...] = s6.c * 0.5
    s6.invalidate()
    b2, _ = sweep(sol6, om6)
    monkeypatch.setenv("ABZ_IAI_LANES", "1")
    b1, _ = sweep(sol6, om6)
    assert np.array_equal(b1, b2) and not np.array_equal(a1, b1)
    # ... and so must coefficients handed to ONE device copy (DeviceSeries.update(c)): sweep, update, sweep
    monkeypatch.setenv("ABZ_IAI_LANES", "2")
    s6.device().update(s6.c * 3.0)
    c2, _ = sweep(sol6, om6)
    monkeypatch.setenv("ABZ_IAI_LANES", "1")
    c1, _ = sweep(sol6, om6)
    assert np.array_equal(c1, c2) and not np.array_equal(b1, c1)


def test_kshard_partial_rules_sum_to_the_full_rule(abz, svo):
    """a solve sharded over k.  The W ranks' rules (slabs of the outermost variable of a
    full grid / blocks of the irreducible nodes) are built one after another on this GPU; their partial
    reductions must add up to the un-sharded rule value, for the matrix integrand, the coordinate
    dependent one (grid indices of a slab carry its offset) and GGR; slab nodes = rows of the full export."""
    L = abz._lib
    rng = np.random.default_rng(31)
    c, first = rand_series(rng, (3, 5, 3), 2, hermitian=True)
    s, _ = both(abz, c, first)
    dev = s.device()
    npt, W = 10, 3
    syms = abz.load_bz(abz.CubicSymIBZ(), np.eye(3)).syms
    for sy in (None, syms):
        full = dev.rule(npt, sy, L.WANT_H)
        ref_dos = full.reduce(L.F_DOS, [0.2], [0.1, 0.7])
        exp_full = full.export(x=True, w=True, H=True)
        tot_dos, rows = 0, []
        for r in range(W):
            dev.kshard, dev.allreduce = (r, W), (lambda a: a)  # keep the partial sums
            try:
                part = dev.rule(npt, sy, L.WANT_H)
                assert part.nk == full.nk and part is not full
                tot_dos = tot_dos + part.reduce(L.F_DOS, [0.2], [0.1, 0.7])
                rows.append(part.export(x=True, w=True, H=True))
            finally:
                dev.kshard, dev.allreduce = None, None
        assert sum(len(e["w"]) for e in rows) == full.nk
        for key in ("x", "w", "H"):
            assert np.array_equal(np.concatenate([e[key] for e in rows]), exp_full[key]), key
        assert np.allclose(tot_dos, ref_dos, rtol=1e-13, atol=1e-15)
    # coordinate-dependent integrand on a scalar series: grid indices of a slab carry its offset
    so = orc.integer_lattice(3)
    s1 = abz.FourierSeries(so.c[..., 0, 0], period=1.0, first=so.first, ndim=3)
    d1 = s1.device()
    ref_x = d1.rule(7, None, L.WANT_H).reduce(L.F_LINEAR_X, [1.3, 0.25])
    tot_x = 0
    for r in range(W):
        d1.kshard, d1.allreduce = (r, W), (lambda a: a)
        try:
            tot_x = tot_x + d1.rule(7, None, L.WANT_H).reduce(L.F_LINEAR_X, [1.3, 0.25])
        finally:
            d1.kshard, d1.allreduce = None, None
    assert np.allclose(tot_x, ref_x, rtol=1e-13, atol=1e-15) and np.abs(ref_x).min() > 0.1
    # GGR (eigenvalues + velocities of a slab) on the SVO model
    svo = svo[0]
    sdev = svo.device()
    Es = np.array([11.8, 12.6, 13.4])
    ref = sdev.rule(12, None, L.WANT_EIG | L.WANT_VEL).ggr(Es)
    tot = 0
    for r in range(4):
        sdev.kshard, sdev.allreduce = (r, 4), (lambda a: a)
        try:
            tot = tot + sdev.rule(12, None, L.WANT_EIG | L.WANT_VEL).ggr(Es)
        finally:
            sdev.kshard, sdev.allreduce = None, None
    assert np.allclose(tot, ref, rtol=1e-12)
    # round 5: the same through the kernels of 5...8 bands (one node per lane), 9...32 (row-layout GGR build) and 33...64
    # (kernels_big.hip): slabs of a full grid and blocks of a symmetric node list reproduce the rows of the whole rule
    for nb, sy in ((6, None), (6, syms), (12, None), (12, syms), (40, None)):
        cb, fb = rand_series(np.random.default_rng(300 + nb), (3, 3, 3), nb, hermitian=True)
        sb = abz.FourierSeries(cb / np.sqrt(nb), period=1.0, first=fb)
        db = sb.device()
        want = (L.WANT_H | L.WANT_EIG) if nb > 32 else (L.WANT_EIG | L.WANT_VEL)
        fullb = db.rule(8, sy, want)
        expf = fullb.export(x=True, w=True, H=bool(want & L.WANT_H), eig=True, vel=bool(want & L.WANT_VEL))
        partsb = []
        for r in range(3):
            db.kshard, db.allreduce = (r, 3), (lambda a: a)
            try:
                pb = db.rule(8, sy, want)
                partsb.append(pb.export(x=True, w=True, H=bool(want & L.WANT_H), eig=True, vel=bool(want & L.WANT_VEL)))
            finally:
                db.kshard, db.allreduce = None, None
        for key in expf:
            if expf[key] is not None:
                assert np.array_equal(np.concatenate([e[key] for e in partsb]), expf[key]), (nb, sy is not None, key)
    # a whole AutoPTR solve under dist.kshard with a single-rank group degenerates to the plain solve
    from autobzcore.jl_amd.dist import kshard
    bz = abz.load_bz(abz.FBZ(), 3.85856 * np.eye(3))
    solver = abz.IntegralSolver(abz.FourierIntegrand(abz.DOSIntegrand(), svo, 0.1), bz, abz.PTR(npt=24))
    with kshard(svo):
        assert solver(12.5) == solver(12.5)


def test_batchsolve_archive_matches_batchsolve(abz, svo, tmp_path):
    """ref: ext/HDF5Ext.jl:123-158 (sweep output with partial-result persistence)."""
    s, _ = svo
    bz = abz.load_bz(abz.CubicSymIBZ(), 3.85856 * np.eye(3))
    solver = abz.IntegralSolver(abz.FourierIntegrand(abz.DOSIntegrand(), s, 0.1), bz, abz.EvalCounter(abz.PTR(npt=24)))
    om = np.linspace(11.5, 13.5, 9)
    ref = abz.batchsolve(solver, om)
    out = abz.batchsolve_archive(tmp_path / "sweep.npz", solver, om, chunk=4)
    z = abz.SweepArchive.load(tmp_path / "sweep.npz")
    assert np.array_equal(out, ref) and np.array_equal(z["I"], ref) and z["done"].all()
    assert np.allclose(z["p"], om) and np.all(z["numevals"] == z["numevals"][0]) and z["numevals"][0] > 0  # plain parameters: `p`
    from autobzcore.jl_amd import h5lite
    if h5lite.available():  # the same sweep into a real HDF5 file (the reference's container)
        out5 = abz.batchsolve_archive(tmp_path / "sweep.h5", solver, om, chunk=4)
        z5 = abz.read_h5_to_nt(tmp_path / "sweep.h5")
        assert np.array_equal(out5, ref) and np.array_equal(z5["I"], ref) and np.array_equal(z5["numevals"], z["numevals"])
        assert np.array_equal(z5["p"], om) and z5["retcode"].dtype == np.int32 and np.all(z5["retcode"] == 1)


def test_absolute_estimate_ptr_then_iai_on_the_device(abz, svo):
    """ref: src/algorithms.jl:614-653 -- the usual AutoBZ pattern: a coarse PTR rule sizes the DOS, IAI then runs with
    abstol = reltol * |estimate|.  The second stage must be exactly the direct IAI solve at that tolerance, and the
    counter adds both stages."""
    s, _ = svo
    bz = abz.load_bz(abz.CubicSymIBZ(), 3.85856 * np.eye(3))
    prob = abz.IntegralProblem(abz.FourierIntegrand(abz.DOSIntegrand(), s, 0.1), bz, abz.MixedParameters(12.5))
    est = abz.solve(prob, abz.EvalCounter(abz.PTR(npt=16)))
    sol = abz.solve(prob, abz.EvalCounter(abz.AbsoluteEstimate(abz.PTR(npt=16), abz.IAI())), reltol=1e-3)
    direct = abz.solve(prob, abz.EvalCounter(abz.IAI()), abstol=1e-3 * abs(est.u), reltol=0.0)
    assert sol.u == direct.u and sol.resid == direct.resid and sol.numevals == est.numevals + direct.numevals
    assert sol.resid <= 1e-3 * abs(est.u) and abs(sol.u - est.u) < 0.05 * abs(est.u)
    # the reference's named combinations (src/brillouin.jl:464-488)
    both_ = abz.solve(prob, abz.EvalCounter(abz.PTR_IAI(ptr=abz.PTR(npt=16))), reltol=1e-3)
    assert both_.u == sol.u and both_.numevals == sol.numevals
    auto = abz.solve(prob, abz.AutoPTR_IAI(reltol=1.0), reltol=1e-3)
    assert abs(auto.u - sol.u) <= 2e-3 * abs(sol.u)


def test_symrep_extension_point_matrix_valued_on_the_ibz(abz):
    """ref: src/brillouin.jl:73-108 (`SymRep(f)` + `symmetrize_`).  H(k) = diag(cos kx, cos ky) + t (cos kx + cos ky) s_x
    obeys H(S k) = D_S H(k) D_S^T with D_S = s_x for the symmetries that swap the axes, 1 otherwise; with that
    representation attached the Green's function integrated over the IBZ maps to the full-BZ result, for
    PTR, AutoPTR and IAI; without it the solve is repeated on the full BZ (UnknownRep) with a warning."""
    sx = np.array([[0.0, 1.0], [1.0, 0.0]])
    c = np.zeros((3, 3, 2, 2), dtype=np.complex128)
    for (i, j), M in (((0, 1), np.diag([0.5, 0.0]) + 0.15 * sx), ((2, 1), np.diag([0.5, 0.0]) + 0.15 * sx),
                      ((1, 0), np.diag([0.0, 0.5]) + 0.15 * sx), ((1, 2), np.diag([0.0, 0.5]) + 0.15 * sx)):
        c[i, j] = M
    s = abz.FourierSeries(c, period=1.0, first=(-1, -1), ndim=2)
    ibz = abz.load_bz(abz.CubicSymIBZ(), np.eye(2))
    fbz = abz.load_bz(abz.FBZ(), np.eye(2))
    reps = [sx if S[0, 0] == 0 else np.eye(2) for S in ibz.syms]
    g = abz.GlocIntegrand().with_symrep(abz.MatrixRep(reps))
    for alg, kw in ((abz.PTR(npt=24), {}), (abz.AutoPTR(), dict(abstol=1e-6)), (abz.IAI(), dict(abstol=1e-6))):
        ref = abz.IntegralSolver(abz.FourierIntegrand(abz.GlocIntegrand(), s, 0.3), fbz, alg, **kw)(0.2)
        got = abz.IntegralSolver(abz.FourierIntegrand(g, s, 0.3), ibz, alg, **kw)(0.2)
        assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max(), type(alg).__name__
        assert abs(got[0, 1]) > 1e-3  # the off-diagonal element is not zero: the test sees the representation
    with pytest.warns(UserWarning):
        red = abz.IntegralSolver(abz.FourierIntegrand(abz.GlocIntegrand(), s, 0.3), ibz, abz.PTR(npt=24))(0.2)
    ref = abz.IntegralSolver(abz.FourierIntegrand(abz.GlocIntegrand(), s, 0.3), fbz, abz.PTR(npt=24))(0.2)
    assert np.abs(red - ref).max() <= 1e-12 * np.abs(ref).max()
    swp = abz.batchsolve(abz.IntegralSolver(abz.FourierIntegrand(g, s, 0.3), ibz, abz.PTR(npt=24)), [0.2, 0.5])
    assert np.abs(swp[0] - ref).max() <= 1e-12 * np.abs(ref).max()


def test_iai_over_general_convex_zones(abz):
    """ref: ext/SymmetryReduceBZExt.jl:33-58, ext/ibzlims.jl:198-289.  IAI over an explicitly given irreducible
    zone (PolyhedralLimits / PolygonLimits): same panels and numevals as the oracle's restatement, the
    polyhedral tetrahedron agrees with CubicSymIBZ, volumes come out, and the zone + point group reproduce
    the full-BZ integral."""
    from scipy.spatial import ConvexHull
    rng = np.random.default_rng(41)
    c, first = rand_series(rng, (3, 3, 3), 2, hermitian=True)
    s, so = both(abz, c / 3, first)
    # (1) cubic IBZ given as a polyhedron
    V = np.array([[0, 0, 0], [0, 0, .5], [0, .5, .5], [.5, .5, .5]], dtype=float)
    cub = abz.load_bz(abz.CubicSymIBZ(), np.eye(3))
    ibz = abz.load_bz(abz.IBZ(), np.eye(3), hull=V, syms=cub.syms)
    assert isinstance(ibz.lims, abz.PolyhedralLimits) and abz.nsyms(ibz) == 48
    so_c, _ = orc.integer_lattice(3), None
    sc = abz.FourierSeries(so_c.c[..., 0, 0], period=1.0, first=so_c.first, ndim=3)
    f = abz.FourierIntegrand(abz.LinearIntegrand(), sc, 1.3, 1.0)
    a = abz.do_solve(f, cub, abz.MixedParameters(), abz.EvalCounter(abz.IAI()), abstol=1e-7)
    b = abz.do_solve(f, ibz, abz.MixedParameters(), abz.EvalCounter(abz.IAI()), abstol=1e-7)
    assert abs(a.u - b.u) <= 1e-12 * abs(a.u) and a.numevals == b.numevals and abs(b.u - (2 * np.pi) ** 3) < 1e-6
    # (2) a random convex polytope: initial break points at the vertex coordinates, panels as in the oracle
    P = rng.standard_normal((9, 3)) * 0.2 + 0.5
    lim = abz.PolyhedralLimits.from_vertices(P)
    fd = abz.FourierIntegrand(abz.DOSIntegrand(), s, 0.3)
    sol = abz.do_solve(fd, lim, abz.MixedParameters(0.2), abz.EvalCounter(abz.NestedQuad(abz.AuxQuadGKJL())), abstol=1e-4)
    ref = orc.nested_quad(so, orc.PolyhedralLimits(lim.faces), orc.f_dos(0.3, 0.2), abstol=1e-4)
    assert sol.numevals == ref[2] and abs(sol.u - ref[0]) <= 1e-10 * abs(ref[0])
    one = abz.do_solve(abz.FourierIntegrand(abz.UnitIntegrand(), s), lim, abz.MixedParameters(),
                       abz.NestedQuad(abz.AuxQuadGKJL()), abstol=1e-9)
    assert abs(one.u - ConvexHull(P).volume) <= 1e-8
    # (3) 2-D polygon + a user closure on the host path
    th = np.arange(6) * np.pi / 3
    hexv = 0.3 * np.column_stack([np.cos(th), np.sin(th)]) + 0.5
    c2, first2 = rand_series(rng, (3, 3), 2, hermitian=True)
    s2, so2 = both(abz, c2 / 3, first2)
    pg = abz.PolygonLimits(hexv)
    sol2 = abz.do_solve(abz.FourierIntegrand(abz.DOSIntegrand(), s2, 0.3), pg, abz.MixedParameters(0.2),
                        abz.EvalCounter(abz.NestedQuad(abz.AuxQuadGKJL())), abstol=1e-5)
    ref2 = orc.nested_quad(so2, orc.PolygonLimits(hexv), orc.f_dos(0.3, 0.2), abstol=1e-5)
    assert sol2.numevals == ref2[2] and abs(sol2.u - ref2[0]) <= 1e-10 * abs(ref2[0])
    user = lambda v, eta, omega: -np.imag(np.trace(np.linalg.inv((omega + 1j * eta) * np.eye(2) - v.s))) / np.pi
    sol3 = abz.do_solve(abz.FourierIntegrand(user, s2, 0.3), pg, abz.MixedParameters(0.2),
                        abz.NestedQuad(abz.AuxQuadGKJL()), abstol=1e-5)
    assert abs(sol3.u - ref2[0]) <= 1e-9 * abs(ref2[0])


@pytest.mark.parametrize("d,n,npt", [(1, 2, 300), (2, 3, 150), (2, 1, 200), (3, 3, 131), (3, 4, 129), (3, 2, 140)])
def test_store_free_rule_value_equals_rule_reduce(abz, d, n, npt):
    """abz_ptr_sum: rule(f, B) without materialising the rule (grids used once / beyond HBM).  Same numbers as
    abz_rule_reduce on the cached rule for every integrand, sweeps longer than one launch, slabs, and through
    the solver when the rule would exceed `stream_above_bytes`."""
    rng = np.random.default_rng(700 + 10 * d + n)
    c, first = rand_series(rng, (5, 3, 3)[:d], n, hermitian=True)
    s, _ = both(abz, c / 2, first)
    dev = s.device()
    L = abz._lib
    rule = dev.rule(npt, None, L.WANT_H | L.WANT_EIG)
    om = np.linspace(-1.5, 1.5, 11)  # 11 > 8 sweep values: two launches
    for fid in (L.F_DOS, L.F_TRGLOC, L.F_DOS_EIG, L.F_GLOC):
        ref = rule.reduce(fid, [0.3], om)
        got = dev.ptr_sum(npt, fid, [0.3], om)
        assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max(), fid
    assert abs(dev.ptr_sum(npt, L.F_ONE)[0, 0] - 1.0) < 1e-14
    if n == 1:
        for fid in (L.F_LINEAR, L.F_LINEAR_X):
            ref = rule.reduce(fid, [1.3, 0.25])
            got = dev.ptr_sum(npt, fid, [1.3, 0.25])
            assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max(), fid
    if d >= 2:  # slabs add up (k-sharded streaming)
        tot = 0
        for r in range(3):
            dev.kshard, dev.allreduce = (r, 3), (lambda a: a)
            try:
                tot = tot + dev.ptr_sum(npt, L.F_DOS, [0.3], om[:3])
                if n == 1:
                    totx = dev.ptr_sum(npt, L.F_LINEAR_X, [1.3, 0.25]) + (totx if r else 0)
            finally:
                dev.kshard, dev.allreduce = None, None
        assert np.abs(tot - rule.reduce(L.F_DOS, [0.3], om[:3])).max() <= 1e-12
        if n == 1:
            assert np.abs(totx - rule.reduce(L.F_LINEAR_X, [1.3, 0.25])).max() <= 1e-12
    # through the solver: a rule "too large to keep" is summed on the fly, the value is the same
    bz = abz.load_bz(abz.FBZ(), np.eye(d))
    solver = abz.IntegralSolver(abz.FourierIntegrand(abz.DOSIntegrand(), s, 0.3), bz, abz.PTR(npt=npt + 1))
    ref = solver(0.2)
    dev.drop_rules()
    old = dev.stream_above_bytes
    dev.stream_above_bytes = 0
    try:
        got = solver(0.2)
        assert not dev.has_rule(npt + 1, None, L.WANT_H)  # nothing was materialised
    finally:
        dev.stream_above_bytes = old
    assert abs(got - ref) <= 1e-12 * abs(ref)
    # a series that is not Hermitian has no closed-form store-free kernel: the library sums the inverse of every node (round 5;
    # it answered ABZ_ERR_UNSUPPORTED before) -- the same value as the scan of a rule; the mirror keeps using rules for these
    c2, first2 = rand_series(rng, (3, 3, 3)[:d], n, hermitian=False)
    s2, _ = both(abz, c2 / 4, first2)
    d2 = s2.device()
    sf2 = d2.ptr_sum(npt, L.F_TRGLOC, [2.5], [0.1])
    r2 = d2.rule(npt, None, want=L.WANT_H)
    assert np.abs(sf2 - r2.reduce(L.F_TRGLOC, [2.5], [0.1])).max() <= 1e-12 * np.abs(sf2).max()
    r2.close()
    d2.drop_rules()
    d2.stream_above_bytes = 0
    sol2 = abz.IntegralSolver(abz.FourierIntegrand(abz.TrGlocIntegrand(), s2, 2.5), bz, abz.PTR(npt=npt))(0.1)
    assert d2.has_rule(npt, None, L.WANT_H) and np.isfinite(sol2)


def test_plain_c_client_of_the_abi(tmp_path):
    """tests/c/abi_smoke.c: config 1 (1-D cos series, f = 1.3 s + 1 -> 1) through eval_nodes, rule + reduce,
    the store-free sum and IAI, from plain C."""
    import subprocess
    from test_abi_cpu import _build_c_client
    run = subprocess.run([str(_build_c_client(tmp_path))], capture_output=True, text=True)
    assert run.returncode == 0 and "abi_smoke ok" in run.stdout, run.stdout + run.stderr


@pytest.mark.parametrize("d,n,npt", [(1, 6, 40), (2, 12, 21), (2, 20, 9), (3, 16, 7), (3, 27, 8)])
def test_store_free_rule_value_generic_n(abz, d, n, npt):
    """abz_ptr_sum for 5..32 bands (gen_grid_sum_kernel): equals the reduction of the materialised rule, for
    sweeps longer than one launch (4 values) and through the solver, which prefers it for n > 4."""
    rng = np.random.default_rng(800 + 10 * d + n)
    c, first = rand_series(rng, (3, 3, 3)[:d], n, hermitian=True)
    s, _ = both(abz, c / np.sqrt(n), first)
    dev = s.device()
    L = abz._lib
    rule = abz.DeviceRule(dev, npt, None, L.WANT_H)
    om = np.linspace(-1.0, 1.0, 6)
    for fid in (L.F_DOS, L.F_TRGLOC):
        ref = rule.reduce(fid, [0.3], om)
        got = dev.ptr_sum(npt, fid, [0.3], om)
        assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max(), fid
    rule.close()
    if d >= 2:
        tot = 0
        for r in range(2):
            dev.kshard, dev.allreduce = (r, 2), (lambda a: a)
            try:
                tot = tot + dev.ptr_sum(npt, L.F_DOS, [0.3], om[:2])
            finally:
                dev.kshard, dev.allreduce = None, None
        assert np.abs(tot - dev.ptr_sum(npt, L.F_DOS, [0.3], om[:2])).max() <= 1e-13
    bz = abz.load_bz(abz.FBZ(), np.eye(d))
    solver = abz.IntegralSolver(abz.FourierIntegrand(abz.DOSIntegrand(), s, 0.3), bz, abz.PTR(npt=npt))
    val = solver(0.2)
    assert not dev.has_rule(npt, None, L.WANT_H)  # summed on the fly
    full = abz.DeviceRule(dev, npt, None, L.WANT_H)
    ref = full.reduce(L.F_DOS, [0.3], [0.2])[0, 0].real * abs(np.linalg.det(bz.B))
    assert abs(val - ref) <= 1e-12 * abs(ref)


def test_store_free_16_band_3d_against_the_oracle(abz):
    """The route config 5's full-size check leans on (16 bands, 3 variables, store-free PTR: one value by the Gauss-Jordan
    trace, sweeps by the tridiagonal resolvent) against the ORACLE's PTR at a size beyond the 7^3 of the layout tests:
    npt = 20 (8 000 nodes) on a random Hermitian series, and the synthetic Wannier model of config 5 itself (2 197 R) at
    npt = 6."""
    L = abz._lib
    rng = np.random.default_rng(4242)
    c, first = rand_series(rng, (3, 3, 3), 16, hermitian=True)
    s, so = both(abz, c / 4.0, first)
    bz, bzo = abz.load_bz(abz.FBZ(), np.eye(3)), orc.load_bz("FBZ", np.eye(3))
    om = [-0.7, 0.0, 0.4, 1.1, 1.9]
    got1 = s.device().ptr_sum(20, L.F_DOS, [0.1], om[:1])[:, 0].real
    got5 = s.device().ptr_sum(20, L.F_DOS, [0.1], om)[:, 0].real
    for i, w in enumerate(om):
        ref = orc.solve_ptr(so, bzo, orc.f_dos(0.1, w), npt=20).u / abs(np.linalg.det(bzo.B))
        assert abs(got5[i] - ref) <= 1e-11 * abs(ref), (w, got5[i], ref)
        if i == 0:
            assert abs(got1[0] - ref) <= 1e-11 * abs(ref)
    sw, swo = abz.synthetic_wannier(), orc.synthetic_wannier()
    got = sw.device().ptr_sum(6, L.F_DOS, [0.05], [0.2])[0, 0].real
    ref = orc.solve_ptr(swo, bzo, orc.f_dos(0.05, 0.2), npt=6).u / abs(np.linalg.det(bzo.B))
    assert abs(got - ref) <= 1e-11 * abs(ref)


@pytest.mark.parametrize("n,eta", [(16, 0.05), (11, 0.3), (6, 0.01)])
def test_store_free_sweep_by_tridiagonal_resolvent(abz, monkeypatch, n, eta):
    """Sweeps of >= 3 values on 5..16 bands: one Householder tridiagonalisation per node, then tr inv(zI - H) = p'/p from
    the three-term recurrence per swept value (gen_grid_sum_tri_kernel).  40 values = two full passes of 16 + a ragged
    one, small eta (poles close to the axis), against numpy on the exported H(k) and against the inversion-per-value
    kernel (ABZ_GEN_SUM_TRI=0)."""
    rng = np.random.default_rng(1234 + n)
    c, first = rand_series(rng, (3, 3, 3), n, hermitian=True)
    s, so = both(abz, c / np.sqrt(n), first)
    dev = s.device()
    L = abz._lib
    npt = 9
    rule = abz.DeviceRule(dev, npt, None, L.WANT_H)
    H = rule.export(H=True)["H"]
    om = np.linspace(-2.0, 2.0, 40)
    eye = np.eye(n)
    tr = np.array([np.trace(np.linalg.inv((w + 1j * eta) * eye - H), axis1=1, axis2=2).mean() for w in om])
    got_t = dev.ptr_sum(npt, L.F_TRGLOC, [eta], om)[:, 0]
    got_d = dev.ptr_sum(npt, L.F_DOS, [eta], om)[:, 0].real
    assert np.abs(got_t - tr).max() <= 1e-11 * np.abs(tr).max()
    assert np.abs(got_d + tr.imag / np.pi).max() <= 1e-11 * np.abs(tr).max()
    # the scan of the cached rule takes the same route (gen_rows_reduce_tri_kernel): 70 values = one chunk of 64 + 6
    om2 = np.linspace(-1.5, 2.5, 70)
    tr2 = np.array([np.trace(np.linalg.inv((w + 1j * eta) * eye - H), axis1=1, axis2=2).mean() for w in om2])
    sc_t = rule.reduce(L.F_TRGLOC, [eta], om2)[:, 0]
    sc_d = rule.reduce(L.F_DOS, [eta], om2)[:, 0].real
    assert np.abs(sc_t - tr2).max() <= 1e-11 * np.abs(tr2).max()
    assert np.abs(sc_d + tr2.imag / np.pi).max() <= 1e-11 * np.abs(tr2).max()
    # weighted (symmetric) rule through the same kernel
    bz = abz.load_bz(abz.CubicSymIBZ(), np.eye(3))
    obz = orc.load_bz("CubicSymIBZ", np.eye(3))
    solver = abz.IntegralSolver(abz.FourierIntegrand(abz.DOSIntegrand(), s, eta), bz, abz.PTR(npt=npt))
    om3 = np.linspace(-1.0, 1.2, 5)
    got_s = np.asarray(abz.batchsolve(solver, om3), dtype=float)
    for u, w in zip(got_s, om3):
        ref = orc.solve_ptr(so, obz, orc.f_dos(eta, w), npt=npt).u
        assert abs(u - ref) <= 1e-10 * abs(ref)
    # and the inversion-per-value kernel gives the same sums
    monkeypatch.setenv("ABZ_GEN_SUM_TRI", "0")
    inv_t = dev.ptr_sum(npt, L.F_TRGLOC, [eta], om)[:, 0]
    assert np.abs(inv_t - got_t).max() <= 1e-11 * np.abs(tr).max()


# ------------------------------------------------------------------ generic n (wave-per-node kernels)
@pytest.mark.parametrize("d,n", [(1, 5), (2, 8), (3, 6), (3, 16), (2, 17), (3, 24), (2, 32)])
def test_generic_n_eval_and_rules(abz, d, n):
    rng = np.random.default_rng(1000 + 10 * d + n)
    dims = (3, 5, 3)[:d]
    c, first = rand_series(rng, dims, n, hermitian=True)
    c = c / np.sqrt(n)
    s, so = both(abz, c, first)
    k = rng.uniform(0, 1, size=(70, d))
    H, E = s.device().eval_nodes(k, want=3)
    Ho = orc.evaluate_many(so, k)
    scale = np.abs(Ho).max()
    assert np.abs(H - Ho).max() <= 1e-12 * scale
    assert np.abs(E - np.linalg.eigvalsh(Ho, UPLO="U")).max() <= 1e-11 * scale
    npt = 6
    rule = s.device().rule(npt, None, want=3)
    out = rule.export(H=True, eig=True)
    vals = orc.fourier_ptr(so, npt)
    perm = tuple(range(d - 1, -1, -1))
    ref = np.transpose(vals, perm + (d, d + 1)).reshape(-1, n, n)
    assert np.abs(out["H"] - ref).max() <= 1e-12 * np.abs(ref).max()
    assert np.abs(out["eig"] - np.linalg.eigvalsh(ref, UPLO="U")).max() <= 1e-11 * np.abs(ref).max()
    omegas = np.array([-0.4, 0.3])
    eta = 0.25
    dos = rule.reduce(abz._lib.F_DOS, [eta], omegas)[:, 0].real
    dose = rule.reduce(abz._lib.F_DOS_EIG, [eta], omegas)[:, 0].real
    trg = rule.reduce(abz._lib.F_TRGLOC, [eta], omegas)[:, 0]
    g = rule.reduce(abz._lib.F_GLOC, [eta], omegas)
    for i, om in enumerate(omegas):
        r, _ = orc._ptr_rule_sum(so, npt, None, orc.f_dos(eta, om))
        G, _ = orc._ptr_rule_sum(so, npt, None, orc.f_gloc(eta, om))
        assert abs(dos[i] - r) <= 1e-10 * abs(r) and abs(dose[i] - r) <= 1e-9 * abs(r)
        assert abs(trg[i] - np.trace(G)) <= 1e-10 * abs(np.trace(G))
        assert np.abs(g[i].reshape(n, n).T - G).max() <= 1e-10 * np.abs(G).max()


@pytest.mark.parametrize("n", [5, 6, 7, 8])
def test_lane_kernels_5_to_8_bands(abz, n, monkeypatch):
    """5...8 bands on full grids, one node per lane (kernels_lane.hip: LAPACK's zhetd2 unrolled in the lane's registers + the
    per-lane QR iteration; ref src/fourier.jl:127-174, eigen(Hermitian(h)) src/dos_ggr.jl:19): H in both layouts, eigenvalues,
    store-free sums of 1 ... 70 swept values -- against the oracle and against the 8-lane row kernels (ABZ_LANE_KERNELS=0), on
    grids shorter and longer than a wave, with 3 to 13 coefficients per variable; degenerate and diagonal matrices included."""
    L = abz._lib
    rng = np.random.default_rng(500 + n)
    for dims, npt in (((3, 3), 5), ((5, 3, 3), 9), ((13, 3), 70), ((7,), 130)):
        d = len(dims)
        c, first = rand_series(rng, dims, n, hermitian=True)
        s, so = both(abz, c / np.sqrt(n), first)
        vals = orc.fourier_ptr(so, npt)
        perm = tuple(range(d - 1, -1, -1))
        ref = np.transpose(vals, perm + (d, d + 1)).reshape(-1, n, n)
        eref = np.linalg.eigvalsh(ref, UPLO="U")
        om = np.linspace(-1.5, 1.5, 70 if d == 2 and npt == 70 else 3)
        got = {}
        for lane in ("1", "0"):
            monkeypatch.setenv("ABZ_LANE_KERNELS", lane)
            for want in (L.WANT_H | L.WANT_EIG | L.WANT_H_COMPACT, L.WANT_H | L.WANT_EIG, L.WANT_EIG, L.WANT_H):
                rule = abz.DeviceRule(s.device(), npt, None, want)
                out = rule.export(H=bool(want & L.WANT_H), eig=bool(want & L.WANT_EIG))
                rule.close()
                if want & L.WANT_H:
                    assert np.abs(out["H"] - ref).max() <= 1e-12 * np.abs(ref).max(), (lane, want, dims)
                if want & L.WANT_EIG:
                    assert np.abs(out["eig"] - eref).max() <= 1e-11 * np.abs(ref).max(), (lane, want, dims)
            got[lane] = (s.device().ptr_sum(npt, L.F_TRGLOC, [0.2], om)[:, 0], s.device().ptr_sum(npt, L.F_DOS, [0.2], om[:1])[:, 0].real)
            # scans of cached rules (either layout; a symmetric node list where the dimension has one) through lane_scan_kernel
            for want in (L.WANT_H | L.WANT_H_COMPACT, L.WANT_H):
                rule = abz.DeviceRule(s.device(), npt, None, want)
                sc = rule.reduce(L.F_TRGLOC, [0.2], om)[:, 0]
                assert np.abs(sc - got[lane][0]).max() <= 1e-11 * np.abs(sc).max(), (lane, want, dims)
                dd = rule.reduce(L.F_DOS, [0.2], om[:2])[:, 0].real
                assert np.abs(dd + sc[:2].imag / np.pi).max() <= 1e-12 * np.abs(sc).max()
                rule.close()
            if d >= 2:  # the runs of a symmetric node list through the lane kernel: values, eigenvalues, the scan
                bzo = orc.load_bz("InversionSymIBZ", np.eye(d))
                rs_ = abz.DeviceRule(s.device(), npt, bzo.syms, L.WANT_H | L.WANT_EIG)
                outs = rs_.export(x=True, w=True, H=True, eig=True)
                Hs = orc.evaluate_many(so, outs["x"])
                assert np.abs(outs["H"] - Hs).max() <= 1e-12 * np.abs(Hs).max(), (lane, dims)
                assert np.abs(outs["eig"] - np.linalg.eigvalsh(Hs, UPLO="U")).max() <= 1e-11 * np.abs(Hs).max(), (lane, dims)
                r0, _ = orc._ptr_rule_sum(so, npt, bzo.syms, orc.f_dos(0.2, om[1]))
                assert abs(rs_.reduce(L.F_DOS, [0.2], om[1:2])[0, 0].real - r0) <= 1e-10 * abs(r0), (lane, dims)
                rs_.close()
        monkeypatch.delenv("ABZ_LANE_KERNELS")
        assert np.abs(got["1"][0] - got["0"][0]).max() <= 1e-11 * np.abs(got["0"][0]).max(), dims
        assert np.abs(got["1"][1] - got["0"][1]).max() <= 1e-11 * np.abs(got["0"][1]).max(), dims
        t, _ = orc._ptr_rule_sum(so, npt, None, orc.f_gloc(0.2, om[1]))
        assert abs(got["1"][0][1] - np.trace(t)) <= 1e-10 * abs(np.trace(t)), dims
    # degenerate / diagonal / zero matrices (constant series): the reflector-free branches of zlarfg and the QR deflation
    mats = [np.diag(np.arange(n, dtype=float)).astype(complex), np.zeros((n, n), dtype=complex), np.eye(n, dtype=complex) * 2.5]
    q, _ = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))
    mats.append((q * np.array([1.0] * (n - 2) + [2.0, 2.0])) @ q.conj().T)
    mats.append((q * np.array([-1.0, -1.0 + 1e-13] + [0.5] * (n - 2))) @ q.conj().T)
    for A in mats:
        A = 0.5 * (A + A.conj().T)
        s1 = abz.FourierSeries(A[None, :, :], period=1.0, first=(0,))
        rule = abz.DeviceRule(s1.device(), 7, None, L.WANT_EIG)
        E = rule.export(eig=True)["eig"]
        rule.close()
        assert np.abs(E - np.linalg.eigvalsh(A)[None, :]).max() <= 1e-12 * max(1.0, np.abs(A).max())


@pytest.mark.parametrize("d,n", [(2, 33), (3, 48), (2, 64), (1, 40)])
def test_more_than_32_bands(abz, d, n, monkeypatch):
    """33...64 bands (ref: src/fourier.jl:22-58 is generic in the matrix size, eigen(Hermitian(h)) is LAPACK there,
    src/dos_ggr.jl:19): kernels_big.hip -- series values and eigenvalues at arbitrary nodes and on PTR rules (full grid and an
    inversion-symmetric node list), DOS / tr G scans of the cached matrices and eigenvalues, store-free sums, with the
    level-1 evaluation on the vector units and on the matrix cores (ABZ_BIG_MFMA=1) -- against the oracle; 65 bands: ArgumentError."""
    L = abz._lib
    rng = np.random.default_rng(3000 + 10 * d + n)
    dims = (3, 5, 3)[:d] if d > 1 else (7,)
    c, first = rand_series(rng, dims, n, hermitian=True)
    c = c / np.sqrt(n)
    s, so = both(abz, c, first)
    k = rng.uniform(0, 1, size=(37, d))
    H, E = s.device().eval_nodes(k, want=3)
    Ho = orc.evaluate_many(so, k)
    scale = np.abs(Ho).max()
    assert np.abs(H - Ho).max() <= 1e-12 * scale
    assert np.abs(E - np.linalg.eigvalsh(Ho, UPLO="U")).max() <= 1e-10 * scale
    npt = 5 if d == 3 else (9 if d == 2 else 21)
    vals = orc.fourier_ptr(so, npt)
    perm = tuple(range(d - 1, -1, -1))
    ref = np.transpose(vals, perm + (d, d + 1)).reshape(-1, n, n)
    omegas = np.array([-0.4, 0.3, 0.9])
    eta = 0.25
    refs = [orc._ptr_rule_sum(so, npt, None, orc.f_gloc(eta, om))[0] for om in omegas]
    for mfma in ("0", "1"):
        monkeypatch.setenv("ABZ_BIG_MFMA", mfma)
        rule = abz.DeviceRule(s.device(), npt, None, 3)
        out = rule.export(H=True, eig=True)
        assert np.abs(out["H"] - ref).max() <= 1e-12 * np.abs(ref).max(), mfma
        assert np.abs(out["eig"] - np.linalg.eigvalsh(ref, UPLO="U")).max() <= 1e-10 * np.abs(ref).max(), mfma
        dos = rule.reduce(L.F_DOS, [eta], omegas)[:, 0].real
        dose = rule.reduce(L.F_DOS_EIG, [eta], omegas)[:, 0].real
        trg = rule.reduce(L.F_TRGLOC, [eta], omegas)[:, 0]
        sf = s.device().ptr_sum(npt, L.F_TRGLOC, [eta], omegas)[:, 0]
        for i in range(len(omegas)):
            t = np.trace(refs[i])
            assert abs(trg[i] - t) <= 1e-10 * abs(t) and abs(sf[i] - t) <= 1e-10 * abs(t), mfma
            assert abs(dos[i] + t.imag / np.pi) <= 1e-10 * abs(t) and abs(dose[i] + t.imag / np.pi) <= 1e-9 * abs(t), mfma
        rule.close()
    monkeypatch.delenv("ABZ_BIG_MFMA")
    if d >= 2:  # a symmetric node list
        bzo = orc.load_bz("InversionSymIBZ", np.eye(d))
        rs = s.device().rule(npt, bzo.syms, want=3)
        outs = rs.export(x=True, w=True, H=True, eig=True)
        Hs = orc.evaluate_many(so, outs["x"])
        assert np.abs(outs["H"] - Hs).max() <= 1e-12 * np.abs(Hs).max()
        assert np.abs(outs["eig"] - np.linalg.eigvalsh(Hs, UPLO="U")).max() <= 1e-10 * np.abs(Hs).max()
        rsum, _ = orc._ptr_rule_sum(so, npt, bzo.syms, orc.f_dos(eta, omegas[1]))
        assert abs(rs.reduce(L.F_DOS, [eta], omegas[1:2])[0, 0].real - rsum) <= 1e-10 * abs(rsum)
    if d == 2:  # IAI through the node path
        bz = abz.load_bz(abz.FBZ(), np.eye(2))
        sol = abz.do_solve(abz.FourierIntegrand(abz.DOSIntegrand(), s, 0.5), bz, abz.MixedParameters(0.2), abz.EvalCounter(abz.IAI()),
                           abstol=1e-2)
        refi = orc.solve_iai(so, orc.load_bz("FBZ", np.eye(2)), orc.f_dos(0.5, 0.2), abstol=1e-2)
        assert sol.numevals == refi.numevals and abs(sol.u - refi.u) <= 1e-9 * abs(refi.u)


@pytest.mark.parametrize("n", [33, 48, 64])
def test_more_than_32_bands_matrix_valued_and_non_hermitian(abz, n):
    """33...64 bands, what the tridiagonal cannot serve (big_inverse_kernel: one workgroup per node, Gauss-Jordan in LDS): the
    matrix-valued Green's function of a Hermitian series and G, tr G, DOS of a series that is NOT Hermitian -- scans of a cached
    rule (full grid and weighted node list) and IAI through the node path, against the oracle.
    ref: src/fourier.jl:22-58 with the integrands of test/fourier.jl."""
    L = abz._lib
    rng = np.random.default_rng(4000 + n)
    c, first = rand_series(rng, (3, 3), n, hermitian=True)
    extra, _ = rand_series(rng, (3, 3), n, hermitian=False)
    omegas = np.array([-0.7, 0.2, 1.1])
    eta = 0.5
    for herm in (True, False):
        s, so = both(abz, (c + (0.0 if herm else 0.05) * extra) / np.sqrt(n), first)
        npt = 7
        refs = [orc._ptr_rule_sum(so, npt, None, orc.f_gloc(eta, om))[0] for om in omegas]
        rule = s.device().rule(npt, None, want=1)
        g = rule.reduce(L.F_GLOC, [eta], omegas)
        tr = rule.reduce(L.F_TRGLOC, [eta], omegas)[:, 0]
        dos = rule.reduce(L.F_DOS, [eta], omegas)[:, 0].real
        assert s.device().ptr_sum_supported(npt, L.F_GLOC) and s.device().ptr_sum_supported(npt, L.F_TRGLOC)
        gsf = s.device().ptr_sum(npt, L.F_GLOC, [eta], omegas)  # store-free: no rule, the inverse of every node of a chunk
        tsf = s.device().ptr_sum(npt, L.F_TRGLOC, [eta], omegas)[:, 0]
        for i in range(len(omegas)):
            t = np.trace(refs[i])
            assert np.abs(g[i].reshape(n, n).T - refs[i]).max() <= 1e-10 * np.abs(refs[i]).max(), herm
            assert np.abs(gsf[i].reshape(n, n).T - refs[i]).max() <= 1e-10 * np.abs(refs[i]).max(), herm
            assert abs(tr[i] - t) <= 1e-10 * abs(t) and abs(tsf[i] - t) <= 1e-10 * abs(t), herm
            assert abs(dos[i] + t.imag / np.pi) <= 1e-10 * abs(t), herm
        rule.close()
        bzo = orc.load_bz("InversionSymIBZ", np.eye(2))
        rs = s.device().rule(npt, bzo.syms, want=1)
        gs = rs.reduce(L.F_GLOC, [eta], omegas[:1])
        rsum, _ = orc._ptr_rule_sum(so, npt, bzo.syms, orc.f_gloc(eta, omegas[0]))
        assert np.abs(gs[0].reshape(n, n).T - rsum).max() <= 1e-10 * np.abs(rsum).max(), herm
        rs.close()
    # IAI, one dimension: G(omega) of the Hermitian series (n^2 components), tr G of the other
    c1, first1 = rand_series(rng, (5,), n, hermitian=True)
    e1, _ = rand_series(rng, (5,), n, hermitian=False)
    bz = abz.load_bz(abz.FBZ(), np.eye(1))
    s1, so1 = both(abz, c1 / np.sqrt(n), first1)
    sol = abz.do_solve(abz.FourierIntegrand(abz.GlocIntegrand(), s1, eta), bz, abz.MixedParameters(0.2), abz.EvalCounter(abz.IAI()), abstol=1e-3)
    ref = orc.solve_iai(so1, orc.load_bz("FBZ", np.eye(1)), orc.f_gloc(eta, 0.2), abstol=1e-3)
    assert sol.numevals == ref.numevals and np.abs(np.asarray(sol.u) - ref.u).max() <= 1e-9 * np.abs(ref.u).max()
    s2, so2 = both(abz, (c1 + 0.05 * e1) / np.sqrt(n), first1)
    sol = abz.do_solve(abz.FourierIntegrand(abz.TrGlocIntegrand(), s2, eta), bz, abz.MixedParameters(0.2), abz.EvalCounter(abz.IAI()), abstol=1e-3)
    f_tr = lambda x, h: np.trace(orc.f_gloc(eta, 0.2)(x, h), axis1=-2, axis2=-1)
    ref = orc.solve_iai(so2, orc.load_bz("FBZ", np.eye(1)), f_tr, abstol=1e-3)
    assert sol.numevals == ref.numevals and abs(sol.u - ref.u) <= 1e-9 * abs(ref.u)


def test_more_than_32_bands_cached_tridiagonal_follows_the_values(abz):
    """33...64 bands: a rule of a Hermitian series keeps the tridiagonal forms of its nodes after its first DOS / tr G scan (the
    Householder pass is most of a scan).  Scans repeat to the bit; new coefficients + rebuild must be seen by the next scan;
    a series that stops being Hermitian must leave the cached route."""
    L = abz._lib
    rng = np.random.default_rng(4242)
    n, npt, eta = 40, 9, 0.3
    c, first = rand_series(rng, (3, 3), n, hermitian=True)
    c = c / np.sqrt(n)
    s, so = both(abz, c, first)
    om = np.array([-0.6, 0.1, 0.8])
    rule = s.device().rule(npt, None, want=1)
    ref = lambda so_: np.array([np.trace(orc._ptr_rule_sum(so_, npt, None, orc.f_gloc(eta, w))[0]) for w in om])
    a1 = rule.reduce(L.F_TRGLOC, [eta], om)[:, 0]
    a2 = rule.reduce(L.F_TRGLOC, [eta], om)[:, 0]
    d1 = rule.reduce(L.F_DOS, [eta], om)[:, 0].real
    t = ref(so)
    assert np.array_equal(a1, a2) and np.abs(a1 - t).max() <= 1e-10 * np.abs(t).max()
    assert np.abs(d1 + t.imag / np.pi).max() <= 1e-10 * np.abs(t).max()
    c2 = c.copy()
    c2[1, 1] += np.diag(np.linspace(-0.2, 0.2, n))  # R = 0: stays Hermitian
    s.device().update(c2)
    rule.rebuild()
    so2 = orc.FourierSeries(c2, period=1.0, first=first, ndim=2)
    b1 = rule.reduce(L.F_TRGLOC, [eta], om)[:, 0]
    t2 = ref(so2)
    assert np.abs(b1 - t2).max() <= 1e-10 * np.abs(t2).max() and np.abs(b1 - a1).max() > 1e-6 * np.abs(t).max()
    extra, _ = rand_series(rng, (3, 3), n, hermitian=False)
    c3 = c2 + 0.05 / np.sqrt(n) * extra
    s.device().update(c3)
    rule.rebuild()
    so3 = orc.FourierSeries(c3, period=1.0, first=first, ndim=2)
    e1 = rule.reduce(L.F_TRGLOC, [eta], om)[:, 0]
    t3 = ref(so3)
    assert np.abs(e1 - t3).max() <= 1e-10 * np.abs(t3).max()
    rule.close()


@pytest.mark.parametrize("n", [20, 40])
def test_autoptr_many_bands_through_the_library_loop(abz, n):
    """AutoPTR on 20- and 40-band models (kept rules, repeated scans of the same rule -- above 32 bands through its cached
    tridiagonal forms --, full BZ and inversion-symmetric zone), DOS, tr G and the matrix-valued G against the oracle's
    autosymptr: equal numevals, values to 1e-10.  ref: src/algorithms.jl:418-432"""
    rng = np.random.default_rng(900 + n)
    c, first = rand_series(rng, (3, 3), n, hermitian=True)
    s, so = both(abz, c / np.sqrt(n), first)
    eta, seq = 0.4, dict(nmin=4, nmax=40, n0=6.0, dn=4.0)
    for kind, bzk in (("FBZ", abz.FBZ()), ("InversionSymIBZ", abz.InversionSymIBZ())):
        bz, bzo = abz.load_bz(bzk, np.eye(2)), orc.load_bz(kind, np.eye(2))
        alg = abz.AutoPTR(a=1.0, **seq)
        for fi, fo in ((abz.DOSIntegrand(), lambda om: orc.f_dos(eta, om)),
                       (abz.TrGlocIntegrand(), lambda om: (lambda x, h: np.trace(orc.f_gloc(eta, om)(x, h), axis1=-2, axis2=-1))),
                       (abz.GlocIntegrand(), lambda om: orc.f_gloc(eta, om))):
            if kind != "FBZ" and isinstance(fi, abz.GlocIntegrand):
                continue  # (a matrix-valued integrand without a SymRep is repeated on the full zone: src/brillouin.jl:76-108)
            f = abz.FourierIntegrand(fi, s, eta)
            for om in (-0.3, 0.2, 0.7):  # (the second and third solves scan rules the first one left behind)
                got = abz.solve(abz.IntegralProblem(f, bz, abz.MixedParameters(om)), abz.EvalCounter(alg), abstol=1e-3)
                ref = orc.solve_autoptr(so, bzo, fo(om), abstol=1e-3, **seq)
                assert got.numevals == ref.numevals, (kind, type(fi).__name__, om)
                assert np.abs(np.asarray(got.u) - ref.u).max() <= 1e-10 * np.abs(ref.u).max(), (kind, type(fi).__name__, om)


def test_more_than_32_bands_and_more_than_64_coefficients(abz):
    """33...64 bands with more than 64 coefficients along a variable (big_series_kernel adds them up in pieces of 64): values,
    eigenvalues, band velocities, the matrix-valued G and a store-free sum against the oracle."""
    L = abz._lib
    rng = np.random.default_rng(6464)
    n, npt = 34, 9
    c, first = rand_series(rng, (131,), n, hermitian=True)
    c = c * np.exp(-0.05 * np.abs(np.arange(131) - 65))[:, None, None] / np.sqrt(n)
    s, so = both(abz, c, first)
    w, e, v = orc.get_ggr_data(so, npt, None)
    rule = abz.DeviceRule(s.device(), npt, None, 1 | 2 | 4)
    o = rule.export(H=True, eig=True, vel=True)
    href = orc.fourier_ptr(so, npt).reshape(-1, n, n)
    assert np.abs(o["H"] - href).max() <= 1e-12 * np.abs(href).max()
    assert np.abs(o["eig"] - e).max() <= 1e-11 * np.abs(e).max()
    sep = np.min(np.diff(e, axis=1), axis=1) > 1e-6 * np.abs(e).max()
    assert np.abs(o["vel"][sep] - v[sep]).max() <= 1e-8 * np.abs(v).max()
    om = np.array([0.15])
    gref, _ = orc._ptr_rule_sum(so, npt, None, orc.f_gloc(0.3, om[0]))
    g = rule.reduce(L.F_GLOC, [0.3], om)[0].reshape(n, n).T
    rule.close()
    assert np.abs(g - gref).max() <= 1e-10 * np.abs(gref).max()
    t = s.device().ptr_sum(npt, L.F_TRGLOC, [0.3], om)[0, 0]
    assert abs(t - np.trace(gref)) <= 1e-10 * abs(np.trace(gref))


def test_more_than_64_bands_is_an_argument_error(abz):
    rng = np.random.default_rng(65)
    c, first = rand_series(rng, (3,), 65, hermitian=True)
    with pytest.raises(ValueError):
        abz.FourierSeries(c, period=1.0, first=first).device()


def test_generic_n_set_above_150_kb_of_lds(abz):
    """28 bands x 11 coefficients: the unpadded level-1 set + store tile take 150.7 KB of LDS -- above the 150 KB the padded
    layouts are held to, inside the 160 KB of a CU that the unpadded 32-lane kernels may use (one workgroup per CU)."""
    rng = np.random.default_rng(2828)
    n, npt = 28, 7
    c, first = rand_series(rng, (11, 3), n, hermitian=True)
    s, so = both(abz, c / np.sqrt(n), first)
    L = abz._lib
    rule = s.device().rule(npt, None, want=3)
    out = rule.export(H=True, eig=True)
    vals = orc.fourier_ptr(so, npt)
    ref = np.transpose(vals, (1, 0, 2, 3)).reshape(-1, n, n)
    assert np.abs(out["H"] - ref).max() <= 1e-12 * np.abs(ref).max()
    assert np.abs(out["eig"] - np.linalg.eigvalsh(ref, UPLO="U")).max() <= 1e-11 * np.abs(ref).max()
    om = np.linspace(-1.0, 1.0, 4)
    a = rule.reduce(L.F_DOS, [0.2], om)
    b = s.device().ptr_sum(npt, L.F_DOS, [0.2], om)
    assert np.abs(a - b).max() <= 1e-12 * np.abs(a).max()


def test_generic_n_set_staged_in_chunks(abz):
    """A level-1 set that does not fit the LDS at all (20 bands x 45 coefficients = 288 KB): the rule-build and sweep kernels
    stage it in chunks per group of nodes, the phase carried from chunk to chunk -- values and eigenvalues against the
    oracle, the store-free sweep against the scan of the rule and the oracle's rule sum."""
    rng = np.random.default_rng(4520)
    n, npt = 20, 7
    c, first = rand_series(rng, (45, 3), n, hermitian=True)
    s, so = both(abz, c / np.sqrt(n), first)
    L = abz._lib
    rule = s.device().rule(npt, None, want=3)
    out = rule.export(H=True, eig=True)
    vals = orc.fourier_ptr(so, npt)
    ref = np.transpose(vals, (1, 0, 2, 3)).reshape(-1, n, n)
    assert np.abs(out["H"] - ref).max() <= 1e-12 * np.abs(ref).max()
    assert np.abs(out["eig"] - np.linalg.eigvalsh(ref, UPLO="U")).max() <= 1e-11 * np.abs(ref).max()
    om = np.linspace(-1.0, 1.0, 4)
    a = rule.reduce(L.F_DOS, [0.2], om)
    b = s.device().ptr_sum(npt, L.F_DOS, [0.2], om)
    assert np.abs(a - b).max() <= 1e-12 * np.abs(a).max()
    r0, _ = orc._ptr_rule_sum(so, npt, None, orc.f_dos(0.2, om[2]))
    assert abs(b[2, 0].real - r0) <= 1e-10 * abs(r0)


@pytest.mark.parametrize("n,M,npt", [(20, 45, 11), (32, 13, 19), (16, 45, 19), (8, 171, 35)])
def test_generic_n_chunked_sets_second_groups_idle_waves_and_node_lists(abz, n, M, npt):
    """The chunked staging (sets beyond the LDS) where the round-4 test did not reach: grids longer than one group of nodes
    and not a multiple of it (a line's chunks are staged again for its second group; the last group leaves waves
    without a node that only keep the staging barriers), the 8- and 16-lane instances, and the runs of an
    inversion-symmetric node list through the same branch -- values, eigenvalues and sums against the oracle."""
    rng = np.random.default_rng(100 * n + M)
    c, first = rand_series(rng, (M, 3), n, hermitian=True)
    s, so = both(abz, c / np.sqrt(n * M / 10), first)
    L = abz._lib
    rule = s.device().rule(npt, None, want=3)
    out = rule.export(H=True, eig=True)
    vals = orc.fourier_ptr(so, npt)
    ref = np.transpose(vals, (1, 0, 2, 3)).reshape(-1, n, n)
    assert np.abs(out["H"] - ref).max() <= 1e-12 * np.abs(ref).max()
    assert np.abs(out["eig"] - np.linalg.eigvalsh(ref, UPLO="U")).max() <= 1e-11 * np.abs(ref).max()
    om = np.linspace(-1.0, 1.0, 5)
    a = rule.reduce(L.F_DOS, [0.2], om)
    b = s.device().ptr_sum(npt, L.F_DOS, [0.2], om)
    assert np.abs(a - b).max() <= 1e-11 * np.abs(a).max()
    r0, _ = orc._ptr_rule_sum(so, npt, None, orc.f_dos(0.2, om[2]))
    assert abs(b[2, 0].real - r0) <= 1e-10 * abs(r0)
    # the symmetric (inversion) node list of the same series: runs of nodes per level-1 set
    bzo = orc.load_bz("InversionSymIBZ", np.eye(2))
    rs = s.device().rule(npt, bzo.syms, want=3)
    outs = rs.export(x=True, w=True, H=True, eig=True)
    Ho = orc.evaluate_many(so, outs["x"])
    assert np.abs(outs["H"] - Ho).max() <= 1e-12 * np.abs(Ho).max()
    assert np.abs(outs["eig"] - np.linalg.eigvalsh(Ho, UPLO="U")).max() <= 1e-11 * np.abs(Ho).max()
    rsum, _ = orc._ptr_rule_sum(so, npt, bzo.syms, orc.f_dos(0.2, om[1]))
    got = rs.reduce(L.F_DOS, [0.2], om[1:2])[0, 0].real
    assert abs(got - rsum) <= 1e-10 * abs(rsum)


def test_generic_n_unpadded_32_lane_layout(abz):
    """17...32 bands with a level-1 set too long for the zero-padded LDS layout (11 coefficients x 32 x 32 x 16 B = 180 KB):
    the unpadded instances of the 32-lane row kernels -- rule values and eigenvalues against the oracle, the store-free
    sweep (tridiagonal resolvent) against the scan of the rule."""
    rng = np.random.default_rng(2024)
    n, npt = 20, 9
    c, first = rand_series(rng, (11, 3), n, hermitian=True)
    s, so = both(abz, c / np.sqrt(n), first)
    L = abz._lib
    rule = s.device().rule(npt, None, want=3)
    out = rule.export(H=True, eig=True)
    vals = orc.fourier_ptr(so, npt)
    ref = np.transpose(vals, (1, 0, 2, 3)).reshape(-1, n, n)
    assert np.abs(out["H"] - ref).max() <= 1e-12 * np.abs(ref).max()
    assert np.abs(out["eig"] - np.linalg.eigvalsh(ref, UPLO="U")).max() <= 1e-11 * np.abs(ref).max()
    om = np.linspace(-1.0, 1.0, 5)
    for fid in (L.F_DOS, L.F_TRGLOC):
        a = rule.reduce(fid, [0.2], om)
        b = s.device().ptr_sum(npt, fid, [0.2], om)
        assert np.abs(a - b).max() <= 1e-12 * np.abs(a).max()
    r0, _ = orc._ptr_rule_sum(so, npt, None, orc.f_dos(0.2, om[1]))
    assert abs(rule.reduce(L.F_DOS, [0.2], om)[1, 0].real - r0) <= 1e-10 * abs(r0)


@pytest.mark.parametrize("n3,copies", [(3, 2), (5, 3), (2, 8), (3, 7), (4, 8)])
def test_generic_n_eigenvalues_degenerate_and_diagonal(abz, n3, copies):
    """The eigenvalue builds of rules for 5..32 bands (Householder in the row layout + the per-lane QR kernel) on exactly degenerate spectra (block-diagonal copies of
    one series: every eigenvalue `copies`-fold) and on an already diagonal H(k)."""
    rng = np.random.default_rng(31 + n3)
    c3, first = rand_series(rng, (3, 5), n3, hermitian=True)
    n = n3 * copies
    c = np.zeros(c3.shape[:2] + (n, n), dtype=complex)
    for b in range(copies):
        c[..., b * n3:(b + 1) * n3, b * n3:(b + 1) * n3] = c3
    s, so = both(abz, c, first)
    npt = 10
    out = s.device().rule(npt, None, want=3).export(H=True, eig=True)
    e3 = np.linalg.eigvalsh(out["H"][:, :n3, :n3], UPLO="U")
    ref = np.sort(np.repeat(e3, copies, axis=1), axis=1)
    assert np.abs(out["eig"] - ref).max() <= 1e-12 * np.abs(ref).max()
    # diagonal H(k): nothing to rotate
    cd = np.zeros_like(c)
    for a in range(n):
        cd[..., a, a] = c3[..., 0, 0] * (1.0 + 0.1 * a)
    sd, sod = both(abz, cd, first)
    outd = sd.device().rule(npt, None, want=3).export(H=True, eig=True)
    diag = np.real(np.einsum("kaa->ka", outd["H"]))
    assert np.abs(outd["eig"] - np.sort(diag, axis=1)).max() <= 1e-13 * np.abs(diag).max()


def test_generic_n_gloc_long_sweep(abz):
    """Matrix-valued G_loc scan of a 16-band rule over 70 swept values (more than one launch of the row kernel: its
    partial sums are capped at 64 MB) against numpy on the exported H(k) and, at both ends, the oracle's rule sum."""
    rng = np.random.default_rng(909)
    n, npt, eta = 16, 16, 0.15
    c, first = rand_series(rng, (3, 3, 3), n, hermitian=True)
    c = c / np.sqrt(n)
    s, so = both(abz, c, first)
    rule = s.device().rule(npt, None, want=1)
    H = rule.export(H=True)["H"]
    omegas = np.linspace(-1.0, 1.0, 70)
    g = rule.reduce(abz._lib.F_GLOC, [eta], omegas)
    eye = np.eye(n)
    for i in (0, 33, 63, 64, 69):
        ref = np.linalg.inv((omegas[i] + 1j * eta) * eye - H).mean(axis=0)
        assert np.abs(g[i].reshape(n, n).T - ref).max() <= 1e-11 * np.abs(ref).max()
    for i in (0, 69):
        G, _ = orc._ptr_rule_sum(so, npt, None, orc.f_gloc(eta, omegas[i]))
        assert np.abs(g[i].reshape(n, n).T - G).max() <= 1e-10 * np.abs(G).max()


@pytest.mark.parametrize("n,kind", [(6, "CubicSymIBZ"), (11, "InversionSymIBZ"), (16, "CubicSymIBZ")])
def test_generic_n_symmetric_ptr_sweep(abz, n, kind):
    """PTR on the irreducible nodes with integer weights for more than four bands (symmetric rule built by the
    wave-per-node kernel, scanned by the row kernel, 4 swept values per pass: 7 omegas = one full + one ragged
    pass) against the oracle's symmetric rule sum; ref: src/fourier.jl:210-292, src/brillouin.jl:337-355."""
    rng = np.random.default_rng(500 + n)
    c, first = rand_series(rng, (3, 3, 3), n, hermitian=True)
    c = c / np.sqrt(n)
    s, so = both(abz, c, first)
    kinds = {"InversionSymIBZ": abz.InversionSymIBZ(), "CubicSymIBZ": abz.CubicSymIBZ()}
    bz = abz.load_bz(kinds[kind], np.eye(3))
    obz = orc.load_bz(kind, np.eye(3))
    eta, npt = 0.2, 7
    omegas = np.linspace(-0.8, 0.9, 7)
    solver = abz.IntegralSolver(abz.FourierIntegrand(abz.DOSIntegrand(), s, eta), bz, abz.PTR(npt=npt))
    got = np.asarray(abz.batchsolve(solver, omegas), dtype=float)
    for u, om in zip(got, omegas):
        ref = orc.solve_ptr(so, obz, orc.f_dos(eta, om), npt=npt).u
        assert abs(u - ref) <= 1e-10 * abs(ref)


@pytest.mark.parametrize("n", [6, 12, 20])
def test_generic_n_iai_matches_oracle(abz, n):
    """n = 6 / 12 / 20 take the 8- / 16- / 32-lane rows of the panel kernel (gen_panel_kernel)."""
    rng = np.random.default_rng(77)
    c, first = rand_series(rng, (3, 3), n, hermitian=True)
    c = c / (n / 2)
    s, so = both(abz, c, first)
    bz = abz.load_bz(abz.FBZ(), np.eye(2))
    f = abz.FourierIntegrand(abz.DOSIntegrand(), s, 0.3)
    sol = abz.do_solve(f, bz, abz.MixedParameters(0.2), abz.EvalCounter(abz.IAI()), abstol=1e-3, _panels=True)
    rec = []
    ref = orc.solve_iai(so, orc.load_bz("FBZ", np.eye(2)), orc.f_dos(0.3, 0.2), abstol=1e-3, record=rec)
    assert np.array_equal(sol.extra["panels"], np.array(rec)) and sol.numevals == ref.numevals
    assert abs(sol.u - ref.u) <= 1e-9 * abs(ref.u)


def test_generic_n_iai_3d_16_band_matches_oracle(abz):
    """The config-5 code path (16 bands, THREE variables: host-driven outer level, middle level, workgroup-per-
    integral innermost kernel) pinned to the oracle rather than to itself: outermost panel tree bit-exact, equal
    `numevals`, value to 1e-9.  3 x 3 x 3 coefficients so that the pure-Python oracle finishes in ~15 s; the
    tolerance is loose enough for that and tight enough to refine the outer and the middle level
    (3 outer panels, 7.3e5 nodes).  ref: src/fourier.jl:394-510."""
    rng = np.random.default_rng(1616)
    n = 16
    c, first = rand_series(rng, (3, 3, 3), n, hermitian=True)
    c = c / (n / 2)
    s, so = both(abz, c, first)
    bz = abz.load_bz(abz.FBZ(), np.eye(3))
    f = abz.FourierIntegrand(abz.DOSIntegrand(), s, 0.3)
    sol = abz.do_solve(f, bz, abz.MixedParameters(0.2), abz.EvalCounter(abz.IAI()), abstol=0.5, _panels=True)
    rec = []
    ref = orc.solve_iai(so, orc.load_bz("FBZ", np.eye(3)), orc.f_dos(0.3, 0.2), abstol=0.5, record=rec)
    assert len(rec) >= 3 and ref.numevals > 15**3
    assert np.array_equal(sol.extra["panels"], np.array(rec)) and sol.numevals == ref.numevals
    assert abs(sol.u - ref.u) <= 1e-9 * abs(ref.u)
    assert abs(sol.resid - ref.resid) <= 1e-6 * abs(ref.resid) + 1e-12


@pytest.mark.parametrize("n,dims", [(16, (5, 5)), (12, (3, 3, 3)), (9, (5, 3))])
def test_16_lane_panel_kernel_variants_agree(abz, monkeypatch, n, dims):
    """The workgroup-per-integral kernel of 9..16 bands has three generations that stay selectable: the unfolded series
    (ABZ_IPANEL_FOLD=0), the folded series with separate pivot-row broadcasts (ABZ_IPANEL_FMAC=0) and the default (pivot
    rows broadcast inside `v_fmac_f64_dpp`, pivot-row scaling deferred to the trace, polynomial sincospi); the adaptive
    step runs on one lane (ABZ_ADAPT_PAIR=0) or two.  The step variants must agree to the bit; the arithmetic variants
    round differently and must agree to 1e-12 with the same panels.  n < 16: the padded identity rows take part."""
    rng = np.random.default_rng(900 + n)
    c, first = rand_series(rng, dims, n, hermitian=True)
    c = c / (n / 2)
    d = len(dims)
    s = abz.FourierSeries(c, period=1.0, first=first, ndim=d)
    bz = abz.load_bz(abz.FBZ(), np.eye(d))
    f = abz.FourierIntegrand(abz.DOSIntegrand(), s, 0.1)
    keys = ("ABZ_IPANEL_FOLD", "ABZ_IPANEL_FMAC", "ABZ_ADAPT_PAIR")
    runs = {}
    for tag, env in (("default", {}), ("one_lane_step", {"ABZ_ADAPT_PAIR": "0"}), ("dpp_moves", {"ABZ_IPANEL_FMAC": "0"}),
                     ("unfolded", {"ABZ_IPANEL_FOLD": "0"})):
        for k in keys:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        sol = abz.do_solve(f, bz, abz.MixedParameters(0.15), abz.EvalCounter(abz.IAI()), abstol=1e-4 if d == 2 else 1e-2, reltol=0.0,
                           _panels=True)
        runs[tag] = (sol.u, sol.resid, sol.numevals, sol.extra["panels"])
    for k in keys:
        monkeypatch.delenv(k, raising=False)
    ref = runs["default"]
    assert ref[2] > 15**d * 4
    assert runs["one_lane_step"][0] == ref[0] and runs["one_lane_step"][1] == ref[1] and runs["one_lane_step"][2] == ref[2]
    assert np.array_equal(runs["one_lane_step"][3], ref[3])
    for tag in ("dpp_moves", "unfolded"):
        assert abs(runs[tag][0] - ref[0]) <= 1e-12 * abs(ref[0]), tag
        assert runs[tag][2] == ref[2] and np.allclose(runs[tag][3], ref[3], rtol=1e-10, atol=1e-13), tag


@pytest.mark.parametrize("n,dims,eta,abstol", [(3, (3, 3, 3), 0.2, 0.3), (16, (3, 3, 3), 0.3, 0.5), (2, (7, 5), 0.05, 1e-3)])
def test_iai_speculative_requests_change_nothing(abz, monkeypatch, n, dims, eta, abstol):
    """The driver requests the halves of every panel that is certain to be popped in one round; pops are replayed in
    QuadGK's order.  Value bits, error, numevals and panels equal those of the one-panel-per-round driver
    (ABZ_IAI_SPECULATE=0 = the round-1 behaviour), also with the set pools cut into many chunks."""
    rng = np.random.default_rng(4242 + n)
    c, first = rand_series(rng, dims, n, hermitian=True)
    c = c / max(1.0, n / 2)
    s = abz.FourierSeries(c, period=1.0, first=first, ndim=len(dims))
    d = len(dims)
    bz = abz.load_bz(abz.FBZ(), np.eye(d))
    f = abz.FourierIntegrand(abz.DOSIntegrand(), s, eta)
    runs = {}
    for tag, env in (("spec", {}), ("serial", {"ABZ_IAI_SPECULATE": "0"}), ("chunks", {"ABZ_IAI_POOL_MB": "1"}),
                     ("onelane", {"ABZ_ADAPT_PAIR": "0"}),  # n <= 4: the LDS kernel with its one-lane step instead of the register-resident one
                     ("nodes", {"ABZ_IAI_PANELS": "0"}),  # the level above the innermost one ships nodes, the host applies its GK rule
                     ("nodes_chunks", {"ABZ_IAI_PANELS": "0", "ABZ_IAI_POOL_MB": "1"}),
                     ("fullrows", {"ABZ_IAI_PACKED": "0"})):  # n <= 4: the chain on full instead of packed Hermitian rows
        for k in ("ABZ_IAI_SPECULATE", "ABZ_IAI_POOL_MB", "ABZ_ADAPT_PAIR", "ABZ_IAI_PACKED", "ABZ_IAI_PANELS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        sol = abz.do_solve(f, bz, abz.MixedParameters(0.1), abz.EvalCounter(abz.IAI()), abstol=abstol, reltol=0.0, _panels=True)
        runs[tag] = (sol.u, sol.resid, sol.numevals, sol.extra["panels"])
    for tag in ("serial", "chunks", "onelane", "nodes", "nodes_chunks"):
        assert runs[tag][0] == runs["spec"][0] and runs[tag][1] == runs["spec"][1] and runs[tag][2] == runs["spec"][2], tag
        assert np.array_equal(runs[tag][3], runs["spec"][3])
    # packed and full rows sum the same terms in a different order: same panels and counts, values equal to rounding
    assert runs["fullrows"][2] == runs["spec"][2] and np.array_equal(runs["fullrows"][3], runs["spec"][3])
    assert abs(runs["fullrows"][0] - runs["spec"][0]) <= 1e-12 * abs(runs["spec"][0])
    assert runs["spec"][2] > 15**d and len(runs["spec"][3]) >= 2


@pytest.mark.parametrize("d,kind", [(2, None), (3, None), (3, "cubic")])
def test_autosymptrjl_matches_oracle(abz, d, kind):
    """AutoSymPTRJL on a Basis domain (ref: src/algorithms.jl:393-432, used at test/fourier.jl:25-37) against the
    oracle's autosymptr loop: same npt sequence, so equal `numevals`, value to 1e-11."""
    rng = np.random.default_rng(50 + d)
    c, first = rand_series(rng, (3, 5, 3)[:d], 2, hermitian=True)
    if kind == "cubic":  # a series with the cubic point group: H(k) = sum_j cos(2 pi k_j) * A + B
        A = rng.standard_normal((2, 2))
        A = A + A.T
        c = np.zeros((3, 3, 3, 2, 2), dtype=complex)
        for j in range(3):
            for sgn in (0, 2):
                idx = [1, 1, 1]
                idx[j] = sgn
                c[tuple(idx)] = 0.5 * A
        c[1, 1, 1] = np.diag([0.3, -0.2])
        first = (-1, -1, -1)
    s, so = both(abz, c, first)
    eta, om = 0.4, 0.1
    obz = orc.load_bz("CubicSymIBZ" if kind else "FBZ", 2 * np.pi * np.eye(d))  # B = I
    syms = obz.syms if kind else None
    f = abz.FourierIntegrand(abz.DOSIntegrand(), s, eta)
    alg = abz.AutoSymPTRJL(a=eta, nmin=10, syms=syms)
    nsym = 1 if syms is None else len(syms)
    # the bare algorithm converges on the un-symmetrised rule value (1/nsyms of the BZ solve's): same decisions at abstol/nsyms
    sol = abz.do_solve(f, abz.Basis(np.eye(d)), abz.MixedParameters(om), abz.EvalCounter(alg), abstol=1e-5 / nsym)
    ref = orc.solve_autoptr(so, obz, orc.f_dos(eta, om), abstol=1e-5, a=eta, nmin=10)
    assert sol.numevals == ref.numevals
    # the oracle's BZ solve symmetrises (TrivialRep: x nsyms); AutoSymPTRJL on a Basis returns the bare rule value
    assert abs(sol.u * (nsym if kind else 1) - ref.u) <= 1e-11 * abs(ref.u)


def test_plain_batch_integrand_of_fourier_values(abz):
    """The reference's GPU hook (src/batch.jl:4-6): a plain BatchIntegrand whose x are FourierValue batches produced
    by abz_eval_nodes (`fourier_batch`), through MonkhorstPack, AutoSymPTRJL (src/algorithms.jl:370-372,421-423),
    AuxQuadGKJL (:227-233, a 1-D series) and NestedQuad (:469-471,517-518) -- against the oracle's rule sums /
    nested quadrature on the same integrand, and the batches really are batches (one GPU call per <= max_batch nodes)."""
    rng = np.random.default_rng(2718)
    eta, om = 0.35, 0.15
    g = lambda v, p: -np.imag(np.trace(np.linalg.inv((p + 1j * eta) * np.eye(2) - np.atleast_2d(v.s)))) / np.pi
    # 2-D, PTR family on a Basis
    c, first = rand_series(rng, (3, 5), 2, hermitian=True)
    s, so = both(abz, c, first)
    obz = orc.load_bz("FBZ", 2 * np.pi * np.eye(2))  # B = I
    sizes = []
    f = abz.fourier_batch(lambda v, p: (sizes.append(1), g(v, p))[1], s, max_batch=1000)
    sol = abz.solve(abz.IntegralProblem(f, abz.Basis(np.eye(2)), om), abz.EvalCounter(abz.MonkhorstPack(npt=24)))
    ref = orc.solve_ptr(so, obz, orc.f_dos(eta, om), npt=24)
    assert sol.numevals == 24 * 24 == len(sizes) and abs(sol.u - ref.u) <= 1e-11 * abs(ref.u)
    sol = abz.solve(abz.IntegralProblem(f, abz.Basis(np.eye(2)), om), abz.EvalCounter(abz.AutoSymPTRJL(a=eta, nmin=10)), abstol=1e-6)
    ref = orc.solve_autoptr(so, obz, orc.f_dos(eta, om), abstol=1e-6, a=eta, nmin=10)
    assert sol.numevals == ref.numevals and abs(sol.u - ref.u) <= 1e-11 * abs(ref.u)
    # NestedQuad over the unit square: innermost integral batched (scalar refinement at the outer level)
    sol = abz.solve(abz.IntegralProblem(f, abz.CubicLimits(np.zeros(2), np.ones(2)), om),
                    abz.EvalCounter(abz.NestedQuad(abz.AuxQuadGKJL())), abstol=1e-5)
    dev = abz.solve(abz.IntegralProblem(abz.FourierIntegrand(abz.DOSIntegrand(), s, eta), abz.CubicLimits(np.zeros(2), np.ones(2)),
                                        abz.MixedParameters(om)), abz.NestedQuad(abz.AuxQuadGKJL()), abstol=1e-5)
    assert abs(sol.u - dev.u) <= 3e-5 and sol.numevals >= 225
    # 1-D series under AuxQuadGKJL: batch refinement == the oracle's, bit-identical decisions
    c1, first1 = rand_series(rng, (5,), 2, hermitian=True)
    s1, so1 = both(abz, c1, first1)
    f1 = abz.fourier_batch(g, s1, max_batch=90)
    sol = abz.solve(abz.IntegralProblem(f1, (0.0, 1.0), om), abz.EvalCounter(abz.AuxQuadGKJL()), abstol=1e-8)
    I, E, nev = orc.auxquadgk(lambda xs: orc.f_dos(eta, om)(xs[:, None], orc.evaluate_many(so1, xs[:, None])), (0.0, 1.0),
                              atol=1e-8, batch=True, max_batch=90)
    assert sol.numevals == nev and abs(sol.u - I) <= 1e-12 * abs(I)


def test_config5_synthetic_16_band(abz):
    """BASELINE configs[4]: synthetic 16-band 3-D Wannier model (2197 R vectors, SURVEY 8d recipe),
    IAI DOS with host-driven panel re-batching on the GPU; cross-checked against PTR on the same
    device (as the reference cross-checks algorithms) and against the oracle's PTR rule."""
    so = orc.synthetic_wannier()
    assert so.c.shape == (13, 13, 13, 16, 16)
    s = abz.FourierSeries(so.c, period=1.0, first=so.first, ndim=3)
    H = s.device().eval_nodes(np.array([[0.1, 0.2, 0.3]]))[0]
    assert np.abs(H - orc.evaluate(so, [0.1, 0.2, 0.3])).max() < 1e-12
    assert np.abs(H - H.conj().T).max() < 1e-13
    bz = abz.load_bz(abz.FBZ(), np.eye(3))
    f = abz.FourierIntegrand(abz.DOSIntegrand(), s, 0.05)
    ptr = abz.solve(abz.IntegralProblem(f, bz, abz.MixedParameters(0.2)), abz.PTR(npt=12)).u
    ref = orc.solve_ptr(so, orc.load_bz("FBZ", np.eye(3)), orc.f_dos(0.05, 0.2), npt=12).u
    assert abs(ptr - ref) <= 1e-9 * abs(ref)
    # IAI at SURVEY 8d's / BASELINE's abstol = 1e-3 (2e-6 of the value ~ 481 = DOS * |det B|; 5.5e9 inner nodes in ~10 s)
    # against a reference that is converged an order of magnitude below that tolerance: store-free PTR sums on the 500^3 and
    # 600^3 grids (0.2 / 0.4 s) agree to 3e-8 per unit BZ volume = 8e-6 in the units of the solution (tools/c5_reference.py:
    # 240 -> 300 -> 400 -> 500 -> 600 -> 720 differ by 3e-6, 2e-6, 4e-7, 3e-8, 3e-8).  north_star: "integrals within the
    # solver's own abstol" -- the bar is abstol itself (measured error 4.9e-4).  Whether the panel decisions are the
    # reference's is tested where the oracle can follow (3 x 3 x 3 R-vectors, 16 bands: panels array_equal, equal numevals).
    sol = abz.solve(abz.IntegralProblem(f, bz, abz.MixedParameters(0.2)), abz.EvalCounter(abz.IAI()), abstol=1e-3, reltol=0.0)
    dev = s.device()
    detB = abs(np.linalg.det(bz.B))
    big = dev.ptr_sum(600, abz._lib.F_DOS, [0.05], [0.2])[0, 0].real * detB
    mid = dev.ptr_sum(500, abz._lib.F_DOS, [0.05], [0.2])[0, 0].real * detB
    assert abs(big - mid) < 1e-4
    assert abs(sol.u - big) <= 1e-3 and sol.resid <= 1e-3
    assert sol.numevals > 10**9
    # a looser tolerance keeps its promise too (measured 3.1e-3 at abstol 1e-2)
    sol2 = abz.solve(abz.IntegralProblem(f, bz, abz.MixedParameters(0.2)), abz.IAI(), abstol=1e-2, reltol=0.0)
    assert abs(sol2.u - big) <= 1e-2


# ------------------------------------------------------------------ SVO (configs 3 / 4)
@pytest.fixture(scope="module")
def svo(abz):
    s = abz.load_w90_series(os.path.join(GOLD, "svo_hr.dat.gz"))
    meta = json.load(open(os.path.join(GOLD, "svo_meta.json")))
    return s, meta


def test_svo_known_eigenvalues(abz, svo):
    """SURVEY Appendix B known answers (independent NumPy evaluation of the reference's data file)."""
    s, meta = svo
    assert s.c.shape == (11, 11, 11, 3, 3) and s.first == (-5, -5, -5)
    assert abs(np.abs(s.c.real).max() - meta["max_abs_re"]) < 1e-6
    for name, (k, e) in meta["eig_known"].items():
        H, E = s.device().eval_nodes(np.array([k]), want=3)
        assert np.abs(E[0] - np.array(e)).max() < 2e-6, name
        assert np.abs(H[0] - H[0].conj().T).max() < 1e-12


def test_svo_dos_sweep_matches_oracle(abz, svo):
    """Config 4 in miniature: PTR DOS sweep on FBZ and CubicSymIBZ, fused on the GPU, against the
    oracle's per-omega quadsum; batchsolve == serial map (ref: test/brillouin.jl:93-110)."""
    s, meta = svo
    so = orc.FourierSeries(s.c, period=1.0, first=s.first, ndim=3)
    A = meta["a_angstrom"] * np.eye(3)
    omegas = np.linspace(10, 15, 6)
    eta = 0.1
    for kind, bzk in (("FBZ", abz.FBZ()), ("CubicSymIBZ", abz.CubicSymIBZ())):
        bz = abz.load_bz(bzk, A)
        bzo = orc.load_bz(kind, A)
        assert abs(abs(np.linalg.det(bz.B)) - 4.31781) < 1e-4
        solver = abz.IntegralSolver(abz.FourierIntegrand(abz.DOSIntegrand(), s, eta), bz, abz.PTR(npt=24), abstol=1e-3)
        got = abz.batchsolve(solver, omegas)
        serial = np.array([solver(om) for om in omegas])
        assert np.array_equal(got, serial)
        for om, g in zip(omegas, got):
            ref = orc.solve_ptr(so, bzo, orc.f_dos(eta, om), npt=24).u
            assert abs(g - ref) <= 1e-10 * abs(ref)


def test_svo_autoptr_off_band_value(abz, svo):
    """Config 3: AutoPTR DOS at omega = 11.0 eV (off band, smooth): SURVEY 8d scratch value
    0.16325288 = 0.037809138 * |det B|, abstol 1e-3; grids 50 -> 100."""
    s, meta = svo
    bz = abz.load_bz(abz.CubicSymIBZ(), meta["a_angstrom"] * np.eye(3))
    f = abz.FourierIntegrand(abz.DOSIntegrand(), s, 0.1)
    sol = abz.solve(abz.IntegralProblem(f, bz, abz.MixedParameters(11.0)), abz.EvalCounter(abz.AutoPTR()), abstol=1e-3)
    assert abs(sol.u - 0.16325288) < 1e-3
    assert sol.numevals == 3276 + 23426  # irreducible nodes of npt = 50 and 100 (integer pin)
    sol_f = abz.solve(abz.IntegralProblem(f, abz.load_bz(abz.FBZ(), meta["a_angstrom"] * np.eye(3)), abz.MixedParameters(11.0)),
                      abz.AutoPTR(), abstol=1e-3)
    assert abs(sol_f.u - sol.u) < 1e-6


@pytest.mark.parametrize("kind,d", [("FBZ", 2), ("InversionSymIBZ", 2), ("CubicSymIBZ", 2), ("FBZ", 3)])
def test_autoptr_loop_inside_the_library(abz, kind, d, monkeypatch):
    """abz_autoptr_solve_many (grid sequence, rules kept by the series, store-free sums for grids used once, error test,
    numevals -- ref: src/algorithms.jl:418-432, src/fourier.jl:381-389) against the oracle's autosymptr and against the host
    loop that drives the same rules one call at a time (solver._autoptr_many with the library path switched off):
    equal integer `numevals` and grid sequence, values to 1e-11; a batch in lock-step takes the decisions of the solves one by
    one (values to the rounding of a scan that carries several swept values); new coefficients reach the kept rules;
    `keepmost` changes no value.  The 3-D case walks npt = 60, 100, 140: its third grid is summed on the fly."""
    from autobzcore.jl_amd import solver as S
    rng = np.random.default_rng(77)
    A = rng.standard_normal((3, 3)) + 1j * rng.standard_normal((3, 3))
    A = 0.5 * (A + A.conj().T)
    c = np.zeros((3,) * d + (3, 3), dtype=np.complex128)  # H(k) = sum_j cos(2 pi k_j) A + B: the cubic point group
    for i in range(d):
        for e_ in (0, 2):
            idx = [1] * d
            idx[i] = e_
            c[tuple(idx)] += 0.5 * A
    c[(1,) * d] += np.diag([0.3, -0.2, 0.1])
    s, so = both(abz, c, (-1,) * d)
    bzk = {"FBZ": abz.FBZ(), "InversionSymIBZ": abz.InversionSymIBZ(), "CubicSymIBZ": abz.CubicSymIBZ()}[kind]
    bz, bzo = abz.load_bz(bzk, 1.3 * np.eye(d)), orc.load_bz(kind, 1.3 * np.eye(d))
    if d == 2:
        seq = dict(nmin=5, n0=7.0, dn=6.0)  # npt = 7, 13, 19, ...: some thirty refinements, the oracle stays fast
        eta, omegas, tol = 0.12, [-0.4, 0.1, 0.8, 2.5], 1e-4
    else:
        seq = dict(nmin=40, n0=60.0, dn=40.0)  # |I(100) - I(60)| = 1.9e-4 at omega = 0.1 and 3.7e-5 at 2.5: three grids and two
        eta, omegas, tol = 0.2, [0.1, 2.5], 1e-4 * abs(np.linalg.det(bz.B))  # (abstol is divided by |det B|, src/brillouin.jl:433)
    alg = abz.AutoPTR(a=1.0, keepmost=2, **seq)
    f = abz.FourierIntegrand(abz.DOSIntegrand(), s, eta)
    lib = [abz.solve(abz.IntegralProblem(f, bz, abz.MixedParameters(om)), abz.EvalCounter(alg), abstol=tol) for om in omegas]
    monkeypatch.setattr(S, "_autoptr_library", lambda *a, **k: None)
    host = [abz.solve(abz.IntegralProblem(f, bz, abz.MixedParameters(om)), abz.EvalCounter(alg), abstol=tol) for om in omegas]
    monkeypatch.undo()
    grids = set()
    for om, a, b in zip(omegas, lib, host):
        assert a.numevals == b.numevals and a.extra["npt"] == b.extra["npt"] and abs(a.u - b.u) <= 1e-13 * abs(b.u)
        grids.add(a.extra["npt"])
        if d == 3 and om != omegas[0]:
            continue  # (the numpy oracle needs ~10 s per million nodes of a 3-D grid)
        ref = orc.solve_autoptr(so, bzo, orc.f_dos(eta, om), abstol=tol, **seq)
        assert a.numevals == ref.numevals, (om, a.numevals, ref.numevals)
        assert a.extra["npt"] == ref.extra["grids"][-1]
        assert abs(a.u - ref.u) <= 1e-11 * abs(ref.u)
        assert abs(a.resid - ref.resid) <= 1e-9 * abs(ref.u) + 1e-3 * ref.resid
    assert len(grids) >= 2  # the solves stop at different grids: the lock-step really shrinks
    if d == 3:
        assert lib[0].extra["npt"] == 140 and lib[0].numevals == 60**3 + 100**3 + 140**3
    solver = abz.IntegralSolver(f, bz, abz.EvalCounter(alg), abstol=tol)
    batch = abz.batchsolve(solver, np.array(omegas))
    assert all(abs(batch[i] - lib[i].u) <= 1e-13 * abs(lib[i].u) for i in range(len(omegas)))
    # the same sweep again: the scans of the kept grids beyond the second are launched ahead of the decisions (the solves of
    # the batch stop at different grids, so the sums launched ahead are picked per active solve) -- the same bits
    meta = []
    batch2 = abz.batchsolve(solver, np.array(omegas), callback=lambda s_, i, k, p, sol, t: meta.append((int(np.ravel(i)[0]), sol.numevals, sol.extra["npt"])))
    assert np.array_equal(np.asarray(batch), np.asarray(batch2))
    assert sorted(meta) == [(i, lib[i].numevals, lib[i].extra["npt"]) for i in range(len(omegas))]
    # keepmost = 0 (nothing kept: every grid on the fly or built and dropped) and a large one: the same numbers
    for km in (0, 9):
        alg2 = abz.AutoPTR(a=1.0, keepmost=km, **seq)
        s.device().drop_rules()
        for om, a in zip(omegas, lib):
            got = abz.solve(abz.IntegralProblem(f, bz, abz.MixedParameters(om)), abz.EvalCounter(alg2), abstol=tol)
            assert got.numevals == a.numevals and abs(got.u - a.u) <= 1e-13 * abs(a.u), (km, om)
    if d == 3:
        return
    # new coefficients of the same shape: the rules the series keeps refill themselves before their next use
    c2 = c.copy()
    c2[(1,) * d] += np.diag([0.05, 0.0, -0.05])
    s.device().update(c2)
    so2 = orc.FourierSeries(c2, period=1.0, first=(-1,) * d, ndim=d)
    got = abz.solve(abz.IntegralProblem(f, bz, abz.MixedParameters(0.1)), abz.EvalCounter(alg), abstol=tol)
    ref = orc.solve_autoptr(so2, bzo, orc.f_dos(eta, 0.1), abstol=tol, **seq)
    assert got.numevals == ref.numevals and abs(got.u - ref.u) <= 1e-11 * abs(ref.u)
    assert abs(got.u - lib[omegas.index(0.1)].u) > 1e-6 * abs(ref.u)  # ... and the value did move


def test_full_size_properties_config3(abz, svo):
    """BASELINE full size (SVO, 150^3 = 3.375 M nodes, 567 MB of rule values): size-independent
    properties instead of a node-by-node oracle comparison.
      * checksum of checksums: the grid mean of sum_b e_b(k) equals Re tr H_{R=0} exactly (PTR kills every
        R != 0 term), same for the mean of tr H(k) through the exported planes of one line;
      * symmetry: the cubic-irreducible rule (76 076 nodes, integer weights summing to 150^3) gives the
        same DOS sweep as the full grid;
      * the two integrand forms (matrix inverse vs cached eigenvalues) agree; rebuild is idempotent."""
    s, meta = svo
    npt = 150
    dev = s.device()
    rule = dev.rule(npt, None, want=3)
    omegas = np.linspace(10, 15, 9)
    a = rule.reduce(abz._lib.F_DOS, [0.1], omegas)[:, 0].real
    b = rule.reduce(abz._lib.F_DOS_EIG, [0.1], omegas)[:, 0].real
    assert np.abs(a - b).max() <= 1e-11 * np.abs(a).max()
    rule.rebuild()
    a2 = rule.reduce(abz._lib.F_DOS, [0.1], omegas)[:, 0].real
    assert np.array_equal(a, a2)
    eig = rule.export(x=False, w=False, eig=True)["eig"]
    assert eig.shape == (npt**3, 3) and np.all(np.diff(eig, axis=1) >= 0)
    tr0 = np.trace(s.c[5, 5, 5]).real
    assert abs(eig.sum(axis=1).mean() - tr0) <= 1e-11 * abs(tr0)
    cub = abz.load_bz(abz.CubicSymIBZ(), meta["a_angstrom"] * np.eye(3))
    rsym = dev.rule(npt, cub.syms, want=1)
    assert rsym.nk == math.comb(75 + 3, 3)
    w = rsym.export(x=False, w=True)["w"]
    assert w.sum() == npt**3
    c = rsym.reduce(abz._lib.F_DOS, [0.1], omegas)[:, 0].real * 48  # symmetrize: nsyms * u
    assert np.abs(c - a).max() <= 1e-10 * np.abs(a).max()
    dev.drop_rules()


def test_config2_one_band_64cubed(abz):
    """BASELINE configs[1]: 3-D 1-band tight binding, fixed 64^3 PTR grid, one omega:
    against the oracle node for node and for the DOS value."""
    so = orc.tb_integer(3)
    s = abz.FourierSeries(so.c, period=1.0, first=so.first, ndim=3)
    rule = s.device().rule(64, None, want=3)
    out = rule.export(H=True, eig=True)
    ref = np.transpose(orc.fourier_ptr(so, 64), (2, 1, 0, 3, 4)).reshape(-1, 1, 1)
    assert np.abs(out["H"] - ref).max() < 1e-13 and np.abs(out["eig"][:, 0] - ref[:, 0, 0].real).max() < 1e-13
    bz = abz.load_bz(abz.FBZ(), np.eye(3))
    u = abz.solve(abz.IntegralProblem(abz.FourierIntegrand(abz.DOSIntegrand(), s, 0.1), bz, abz.MixedParameters(0.5)), abz.PTR(npt=64)).u
    r = orc.solve_ptr(so, orc.load_bz("FBZ", np.eye(3)), orc.f_dos(0.1, 0.5), npt=64).u
    assert abs(u - r) <= 1e-11 * abs(r)


# ------------------------------------------------------------------ GGR
def test_ggr_matches_oracle_and_exact(abz):
    """ref: test/dos.jl:88-111 at the reference's npt = 200 (1-D, 2-D) and vs the oracle (3-D)."""
    from test_oracle_pins import dos_graphene_exact, dos_integer_1d_exact, dos_integer_2d_exact
    cases = [(orc.tb_graphene(), dos_graphene_exact, 4, "FBZ"), (orc.tb_integer(1), dos_integer_1d_exact, 2, "InversionSymIBZ"),
             (orc.tb_integer(2), dos_integer_2d_exact, 4, "CubicSymIBZ")]
    kinds = {"FBZ": abz.FBZ(), "InversionSymIBZ": abz.InversionSymIBZ(), "CubicSymIBZ": abz.CubicSymIBZ()}
    for so, exact, B, kind in cases:
        s = abz.FourierSeries(so.c, period=1.0, first=so.first, ndim=so.d)
        bz = abz.load_bz(kinds[kind], np.eye(so.d))
        Es = [-B - 1, -0.8 * B, -0.6 * B, -0.2 * B, 0.1 * B, 0.3 * B, 0.5 * B, 0.7 * B, 0.9 * B, B + 2]
        cache = abz.dos.init(abz.DOSProblem(s, 0.0, bz), abz.GGR(npt=200))
        ref = orc.dos_ggr(so, orc.load_bz(kind, np.eye(so.d)), Es, npt=200)
        for e, r in zip(Es, ref):
            cache.domain = e
            u = abz.dos.solve_(cache).u
            assert abs(u - exact(e)) < 1e-2
            assert abs(u - r) <= 1e-9 * max(1.0, abs(r))
    # 3-D, 3 bands, all symmetry kinds vs the oracle
    rng = np.random.default_rng(2)
    so3 = orc.tb_integer(3)
    for kind in ("FBZ", "InversionSymIBZ", "CubicSymIBZ"):
        s = abz.FourierSeries(so3.c, period=1.0, first=so3.first, ndim=3)
        Es = np.linspace(-5.5, 5.5, 9)
        u = abz.dos.solve(abz.DOSProblem(s, Es, abz.load_bz(kinds[kind], np.eye(3))), abz.GGR(npt=24)).u
        ref = orc.dos_ggr(so3, orc.load_bz(kind, np.eye(3)), Es, npt=24)
        assert np.abs(u - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())


def test_ggr_velocities_match_oracle(abz, svo):
    s, _ = svo
    so = orc.FourierSeries(s.c, period=1.0, first=s.first, ndim=3)
    rule = s.device().rule(6, None, want=2 | 4)
    out = rule.export(eig=True, vel=True)
    w, e, v = orc.get_ggr_data(so, 6, None)
    assert np.abs(out["eig"] - e).max() < 1e-11
    # velocities are basis dependent inside degenerate subspaces: compare sums over
    # (near-)degenerate sets, i.e. sort-insensitive per-node totals, and exact values elsewhere
    gap = np.min(np.diff(e, axis=1), axis=1)
    ok = gap > 1e-6
    assert np.abs(out["vel"][ok] - v[ok]).max() < 1e-8
    assert np.abs(out["vel"].sum(axis=2) - v.sum(axis=2)).max() < 1e-8


@pytest.mark.parametrize("n", [6, 12, 16, 17, 24, 32])
def test_ggr_more_than_four_bands(abz, n, monkeypatch):
    """GGR for n > 4 (ref: src/dos_ggr.jl:1-44 falls back to LAPACK's eigen there): eigenvalues, band velocities
    and the scanned DOS of 6-, 12- and 16-band models against the oracle (up to 8 bands: row-layout Jacobi with accumulated
    rotations; 9..16: Householder + bisection eigenvalues and inverse-iteration eigenvectors)."""
    so = orc.synthetic_wannier(n=n, rmax=2, seed=7)
    s = abz.FourierSeries(so.c, period=1.0, first=so.first, ndim=3)
    w, e, v = orc.get_ggr_data(so, 6, None)
    ok = np.min(np.diff(e, axis=1), axis=1) > 1e-6
    assert ok.sum() > 100
    # the fused row-layout build (kernels_ggr_rows.hip) and the unfused one (eigenvectors + a velocity launch per variable)
    for env in ({}, {"ABZ_GGR_FUSED": "0"}):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        rule = abz.DeviceRule(s.device(), 6, None, 2 | 4)
        out = rule.export(eig=True, vel=True)
        rule.close()
        for k_ in env:
            monkeypatch.delenv(k_)
        assert np.abs(out["eig"] - e).max() < 1e-11, env
        assert np.abs(out["vel"][ok] - v[ok]).max() < 1e-8, env
        assert np.abs(out["vel"].sum(axis=2) - v.sum(axis=2)).max() < 1e-8, env
    Es = np.linspace(-2.0, 2.0, 9)
    for kind, bzk in (("FBZ", abz.FBZ()), ("InversionSymIBZ", abz.InversionSymIBZ())):
        u = abz.dos.solve(abz.DOSProblem(s, Es, abz.load_bz(bzk, np.eye(3))), abz.GGR(npt=10)).u
        ref = orc.dos_ggr(so, orc.load_bz(kind, np.eye(3)), Es, npt=10)
        assert np.abs(ref).max() > 0.1
        assert np.abs(u - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("n", [33, 40, 48, 57, 64])
def test_ggr_33_to_64_bands(abz, n):
    """GGR builds for 33...64 bands (ref: src/dos_ggr.jl:14-44, LAPACK there): kernels_big.hip (H, tridiagonal with the
    reflectors kept, eigenvalues, derivative matrices) + kernels_big_vec.hip (per-lane inverse iteration, back-transformation
    and quadratic forms, two waves per node) -- eigenvalues, band velocities and the scanned DOS on the full grid and on an
    inversion-symmetric node list against the oracle."""
    so = orc.synthetic_wannier(n=n, rmax=2, seed=7)
    s = abz.FourierSeries(so.c, period=1.0, first=so.first, ndim=3)
    w, e, v = orc.get_ggr_data(so, 6, None)
    scale, vscale = np.abs(e).max(), np.abs(v).max()
    ok = np.min(np.diff(e, axis=1), axis=1) > 1e-6 * scale
    assert ok.sum() > 100
    rule = abz.DeviceRule(s.device(), 6, None, 2 | 4)
    out = rule.export(eig=True, vel=True)
    rule.close()
    assert np.abs(out["eig"] - e).max() < 1e-11 * scale
    assert np.abs(out["vel"][ok] - v[ok]).max() < 1e-8 * vscale
    assert np.abs(out["vel"].sum(axis=2) - v.sum(axis=2)).max() < 1e-8 * vscale * n
    if n in (33, 64):  # the matrices as well (a rule that serves G scans and GGR)
        rule = abz.DeviceRule(s.device(), 6, None, 1 | 2 | 4)
        o7 = rule.export(H=True, eig=True, vel=True)
        rule.close()
        href = np.transpose(orc.fourier_ptr(so, 6), (2, 1, 0, 3, 4)).reshape(-1, n, n)
        assert np.abs(o7["H"] - href).max() <= 1e-12 * np.abs(href).max()
        assert np.array_equal(o7["eig"], out["eig"]) and np.array_equal(o7["vel"], out["vel"])
    Es = np.linspace(-2.0, 2.0, 9)
    for kind, bzk in (("FBZ", abz.FBZ()), ("InversionSymIBZ", abz.InversionSymIBZ())):
        u = abz.dos.solve(abz.DOSProblem(s, Es, abz.load_bz(bzk, np.eye(3))), abz.GGR(npt=8)).u
        ref = orc.dos_ggr(so, orc.load_bz(kind, np.eye(3)), Es, npt=8)
        assert np.abs(ref).max() > 0.1
        assert np.abs(u - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())
    # one and two dimensions
    for dims in ((5,), (3, 5)):
        rng = np.random.default_rng(100 * n + len(dims))
        c, first = rand_series(rng, dims, n, hermitian=True)
        s2, so2 = both(abz, c / np.sqrt(n), first)
        npt = 11 if len(dims) == 1 else 7
        w2, e2, v2 = orc.get_ggr_data(so2, npt, None)
        rule = abz.DeviceRule(s2.device(), npt, None, 2 | 4)
        o2 = rule.export(eig=True, vel=True)
        rule.close()
        sep = np.min(np.diff(e2, axis=1), axis=1) > 1e-6 * np.abs(e2).max()
        assert np.abs(o2["eig"] - e2).max() < 1e-11 * np.abs(e2).max()
        assert np.abs(o2["vel"][sep] - v2[sep]).max() < 1e-8 * np.abs(v2).max()
        # ... and the inversion-symmetric node list in the same dimension (one variable: a list without level-1 parents)
        sy = orc.load_bz("InversionSymIBZ", np.eye(len(dims))).syms
        w3, e3, v3 = orc.get_ggr_data(so2, npt, sy)
        rule = abz.DeviceRule(s2.device(), npt, sy, 2 | 4)
        o3 = rule.export(eig=True, vel=True)
        rule.close()
        sep3 = np.min(np.diff(e3, axis=1), axis=1) > 1e-6 * np.abs(e3).max()
        assert np.abs(o3["eig"] - e3).max() < 1e-11 * np.abs(e3).max()
        assert np.abs(o3["vel"][sep3] - v3[sep3]).max() < 1e-8 * np.abs(v3).max()


def test_33_to_64_bands_in_many_chunks(abz, monkeypatch):
    """33...64 bands work through their nodes in chunks that fit a scratch budget (ABZ_BIG_CHUNK_MB: 256 MB, 2 GB for GGR builds --
    more than any other test's grid needs): with 1 MB -- a few grid lines per chunk, and for the node list chunks that end inside
    a run -- every node's values must be those of a single chunk, bit for bit, and the sums equal to rounding (tridiagonalisation
    with one and with four waves)."""
    L = abz._lib
    so = orc.synthetic_wannier(n=40, rmax=1, seed=11)
    s = abz.FourierSeries(so.c, period=1.0, first=so.first, ndim=3)
    syms = orc.load_bz("InversionSymIBZ", np.eye(3)).syms
    om = np.array([-0.5, 0.4])

    def everything():
        res = []
        for sy in (None, syms):
            r = abz.DeviceRule(s.device(), 7, sy, 1 | 2 | 4)
            o = r.export(H=True, eig=True, vel=True)
            res += [o["H"], o["eig"], o["vel"], r.reduce(L.F_GLOC, [0.3], om), r.reduce(L.F_TRGLOC, [0.3], om)]
            r.close()
        res.append(s.device().ptr_sum(7, L.F_DOS, [0.3], om))
        return res

    for waves in ("1", "4"):
        monkeypatch.setenv("ABZ_BIG_TRI_WAVES", waves)
        monkeypatch.delenv("ABZ_BIG_CHUNK_MB", raising=False)
        whole = everything()
        monkeypatch.setenv("ABZ_BIG_CHUNK_MB", "1")
        parts = everything()
        for a, b in zip(whole, parts):  # values per node: the same bits; sums: another order of the partial sums
            assert np.array_equal(a, b) if a.shape[0] > 2 else np.abs(a - b).max() <= 1e-13 * np.abs(a).max(), waves
    w, e, v = orc.get_ggr_data(so, 7, None)
    assert np.abs(whole[1] - e).max() < 1e-11 * np.abs(e).max()


@pytest.mark.parametrize("n3,mult", [(11, 3), (5, 8), (8, 8), (32, 2)])
def test_ggr_33_to_64_bands_degenerate(abz, n3, mult):
    """Exactly degenerate levels everywhere at 33...64 bands (H = Q (I_mult x h(k)) Q^H): the cluster rounds of kernels_big_vec.hip
    -- eigenvalues, velocity sums over each degenerate set and the scanned DOS against the oracle."""
    rng = np.random.default_rng(70 * n3 + mult)
    c3, first = rand_series(rng, (3, 3, 3), n3, hermitian=True)
    n = n3 * mult
    q, _ = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))
    c = np.einsum("ab,...bc,dc->...ad", q, np.kron(np.eye(mult), c3 / np.sqrt(n3)), q.conj())
    c = 0.5 * (c + np.conj(np.swapaxes(c[::-1, ::-1, ::-1], -1, -2)))  # c(-R) = c(R)^H to the bit: the Hermitian kernels' condition
    s, so = both(abz, c, first)
    npt = 5
    w, e, v = orc.get_ggr_data(so, npt, None)
    rule = abz.DeviceRule(s.device(), npt, None, 2 | 4)
    out = rule.export(eig=True, vel=True)
    rule.close()
    scale, vscale = np.abs(e).max(), np.abs(v).max()
    assert np.abs(out["eig"] - e).max() <= 1e-11 * scale
    grp = lambda a: a.reshape(a.shape[0], a.shape[1], n3, mult).sum(axis=3)
    e3 = e.reshape(len(e), n3, mult)[:, :, 0]
    sep = np.min(np.diff(e3, axis=1), axis=1) > 1e-6 * scale
    assert sep.mean() > 0.5
    assert np.abs(grp(out["vel"])[sep] - grp(v)[sep]).max() <= 1e-8 * vscale * mult
    assert np.abs(out["vel"].sum(axis=2) - v.sum(axis=2)).max() <= 1e-8 * vscale * n
    Es = np.linspace(-3.0, 3.0, 7)
    u = abz.dos.solve(abz.DOSProblem(s, Es, abz.load_bz(abz.FBZ(), np.eye(3))), abz.GGR(npt=6)).u
    ref = orc.dos_ggr(so, orc.load_bz("FBZ", np.eye(3)), Es, npt=6)
    assert np.abs(u - ref).max() <= 1e-8 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("n3,mult", [(3, 2), (2, 4), (3, 4), (5, 2), (3, 8), (7, 4)])
def test_ggr_rows_degenerate_bands(abz, n3, mult):
    """Exactly degenerate levels everywhere (H = Q (I_mult x h(k)) Q^H with a fixed unitary Q): the per-lane inverse iteration
    of the row-layout GGR build must return an orthonormal basis of every eigenspace -- eigenvalues against LAPACK, velocity
    SUMS over each degenerate set against the oracle (any basis of the set is an answer, ref src/dos_ggr.jl:31-44), and the
    scanned DOS; grid sizes that leave waves without nodes and lines shorter / longer than a pass."""
    rng = np.random.default_rng(7 * n3 + mult)
    c3, first = rand_series(rng, (3, 3, 3), n3, hermitian=True)
    n = n3 * mult
    q, _ = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))
    c = np.einsum("ab,...bc,dc->...ad", q, np.kron(np.eye(mult), c3), q.conj())
    s, so = both(abz, c, first)
    for npt in (5, 19):
        w, e, v = orc.get_ggr_data(so, npt, None)
        rule = abz.DeviceRule(s.device(), npt, None, 2 | 4)
        out = rule.export(eig=True, vel=True)
        rule.close()
        scale, vscale = np.abs(e).max(), np.abs(v).max()
        assert np.abs(out["eig"] - e).max() <= 1e-11 * scale
        grp = lambda a: a.reshape(a.shape[0], a.shape[1], n3, mult).sum(axis=3) if a.ndim == 3 else None
        # bands come in runs of `mult` equal values: sums over each run (distinct levels separated by > 1e-6 only)
        e3 = e.reshape(len(e), n3, mult)[:, :, 0]
        sep = np.min(np.diff(e3, axis=1), axis=1) > 1e-6 * scale if n3 > 1 else np.ones(len(e), dtype=bool)
        assert sep.mean() > 0.9
        assert np.abs(grp(out["vel"])[sep] - grp(v)[sep]).max() <= 1e-8 * vscale * mult
        assert np.abs(out["vel"].sum(axis=2) - v.sum(axis=2)).max() <= 1e-8 * vscale * n
    Es = np.linspace(-3.0, 3.0, 7)
    u = abz.dos.solve(abz.DOSProblem(s, Es, abz.load_bz(abz.FBZ(), np.eye(3))), abz.GGR(npt=12)).u
    ref = orc.dos_ggr(so, orc.load_bz("FBZ", np.eye(3)), Es, npt=12)
    assert np.abs(u - ref).max() <= 1e-8 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("d,n,dims", [(1, 8, (5,)), (2, 6, (3, 5)), (2, 12, (5, 3)), (3, 7, (3, 3, 5)), (3, 32, (11, 3, 3)), (3, 20, (13, 3, 3))])
def test_ggr_rows_dimensions_and_chunked_sets(abz, d, n, dims):
    """The row-layout GGR build in 1, 2 and 3 dimensions (ref: src/dos_ggr.jl:14-44 for any N), with periods other than one
    (the velocities carry the period, src/dos_ggr.jl:20,35), on full grids and on an inversion-symmetric node list, and for
    level-1 sets that do not fit the LDS whole (32 bands x 11 coefficients: staged in chunks) -- against the oracle."""
    rng = np.random.default_rng(1000 * d + n)
    c, first = rand_series(rng, dims, n, hermitian=True)
    period = (1.0, 2.0, 0.5)[:d]
    s, so = both(abz, c, first, period, ndim=d)
    for npt in ((9, 40) if d == 1 else (7,)):
        w, e, v = orc.get_ggr_data(so, npt, None)
        rule = abz.DeviceRule(s.device(), npt, None, 2 | 4)
        out = rule.export(eig=True, vel=True)
        rule.close()
        scale, vscale = np.abs(e).max(), np.abs(v).max()
        ok = np.min(np.diff(e, axis=1), axis=1) > 1e-6 * scale
        assert ok.mean() > 0.9
        assert np.abs(out["eig"] - e).max() <= 1e-11 * scale
        assert np.abs(out["vel"][ok] - v[ok]).max() <= 1e-8 * vscale
        assert np.abs(out["vel"].sum(axis=2) - v.sum(axis=2)).max() <= 1e-9 * vscale * n
    if d >= 2 and n <= 12:  # BZ solves: period one (the reference's BZ integrals assume it)
        s, so = both(abz, c, first, 1.0, ndim=d)
        Es = np.linspace(-4.0, 4.0, 5)
        for kind, bzk in (("FBZ", abz.FBZ()), ("InversionSymIBZ", abz.InversionSymIBZ())):
            u = abz.dos.solve(abz.DOSProblem(s, Es, abz.load_bz(bzk, np.eye(d))), abz.GGR(npt=8)).u
            ref = orc.dos_ggr(so, orc.load_bz(kind, np.eye(d)), Es, npt=8)
            assert np.abs(u - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), kind


def test_config4_share_at_full_size_three_forms_agree(abz):
    """BASELINE configs[3] at the size the bench runs it (SVO, 150^3 PTR grid, 256 omega in [10, 15] eV, eta = 0.1): the sweep
    from the cached matrices (the reference's own form, src/fourier.jl:204-207), from the cached eigenvalues, and over the
    cubic IBZ (48 symmetries, weights from symptr_rule, src/fourier.jl:289-292) must be the same 256 numbers -- a
    size-independent property of the path, checked where the oracle is too slow to follow."""
    L = abz._lib
    s = abz.load_w90_series(os.path.join(GOLD, "svo_hr.dat.gz"))
    dev = s.device()
    om = np.linspace(10.0, 15.0, 256)
    rule = dev.rule(150, None, want=L.WANT_H | L.WANT_EIG)
    a = rule.reduce(L.F_DOS, [0.1], om)[:, 0].real
    b = rule.reduce(L.F_DOS_EIG, [0.1], om)[:, 0].real
    cub = abz.load_bz(abz.CubicSymIBZ(), 3.85856 * np.eye(3))
    rs = dev.rule(150, cub.syms, want=L.WANT_H | L.WANT_EIG)
    assert rs.nk_local == 76076  # C(npt // 2 + 3, 3) irreducible nodes
    # (a symmetric rule's value carries dvol = 1 / (npt^d nsyms), src/fourier.jl:289-292; TrivialRep multiplies by nsyms, src/brillouin.jl:107)
    c = rs.reduce(L.F_DOS, [0.1], om)[:, 0].real * len(cub.syms)
    scale = np.abs(a).max()
    assert scale > 0.1
    assert np.abs(a - b).max() <= 1e-10 * scale
    assert np.abs(a - c).max() <= 1e-10 * scale
    # and the store-free sum of the same grid (abz_ptr_sum) for a few of the values
    d_ = dev.ptr_sum(150, L.F_DOS, [0.1], om[::37])[:, 0].real
    assert np.abs(d_ - a[::37]).max() <= 1e-10 * scale
    dev.drop_rules()


def _ggr_rule_data(abz, s, npt, syms=None):
    from autobzcore.jl_amd import _lib as L
    rule = abz.DeviceRule(s.device(), npt, syms, L.WANT_EIG | L.WANT_VEL)
    out = rule.export(x=False, w=False, eig=True, vel=True)
    return rule, out["eig"], out["vel"]


@pytest.mark.parametrize("d,n", [(1, 1), (1, 3), (2, 2), (2, 3), (2, 4), (3, 1), (3, 2), (3, 3), (3, 4)])
def test_fused_ggr_build_matches_oracle_and_unfused_build(abz, d, n, monkeypatch):
    """The one-kernel GGR build (kernels_ggr.hip: packed Hermitian sets, projector velocities) against the oracle's
    get_ggr_data (ref: src/dos_ggr.jl:14-44) and against the unfused round-2 build, for every band count and dimension it
    covers, grids that exercise whole passes, one-node-per-lane tails, odd line counts and the padding columns; also the
    variant that reads level-1 families instead of contracting in the kernel (ABZ_GGR_FUSE2=0)."""
    rng = np.random.default_rng(100 * d + n)
    dims = (5, 7, 3)[:d]
    c, first = rand_series(rng, dims, n, hermitian=True)
    period = (1.0, 2.0, 0.5)[:d]
    s, so = both(abz, c, first, period, ndim=d)
    for npt in ((7, 33, 70, 100, 150) if d < 3 else (6, 33, 70)):
        w, e, v = orc.get_ggr_data(so, npt, None)
        scale, vscale = np.abs(e).max(), np.abs(v).max()
        ok = np.min(np.diff(e, axis=1), axis=1) > 1e-6 * scale if n > 1 else np.ones(len(e), dtype=bool)
        assert ok.mean() > 0.99
        got = {}
        for name, env in (("fused", {}), ("lines", {"ABZ_GGR_FUSE2": "0"}), ("unfused", {"ABZ_GGR_FUSED": "0"})):
            for k_, v_ in env.items():
                monkeypatch.setenv(k_, v_)
            rule, E, V = _ggr_rule_data(abz, s, npt)
            rule.close()
            for k_ in env:
                monkeypatch.delenv(k_)
            got[name] = (E, V)
            assert np.abs(E - e).max() <= 1e-12 * scale, (name, npt)
            assert np.abs(V[ok] - v[ok]).max() <= 1e-9 * vscale, (name, npt)
            assert np.abs(V.sum(axis=2) - v.sum(axis=2)).max() <= 1e-10 * vscale * n, (name, npt)
        assert np.abs(got["fused"][0] - got["unfused"][0]).max() <= 1e-12 * scale
        assert np.abs(got["fused"][1][ok] - got["unfused"][1][ok]).max() <= 1e-9 * vscale
        assert np.array_equal(got["fused"][0], got["lines"][0]) or np.abs(got["fused"][0] - got["lines"][0]).max() <= 1e-13 * scale


def test_fused_ggr_build_above_64_kb_of_lds(abz, monkeypatch):
    """A 4-band, 3-D model with 11 coefficients per variable makes the fused build ask for 66-68 KB of dynamic LDS
    (16 (2*11*90 + npt + 24*90) B): the launch needs hipFuncAttributeMaxDynamicSharedMemorySize.  Against the
    oracle on a small grid and against the unfused build at npt = 100."""
    rng = np.random.default_rng(411)
    c, first = rand_series(rng, (11, 11, 11), 4, hermitian=True)
    c *= np.exp(-0.5 * np.abs(np.arange(-5, 6)))[:, None, None, None, None]
    s, so = both(abz, c, first, 1.0, ndim=3)
    w, e, v = orc.get_ggr_data(so, 12, None)
    scale, vscale = np.abs(e).max(), np.abs(v).max()
    ok = np.min(np.diff(e, axis=1), axis=1) > 1e-6 * scale
    rule, E, V = _ggr_rule_data(abz, s, 12)
    rule.close()
    assert np.abs(E - e).max() <= 1e-12 * scale
    assert np.abs(V[ok] - v[ok]).max() <= 1e-9 * vscale
    rule, E, V = _ggr_rule_data(abz, s, 100)
    rule.close()
    monkeypatch.setenv("ABZ_GGR_FUSED", "0")
    rule, E0, V0 = _ggr_rule_data(abz, s, 100)
    rule.close()
    monkeypatch.delenv("ABZ_GGR_FUSED")
    ok = np.min(np.diff(E0, axis=1), axis=1) > 1e-6 * scale
    assert ok.mean() > 0.99
    assert np.abs(E - E0).max() <= 1e-12 * scale
    assert np.abs(V[ok] - V0[ok]).max() <= 1e-9 * vscale


def test_fused_ggr_build_degenerate_and_clustered_bands(abz):
    """Eigenvalues of the fused build to 1e-12 ||H|| for exactly / nearly degenerate 3-band spectra (constant series:
    H(k) = c[0], velocities zero), and velocities that stay finite and sum to tr dH/dk_j on a model with symmetry-enforced
    degeneracies (the SVO bands along the cube axes and diagonals)."""
    rng = np.random.default_rng(7)
    for gap in (0.0, 1e-14, 1e-10, 1e-7, 1e-5, 1e-4, 2e-3, 1e-2, 1.0):
        for trip in range(4):
            q, _ = np.linalg.qr(rng.standard_normal((3, 3)) + 1j * rng.standard_normal((3, 3)))
            base = rng.uniform(-5, 5)
            for ev in ([base, base + gap, base + 3.0], [base - 2.0, base, base + gap], [base, base + gap, base + 2 * gap]):
                A = (q * np.array(ev)) @ q.conj().T
                A = 0.5 * (A + A.conj().T)
                s = abz.FourierSeries(A.reshape(1, 1, 3, 3), period=1.0, first=(0, 0), ndim=2)
                rule, E, V = _ggr_rule_data(abz, s, 4)
                rule.close()
                ref = np.linalg.eigvalsh(A)
                # 1e-12 ||H|| like every eigenvalue of the suite: roots of the characteristic cubic carry eps ||B||^3 / |p'(w)|,
                # pairs closer than ~2e-3 of the scale go through the Jacobi iteration instead
                assert np.abs(E - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max()), (gap, ev)
                assert np.abs(V).max() == 0.0
    s = abz.load_w90_series(os.path.join(GOLD, "svo_hr.dat.gz"))
    so = orc.FourierSeries(s.c, period=1.0, first=s.first, ndim=3)
    rule, E, V = _ggr_rule_data(abz, s, 24)
    rule.close()
    w, e, v = orc.get_ggr_data(so, 24, None)
    assert np.abs(E - e).max() < 1e-11
    ok = np.min(np.diff(e, axis=1), axis=1) > 1e-6
    assert 0.5 < ok.mean() < 1.0  # the grid does hit degenerate nodes
    assert np.isfinite(V).all()
    assert np.abs(V[ok] - v[ok]).max() < 1e-8
    assert np.abs(V.sum(axis=2) - v.sum(axis=2)).max() < 1e-8


def test_ggr_windowed_scan_equals_the_formula_on_every_pair(abz, svo, monkeypatch):
    """abz_rule_ggr: the windowed scan (each band's energy window, per-wave histograms) gives the sums of the oracle's
    ggr_formula evaluated on EVERY (node, band, energy) of the rule's own (e, v, w), for unsorted energy lists with duplicates
    and values outside the band, beyond one 1024-energy chunk, with symmetric-rule weights and for 1 / 2 / 3 dimensions.
    ref: src/dos_ggr.jl:58-104."""
    from autobzcore.jl_amd import _lib as L
    s, _ = svo
    rng = np.random.default_rng(3)
    cases = [(s, 20, None), (s, 20, abz.load_bz(abz.CubicSymIBZ(), np.eye(3)).syms)]
    for dd in (1, 2):
        so = orc.tb_integer(dd)
        cases.append((abz.FourierSeries(so.c, period=1.0, first=so.first, ndim=dd), 64, None))
    for ser, npt, syms in cases:
        rule = abz.DeviceRule(ser.device(), npt, syms, L.WANT_EIG | L.WANT_VEL)
        ex = rule.export(x=False, w=True, eig=True, vel=True)
        E, V, W = ex["eig"], ex["vel"], np.asarray(ex["w"], dtype=np.float64)
        lo, hi = E.min(), E.max()
        for nE in (1, 7, 300, 2500):
            Es = rng.uniform(lo - 0.3, hi + 0.3, size=nE)
            if nE > 3:
                Es[3] = Es[1]
                Es[2] = lo - 10.0
            a = rule.ggr(Es)
            b = np.array([orc.sum_ggr(ser.d, npt, En, W, E, V) for En in Es[:400]])
            assert np.isfinite(a).all() and (nE < 300 or np.abs(b).max() > 0)
            assert np.abs(a[:400] - b).max() <= 1e-11 * max(np.abs(b).max(), 1e-300), (npt, nE)
            if nE > 3:
                assert a[3] == a[1] and a[2] == 0.0
        # equispaced sweeps: the window's first index comes from arithmetic instead of the search -- the same sums to the bit
        for Es in (np.linspace(lo - 0.2, hi + 0.2, 257), np.linspace(lo + 0.3 * (hi - lo), lo + 0.31 * (hi - lo), 1500)):
            a = rule.ggr(Es)
            monkeypatch.setenv("ABZ_GGR_UNIFORM", "0")
            b = rule.ggr(Es)
            monkeypatch.delenv("ABZ_GGR_UNIFORM")
            assert np.array_equal(a, b) and np.abs(a).max() > 0
        rule.close()


@pytest.mark.parametrize("kind,d", [("InversionSymIBZ", 1), ("InversionSymIBZ", 2), ("CubicSymIBZ", 2), ("InversionSymIBZ", 3), ("CubicSymIBZ", 3)])
def test_symmetric_rule_built_on_the_device_equals_the_host_built_rule(abz, kind, d, monkeypatch):
    """abz_ptr_rule_build_sym (orbit tables, contraction plan and values on the device; ref: the FourierMonkhorstPack
    constructor, src/fourier.jl:265-277) against abz_symptr_rule + abz_ptr_rule_build with the host's node list: nodes and
    integer weights array_equal (and equal to the oracle's symptr_rule), values and rule sums bit-identical; grids with
    even / odd npt, lines without nodes, one-node lines; a second series re-uses the cached tables."""
    from autobzcore.jl_amd import _lib as L
    rng = np.random.default_rng(5 * d + len(kind))
    c, first = rand_series(rng, (3, 5, 3)[:d], 3, hermitian=True)
    s, so = both(abz, c, first, 1.0, ndim=d)
    kinds = {"InversionSymIBZ": abz.InversionSymIBZ(), "CubicSymIBZ": abz.CubicSymIBZ()}
    syms = abz.load_bz(kinds[kind], np.eye(d)).syms
    osyms = orc.load_bz(kind, np.eye(d)).syms
    om = np.linspace(-1.0, 1.0, 5)
    for npt in ((1, 2, 7, 24, 65, 150) if d < 3 else (1, 2, 7, 24, 33)):
        for want in (L.WANT_H | L.WANT_EIG, L.WANT_EIG | L.WANT_VEL):
            rd = abz.DeviceRule(s.device(), npt, syms, want)
            monkeypatch.setenv("ABZ_SYM_DEVICE", "0")
            rh = abz.DeviceRule(s.device(), npt, syms, want)
            monkeypatch.delenv("ABZ_SYM_DEVICE")
            assert rd.nk == rh.nk
            a = rd.export(H=bool(want & L.WANT_H), eig=True, vel=bool(want & L.WANT_VEL))
            b = rh.export(H=bool(want & L.WANT_H), eig=True, vel=bool(want & L.WANT_VEL))
            for key in b:
                assert np.array_equal(a[key], b[key]), (npt, want, key)
            if want & L.WANT_H:
                assert np.array_equal(rd.reduce(L.F_DOS, [0.3], om), rh.reduce(L.F_DOS, [0.3], om))
            else:
                assert np.array_equal(rd.ggr(om), rh.ggr(om))
            rd.close()
            rh.close()
        wo, xo, _, idxo = orc.fourier_symptr(so, npt, osyms)
        assert np.array_equal(a["x"], xo) and np.array_equal(a["w"], wo.astype(float))  # integer parity with the oracle
    c2, _ = rand_series(rng, (3, 5, 3)[:d], 3, hermitian=True)
    s2, _ = both(abz, c2, first, 1.0, ndim=d)
    r2 = abz.DeviceRule(s2.device(), 24, syms, L.WANT_H)
    monkeypatch.setenv("ABZ_SYM_DEVICE", "0")
    r3 = abz.DeviceRule(s2.device(), 24, syms, L.WANT_H)
    monkeypatch.delenv("ABZ_SYM_DEVICE")
    assert np.array_equal(r2.reduce(L.F_DOS, [0.3], om), r3.reduce(L.F_DOS, [0.3], om))


def test_ggr_cache_invalidation(abz):
    """ref: test/dos.jl:114-132."""
    h = abz.FourierSeries(np.array([0.5, 0.0, 0.5]).reshape(3, 1, 1), period=1.0, offset=-2, ndim=1)
    bz = abz.load_bz(abz.FBZ(), [[2 * np.pi]])
    cache = abz.dos.init(abz.DOSProblem(h, 0.3, bz), abz.GGR(npt=100))
    sol1 = abz.dos.solve_(cache).u
    h.c *= 2
    cache.isfresh = True
    cache.domain = 0.6
    sol2 = abz.dos.solve_(cache).u
    assert sol1 > 0 and abs(sol1 / 2 - sol2) < 1e-12
    cache.H = abz.FourierSeries(2 * h.c, period=h.t, offset=h.o[0], ndim=1)
    cache.domain = 1.2
    sol3 = abz.dos.solve_(cache).u
    assert abs(sol2 / 2 - sol3) < 1e-12


def test_coefficient_update_refreshes_cached_rules_and_handles_stay_safe(abz):
    """A cached PTR rule follows `update()` / `invalidate()` (the reference rebuilds its rule from the current series on
    every solve, src/interfaces.jl:174-179): PTR, AutoPTR and GGR values after the update equal the oracle's on the new
    coefficients.  Handles retained across an invalidation stay valid (two DOS caches on one series, a user-held rule),
    closed handles raise instead of passing freed pointers, and the C destroy calls may come in any order."""
    rng = np.random.default_rng(99)
    c, first = rand_series(rng, (3, 3, 3), 3, hermitian=True)
    s, so = both(abz, c, first)
    bz, obz = abz.load_bz(abz.CubicSymIBZ(), np.eye(3)), orc.load_bz("CubicSymIBZ", np.eye(3))
    eta, om = 0.3, 0.2
    solver = abz.IntegralSolver(abz.FourierIntegrand(abz.DOSIntegrand(), s, eta), bz, abz.PTR(npt=12))
    u0 = solver(om)
    assert abs(u0 - orc.solve_ptr(so, obz, orc.f_dos(eta, om), npt=12).u) <= 1e-10 * abs(u0)
    held = s.device().rule(12, bz.syms, abz._lib.WANT_H)  # a user-held handle to the cached rule
    c2 = 0.5 * c
    c2[1, 1, 1] += np.diag([0.4, -0.1, 0.2])
    s.device().update(c2)
    so2 = orc.FourierSeries(c2, period=1.0, first=first, ndim=3)
    u1 = solver(om)
    ref1 = orc.solve_ptr(so2, obz, orc.f_dos(eta, om), npt=12).u
    assert abs(u1 - ref1) <= 1e-10 * abs(ref1) and abs(u1 - u0) > 1e-3 * abs(u0)
    assert abs(held.reduce(abz._lib.F_DOS, [eta], [om], nsyms=48)[0, 0].real * 48 * abs(np.linalg.det(bz.B)) - ref1) <= 1e-10 * abs(ref1)
    # a non-Hermitian update flips the rule's Hermitian flag with the refill
    c3 = c2.copy()
    c3[0, 0, 0, 0, 1] += 0.3
    s.device().update(c3)
    so3 = orc.FourierSeries(c3, period=1.0, first=first, ndim=3)
    fbz, ofbz = abz.load_bz(abz.FBZ(), np.eye(3)), orc.load_bz("FBZ", np.eye(3))
    sol = abz.IntegralSolver(abz.FourierIntegrand(abz.TrGlocIntegrand(), s, eta), fbz, abz.PTR(npt=8))
    refg = orc.solve_ptr(so3, ofbz, orc.f_trgloc(eta, om) if hasattr(orc, "f_trgloc") else (lambda x, v: np.trace(
        np.linalg.inv((om + 1j * eta) * np.eye(3) - v), axis1=1, axis2=2)), npt=8).u
    assert abs(sol(om) - refg) <= 1e-10 * abs(refg)
    # two DOS caches on the same series: re-initialising one must not free what the other holds
    h = abz.FourierSeries(c, period=1.0, first=first, ndim=3)
    ca = abz.dos.init(abz.DOSProblem(h, 0.1, fbz), abz.GGR(npt=10))
    cb = abz.dos.init(abz.DOSProblem(h, 0.1, fbz), abz.GGR(npt=10))
    ua = abz.dos.solve_(ca).u
    h.c *= 2
    cb.isfresh = True
    cb.domain = 0.2
    ub = abz.dos.solve_(cb).u
    ca.domain = 0.2
    assert abs(abz.dos.solve_(ca).u - ub) <= 1e-12 * abs(ub) and abs(ub - ua / 2) <= 1e-9 * abs(ua)
    # closed handles raise
    r = abz.DeviceRule(h.device(), 6, None, 1)
    r.close()
    with pytest.raises(abz.AbzError):
        r.reduce(abz._lib.F_DOS, [eta], [om])
    # C ABI: destroy in "finalizer order" (context first, then series, then rule); the rule still works in between
    import ctypes as C
    L = abz._lib
    lib = L.lib()
    ctx, ser, rule = C.c_void_p(), C.c_void_p(), C.c_void_p()
    L.check(lib.abz_ctx_create(0, C.byref(ctx)))
    coef = np.ascontiguousarray(abz.series.julia_coefficient_order(c, 3).view(np.float64))
    dims, frst, per = np.array([3, 3, 3], dtype=np.int32), np.array(first, dtype=np.int32), np.ones(3)
    L.check(lib.abz_series_create(ctx, coef.ctypes.data_as(L.c_f64p), 3, dims.ctypes.data_as(L.c_i32p),
                                  frst.ctypes.data_as(L.c_i32p), per.ctypes.data_as(L.c_f64p), 3, C.byref(ser)))
    L.check(lib.abz_ptr_rule_build(ser, 6, 0, None, None, 1, C.byref(rule)))
    out = np.zeros((1, 1, 2))
    sw = np.array([om])
    par = np.array([eta])
    L.check(lib.abz_rule_reduce(rule, L.F_DOS, par.ctypes.data_as(L.c_f64p), 1, sw.ctypes.data_as(L.c_f64p), 1, 1, out.ctypes.data_as(L.c_f64p)))
    v0 = out[0, 0, 0]
    assert lib.abz_ctx_destroy(ctx) == 0 and lib.abz_series_destroy(ser) == 0
    assert lib.abz_rule_reduce(rule, L.F_DOS, par.ctypes.data_as(L.c_f64p), 1, sw.ctypes.data_as(L.c_f64p), 1, 1,
                               out.ctypes.data_as(L.c_f64p)) == L.ERR_ARG  # closed owner: an error, not a fault
    assert lib.abz_rule_destroy(rule) == 0 and v0 > 0


def test_context_on_a_borrowed_stream_and_device_resident_sums(abz, svo):
    """abz_ctx_create_on_stream + abz_rule_reduce_device + abz_rule_values_ptr: the library enqueues on a stream the
    harness owns (a torch stream), the sweep's sums stay in HBM, and -- without any host synchronisation in between --
    equal abz_rule_reduce's host results bit for bit; the rule's value block is visible as a zero-copy device view."""
    import torch
    s, _ = svo
    L = abz._lib
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        ctx = abz.Context(0, stream=st.cuda_stream)
        dev = abz.DeviceSeries(s, ctx)
        rule = abz.DeviceRule(dev, 24, None, L.WANT_H | L.WANT_EIG)
        om = np.linspace(11.0, 14.0, 9)
        om_dev = torch.from_numpy(om).cuda()
        for fid, ncomp in ((L.F_DOS, 1), (L.F_DOS_EIG, 1), (L.F_TRGLOC, 1), (L.F_GLOC, 9)):
            out = torch.zeros(len(om), ncomp, 2, dtype=torch.float64, device="cuda")
            rule.rebuild()  # async, same stream
            rule.reduce_device(fid, [0.1], om_dev.data_ptr(), len(om), out.data_ptr())
            twice = out * 2  # a torch op on the same stream consumes the sums in order
            got = (twice / 2).cpu().numpy().view(np.complex128).reshape(len(om), ncomp)
            ref = rule.reduce(fid, [0.1], om)
            assert np.array_equal(got, ref)
        base, nbytes = rule.values_ptr()
        assert base != 0 and nbytes >= 24**3 * (18 + 3) * 8
        rule.close()
        dev.close()
        ctx.close()
    assert st.query() or True  # the borrowed stream is still usable / not destroyed by abz_ctx_destroy
    torch.zeros(4, device="cuda").sum().item()


def test_single_iai_solve_sharded_over_two_ranks():
    """one IAI solve on several GPUs -- the innermost integrals of every round dealt to the ranks, one
    all-gather per round (abz_iai_set_exchange / dist.iaishard).  Two rank processes share this box's GPU (gloo): both
    return the unsharded solve's value, error and numevals bit for bit (3 and 16 bands)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(root, "tests", "dist_iai_worker.py")],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert p.stdout.count("sharded IAI ok") == 2


@pytest.mark.parametrize("n", [1, 2, 3, 4, 6, 16])
def test_hermitian_compact_rule_layout_equals_full_layout(abz, n, monkeypatch):
    """ABZ_WANT_H_COMPACT (rules of a Hermitian series keep H(k) as its upper triangle; the host mirror asks for it by
    default): exported matrices, eigenvalues and every built-in rule sum are BIT-identical to the reference's full
    SMatrix layout (FourierPTR's vals, src/fourier.jl:127-174) on full grids and equal to rounding on symmetric rules,
    with and without the packed chain; the value block shrinks to n^2 + n planes; a series that stops being Hermitian drops such rules; a
    non-Hermitian series never gets the layout.  6 and 16 bands: the row kernel of kernels_generic.hip writes both layouts from
    the same upper triangle (through its LDS tile), every scan reads either.  ref: src/fourier.jl:127-174."""
    from autobzcore.jl_amd import _lib as L
    rng = np.random.default_rng(900 + n)
    c, first = rand_series(rng, (3, 5, 3), n, hermitian=True)
    c = c / max(1.0, n / 3)
    s, so = both(abz, c, first)
    dev = s.device()
    cubic = abz.load_bz(abz.CubicSymIBZ(), np.eye(3)).syms
    omegas = np.linspace(-2.5, 2.5, 19)
    fids = [(L.F_DOS, [0.3], omegas), (L.F_TRGLOC, [0.3], omegas), (L.F_GLOC, [0.3], omegas[:5]), (L.F_DOS_EIG, [0.3], omegas)]
    if n == 1:
        fids += [(L.F_LINEAR, [1.3, 1.0], None), (L.F_LINEAR_X, [1.3, 0.5], None)]
    for npt, syms in ((10, None), (150, None), (12, cubic)):
        if npt == 150 and n not in (1, 3):
            continue
        full = abz.DeviceRule(dev, npt, syms, L.WANT_H | L.WANT_EIG)
        comp = abz.DeviceRule(dev, npt, syms, L.WANT_H | L.WANT_EIG | L.WANT_H_COMPACT)
        assert comp.want & L.WANT_H_COMPACT and not full.want & L.WANT_H_COMPACT
        bf, bc = full.values_ptr()[1], comp.values_ptr()[1]
        assert bc * (2 * n * n + n) == bf * (n * n + n)
        ef = full.export(x=False, w=False, H=True, eig=True)
        ec = comp.export(x=False, w=False, H=True, eig=True)
        # full grids: the grid kernel mirrors the upper triangle, so both layouts hold the same bits.  Node lists: the
        # full layout keeps an independently rounded lower triangle (and a diagonal imaginary part of rounding size)
        same = (lambda a, b: np.array_equal(a, b)) if syms is None else \
            (lambda a, b: np.abs(a - b).max() <= 1e-14 * max(np.abs(a).max(), 1e-300))
        assert same(ef["H"], ec["H"]) and np.array_equal(ef["eig"], ec["eig"])
        assert np.array_equal(ec["H"], np.conj(np.swapaxes(ec["H"], -1, -2))) or n == 1
        for fid, params, sw in fids:
            a, b = full.reduce(fid, params, sw), comp.reduce(fid, params, sw)
            assert same(a, b), (n, npt, fid)
        if npt == 10:  # the unpacked chain gives the same planes
            monkeypatch.setenv("ABZ_EVAL_PACKED", "0")
            comp.rebuild()
            monkeypatch.delenv("ABZ_EVAL_PACKED")
            e2 = comp.export(x=False, w=False, H=True)
            assert np.abs(e2["H"] - ec["H"]).max() <= 1e-13 * np.abs(ec["H"]).max()
        full.close()
        comp.close()
    # the host mirror asks for the compact layout by itself, and drops such rules when the series stops being Hermitian
    r = dev.rule(8, None, want=L.WANT_H)
    assert bool(r.want & L.WANT_H_COMPACT) == (os.environ.get("ABZ_RULE_COMPACT", "1") != "0")
    ref = np.array([orc._ptr_rule_sum(so, 8, None, orc.f_dos(0.3, om))[0] for om in omegas[:3]])
    assert np.abs(r.reduce(L.F_DOS, [0.3], omegas[:3])[:, 0].real - ref).max() <= 1e-11 * np.abs(ref).max()
    extra, _ = rand_series(rng, (3, 5, 3), n, hermitian=False)
    with pytest.raises(ValueError):  # the library refuses to refill an upper-triangle rule from such a series
        keep = abz.DeviceRule(dev, 6, None, L.WANT_H | L.WANT_H_COMPACT)
        L.check(L.lib().abz_series_update(dev.h, np.ascontiguousarray(
            abz.series.julia_coefficient_order(c + 0.05 * extra, 3).view(np.float64)).ctypes.data_as(L.c_f64p)))
        keep.rebuild()
    dev.update(c + 0.05 * extra)
    assert not dev.hermitian() and (not dev.rules or os.environ.get("ABZ_RULE_COMPACT", "1") == "0")
    r2 = dev.rule(8, None, want=L.WANT_H)
    assert not r2.want & L.WANT_H_COMPACT
    so2 = orc.FourierSeries(c + 0.05 * extra, period=1.0, first=first, ndim=3)
    ref = np.array([orc._ptr_rule_sum(so2, 8, None, orc.f_dos(0.6, om))[0] for om in omegas[:3]])
    assert np.abs(r2.reduce(L.F_DOS, [0.6], omegas[:3])[:, 0].real - ref).max() <= 1e-10 * np.abs(ref).max()


# ------------------------------------------------------------------ errors
def test_error_behaviour(abz):
    s = abz.FourierSeries(np.zeros((3, 3)), first=-1, ndim=2)
    bz3 = abz.load_bz(abz.FBZ(), np.eye(3))
    with pytest.raises(ValueError):  # ref: src/fourier.jl:506
        abz.solve(abz.IntegralProblem(abz.FourierIntegrand(abz.UnitIntegrand(), s), bz3), abz.IAI())
    with pytest.raises(ValueError):  # ref: src/interfaces.jl:64-69
        abz.IntegralSolver(abz.FourierIntegrand(abz.UnitIntegrand(), s), bz3, abz.PTR(), tol=1)
    with pytest.raises(ValueError):
        s.device().rule(0, None)
