"""Randomised parity sweeps of the HIP path against the numpy oracle, over sizes the unit tests do not enumerate
(fixed seeds in the suite; `python tools/fuzz_parity.py SEED` / `fuzz_parity2.py SEED` run the same cases with another seed).

Bars: floating point <= 1e-12 relative for series values and rule sums (eigenvalues of 5...32 bands: 1e-10, cyclic Jacobi /
bisection against LAPACK), IAI values 1e-9 of the oracle's; every integer (evaluation counts) equal.
"""
import numpy as np
import pytest

import abz_oracle as orc

pytestmark = pytest.mark.gpu

TOL = 1e-12


def _herm_series(rng, dims, n, scale=1.0):
    c = rng.standard_normal(dims + (n, n)) + 1j * rng.standard_normal(dims + (n, n))
    flip = c[tuple(slice(None, None, -1) for _ in dims)]
    c = 0.5 * (c + np.conj(np.swapaxes(flip, -1, -2))) * scale
    return c, tuple(-(m // 2) for m in dims)


def _grid_ref(so, npt, d, n):
    ref = orc.fourier_ptr(so, npt)  # [i1..id, n, n]
    return np.transpose(ref, tuple(range(d - 1, -1, -1)) + (d, d + 1)).reshape(-1, n, n)  # i1 fastest


def fuzz_small_band_rules(abz, seed, emit=None):
    """Full-grid rules of 1...4 bands (grid kernels incl. multi-pass lines, contraction levels, eigenvalues, fused scans,
    the Hermitian-compact layout and the 3-band DOS sweep kernel).  Returns (worst error, list of failing cases)."""
    L = abz._lib
    rng = np.random.default_rng(seed)
    cases = []
    for npt in (1, 2, 3, 63, 64, 65, 127, 128, 129, 150, 192, 193, 200, 257, 300, 400, 1000):
        cases.append((1, int(rng.integers(1, 5)), npt))
    for npt in (1, 2, 5, 63, 64, 65, 128, 129, 193, 200, 260, 300):
        cases.append((2, int(rng.integers(1, 5)), npt))
    for npt in (1, 2, 7, 33, 64, 65, 70):
        cases.append((3, int(rng.integers(1, 5)), npt))
    worst, bad = 0.0, []
    for d, n, npt in cases:
        for herm in (True, False):
            dims = (tuple(int(rng.choice([1, 3, 5, 7, 9, 11, 13])) for _ in range(d)) if herm
                    else tuple(int(rng.integers(1, 9)) for _ in range(d)))
            if herm:
                c, first = _herm_series(rng, dims, n)
            else:
                c = rng.standard_normal(dims + (n, n)) + 1j * rng.standard_normal(dims + (n, n))
                first = tuple(int(rng.integers(-4, 3)) for _ in dims)
            s = abz.FourierSeries(c, period=1.0, first=first, ndim=d)
            so = orc.FourierSeries(c, period=1.0, first=first, ndim=d)
            want = L.WANT_H | (L.WANT_EIG if herm else 0)
            rule = abz.DeviceRule(s.device(), npt, None, want)
            ex = rule.export(x=False, w=False, H=True, eig=herm)
            ref = _grid_ref(so, npt, d, n)
            scale = max(np.abs(ref).max(), 1e-300)
            errH = np.abs(ex["H"].reshape(-1, n, n) - ref).max() / scale
            errE = np.abs(ex["eig"] - np.linalg.eigvalsh(ref)).max() / scale if herm else 0.0
            om, eta = np.array([-0.3, 0.4]), 0.35
            got = rule.reduce(L.F_TRGLOC, [eta], om)[:, 0]
            zi = np.linalg.inv((om + 1j * eta)[:, None, None, None] * np.eye(n) - ref[None])
            tr = np.trace(zi, axis1=-2, axis2=-1).mean(axis=1)
            errR = np.abs(got - tr).max() / max(np.abs(tr).max(), 1e-300)
            # conditioning of the case itself: a non-Hermitian H(k) can make z - H(k) nearly singular at a node; the
            # rule value then amplifies the 1e-16 differences of H by |G|^2 |H| / |mean tr G|
            amp = (np.abs(zi).max() ** 2) * max(np.abs(ref).max(), 1.0) / max(np.abs(tr).max(), 1e-300)
            errR = errR / max(1.0, amp)
            if herm and npt <= 300:
                comp = abz.DeviceRule(s.device(), npt, None, want | L.WANT_H_COMPACT)
                exc = comp.export(x=False, w=False, H=True)
                errH = max(errH, np.abs(exc["H"].reshape(-1, n, n) - ref).max() / scale)
                om3 = np.linspace(-0.8, 0.9, 19)
                z3 = (om3 + 1j * eta)[:, None, None, None] * np.eye(n) - ref[None]
                dos = -np.trace(np.linalg.inv(z3), axis1=-2, axis2=-1).imag.mean(axis=1) / np.pi
                for fid in (L.F_DOS, L.F_DOS_EIG):
                    gd = comp.reduce(fid, [eta], om3)[:, 0].real
                    # the eigenvalue form divides by (omega - e)^2 + eta^2 with e good to 1e-12 |H|: one more digit of slack
                    errR = max(errR, np.abs(gd - dos).max() / max(np.abs(dos).max(), 1e-300) / (1.0 if fid == L.F_DOS else 10.0))
                comp.close()
            rule.close()
            e = max(errH, errE, errR)
            worst = max(worst, e)
            line = f"d={d} n={n} npt={npt:4d} dims={dims} herm={int(herm)}: H {errH:.1e} eig {errE:.1e} reduce {errR:.1e}"
            if not e <= TOL:
                bad.append(line)
            if emit:
                emit(line + ("" if e <= TOL else "   <-- CHECK"))
        s.device().drop_rules()
    return worst, bad


def fuzz_many_band_rules_iai_and_symmetric_rules(abz, seed, emit=None, quick=False):
    """5...32-band rules / scans / tridiagonal sweeps, IAI panel kernels with device-side inner loops, and symmetric
    (irreducible-node) rules, against the oracle.  Returns (worst error, list of failing cases)."""
    L = abz._lib
    rng = np.random.default_rng(seed)
    worst, bad = 0.0, []

    def note(line, ok, e):
        nonlocal worst
        worst = max(worst, e)
        if not ok:
            bad.append(line)
        if emit:
            emit(line + ("" if ok else "   <-- CHECK"))

    for n in (5, 7, 8, 9, 12, 16, 17, 24, 32):
        for d, npt in ((1, 37), (2, 11), (2, 33)):
            dims = tuple(int(rng.choice([1, 3, 5])) for _ in range(d))
            c, first = _herm_series(rng, dims, n, 1.0 / np.sqrt(n))
            s = abz.FourierSeries(c, period=1.0, first=first, ndim=d)
            so = orc.FourierSeries(c, period=1.0, first=first, ndim=d)
            rule = abz.DeviceRule(s.device(), npt, None, L.WANT_H | L.WANT_EIG)
            ex = rule.export(x=False, w=False, H=True, eig=True)
            ref = _grid_ref(so, npt, d, n)
            scale = np.abs(ref).max()
            eH = np.abs(ex["H"].reshape(-1, n, n) - ref).max() / scale
            eE = np.abs(ex["eig"] - np.linalg.eigvalsh(ref)).max() / scale
            om, eta = np.array([-0.3, 0.4]), 0.35
            got = rule.reduce(L.F_TRGLOC, [eta], om)[:, 0]
            z = (om + 1j * eta)[:, None, None, None] * np.eye(n) - ref[None]
            tr = np.trace(np.linalg.inv(z), axis1=-2, axis2=-1).mean(axis=1)
            eR = np.abs(got - tr).max() / np.abs(tr).max()
            gote = rule.reduce(L.F_DOS_EIG, [eta], om)[:, 0].real
            eD = np.abs(gote + tr.imag / np.pi).max() / np.abs(tr).max()
            # sweeps of >= 3 values take the tridiagonal routes for n <= 16 (scan of the cached rule and store-free sum)
            nsw = int(rng.integers(3, 41))
            om5, eta5 = np.sort(rng.uniform(-1.5, 1.5, nsw)), float(rng.choice([0.02, 0.1, 0.4]))
            z5 = (om5 + 1j * eta5)[:, None, None, None] * np.eye(n) - ref[None]
            tr5 = np.trace(np.linalg.inv(z5), axis1=-2, axis2=-1).mean(axis=1)
            sc = rule.reduce(L.F_TRGLOC, [eta5], om5)[:, 0]
            sf = s.device().ptr_sum(npt, L.F_DOS, [eta5], om5)[:, 0].real
            eS = max(np.abs(sc - tr5).max(), np.abs(sf + tr5.imag / np.pi).max()) / np.abs(tr5).max()
            rule.close()
            s.device().drop_rules()
            # eta = 0.02 puts 1/eta^2 of amplification on the 1e-16 |H| of the series values
            ok = max(eH, eR, eD) <= TOL and eS <= 1e-11 and eE <= 1e-10
            note(f"gen n={n:2d} d={d} npt={npt}: H {eH:.1e} eig {eE:.1e} trgloc {eR:.1e} dos_eig {eD:.1e} "
                 f"sweep[{nsw}, eta {eta5}] {eS:.1e}", ok, max(eH, eR, eD, eS))

    # IAI (panel kernels, device-side inner loops), 2-D so that the Python oracle stays fast.  (20, (11, 3)): the
    # zero-padded coefficient set of the inner variable (11 x 32 x 32 complex) does not fit the LDS, so the kernels run
    # on the unpadded layout with register-only identity padding
    for n, dims in ((5, (3, 3)), (8, (3, 3)), (11, (3, 3)), (16, (3, 3)), (19, (3, 3)), (32, (3, 3)), (20, (11, 3))):
        c, first = _herm_series(rng, dims, n, 1.0 / np.sqrt(n))
        s = abz.FourierSeries(c, period=1.0, first=first, ndim=2)
        so = orc.FourierSeries(c, period=1.0, first=first, ndim=2)
        bz = abz.load_bz(abz.FBZ(), np.eye(2))
        slow = quick and n >= 20
        atol = 1e-1 if slow else 1e-2
        for integ, f in ((abz.DOSIntegrand(), orc.f_dos(0.3, 0.1)),
                         (abz.TrGlocIntegrand(), lambda x, h: np.trace(orc.f_gloc(0.3, 0.1)(x, h), axis1=-2, axis2=-1)))[:1 if slow else 2]:
            sol = abz.do_solve(abz.FourierIntegrand(integ, s, 0.3), bz, abz.MixedParameters(0.1),
                               abz.EvalCounter(abz.IAI()), abstol=atol)
            ref = orc.solve_iai(so, orc.load_bz("FBZ", np.eye(2)), f, abstol=atol)
            e = abs(sol.u - ref.u) / abs(ref.u)
            note(f"iai n={n:2d} {type(integ).__name__:16s}: rel {e:.1e} numevals {sol.numevals} vs {ref.numevals}",
                 e <= 1e-9 and sol.numevals == ref.numevals, e)

    # symmetric rules (irreducible nodes + integer weights): n <= 4 and, on the row kernels, 5...16 bands
    for kind, bzk in (("InversionSymIBZ", abz.InversionSymIBZ()), ("CubicSymIBZ", abz.CubicSymIBZ())):
        for d, npt in ((1, 17), (2, 9), (2, 30), (3, 8), (3, 21), (2, 12), (3, 9)):
            n = int(rng.integers(1, 5)) if npt not in (12, 9) or d == 2 and npt == 9 else int(rng.integers(5, 17))
            # a series with the symmetry of the lattice: s(k) = sum_i cos(2 pi k_i) * A  (A Hermitian)
            A = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
            A = 0.5 * (A + A.conj().T)
            c = np.zeros((3,) * d + (n, n), dtype=np.complex128)
            for i in range(d):
                for e_ in (0, 2):
                    idx = [1] * d
                    idx[i] = e_
                    c[tuple(idx)] += 0.5 * A
            first = (-1,) * d
            s = abz.FourierSeries(c, period=1.0, first=first, ndim=d)
            so = orc.FourierSeries(c, period=1.0, first=first, ndim=d)
            bz = abz.load_bz(bzk, np.eye(d))
            bzo = orc.load_bz(kind, np.eye(d))
            sol = abz.do_solve(abz.FourierIntegrand(abz.DOSIntegrand(), s, 0.3), bz, abz.MixedParameters(0.2),
                               abz.EvalCounter(abz.PTR(npt=npt)))
            ref = orc.solve_ptr(so, bzo, orc.f_dos(0.3, 0.2), npt=npt)
            full = orc.solve_ptr(so, orc.load_bz("FBZ", np.eye(d)), orc.f_dos(0.3, 0.2), npt=npt)
            e = abs(sol.u - ref.u) / abs(ref.u)
            ef = abs(sol.u - full.u) / abs(full.u)
            note(f"sym {kind:16s} d={d} n={n} npt={npt}: rel {e:.1e} (vs FBZ {ef:.1e}) numevals {sol.numevals} vs {ref.numevals}",
                 e <= TOL and ef <= 1e-11 and sol.numevals == ref.numevals, e)
    return worst, bad


def fuzz_round5_kernels(abz, seed, emit=None):
    """The kernels of round 5 over random shapes: GGR builds of 5...32 bands (kernels_ggr_rows.hip: eigenvalues, band velocities on
    separated bands, their sums per node everywhere) in 1...3 dimensions on full grids and inversion-symmetric node lists;
    5...8 bands one node per lane (kernels_lane.hip: values in either layout, eigenvalues, store-free sums, scans); 33...64 bands
    (kernels_big.hip: values, eigenvalues, tr G sums, the matrix-valued G; GGR builds there through kernels_big_vec.hip).  Returns (worst error, failing cases)."""
    L = abz._lib
    rng = np.random.default_rng(seed)
    worst, bad = 0.0, []

    def note(tag, err, bar):
        nonlocal worst
        worst = max(worst, err / bar * TOL)
        if not (err <= bar):
            bad.append((tag, err))
        if emit:
            emit(tag, err)

    # --- GGR builds
    for _ in range(14):
        d = int(rng.integers(1, 4))
        n = int(rng.choice([5, 6, 7, 8, 9, 11, 13, 16, 17, 19, 23, 28, 32, 33, 41, 47, 56, 63]))  # (> 32: kernels_big_vec.hip)
        dims = tuple(int(rng.choice([1, 3, 5, 7] if n <= 16 else [1, 3, 5])) for _ in range(d))
        npt = int(rng.integers(1, (12 if n <= 32 else 7) if d == 3 else (20 if d == 2 else 70)))
        c, first = _herm_series(rng, dims, n, scale=1.0 / np.sqrt(n))
        per = tuple(float(rng.choice([1.0, 2.0, 0.5])) for _ in range(d))
        s = abz.FourierSeries(c, period=per, first=first, ndim=d)
        so = orc.FourierSeries(c, period=per, first=first, ndim=d)
        syms = orc.load_bz("InversionSymIBZ", np.eye(d)).syms if (d >= 2 and rng.integers(0, 2)) else None
        w, e, v = orc.get_ggr_data(so, npt, syms)
        rule = abz.DeviceRule(s.device(), npt, syms, L.WANT_EIG | L.WANT_VEL)
        out = rule.export(x=False, w=False, eig=True, vel=True)
        rule.close()
        scale, vscale = max(np.abs(e).max(), 1e-300), max(np.abs(v).max(), 1e-300)
        tag = ("ggr", d, n, dims, npt, syms is not None)
        note(tag + ("eig",), np.abs(out["eig"] - e).max() / scale, 1e-11)
        note(tag + ("velsum",), np.abs(out["vel"].sum(axis=2) - v.sum(axis=2)).max() / (vscale * n), 1e-9)
        ok = np.min(np.diff(e, axis=1), axis=1) > 1e-5 * scale
        if ok.any():
            note(tag + ("vel",), np.abs(out["vel"][ok] - v[ok]).max() / vscale, 1e-8)
    # --- 5...8 bands, one node per lane
    for _ in range(10):
        d = int(rng.integers(1, 4))
        n = int(rng.integers(5, 9))
        dims = tuple(int(rng.choice([1, 3, 5, 7, 9, 11, 13])) for _ in range(d))
        npt = int(rng.choice([1, 2, 3, 31, 63, 64, 65, 100, 129] if d == 1 else ([1, 2, 9, 33, 64, 70] if d == 2 else [1, 2, 5, 9, 17])))
        c, first = _herm_series(rng, dims, n, scale=1.0 / np.sqrt(n))
        s = abz.FourierSeries(c, period=1.0, first=first, ndim=d)
        so = orc.FourierSeries(c, period=1.0, first=first, ndim=d)
        ref = _grid_ref(so, npt, d, n)
        scale = max(np.abs(ref).max(), 1e-300)
        compact = bool(rng.integers(0, 2))
        rule = abz.DeviceRule(s.device(), npt, None, L.WANT_H | L.WANT_EIG | (L.WANT_H_COMPACT if compact else 0))
        ex = rule.export(x=False, w=False, H=True, eig=True)
        tag = ("lane", d, n, dims, npt, compact)
        note(tag + ("H",), np.abs(ex["H"].reshape(-1, n, n) - ref).max() / scale, TOL)
        note(tag + ("eig",), np.abs(ex["eig"] - np.linalg.eigvalsh(ref)).max() / scale, 1e-11)
        om = rng.uniform(-1.5, 1.5, size=int(rng.integers(1, 40)))
        eta = float(rng.uniform(0.05, 0.5))
        z = om[None, :] + 1j * eta
        tref = (1.0 / (z[:, :, None] - np.linalg.eigvalsh(ref)[:, None, :])).sum(axis=2).mean(axis=0)
        sc = rule.reduce(L.F_TRGLOC, [eta], om)[:, 0]
        sf = s.device().ptr_sum(npt, L.F_TRGLOC, [eta], om)[:, 0]
        rule.close()
        note(tag + ("scan",), np.abs(sc - tref).max() / np.abs(tref).max(), 1e-10)
        note(tag + ("sum",), np.abs(sf - tref).max() / np.abs(tref).max(), 1e-10)
    # --- 33...64 bands
    for _ in range(5):
        d = int(rng.integers(1, 3))
        n = int(rng.integers(33, 65))
        dims = tuple(int(rng.choice([1, 3, 5])) for _ in range(d))
        npt = int(rng.integers(1, 9 if d == 2 else 40))
        c, first = _herm_series(rng, dims, n, scale=1.0 / np.sqrt(n))
        s = abz.FourierSeries(c, period=1.0, first=first, ndim=d)
        so = orc.FourierSeries(c, period=1.0, first=first, ndim=d)
        ref = _grid_ref(so, npt, d, n)
        scale = max(np.abs(ref).max(), 1e-300)
        rule = abz.DeviceRule(s.device(), npt, None, L.WANT_H | L.WANT_EIG)
        ex = rule.export(x=False, w=False, H=True, eig=True)
        tag = ("big", d, n, dims, npt)
        note(tag + ("H",), np.abs(ex["H"].reshape(-1, n, n) - ref).max() / scale, TOL)
        note(tag + ("eig",), np.abs(ex["eig"] - np.linalg.eigvalsh(ref)).max() / scale, 1e-10)
        om = rng.uniform(-1.5, 1.5, size=3)
        z = om[None, :] + 0.3j
        tref = (1.0 / (z[:, :, None] - np.linalg.eigvalsh(ref)[:, None, :])).sum(axis=2).mean(axis=0)
        sc = rule.reduce(L.F_TRGLOC, [0.3], om)[:, 0]
        sf = s.device().ptr_sum(npt, L.F_TRGLOC, [0.3], om)[:, 0]
        g = rule.reduce(L.F_GLOC, [0.3], om[:1])[0].reshape(n, n).T  # the full inverse (big_inverse_kernel)
        gref = np.linalg.inv((om[0] + 0.3j) * np.eye(n)[None] - ref).mean(axis=0)
        rule.close()
        note(tag + ("gloc",), np.abs(g - gref).max() / np.abs(gref).max(), 1e-10)
        note(tag + ("scan",), np.abs(sc - tref).max() / np.abs(tref).max(), 1e-10)
        note(tag + ("sum",), np.abs(sf - tref).max() / np.abs(tref).max(), 1e-10)
    # --- series that are NOT Hermitian, 2...64 bands: G, tr G and DOS of cached rules (full grids and inversion-symmetric lists)
    # against plain inverses, IAI in one and two dimensions against the oracle's adaptive loop (equal numevals)
    for it in range(10):
        d = int(rng.integers(1, 3))
        n = int(rng.choice([2, 3, 4, 5, 7, 8, 9, 13, 16, 17, 21, 29, 32, 33, 45, 64]))
        dims = tuple(int(rng.choice([1, 3, 5])) for _ in range(d))
        npt = int(rng.integers(1, 9 if d == 2 else 30))
        c, first = _herm_series(rng, dims, n, scale=1.0 / np.sqrt(n))
        c = c + 0.05 / np.sqrt(n) * (rng.standard_normal(c.shape) + 1j * rng.standard_normal(c.shape))
        s = abz.FourierSeries(c, period=1.0, first=first, ndim=d)
        so = orc.FourierSeries(c, period=1.0, first=first, ndim=d)
        om = rng.uniform(-1.0, 1.0, size=2)
        eta = float(rng.uniform(0.4, 0.8))
        syms = orc.load_bz("InversionSymIBZ", np.eye(d)).syms if (d == 2 and rng.integers(0, 2)) else None
        rule = s.device().rule(npt, syms, want=1)
        g = rule.reduce(L.F_GLOC, [eta], om)
        tr = rule.reduce(L.F_TRGLOC, [eta], om)[:, 0]
        dos = rule.reduce(L.F_DOS, [eta], om)[:, 0].real
        rule.close()
        tag = ("nonherm", d, n, dims, npt, syms is not None)
        for i in range(2):
            gref, _ = orc._ptr_rule_sum(so, npt, syms, orc.f_gloc(eta, om[i]))
            t = np.trace(gref)
            note(tag + ("gloc", i), np.abs(g[i].reshape(n, n).T - gref).max() / np.abs(gref).max(), 1e-10)
            note(tag + ("trg", i), abs(tr[i] - t) / abs(t), 1e-10)
            note(tag + ("dos", i), abs(dos[i] + t.imag / np.pi) / abs(t), 1e-10)
        if it < 5:
            bz = abz.load_bz(abz.FBZ(), np.eye(d))
            sol = abz.do_solve(abz.FourierIntegrand(abz.TrGlocIntegrand(), s, eta), bz, abz.MixedParameters(float(om[0])),
                               abz.EvalCounter(abz.IAI()), abstol=1e-2)
            f_tr = lambda x, h: np.trace(orc.f_gloc(eta, float(om[0]))(x, h), axis1=-2, axis2=-1)
            ref = orc.solve_iai(so, orc.load_bz("FBZ", np.eye(d)), f_tr, abstol=1e-2)
            note(tag + ("iai numevals",), float(abs(sol.numevals - ref.numevals)), 0.5)
            note(tag + ("iai",), abs(sol.u - ref.u) / abs(ref.u), 1e-9)
    return worst, bad


@pytest.fixture(scope="module")
def abz():
    import autobzcore.jl_amd as m
    return m


def test_fuzz_small_band_rules(abz):
    worst, bad = fuzz_small_band_rules(abz, seed=1)
    assert not bad, "\n".join(bad)
    assert worst <= TOL


def test_fuzz_many_band_rules_iai_and_symmetric_rules(abz):
    worst, bad = fuzz_many_band_rules_iai_and_symmetric_rules(abz, seed=1, quick=True)
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("seed", [501, 502])
def test_fuzz_round5_kernels(abz, seed):
    worst, bad = fuzz_round5_kernels(abz, seed)
    assert not bad, bad
