"""Second parity sweep: generic n (5..32 bands) rules / reductions / IAI panel kernels, and symmetric
(irreducible-node) rules, against the numpy oracle."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np

import abz_oracle as orc
import autobzcore.jl_amd as abz

L = abz._lib
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
worst = 0.0
t00 = time.time()


def herm_series(dims, n, scale=1.0):
    c = rng.standard_normal(dims + (n, n)) + 1j * rng.standard_normal(dims + (n, n))
    flip = c[tuple(slice(None, None, -1) for _ in dims)]
    c = 0.5 * (c + np.conj(np.swapaxes(flip, -1, -2))) * scale
    first = tuple(-(m // 2) for m in dims)
    return c, first


# ---- generic n: rules, eigenvalues, fused scan
for n in (5, 7, 8, 9, 12, 16, 17, 24, 32):
    for d, npt in ((1, 37), (2, 11), (2, 33)):
        dims = tuple(int(rng.choice([1, 3, 5])) for _ in range(d))
        c, first = herm_series(dims, n, 1.0 / np.sqrt(n))
        s = abz.FourierSeries(c, period=1.0, first=first, ndim=d)
        so = orc.FourierSeries(c, period=1.0, first=first, ndim=d)
        rule = abz.DeviceRule(s.device(), npt, None, L.WANT_H | L.WANT_EIG)
        ex = rule.export(x=False, w=False, H=True, eig=True)
        ref = orc.fourier_ptr(so, npt)
        ref = np.transpose(ref, tuple(range(d - 1, -1, -1)) + (d, d + 1)).reshape(-1, n, n)
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
        worst = max(worst, eH, eE, eR, eD, eS)
        flag = "" if max(eH, eR, eD, eS) < 1e-11 and eE < 1e-10 else "   <-- CHECK"
        print(f"gen n={n:2d} d={d} npt={npt}: H {eH:.1e} eig {eE:.1e} trgloc {eR:.1e} dos_eig {eD:.1e} sweep[{nsw}, eta {eta5}] {eS:.1e}{flag}", flush=True)

# ---- generic n: IAI (panel kernels, device-side inner loops), 2-D so that the Python oracle stays fast
# (20, (11, 3)): the zero-padded coefficient set of the inner variable (11 x 32 x 32 complex) does not fit the LDS,
# so the kernels run on the unpadded layout with register-only identity padding
for n, dims in ((5, (3, 3)), (8, (3, 3)), (11, (3, 3)), (16, (3, 3)), (19, (3, 3)), (32, (3, 3)), (20, (11, 3))):
    c, first = herm_series(dims, n, 1.0 / np.sqrt(n))
    s = abz.FourierSeries(c, period=1.0, first=first, ndim=2)
    so = orc.FourierSeries(c, period=1.0, first=first, ndim=2)
    bz = abz.load_bz(abz.FBZ(), np.eye(2))
    for integ, f in ((abz.DOSIntegrand(), orc.f_dos(0.3, 0.1)),
                     (abz.TrGlocIntegrand(), lambda x, h: np.trace(orc.f_gloc(0.3, 0.1)(x, h), axis1=-2, axis2=-1))):
        sol = abz.do_solve(abz.FourierIntegrand(integ, s, 0.3), bz, abz.MixedParameters(0.1), abz.EvalCounter(abz.IAI()),
                           abstol=1e-2)
        ref = orc.solve_iai(so, orc.load_bz("FBZ", np.eye(2)), f, abstol=1e-2)
        e = abs(sol.u - ref.u) / abs(ref.u)
        worst = max(worst, e)
        flag = "" if e < 1e-9 and sol.numevals == ref.numevals else "   <-- CHECK"
        print(f"iai n={n:2d} {type(integ).__name__:16s}: rel {e:.1e} numevals {sol.numevals} vs {ref.numevals}{flag}", flush=True)

# ---- symmetric rules (irreducible nodes + integer weights): n <= 4 and, on the row kernels, 5..16 bands
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
        worst = max(worst, e)
        flag = "" if e < 1e-11 and ef < 1e-11 and sol.numevals == ref.numevals else "   <-- CHECK"
        print(f"sym {kind:16s} d={d} n={n} npt={npt}: rel {e:.1e} (vs FBZ {ef:.1e}) numevals {sol.numevals} vs {ref.numevals}{flag}", flush=True)
print(f"worst relative error {worst:.2e} in {time.time() - t00:.1f} s")
