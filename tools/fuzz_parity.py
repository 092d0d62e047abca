"""Randomised parity sweep of the rule path (grid kernels incl. multi-pass lines, contraction levels,
eigenvalues, fused reduce) against the numpy oracle, over sizes the unit tests do not enumerate."""
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
cases = []
for npt in (1, 2, 3, 63, 64, 65, 127, 128, 129, 150, 192, 193, 200, 257, 300, 400, 1000):
    cases.append((1, int(rng.integers(1, 5)), npt))
for npt in (1, 2, 5, 63, 64, 65, 128, 129, 193, 200, 260, 300):
    cases.append((2, int(rng.integers(1, 5)), npt))
for npt in (1, 2, 7, 33, 64, 65, 70):
    cases.append((3, int(rng.integers(1, 5)), npt))
worst = 0.0
t00 = time.time()
for d, n, npt in cases:
    for herm in (True, False):
        dims = tuple(int(rng.choice([1, 3, 5, 7, 9, 11, 13])) for _ in range(d)) if herm else tuple(int(rng.integers(1, 9)) for _ in range(d))
        c = rng.standard_normal(dims + (n, n)) + 1j * rng.standard_normal(dims + (n, n))
        first = tuple(-(m // 2) for m in dims) if herm else tuple(int(rng.integers(-4, 3)) for _ in dims)
        if herm:
            flip = c[tuple(slice(None, None, -1) for _ in dims)]
            c = 0.5 * (c + np.conj(np.swapaxes(flip, -1, -2)))
        s = abz.FourierSeries(c, period=1.0, first=first, ndim=d)
        so = orc.FourierSeries(c, period=1.0, first=first, ndim=d)
        want = L.WANT_H | (L.WANT_EIG if herm else 0)
        rule = abz.DeviceRule(s.device(), npt, None, want)
        ex = rule.export(x=False, w=False, H=True, eig=herm)
        ref = orc.fourier_ptr(so, npt)  # [i1..id, n, n]
        ref = np.transpose(ref, tuple(range(d - 1, -1, -1)) + (d, d + 1)).reshape(-1, n, n)  # i1 fastest
        scale = max(np.abs(ref).max(), 1e-300)
        errH = np.abs(ex["H"].reshape(-1, n, n) - ref).max() / scale
        errE = 0.0
        if herm:
            errE = np.abs(ex["eig"] - np.linalg.eigvalsh(ref)).max() / scale
        # fused scan vs numpy on the exported values
        om, eta = np.array([-0.3, 0.4]), 0.35
        got = rule.reduce(L.F_TRGLOC, [eta], om)[:, 0]
        z = (om + 1j * eta)[:, None, None, None] * np.eye(n) - ref[None]
        zi = np.linalg.inv(z)
        tr = np.trace(zi, axis1=-2, axis2=-1).mean(axis=1)
        errR = np.abs(got - tr).max() / max(np.abs(tr).max(), 1e-300)
        # conditioning of the test itself: a non-Hermitian H(k) can make z - H(k) nearly singular at a node;
        # the rule value then amplifies the 1e-16 differences of H by |G|^2 |H| / |mean tr G|
        amp = (np.abs(zi).max() ** 2) * max(np.abs(ref).max(), 1.0) / max(np.abs(tr).max(), 1e-300)
        errR = errR / max(1.0, amp)
        if herm and npt <= 300:
            # the Hermitian-compact layout (upper-triangle planes) and the DOS scans on it (n = 3: the sweep kernel)
            comp = abz.DeviceRule(s.device(), npt, None, want | L.WANT_H_COMPACT)
            exc = comp.export(x=False, w=False, H=True)
            errH = max(errH, np.abs(exc["H"].reshape(-1, n, n) - ref).max() / scale)
            om3 = np.linspace(-0.8, 0.9, 19)
            z3 = (om3 + 1j * eta)[:, None, None, None] * np.eye(n) - ref[None]
            dos = -np.trace(np.linalg.inv(z3), axis1=-2, axis2=-1).imag.mean(axis=1) / np.pi
            for fid in (L.F_DOS, L.F_DOS_EIG):
                gd = comp.reduce(fid, [eta], om3)[:, 0].real
                errR = max(errR, np.abs(gd - dos).max() / max(np.abs(dos).max(), 1e-300) / (1.0 if fid == L.F_DOS else 10.0))
            comp.close()
        rule.close()
        worst = max(worst, errH, errE, errR)
        flag = "" if max(errH, errR) < 1e-11 and errE < 1e-11 else "   <-- CHECK"
        print(f"d={d} n={n} npt={npt:4d} dims={dims} herm={int(herm)}: H {errH:.1e} eig {errE:.1e} reduce {errR:.1e}{flag}", flush=True)
    s.device().drop_rules()
print(f"worst relative error {worst:.2e} over {2 * len(cases)} cases in {time.time() - t00:.1f} s")
