import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import abz_oracle as orc
import autobzcore.jl_amd as abz
L = abz._lib
rng = np.random.default_rng(3)
# A: d=1 n=2 npt=400 dims=(7,) hermitian
for npt in (400, 384, 385, 448, 449, 512, 640):
    for n in (2, 3):
        dims = (7,)
        c = rng.standard_normal(dims + (n, n)) + 1j * rng.standard_normal(dims + (n, n))
        flip = c[::-1]
        c = 0.5 * (c + np.conj(np.swapaxes(flip, -1, -2)))
        s = abz.FourierSeries(c, period=1.0, first=(-3,), ndim=1)
        so = orc.FourierSeries(c, period=1.0, first=(-3,), ndim=1)
        rule = abz.DeviceRule(s.device(), npt, None, L.WANT_H)
        H = rule.export(x=False, w=False, H=True)["H"].reshape(-1, n, n)
        ref = orc.fourier_ptr(so, npt).reshape(-1, n, n)
        bad = np.where(np.abs(H - ref).max(axis=(1, 2)) > 1e-10)[0]
        print("A npt", npt, "n", n, "bad nodes:", len(bad), bad[:6], bad[-3:] if len(bad) else "")
# B: constant non-Hermitian series, reduce
for npt, n, dims, d in ((300, 2, (1,), 1), (200, 2, (1,), 1), (257, 2, (1,), 1), (1000, 3, (1,), 1), (300, 1, (6, 1), 2), (300, 2, (3,), 1)):
    c = rng.standard_normal(dims + (n, n)) + 1j * rng.standard_normal(dims + (n, n))
    first = tuple(0 for _ in dims)
    s = abz.FourierSeries(c, period=1.0, first=first, ndim=d)
    so = orc.FourierSeries(c, period=1.0, first=first, ndim=d)
    rule = abz.DeviceRule(s.device(), npt, None, L.WANT_H)
    ref = orc.fourier_ptr(so, npt)
    ref = np.transpose(ref, tuple(range(d - 1, -1, -1)) + (d, d + 1)).reshape(-1, n, n)
    om, eta = np.array([-0.3, 0.4]), 0.35
    got = rule.reduce(L.F_TRGLOC, [eta], om)[:, 0]
    z = (om + 1j * eta)[:, None, None, None] * np.eye(n) - ref[None]
    tr = np.trace(np.linalg.inv(z), axis1=-2, axis2=-1).mean(axis=1)
    one = rule.reduce(L.F_ONE)[0, 0]
    print("B npt", npt, "n", n, dims, "got", got, "ref", tr, "sum(1) =", one)
