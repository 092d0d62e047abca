"""H(k) + eigenvalue rule builds for 1..4 bands on a grid beyond the Infinity Cache: GB/s of H + eig written."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
L = abz._lib
rng = np.random.default_rng(0)
for n, npt in ((4, 160), (3, 160), (2, 208)):
    c = rng.standard_normal((11, 11, 11, n, n)) + 1j * rng.standard_normal((11, 11, 11, n, n))
    c = 0.5 * (c + np.conj(np.swapaxes(c[::-1, ::-1, ::-1], -1, -2)))
    s = abz.FourierSeries(c, period=1.0, first=(-5, -5, -5), ndim=3)
    dev = s.device()
    for want, name in ((L.WANT_H, "H only"), (L.WANT_H | L.WANT_EIG, "H + eig")):
        rule = abz.DeviceRule(dev, npt, None, want)
        for _ in range(5):
            rule.rebuild()
        dev.ctx.sync()
        dev.ctx.prof_enable(True, kernels=[L.K_EVAL]); dev.ctx.prof_reset()
        for _ in range(30):
            rule.rebuild()
        dev.ctx.sync()
        ms, cnt = dev.ctx.prof_read(L.K_EVAL); dev.ctx.prof_enable(False)
        nk = npt**3
        by = nk * (16 * n * n + (8 * n if want & L.WANT_EIG else 0))
        print(f"n = {n}, {npt}^3, {name}: eval kernel {ms/cnt:.4f} ms = {by/(ms/cnt)*1e-6:.0f} GB/s of algorithmic bytes ({by/1e6:.0f} MB)", flush=True)
        rule.close()
