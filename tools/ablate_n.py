"""Fourier-eval kernel across band counts (random Hermitian series, M = 5 per dimension, 200^3 grid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
from autobzcore.jl_amd import _lib as L
rng = np.random.default_rng(0)
npt = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for n in (1, 2, 3, 4):
    dims = (5, 5, 5)
    c = rng.standard_normal(dims + (n, n)) + 1j * rng.standard_normal(dims + (n, n))
    flip = c[::-1, ::-1, ::-1]
    c = 0.5 * (c + np.conj(np.swapaxes(flip, -1, -2)))
    s = abz.FourierSeries(c, period=1.0, first=(-2, -2, -2), ndim=3)
    dev = s.device(); ctx = dev.ctx
    for want, name in ((1, "H"), (3, "H+EIG")):
        rule = abz.DeviceRule(dev, npt, None, want)
        for _ in range(3): rule.rebuild()
        ctx.sync(); ctx.prof_enable(True); ctx.prof_reset()
        for _ in range(10): rule.rebuild()
        ctx.sync(); ms, k = ctx.prof_read(L.K_EVAL); ctx.prof_enable(False)
        nk = npt**3; B = 16*n*n + (8*n if want & 2 else 0)
        print(f"n={n} npt={npt} {name:6s}: {ms/k:.4f} ms  {nk/(ms/k*1e-3)/1e9:6.1f} G k/s  {nk*B/(ms/k*1e-3)/1e12:5.2f} TB/s ({B} B/k)")
        rule.close()
    om = np.linspace(-1, 1, 8)
    t = dev.ptr_sum(npt, L.F_DOS, [0.2], om)
    import time; t0 = time.perf_counter(); dev.ptr_sum(npt, L.F_DOS, [0.2], om); dt = time.perf_counter() - t0
    print(f"      store-free DOS x8 omega: {1e3*dt:.3f} ms  {npt**3/dt/1e9:.1f} G k/s")
