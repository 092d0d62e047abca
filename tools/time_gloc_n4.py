"""4-band matrix-valued G_loc scans of a cached rule (adjugate form for Hermitian rules)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
from autobzcore.jl_amd import _lib as L
rng = np.random.default_rng(3)
n, M = (int(sys.argv[1]) if len(sys.argv) > 1 else 4), 7
c = rng.standard_normal((M, M, M, n, n)) + 1j * rng.standard_normal((M, M, M, n, n))
c = 0.5 * (c + np.conj(np.swapaxes(c[::-1, ::-1, ::-1], -1, -2))) * 0.1
s = abz.FourierSeries(c, period=1.0, first=(-3, -3, -3), ndim=3)
dev = s.device()
r = abz.DeviceRule(dev, 100, None, L.WANT_H)
for nw in (1, 32, 256):
    om = np.linspace(-1, 1, nw)
    r.reduce(L.F_GLOC, [0.05], om)
    t0 = time.perf_counter(); g = r.reduce(L.F_GLOC, [0.05], om); dt = time.perf_counter() - t0
    print(f"n={n} npt=100 G_loc scan n_omega={nw:3d}: {1e3*dt:9.3f} ms  {100**3*nw/dt/1e9:8.2f} G (k,omega)/s  tr={np.trace(g[0].reshape(n, n)):.6f}", flush=True)
