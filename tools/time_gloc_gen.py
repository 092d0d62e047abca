"""Matrix-valued G_loc scans of a cached 16-band rule: ms per call vs number of swept values."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
from autobzcore.jl_amd import _lib as L
for n, rmax in ((5, 3), (16, 6)):
    s = abz.synthetic_wannier(n=n, rmax=rmax)
    dev = s.device()
    r = abz.DeviceRule(dev, 48, None, L.WANT_H)
    for nw in (1, 16, 64):
        om = np.linspace(-1, 1, nw)
        r.reduce(L.F_GLOC, [0.05], om)
        t0 = time.perf_counter(); g = r.reduce(L.F_GLOC, [0.05], om); dt = time.perf_counter() - t0
        print(f"n={n:2d} npt=48 G_loc scan n_omega={nw:3d}: {1e3*dt:9.2f} ms  {48**3*nw/dt/1e6:9.1f} M (k,omega)/s  tr={np.trace(g[0].reshape(n, n)):.6f}", flush=True)
    r.close()
