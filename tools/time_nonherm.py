"""Series that are not Hermitian, 6...24 bands: scans of a cached rule and a store-free sum (big_inverse_kernel)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import autobzcore.jl_amd as abz
from autobzcore.jl_amd import _lib as L
rng = np.random.default_rng(1)
for n in (6, 8, 12, 16, 24):
    s = abz.synthetic_wannier(n=n, rmax=2, seed=7)
    c = s.c + 0.02 * (rng.standard_normal(s.c.shape) + 1j * rng.standard_normal(s.c.shape))
    s2 = abz.FourierSeries(c, period=1.0, first=s.first, ndim=3)
    dev = s2.device()
    r = dev.rule(24, None, want=1); dev.ctx.sync()
    om = np.linspace(-1, 1, 8)
    out = []
    for fid in (L.F_TRGLOC, L.F_GLOC):
        r.reduce(fid, [0.05], om)
        t0 = time.perf_counter(); r.reduce(fid, [0.05], om); out.append(1e3 * (time.perf_counter() - t0))
    dev.ptr_sum(24, L.F_TRGLOC, [0.05], om)
    t0 = time.perf_counter(); dev.ptr_sum(24, L.F_TRGLOC, [0.05], om); out.append(1e3 * (time.perf_counter() - t0))
    print(f"n={n:2d} not Hermitian, 24^3, 8 omega: scan tr G {out[0]:7.3f} ms, scan G {out[1]:7.3f} ms, store-free tr G {out[2]:7.3f} ms", flush=True)
    r.close()
