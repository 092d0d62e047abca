"""The matrix-valued Green's function of a cached 33...64-band rule (big_inverse_kernel): a few scans for rocprofv3."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
from autobzcore.jl_amd import _lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
npt = int(sys.argv[2]) if len(sys.argv) > 2 else 24
nw = int(sys.argv[3]) if len(sys.argv) > 3 else 4
s = abz.synthetic_wannier(n=n, rmax=2, seed=7)
dev = s.device()
r = abz.DeviceRule(dev, npt, None, L.WANT_H); dev.ctx.sync()
om = np.linspace(-1, 1, nw)
for _ in range(3):
    t0 = time.perf_counter(); r.reduce(L.F_GLOC, [0.05], om); dt = time.perf_counter() - t0
print(f"n={n} {npt}^3 G scan of {nw} omega: {1e3*dt:.3f} ms")
