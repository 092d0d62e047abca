"""Symmetric (irreducible-node) rule builds for 16 bands: cubic IBZ, H and H + eigenvalues."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
L = abz._lib
s = abz.synthetic_wannier()
dev = s.device()
bz = abz.load_bz(abz.CubicSymIBZ(), np.eye(3))
for npt in (48, 100):
    for want, name in ((L.WANT_H, "H"), (L.WANT_H | L.WANT_EIG, "H+EIG")):
        r = abz.DeviceRule(dev, npt, bz.syms, want)
        dev.ctx.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            r.rebuild()
        dev.ctx.sync()
        dt = (time.perf_counter() - t0) / 5
        print(f"16 bands, cubic IBZ npt={npt}: {r.nk} irreducible nodes, rebuild {name:6s} {1e3*dt:8.3f} ms = {r.nk/dt/1e6:7.2f} M nodes/s", flush=True)
        r.close()
