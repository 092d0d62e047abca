"""Generic-n (16 bands) PTR path: rule build (H, H + eig) and fused scans, wave-per-node kernels."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
from autobzcore.jl_amd import _lib as L
s = abz.synthetic_wannier()
dev = s.device(); ctx = dev.ctx
for npt in (24, 48):
    nk = npt**3
    for want, name in ((L.WANT_H, "H"), (L.WANT_H | L.WANT_EIG, "H+EIG"), (L.WANT_H | L.WANT_H_COMPACT, "Hc"),
                       (L.WANT_H | L.WANT_EIG | L.WANT_H_COMPACT, "Hc+EIG"), (L.WANT_EIG, "EIG")):
        r = abz.DeviceRule(dev, npt, None, want); ctx.sync()
        for _ in range(3):
            r.rebuild()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(10):
            r.rebuild()
        ctx.sync(); dt = (time.perf_counter() - t0) / 10
        print(f"n=16 npt={npt} rebuild {name:6s}: {1e3*dt:9.2f} ms  {nk/dt/1e6:8.2f} M k/s")
        if want == (L.WANT_H | L.WANT_EIG):
            for fid, nm in ((L.F_DOS, "DOS (inverse)"), (L.F_DOS_EIG, "DOS (eig)")):
                for nw in (1, 16):
                    om = np.linspace(-1, 1, nw)
                    r.reduce(fid, [0.05], om)
                    t0 = time.perf_counter(); r.reduce(fid, [0.05], om); dt = time.perf_counter() - t0
                    print(f"      scan {nm:14s} n_omega={nw:3d}: {1e3*dt:9.2f} ms  {nk*nw/dt/1e6:9.1f} M (k,omega)/s")
        r.close()
for npt in (24, 48):
    for nw in (1, 4, 16):
        om = np.linspace(-1, 1, nw)
        dev.ptr_sum(npt, L.F_DOS, [0.05], om)
        t0 = time.perf_counter(); dev.ptr_sum(npt, L.F_DOS, [0.05], om); dt = time.perf_counter() - t0
        print(f"n=16 npt={npt} store-free DOS n_omega={nw:3d}: {1e3*dt:9.2f} ms  {npt**3*nw/dt/1e6:9.1f} M (k,omega)/s")
