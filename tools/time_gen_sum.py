"""Generic-n (16 bands) store-free PTR sums: ms per call vs number of sweep values."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
from autobzcore.jl_amd import _lib as L
s = abz.synthetic_wannier()
dev = s.device()
for npt in (48, 96):
    for nw in (1, 4, 16):
        om = np.linspace(-1, 1, nw)
        dev.ptr_sum(npt, L.F_DOS, [0.05], om)
        t0 = time.perf_counter(); v = dev.ptr_sum(npt, L.F_DOS, [0.05], om); dt = time.perf_counter() - t0
        print(f"n=16 npt={npt} store-free DOS n_omega={nw:3d}: {1e3*dt:9.2f} ms  {npt**3*nw/dt/1e6:9.1f} M (k,omega)/s  v0={np.ravel(v)[0]:.6f}")
