import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
from autobzcore.jl_amd import _lib as L
s = abz.load_w90_series(os.path.join(ROOT, "tests", "golden", "svo_hr.dat.gz"))
dev = s.device(); ctx = dev.ctx
Es = np.linspace(10, 15, 256)
for npt in (50, 100, 150):
    for rep in range(2):
        t0 = time.perf_counter(); r = abz.DeviceRule(dev, npt, None, L.WANT_EIG | L.WANT_VEL); ctx.sync(); t1 = time.perf_counter()
        g = r.ggr(Es); t2 = time.perf_counter()
        if rep: print(f"GGR npt={npt}: build (eig + velocities) {1e3*(t1-t0):8.2f} ms   scan 256 energies {1e3*(t2-t1):8.2f} ms   dos[128]={g[128]:.6f}")
        r.close()
