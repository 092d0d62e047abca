"""Workload for the rocprofv3 passes of the GGR kernels: `reps` rebuilds of the SVO eigenvalue + velocity rule at npt^3
(abz_rule_rebuild, WANT_EIG | WANT_VEL) and `scans` 256-energy scans (abz_rule_ggr).  Usage: prof_ggr.py [npt reps scans]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
from autobzcore.jl_amd import _lib as L

npt, reps, scans = (int(v) for v in (sys.argv[1:4] + ["150", "20", "5"][len(sys.argv) - 1:]))
s = abz.load_w90_series(os.path.join(ROOT, "tests", "golden", "svo_hr.dat.gz"))
dev = s.device(); ctx = dev.ctx
r = abz.DeviceRule(dev, npt, None, L.WANT_EIG | L.WANT_VEL)
for _ in range(reps): r.rebuild()
ctx.sync()
Es = np.linspace(10, 15, 256)
for _ in range(scans): g = r.ggr(Es)
print("dos[128] =", g[128])
