"""Cached AutoPTR solves of config 3 (SVO DOS at one omega) for a kernel trace: python tools/prof_autoptr.py FBZ|CubicSymIBZ [n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
s = abz.load_w90_series(os.path.join(ROOT, "tests", "golden", "svo_hr.dat.gz"))
kind = sys.argv[1] if len(sys.argv) > 1 else "FBZ"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bz = abz.load_bz(getattr(abz, kind)(), 3.85856 * np.eye(3))
solver = abz.IntegralSolver(abz.FourierIntegrand(abz.DOSIntegrand(), s, 0.1), bz, abz.EvalCounter(abz.AutoPTR()), abstol=1e-3)
r = solver.solve_p(abz.MixedParameters(12.5))
ts = []
for _ in range(n):
    t0 = time.perf_counter(); r = solver.solve_p(abz.MixedParameters(12.5)); ts.append(time.perf_counter() - t0)
print(f"{kind}: cached, {n} solves: min {1e3*min(ts):.3f} median {1e3*sorted(ts)[n//2]:.3f} ms  numevals {r.numevals} npt {r.extra['npt']}", flush=True)
