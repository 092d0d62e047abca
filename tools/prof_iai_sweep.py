"""The 432-omega IAI sweep of the reference example, a few times (for a kernel trace)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
h = abz.load_w90_series(os.path.join(ROOT, "tests", "golden", "svo_hr.dat.gz"))
bz = abz.load_bz(abz.CubicSymIBZ(), 3.85856 * np.eye(3))
solver = abz.IntegralSolver(abz.FourierIntegrand(abz.DOSIntegrand(), h, 0.01), bz, abz.EvalCounter(abz.IAI()), abstol=1e-3)
om = np.linspace(10, 15, 432)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
tot = []
abz.batchsolve(solver, om, callback=lambda s_, i, k, p, sol, t: tot.append(sol.numevals))
print("nodes per sweep", sum(tot))
for rep in range(n):
    t0 = time.perf_counter(); abz.batchsolve(solver, om); print(f"sweep {1e3*(time.perf_counter()-t0):.1f} ms", flush=True)
