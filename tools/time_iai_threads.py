"""432-omega IAI sweep of the reference example against ABZ_HOST_THREADS and ABZ_IAI_LANES."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
h = abz.load_w90_series(os.path.join(ROOT, "tests", "golden", "svo_hr.dat.gz"))
bz = abz.load_bz(abz.CubicSymIBZ(), 3.85856 * np.eye(3))
solver = abz.IntegralSolver(abz.FourierIntegrand(abz.DOSIntegrand(), h, 0.01), bz, abz.IAI(), abstol=1e-3)
om = np.linspace(10, 15, 432)
for lanes in (4, 6):
    row = []
    for th in (1, 2, 4, 8):
        os.environ["ABZ_IAI_LANES"] = str(lanes)
        os.environ["ABZ_HOST_THREADS"] = str(th)
        abz.batchsolve(solver, om)
        best = 1e9
        for rep in range(4):
            t0 = time.perf_counter(); abz.batchsolve(solver, om); best = min(best, time.perf_counter() - t0)
        row.append(f"{th} threads {1e3*best:6.1f} ms")
    print(f"{lanes} lanes: " + "  ".join(row), flush=True)
