"""IAI sweep time against the number of solves and lanes (ABZ_IAI_LANES / ABZ_IAI_LANE_MIN)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
h = abz.load_w90_series(os.path.join(ROOT, "tests", "golden", "svo_hr.dat.gz"))
bz = abz.load_bz(abz.CubicSymIBZ(), 3.85856 * np.eye(3))
solver = abz.IntegralSolver(abz.FourierIntegrand(abz.DOSIntegrand(), h, 0.01), bz, abz.IAI(), abstol=1e-3)
om = np.linspace(10, 15, 432)
for nsol in (432, 216, 108, 54, 27):
    sub = om[:: 432 // nsol][:nsol]
    row = []
    for lanes in (1, 2, 4, 6, 8):
        os.environ["ABZ_IAI_LANES"] = str(lanes)
        os.environ["ABZ_IAI_LANE_MIN"] = "1"
        abz.batchsolve(solver, sub)
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter(); abz.batchsolve(solver, sub); best = min(best, time.perf_counter() - t0)
        row.append(best)
    print(f"{nsol:4d} solves: " + "  ".join(f"{l} lanes {1e3*t:6.1f} ms" for l, t in zip((1, 2, 4, 6, 8), row)), flush=True)
