import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import autobzcore.jl_amd as abz
s = abz.load_w90_series("tests/golden/svo_hr.dat.gz")
f = abz.FourierIntegrand(abz.DOSIntegrand(), s, 0.01)
sol_iai = abz.IntegralSolver(f, abz.load_bz(abz.CubicSymIBZ(), 3.85856 * np.eye(3)), abz.IAI(), abstol=1e-3)
om = np.linspace(10, 15, 432)
abz.batchsolve(sol_iai, om[:8])
for rep in range(3):
    t0 = time.perf_counter(); r = abz.batchsolve(sol_iai, om); print("sweep432", time.perf_counter() - t0, "spec", os.environ.get("ABZ_IAI_SPECULATE", "1"), float(np.sum(r)), flush=True)
