"""IAI (DOS, 3 variables) against the number of bands: nodes per second of the innermost device loops."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
tol = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
for n in (8, 16, 17, 24, 32):
    s = abz.synthetic_wannier(n=n, rmax=2, seed=11)
    f = abz.FourierIntegrand(abz.DOSIntegrand(), s, 0.1)
    prob = abz.IntegralProblem(f, abz.load_bz(abz.FBZ(), np.eye(3)), abz.MixedParameters(0.2))
    abz.solve(prob, abz.EvalCounter(abz.IAI()), abstol=10 * tol, reltol=0.0)
    t0 = time.perf_counter()
    sol = abz.solve(prob, abz.EvalCounter(abz.IAI()), abstol=tol, reltol=0.0)
    dt = time.perf_counter() - t0
    print(f"n={n:2d}: {sol.numevals:12d} nodes in {dt:7.3f} s = {sol.numevals/dt/1e6:8.1f} M nodes/s   u={sol.u:.6f}", flush=True)
