"""Config 5's value by store-free PTR sums on finer and finer grids (the reference a test of IAI's abstol needs), beside
IAI at a few tolerances.  Usage: c5_reference.py [npt ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
s16 = abz.synthetic_wannier()
dev = s16.device()
vals = {}
for npt in [int(v) for v in sys.argv[1:]] or [240, 300, 400, 500, 600, 720]:
    t0 = time.perf_counter()
    v = dev.ptr_sum(npt, abz._lib.F_DOS, [0.05], [0.2])[0, 0].real
    vals[npt] = v
    print(f"PTR npt={npt}: {v!r}  ({time.perf_counter()-t0:.2f} s)", flush=True)
ks = sorted(vals)
for a, b in zip(ks, ks[1:]):
    print(f"  |PTR({b}) - PTR({a})| = {abs(vals[b]-vals[a]):.3e}")
f16 = abz.FourierIntegrand(abz.DOSIntegrand(), s16, 0.05)
prob = abz.IntegralProblem(f16, abz.load_bz(abz.FBZ(), np.eye(3)), abz.MixedParameters(0.2))
for tol in (1e-1, 1e-2, 1e-3):
    t0 = time.perf_counter()
    sol = abz.solve(prob, abz.EvalCounter(abz.IAI()), abstol=tol, reltol=0.0)
    print(f"IAI abstol={tol}: u={sol.u!r} resid={sol.resid:.3e} numevals={sol.numevals} |u - PTR({ks[-1]})| = {abs(sol.u-vals[ks[-1]]):.3e}  ({time.perf_counter()-t0:.1f} s)", flush=True)
