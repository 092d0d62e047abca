import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import autobzcore.jl_amd as abz
L = abz._lib
s16 = abz.synthetic_wannier()
dev = s16.device()
for npt in (96, 120, 160, 200, 240, 300):
    t0 = time.perf_counter()
    v = dev.ptr_sum(npt, L.F_DOS, [0.05], [0.2])[0, 0].real * (2 * np.pi) ** 3
    print(npt, repr(v), time.perf_counter() - t0, flush=True)
# SVO IAI stats
s = abz.load_w90_series("tests/golden/svo_hr.dat.gz")
f = abz.FourierIntegrand(abz.DOSIntegrand(), s, 0.01)
for kind in (abz.CubicSymIBZ(), abz.FBZ()):
    bz = abz.load_bz(kind, 3.85856 * np.eye(3))
    prob = abz.IntegralProblem(f, bz, abz.MixedParameters(12.5))
    abz.solve(prob, abz.IAI(), abstol=1e-3)
    t0 = time.perf_counter()
    sol = abz.solve(prob, abz.EvalCounter(abz.IAI()), abstol=1e-3)
    print(type(kind).__name__, sol.u, sol.numevals, time.perf_counter() - t0, flush=True)
sol_iai = abz.IntegralSolver(f, abz.load_bz(abz.CubicSymIBZ(), 3.85856 * np.eye(3)), abz.IAI(), abstol=1e-3)
om = np.linspace(10, 15, 432)
abz.batchsolve(sol_iai, om[:8])
t0 = time.perf_counter(); abz.batchsolve(sol_iai, om); print("sweep432", time.perf_counter() - t0)
