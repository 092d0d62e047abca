"""432 IAI solves in lock-step (the load of the reference's aps_example): batchsolve time."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from autobzcore.jl_amd import _lib as L
if os.environ.get("ABZ_LIB"):  # older builds of the library lack the newest entry points
    dll = ctypes.CDLL(os.environ["ABZ_LIB"])
    for name in list(L.PROTOTYPES):
        if not hasattr(dll, name):
            del L.PROTOTYPES[name]
import autobzcore.jl_amd as abz
h = abz.load_w90_series(os.path.join(ROOT, "tests", "golden", "svo_hr.dat.gz"))
bz = abz.load_bz(abz.CubicSymIBZ(), 3.85856 * np.eye(3))
solver = abz.IntegralSolver(abz.FourierIntegrand(abz.DOSIntegrand(), h, 0.01), bz, abz.IAI(), abstol=1e-3)
om = np.linspace(10, 15, 432)
abz.batchsolve(solver, om[:8])
for rep in range(2):
    t0 = time.perf_counter(); v = abz.batchsolve(solver, om); dt = time.perf_counter() - t0
    print(f"432-omega IAI sweep: {dt:.3f} s  (sum {v.sum():.6f})", flush=True)
ctx = h.device().ctx
ctx.prof_enable(True); ctx.prof_reset()
t0 = time.perf_counter(); v = abz.batchsolve(solver, om); dt = time.perf_counter() - t0
for name, kid in (("contract", L.K_CONTRACT), ("eval/inner", L.K_EVAL)):
    ms, n = ctx.prof_read(kid); print(f"  {name}: {n} launches, {ms:.1f} ms")
print(f"  wall {1e3*dt:.1f} ms")
