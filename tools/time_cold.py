"""First / rule-less / cached AutoPTR solves of config 3 (SVO DOS at one omega) on the FBZ and the cubic IBZ; with an
argument the symmetric-rule kernels are used once before (their code object is then loaded)."""
import os, sys, time
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd()
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
from autobzcore.jl_amd import _lib as L
s = abz.load_w90_series(os.path.join(ROOT, "tests", "golden", "svo_hr.dat.gz"))
A = 3.85856 * np.eye(3)
if len(sys.argv) > 1:  # the symmetric-rule kernels run once on another grid first: their code object is loaded
    abz.DeviceRule(s.device(), 6, abz.load_bz(abz.CubicSymIBZ(), A).syms, L.WANT_H).close()
for kind, bzk in (("FBZ", abz.FBZ()), ("CubicSymIBZ", abz.CubicSymIBZ())):
    bz = abz.load_bz(bzk, A)
    solver = abz.IntegralSolver(abz.FourierIntegrand(abz.DOSIntegrand(), s, 0.1), bz, abz.EvalCounter(abz.AutoPTR()), abstol=1e-3)
    s.device().drop_rules()
    print("----", kind, "first", flush=True)
    t0 = time.perf_counter(); r = solver.solve_p(abz.MixedParameters(12.5)); t1 = time.perf_counter()
    print(f"{kind}: first {1e3*(t1-t0):.3f} ms", flush=True)
    s.device().drop_rules()
    print("----", kind, "second (tables cached)", flush=True)
    t0 = time.perf_counter(); r = solver.solve_p(abz.MixedParameters(12.5)); t1 = time.perf_counter()
    print(f"{kind}: rules dropped {1e3*(t1-t0):.3f} ms", flush=True)
    t0 = time.perf_counter(); r = solver.solve_p(abz.MixedParameters(12.5)); t1 = time.perf_counter()
    print(f"{kind}: cached {1e3*(t1-t0):.3f} ms   u = {r.u!r} resid {r.resid:.3e} numevals {r.numevals} last npt {r.extra['npt']}", flush=True)
    ts = []
    for _ in range(200):
        t0 = time.perf_counter(); solver.solve_p(abz.MixedParameters(12.5)); ts.append(time.perf_counter() - t0)
    print(f"{kind}: cached, 200 solves: min {1e3*min(ts):.3f} median {1e3*sorted(ts)[100]:.3f} ms", flush=True)
