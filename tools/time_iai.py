"""Timing of IAI solves (SVO example of aps_example/aps_example.jl:29-34) and of config 5."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np

import autobzcore.jl_amd as abz

s = abz.load_w90_series(os.path.join(ROOT, "tests", "golden", "svo_hr.dat.gz"))
A = 3.85856 * np.eye(3)
for kind, bzk in (("CubicSymIBZ", abz.CubicSymIBZ()), ("FBZ", abz.FBZ())):
    bz = abz.load_bz(bzk, A)
    for eta in (0.1, 0.01):
        f = abz.FourierIntegrand(abz.DOSIntegrand(), s, eta)
        for om in (12.5, 11.0):
            t0 = time.perf_counter()
            sol = abz.solve(abz.IntegralProblem(f, bz, abz.MixedParameters(om)), abz.EvalCounter(abz.IAI()), abstol=1e-3)
            dt = time.perf_counter() - t0
            print(f"SVO IAI {kind:12s} eta={eta:5.2f} omega={om:5.1f}: u={sol.u:.6f} err={sol.resid:.2e} numevals={sol.numevals:>10d} "
                  f"t={dt*1e3:8.1f} ms  ({sol.numevals/dt/1e6:7.2f} M nodes/s)", flush=True)
if len(sys.argv) > 1:
    import abz_oracle as orc
    so = orc.synthetic_wannier()
    s16 = abz.FourierSeries(so.c, period=1.0, first=so.first, ndim=3)
    f = abz.FourierIntegrand(abz.DOSIntegrand(), s16, 0.05)
    bz = abz.load_bz(abz.FBZ(), np.eye(3))
    for tol in (1.0, 0.1):
        t0 = time.perf_counter()
        sol = abz.solve(abz.IntegralProblem(f, bz, abz.MixedParameters(0.2)), abz.EvalCounter(abz.IAI()), abstol=tol, reltol=0.0)
        dt = time.perf_counter() - t0
        print(f"C5 16-band IAI abstol={tol}: u={sol.u:.4f} err={sol.resid:.2e} numevals={sol.numevals} t={dt:.2f} s ({sol.numevals/dt/1e6:.2f} M nodes/s)", flush=True)
