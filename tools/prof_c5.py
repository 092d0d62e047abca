"""Config 5 (synthetic 16-band, IAI): wall time vs time inside the library's kernels."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np

import abz_oracle as orc
import autobzcore.jl_amd as abz
from autobzcore.jl_amd import _lib as L

so = orc.synthetic_wannier()
s16 = abz.FourierSeries(so.c, period=1.0, first=so.first, ndim=3)
f = abz.FourierIntegrand(abz.DOSIntegrand(), s16, 0.05)
bz = abz.load_bz(abz.FBZ(), np.eye(3))
ctx = s16.device().ctx
tol = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
abz.solve(abz.IntegralProblem(f, bz, abz.MixedParameters(0.2)), abz.EvalCounter(abz.IAI()), abstol=10.0, reltol=0.0)
for prof in (False, True):
    if prof:
        ctx.prof_enable(True)
        ctx.prof_reset()
    t0 = time.perf_counter()
    sol = abz.solve(abz.IntegralProblem(f, bz, abz.MixedParameters(0.2)), abz.EvalCounter(abz.IAI()), abstol=tol, reltol=0.0)
    dt = time.perf_counter() - t0
    print(f"C5 abstol={tol} prof={prof}: u={sol.u:.5f} numevals={sol.numevals} t={dt:.3f} s ({sol.numevals/dt/1e6:.2f} M nodes/s)", flush=True)
    if prof:
        for name, kid in (("contract", L.K_CONTRACT), ("eval", L.K_EVAL), ("reduce", L.K_REDUCE), ("eig", L.K_EIG)):
            ms, n = ctx.prof_read(kid)
            print(f"   {name:9s}: {n:7d} launches, {ms:9.2f} ms total")
        ctx.prof_enable(False)
