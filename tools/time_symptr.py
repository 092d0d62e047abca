import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
ctx = abz.Context.default()
bz = abz.load_bz(abz.CubicSymIBZ(), np.eye(3))
abz.symptr_rule(8, 3, bz.syms, ctx=ctx)
for npt in (50, 100, 150, 200, 300):
    t = time.perf_counter(); ih, wh = abz.symptr_rule(npt, 3, bz.syms); th = time.perf_counter() - t
    t = time.perf_counter(); ig, wg = abz.symptr_rule(npt, 3, bz.syms, ctx=ctx); tg = time.perf_counter() - t
    print(f"npt={npt}: nirr={len(wg)} host {th*1e3:.1f} ms  device {tg*1e3:.1f} ms  equal={np.array_equal(ih, ig) and np.array_equal(wh, wg)}")
