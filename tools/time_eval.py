"""Fourier-eval kernel timing (HIP events on the library's stream) for a few grid sizes: avg launch ms and fraction of the
8 TB/s HBM peak at 168 B per k-point.  Env switches of the experiment are read by the library."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
L = abz._lib
s = abz.load_w90_series(os.path.join(ROOT, "tests", "golden", "svo_hr.dat.gz"))
dev = s.device()
ctx = dev.ctx
for npt in [int(v) for v in (sys.argv[1:] or ["150", "160", "200"])]:
    rule = abz.DeviceRule(dev, npt, None, L.WANT_H | L.WANT_EIG)
    for _ in range(20):
        rule.rebuild()
    ctx.sync()
    res = []
    for rep in range(3):
        ctx.prof_enable(True, kernels=[L.K_EVAL])
        ctx.prof_reset()
        for _ in range(200):
            rule.rebuild()
        ctx.sync()
        ms, n = ctx.prof_read(L.K_EVAL)
        ctx.prof_enable(False)
        res.append(ms / n)
    nk = npt**3
    base, nb = rule.values_ptr()
    print(f"npt {npt}: eval kernel {min(res):.4f} .. {max(res):.4f} ms -> {nk*168/min(res)*1e-6:.0f} GB/s = {nk*168/min(res)*1e-6/8000:.3f} of peak; "
          f"rule {nb/1e6:.0f} MB (algorithmic {nk*168/1e6:.0f} MB)", flush=True)
    rule.close()
