"""Ablation of the Fourier-eval kernel: time rule rebuilds for each `want` mask (HIP events)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import autobzcore.jl_amd as abz
from autobzcore.jl_amd import _lib as L

npt = int(sys.argv[1]) if len(sys.argv) > 1 else 150
s = abz.load_w90_series(os.path.join(ROOT, "tests", "golden", "svo_hr.dat.gz"))
dev = s.device()
ctx = dev.ctx
for want, name in ((1, "H only"), (2, "EIG only"), (3, "H+EIG")):
    rule = abz.DeviceRule(dev, npt, None, want)
    for _ in range(3):
        rule.rebuild()
    ctx.sync()
    ctx.prof_enable(True)
    ctx.prof_reset()
    for _ in range(20):
        rule.rebuild()
    ctx.sync()
    ms, n = ctx.prof_read(L.K_EVAL)
    cms, cn = ctx.prof_read(L.K_CONTRACT)
    ctx.prof_enable(False)
    nk = npt**3
    print(f"npt={npt} want={name:9s} eval {ms/n:.4f} ms  ({nk/(ms/n*1e-3)/1e9:.2f} G k/s)  contract {cms/cn:.4f} ms x{cn//n}")
    rule.close()
