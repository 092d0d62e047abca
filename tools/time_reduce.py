"""Phase B timing vs number of sweep values (fixed per-sweep cost vs per-omega cost)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
from autobzcore.jl_amd import _lib as L
s = abz.load_w90_series(os.path.join(ROOT, "tests", "golden", "svo_hr.dat.gz"))
dev = s.device(); ctx = dev.ctx
rule = dev.rule(150, None, L.WANT_H | L.WANT_EIG)
for fid, name in ((L.F_DOS, "DOS (matrix-cached)"), (L.F_DOS_EIG, "DOS (eig-cached)"), (L.F_TRGLOC, "TRGLOC"), (L.F_GLOC, "GLOC")):
    for nw in (1, 8, 32, 128, 256):
        if fid == L.F_GLOC and nw > 32: continue
        om = np.linspace(10, 15, nw)
        rule.reduce(fid, [0.1], om)
        ctx.prof_enable(True, kernels=[L.K_REDUCE]); ctx.prof_reset()
        for _ in range(5): rule.reduce(fid, [0.1], om)
        ms, n = ctx.prof_read(L.K_REDUCE); ctx.prof_enable(False)
        print(f"{name:22s} n_omega={nw:4d}: {ms/n:8.4f} ms  -> {150**3*nw/(ms/n*1e-3)/1e9:8.1f} G (k,omega)/s  {nw/(ms/n*1e-3):10.0f} DOS pts/s")
