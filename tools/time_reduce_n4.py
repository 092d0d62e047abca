import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
from autobzcore.jl_amd import _lib as L
rng = np.random.default_rng(0)
for n in (4, 3):
    c = rng.standard_normal((5, 5, 5, n, n)) + 1j * rng.standard_normal((5, 5, 5, n, n))
    c = 0.5 * (c + np.conj(np.swapaxes(c[::-1, ::-1, ::-1], -1, -2)))
    s = abz.FourierSeries(c, period=1.0, first=(-2, -2, -2), ndim=3)
    dev = s.device(); ctx = dev.ctx
    rule = dev.rule(150, None, L.WANT_H | L.WANT_EIG)
    for fid, name in ((L.F_DOS, "DOS"), (L.F_TRGLOC, "TRGLOC"), (L.F_DOS_EIG, "DOS eig")):
        for nw in (1, 32, 256):
            om = np.linspace(-2, 2, nw)
            rule.reduce(fid, [0.2], om)
            ctx.prof_enable(True, kernels=[L.K_REDUCE]); ctx.prof_reset()
            for _ in range(3): v = rule.reduce(fid, [0.2], om)
            ms, k = ctx.prof_read(L.K_REDUCE); ctx.prof_enable(False)
            print(f"n={n} {name:8s} n_omega={nw:4d}: {ms/k:8.4f} ms  {150**3*nw/(ms/k*1e-3)/1e9:8.1f} G (k,omega)/s")
    ref = rule.reduce(L.F_DOS_EIG, [0.2], np.linspace(-2, 2, 5))
    got = rule.reduce(L.F_DOS, [0.2], np.linspace(-2, 2, 5))
    print("   DOS vs eig form max rel diff", np.abs(got - ref).max() / np.abs(ref).max())
