"""rocprofv3 target: rule builds (H + eigenvalues) and store-free sums of a small-band model on the 64^3 grid.  argv: n [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
from autobzcore.jl_amd import _lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rng = np.random.default_rng(5)
M, npt = 5, 64
c = rng.standard_normal((M, M, M, n, n)) + 1j * rng.standard_normal((M, M, M, n, n))
c = c + np.conj(np.swapaxes(c[::-1, ::-1, ::-1], -1, -2))
s = abz.FourierSeries(c / n, period=1.0, first=(-(M // 2),) * 3)
dev = s.device(); ctx = dev.ctx
for want in (L.WANT_H | L.WANT_H_COMPACT, L.WANT_H | L.WANT_EIG | L.WANT_H_COMPACT, L.WANT_EIG):
    r = abz.DeviceRule(dev, npt, None, want); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        r.rebuild()
    ctx.sync(); print("want", want, "ms", 1e3 * (time.perf_counter() - t0) / reps)
    r.close()
for nw in (1, 16):
    om = np.linspace(-1, 1, nw)
    dev.ptr_sum(npt, L.F_DOS, [0.05], om)
    t0 = time.perf_counter()
    for _ in range(reps):
        dev.ptr_sum(npt, L.F_DOS, [0.05], om)
    print("sum", nw, "ms", 1e3 * (time.perf_counter() - t0) / reps)
