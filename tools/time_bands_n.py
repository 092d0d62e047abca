"""Rule builds (H, H + eig) and store-free DOS sums against the number of bands (random Hermitian series, 3 variables, 5^3 R)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
from autobzcore.jl_amd import _lib as L
rng = np.random.default_rng(5)
npt = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for n in (8, 12, 16, 17, 20, 24, 32):
    M = 5
    c = rng.standard_normal((M, M, M, n, n)) + 1j * rng.standard_normal((M, M, M, n, n))
    c = c + np.conj(np.swapaxes(c[::-1, ::-1, ::-1], -1, -2))
    s = abz.FourierSeries(c / n, period=1.0, first=(-(M // 2),) * 3)
    dev = s.device(); ctx = dev.ctx
    row = [f"n={n:2d}"]
    for want, name in ((L.WANT_H, "H"), (L.WANT_H | L.WANT_EIG, "H+EIG")):
        r = abz.DeviceRule(dev, npt, None, want); ctx.sync()
        for _ in range(2):
            r.rebuild()
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(5):
            r.rebuild()
        ctx.sync(); dt = (time.perf_counter() - t0) / 5
        row.append(f"{name} {1e3*dt:8.3f} ms")
        r.close()
    for nw in (1, 16):
        om = np.linspace(-1, 1, nw)
        dev.ptr_sum(npt, L.F_DOS, [0.05], om)
        t0 = time.perf_counter(); dev.ptr_sum(npt, L.F_DOS, [0.05], om); dt = time.perf_counter() - t0
        row.append(f"sum[{nw:2d} w] {1e3*dt:8.3f} ms")
    print("  ".join(row), flush=True)
