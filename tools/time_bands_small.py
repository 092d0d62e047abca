"""Rule builds, store-free sums and IAI against the number of bands around the 4 -> 5 step (closed-form kernels vs row kernels)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
from autobzcore.jl_amd import _lib as L
rng = np.random.default_rng(5)
npt = 64
for n in (2, 3, 4, 5, 6, 8):
    M = 5
    c = rng.standard_normal((M, M, M, n, n)) + 1j * rng.standard_normal((M, M, M, n, n))
    c = c + np.conj(np.swapaxes(c[::-1, ::-1, ::-1], -1, -2))
    s = abz.FourierSeries(c / n, period=1.0, first=(-(M // 2),) * 3)
    dev = s.device(); ctx = dev.ctx
    row = [f"n={n:2d}"]
    for want, name in ((L.WANT_H, "H"), (L.WANT_H | L.WANT_EIG, "H+EIG"), (L.WANT_H | L.WANT_EIG, "H+EIG@160")):
        r = abz.DeviceRule(dev, 160 if name.endswith("160") else npt, None, want); ctx.sync()
        for _ in range(2):
            r.rebuild()
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(5):
            r.rebuild()
        ctx.sync(); dt = (time.perf_counter() - t0) / 5
        row.append(f"{name} {1e3*dt:7.3f} ms")
        if (want & L.WANT_EIG) and not name.endswith("160"):
            om = np.linspace(-1, 1, 16)
            r.reduce(L.F_DOS, [0.05], om); t0 = time.perf_counter(); r.reduce(L.F_DOS, [0.05], om); row.append(f"scan16 {1e3*(time.perf_counter()-t0):7.3f} ms")
        r.close()
    for nw in (1, 16):
        om = np.linspace(-1, 1, nw)
        try:
            for g in ((160,) if n <= 4 else (npt, 160)):  # (5...8 bands: also on the 160^3 grid of the closed-form kernels: per-node cost without the call overhead)
                dev.ptr_sum(g, L.F_DOS, [0.05], om)
                t0 = time.perf_counter(); dev.ptr_sum(g, L.F_DOS, [0.05], om); dt = time.perf_counter() - t0
                row.append(f"sum[{nw:2d}w,{g}^3] {1e3*dt:7.3f} ms")
        except Exception as e:
            row.append(f"sum[{nw}] n/a")
    f = abz.FourierIntegrand(abz.DOSIntegrand(), s, 0.1)
    prob = abz.IntegralProblem(f, abz.load_bz(abz.FBZ(), np.eye(3)), abz.MixedParameters(0.2))
    abz.solve(prob, abz.EvalCounter(abz.IAI()), abstol=1.0, reltol=0.0)
    t0 = time.perf_counter(); sol = abz.solve(prob, abz.EvalCounter(abz.IAI()), abstol=0.03, reltol=0.0); dt = time.perf_counter() - t0
    row.append(f"IAI {sol.numevals/dt/1e6:8.1f} M nodes/s")
    print("  ".join(row), flush=True)
