"""33...64 bands (kernels_big.hip): rule builds on the 24^3 grid -- H only (the level-1 evaluation: vector FMAs against the real
GEMM on v_mfma_f64_16x16x4_f64, ABZ_BIG_MFMA=1), H + eigenvalues, eigenvalues only -- a 16-omega store-free sweep and an IAI solve."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
from autobzcore.jl_amd import _lib as L
npt = int(sys.argv[1]) if len(sys.argv) > 1 else 24
for n, rmax in ((32, 2), (33, 2), (48, 2), (64, 2), (48, 6)):
    s = abz.synthetic_wannier(n=n, rmax=rmax, seed=7)
    dev = s.device(); ctx = dev.ctx
    row = [f"n={n:2d} M={2*rmax+1:2d} {npt}^3:"]
    for mfma in ("0", "1"):
        os.environ["ABZ_BIG_MFMA"] = mfma
        for want, name in ((L.WANT_H, "H"), (L.WANT_H | L.WANT_EIG, "H+eig"), (L.WANT_EIG, "eig")):
            r = abz.DeviceRule(dev, npt, None, want); ctx.sync()
            ts = []
            for _ in range(4):
                t0 = time.perf_counter(); r.rebuild(); ctx.sync(); ts.append(time.perf_counter() - t0)
            row.append(f"{name}[mfma={mfma}] {1e3*min(ts):8.3f} ms")
            r.close()
        if n <= 32:
            break
    os.environ["ABZ_BIG_MFMA"] = "0"
    om = np.linspace(-1, 1, 16)
    dev.ptr_sum(npt, L.F_DOS, [0.05], om)
    t0 = time.perf_counter(); dev.ptr_sum(npt, L.F_DOS, [0.05], om); row.append(f"sum[16w] {1e3*(time.perf_counter()-t0):8.3f} ms")
    if n > 32 and rmax == 2:  # the matrix-valued Green's function of a cached rule (big_inverse_kernel): 4 omega
        r = abz.DeviceRule(dev, npt, None, L.WANT_H); ctx.sync()
        om4 = np.linspace(-1, 1, 4)
        r.reduce(L.F_GLOC, [0.05], om4)
        t0 = time.perf_counter(); r.reduce(L.F_GLOC, [0.05], om4); row.append(f"G scan[4w] {1e3*(time.perf_counter()-t0):8.3f} ms")
        r.close()
    if rmax == 2:
        f = abz.FourierIntegrand(abz.DOSIntegrand(), s, 0.1)
        prob = abz.IntegralProblem(f, abz.load_bz(abz.FBZ(), np.eye(3)), abz.MixedParameters(0.2))
        abz.solve(prob, abz.EvalCounter(abz.IAI()), abstol=10.0, reltol=0.0)
        t0 = time.perf_counter(); sol = abz.solve(prob, abz.EvalCounter(abz.IAI()), abstol=1.0, reltol=0.0); dt = time.perf_counter() - t0
        row.append(f"IAI {sol.numevals/dt/1e6:7.2f} M nodes/s ({sol.numevals} nodes)")
    print("  ".join(row), flush=True)
