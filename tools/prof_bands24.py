"""24-band kernels (32 lanes per node), one short pass each for rocprofv3: rule build with eigenvalues, store-free 16-value sweep, IAI."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
from autobzcore.jl_amd import _lib as L
rng = np.random.default_rng(24)
n, M = 24, 5
c = rng.standard_normal((M, M, M, n, n)) + 1j * rng.standard_normal((M, M, M, n, n))
c = (c + np.conj(np.swapaxes(c[::-1, ::-1, ::-1], -1, -2))) / n
s = abz.FourierSeries(c, period=1.0, first=(-2, -2, -2))
dev = s.device()
r = abz.DeviceRule(dev, 32, None, L.WANT_H | L.WANT_EIG)
for _ in range(4):
    r.rebuild()
dev.ctx.sync()
om = np.linspace(-1, 1, 16)
for _ in range(3):
    dev.ptr_sum(32, L.F_DOS, [0.05], om)
    r.reduce(L.F_DOS, [0.05], om)
f = abz.FourierIntegrand(abz.DOSIntegrand(), s, 0.1)
prob = abz.IntegralProblem(f, abz.load_bz(abz.FBZ(), np.eye(3)), abz.MixedParameters(0.2))
t0 = time.perf_counter(); sol = abz.solve(prob, abz.EvalCounter(abz.IAI()), abstol=0.3, reltol=0.0)
print(f"IAI: {sol.numevals} nodes in {time.perf_counter()-t0:.3f} s")
