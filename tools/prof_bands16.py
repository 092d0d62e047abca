"""The 16-band kernels, one short pass each (for rocprofv3): config-5 IAI (gen_inner_panel_kernel), store-free PTR sum
(gen_grid_sum_kernel), rule build with eigenvalues (gen_grid_eig_kernel), rule scan (gen_rows_reduce_kernel)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
L = abz._lib
tol = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
s16 = abz.synthetic_wannier()
dev = s16.device()
f = abz.FourierIntegrand(abz.DOSIntegrand(), s16, 0.05)
prob = abz.IntegralProblem(f, abz.load_bz(abz.FBZ(), np.eye(3)), abz.MixedParameters(0.2))
t0 = time.perf_counter()
sol = abz.solve(prob, abz.EvalCounter(abz.IAI()), abstol=tol, reltol=0.0)
print(f"IAI abstol {tol}: {sol.numevals} nodes in {time.perf_counter() - t0:.3f} s", flush=True)
om = np.linspace(-1, 1, 16)
for _ in range(2):
    t0 = time.perf_counter()
    dev.ptr_sum(96, L.F_DOS, [0.05], om)
    print(f"store-free 96^3 x 16 omega: {time.perf_counter() - t0:.4f} s", flush=True)
r = abz.DeviceRule(dev, 48, None, L.WANT_H | L.WANT_EIG)
for _ in range(3):
    r.rebuild()
dev.ctx.sync()
for _ in range(2):
    r.reduce(L.F_DOS, [0.05], om)
