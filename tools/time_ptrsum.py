"""Store-free rule values (abz_ptr_sum): k-points/s without materialising H(k); up to 1000^3 = 10^9 k-points."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
from autobzcore.jl_amd import _lib as L
s = abz.load_w90_series(os.path.join(ROOT, "tests", "golden", "svo_hr.dat.gz"))
dev = s.device()
for npt in (150, 300, 600, 1000):
    for nw in (1, 8):
        om = np.linspace(12.0, 13.0, nw)
        dev.ptr_sum(npt, L.F_DOS, [0.1], om)
        t0 = time.perf_counter(); v = dev.ptr_sum(npt, L.F_DOS, [0.1], om); dt = time.perf_counter() - t0
        print(f"npt={npt:5d} n_omega={nw}: {1e3*dt:9.3f} ms  {npt**3/dt/1e9:7.2f} G k-points/s  {npt**3*nw/dt/1e9:8.2f} G (k,omega)/s  dos[0]={v[0,0].real:.8f}", flush=True)
bz = abz.load_bz(abz.FBZ(), 3.85856 * np.eye(3))
solver = abz.IntegralSolver(abz.FourierIntegrand(abz.DOSIntegrand(), s, 0.01), bz, abz.EvalCounter(abz.AutoPTR()), abstol=1e-3)
t0 = time.perf_counter(); r = solver.solve_p(abz.MixedParameters(12.5)); dt = time.perf_counter() - t0
print(f"AutoPTR FBZ eta=0.01 (grids up to npt={r.extra.get('npt')}): u={r.u:.6f} err={r.resid:.2e} numevals={r.numevals} in {dt:.3f} s")
