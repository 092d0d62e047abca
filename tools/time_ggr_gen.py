"""GGR for more than four bands: build (eigenvectors + velocities) and scan times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
for n, rmax, npt in ((6, 2, 24), (8, 2, 24), (16, 3, 24), (17, 2, 24), (24, 2, 24)):
    s = abz.synthetic_wannier(n=n, rmax=rmax, seed=7)
    dev = s.device()
    rule = dev.rule(npt, None, want=2 | 4)
    dev.ctx.sync()
    t0 = time.perf_counter(); rule.rebuild(); dev.ctx.sync(); tb = time.perf_counter() - t0
    Es = np.linspace(-2, 2, 64)
    bz = abz.load_bz(abz.FBZ(), np.eye(3))
    abz.dos.solve(abz.DOSProblem(s, Es, bz), abz.GGR(npt=npt))
    t0 = time.perf_counter(); u = abz.dos.solve(abz.DOSProblem(s, Es, bz), abz.GGR(npt=npt)).u; ts = time.perf_counter() - t0
    print(f"n={n:2d} npt={npt}: rule (eig + velocities) {1e3*tb:8.2f} ms   GGR solve, 64 energies {1e3*ts:8.2f} ms   dos[32]={u[32]:.6f}", flush=True)
    rule.close()
