"""GGR for more than four bands: build (eigenvalues + velocities) and scan times; ABZ_GGR_FUSED=0 times the unfused build."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
cases = ((5, 2, 24), (6, 2, 24), (8, 2, 24), (12, 2, 24), (16, 3, 24), (16, 6, 24), (17, 2, 24), (24, 2, 24), (32, 2, 24), (16, 3, 48), (8, 2, 64))
if len(sys.argv) > 1:
    cases = tuple(tuple(int(v) for v in a.split(",")) for a in sys.argv[1:])
for n, rmax, npt in cases:
    s = abz.synthetic_wannier(n=n, rmax=rmax, seed=7)
    dev = s.device()
    rule = dev.rule(npt, None, want=2 | 4)
    dev.ctx.sync()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); rule.rebuild(); dev.ctx.sync(); ts.append(time.perf_counter() - t0)
    tb = min(ts)
    Es = np.linspace(-2, 2, 64)
    bz = abz.load_bz(abz.FBZ(), np.eye(3))
    abz.dos.solve(abz.DOSProblem(s, Es, bz), abz.GGR(npt=npt))
    t0 = time.perf_counter(); u = abz.dos.solve(abz.DOSProblem(s, Es, bz), abz.GGR(npt=npt)).u; tsolve = time.perf_counter() - t0
    print(f"n={n:2d} M={2*rmax+1:2d} npt={npt}: rule (eig + velocities) {1e3*tb:8.3f} ms = {npt**3/tb/1e6:8.2f} M nodes/s   GGR solve, 64 energies {1e3*tsolve:8.2f} ms   dos[32]={u[32]:.6f}", flush=True)
    rule.close()
