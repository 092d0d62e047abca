"""Config 5 (synthetic 16-band, IAI on the FBZ) at a given abstol: time, nodes/s, launches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
abstol = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-3
s16 = abz.synthetic_wannier()
f16 = abz.FourierIntegrand(abz.DOSIntegrand(), s16, 0.05)
prob = abz.IntegralProblem(f16, abz.load_bz(abz.FBZ(), np.eye(3)), abz.MixedParameters(0.2))
abz.solve(prob, abz.IAI(), abstol=10.0, reltol=0.0)
t0 = time.perf_counter()
sol = abz.solve(prob, abz.EvalCounter(abz.IAI()), abstol=abstol, reltol=0.0)
dt = time.perf_counter() - t0
print(f"abstol {abstol}: u = {sol.u!r} resid {sol.resid:.3e} numevals {sol.numevals} in {dt:.3f} s = {sol.numevals/dt/1e6:.1f} M nodes/s "
      f"(spec={os.environ.get('ABZ_IAI_SPECULATE','1')})", flush=True)
