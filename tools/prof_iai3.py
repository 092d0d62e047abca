"""The reference example's IAI solve (SVO 3 bands, eta = 0.01 eV, abstol 1e-3; aps_example/aps_example.jl:29-34) on the full BZ:
the workload behind the counters of inner_adaptive_kernel<3> (tools/pmc_iai3.sh)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
s = abz.load_w90_series(os.path.join(ROOT, "tests", "golden", "svo_hr.dat.gz"))
f = abz.FourierIntegrand(abz.DOSIntegrand(), s, 0.01)
bz = abz.load_bz(abz.FBZ(), 3.85856 * np.eye(3))
prob = abz.IntegralProblem(f, bz, abz.MixedParameters(12.5))
abz.solve(prob, abz.IAI(), abstol=1e-3)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    t0 = time.perf_counter()
    sol = abz.solve(prob, abz.EvalCounter(abz.IAI()), abstol=1e-3)
    print(f"u={sol.u:.6f} numevals={sol.numevals} {1e3*(time.perf_counter()-t0):.2f} ms", flush=True)
