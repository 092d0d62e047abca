"""Cold construction of symmetric (irreducible-node) rules on the SVO model: orbit tables, plan, uploads, fill.
ABZ_DEBUG_TIMING=1 prints the phases of abz_ptr_rule_build.  Usage: time_symbuild.py [npt ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
from autobzcore.jl_amd import _lib as L
from autobzcore.jl_amd.series import symptr_rule
s = abz.load_w90_series(os.path.join(ROOT, "tests", "golden", "svo_hr.dat.gz"))
dev = s.device(); ctx = dev.ctx
syms = abz.load_bz(abz.CubicSymIBZ(), 3.85856 * np.eye(3)).syms
dev.rule(16, None, L.WANT_H); ctx.sync()
if os.environ.get("FIRST") == "1":  # the very first symmetric solve of the process: no orbit tables cached yet
    bz0 = abz.load_bz(abz.CubicSymIBZ(), 3.85856 * np.eye(3))
    for alg, tag in ((abz.AutoPTR(nmin=30), "warm-up of the code objects on other grids (npt = 30, 60, ...)"), (abz.AutoPTR(), "FIRST solve on npt = 50, 100, 150 (no orbit tables cached)")):
        sol0 = abz.IntegralSolver(abz.FourierIntegrand(abz.DOSIntegrand(), s, 0.1), bz0, abz.EvalCounter(alg), abstol=1e-3)
        t0 = time.perf_counter(); r0 = sol0.solve_p(abz.MixedParameters(12.5)); t1 = time.perf_counter()
        print(f"AutoPTR cubic IBZ, {tag}: {1e3*(t1-t0):.2f} ms  numevals={r0.numevals}", flush=True)
for npt in [int(v) for v in sys.argv[1:]] or [50, 100, 150]:
    for want, name in ((L.WANT_H, "H"), (L.WANT_H | L.WANT_EIG, "H+eig"), (L.WANT_EIG | L.WANT_VEL, "eig+vel")):
        for rep in range(2):
            t0 = time.perf_counter()
            idx, w = symptr_rule(npt, 3, syms, ctx=ctx)
            t1 = time.perf_counter()
            r = abz.DeviceRule(dev, npt, syms, want); ctx.sync()
            t2 = time.perf_counter()
            print(f"cubic npt={npt} {name:8s} rep {rep}: symptr {1e3*(t1-t0):6.2f} ms ({len(w)} nodes)  DeviceRule (symptr again + build) {1e3*(t2-t1):6.2f} ms", flush=True)
            r.close()
bz = abz.load_bz(abz.CubicSymIBZ(), 3.85856 * np.eye(3))
sol = abz.IntegralSolver(abz.FourierIntegrand(abz.DOSIntegrand(), s, 0.1), bz, abz.EvalCounter(abz.AutoPTR()), abstol=1e-3)
for rep in range(3):
    dev.drop_rules()
    t0 = time.perf_counter(); r3 = sol.solve_p(abz.MixedParameters(12.5)); t1 = time.perf_counter()
    r4 = sol.solve_p(abz.MixedParameters(12.5)); t2 = time.perf_counter()
    print(f"AutoPTR cubic IBZ: cold {1e3*(t1-t0):.2f} ms  cached {1e3*(t2-t1):.2f} ms  u={r3.u:.10f} numevals={r3.numevals}")
