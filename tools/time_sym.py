import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
from autobzcore.jl_amd import _lib as L
s = abz.load_w90_series(os.path.join(ROOT, "tests", "golden", "svo_hr.dat.gz"))
A = 3.85856 * np.eye(3)
ctx = s.device().ctx
for kind, bzk in (("InversionSymIBZ", abz.InversionSymIBZ()), ("CubicSymIBZ", abz.CubicSymIBZ()), ("FBZ", abz.FBZ())):
    bz = abz.load_bz(bzk, A)
    solver = abz.IntegralSolver(abz.FourierIntegrand(abz.DOSIntegrand(), s, 0.1), bz, abz.EvalCounter(abz.AutoPTR()), abstol=1e-3)
    s.device().drop_rules()
    t0 = time.perf_counter(); r = solver.solve_p(abz.MixedParameters(12.5)); t1 = time.perf_counter()
    r2 = solver.solve_p(abz.MixedParameters(12.6)); t2 = time.perf_counter()
    om = np.linspace(10, 15, 256)
    t3 = time.perf_counter(); sw = abz.batchsolve(solver, om); t4 = time.perf_counter()
    print(f"{kind:16s}: cold {1e3*(t1-t0):8.2f} ms  warm {1e3*(t2-t1):7.3f} ms  256-omega sweep {1e3*(t4-t3):8.2f} ms  numevals {r.numevals}  u={r.u:.6f}")
    for npt in (150,):
        for _ in range(2):
            t0 = time.perf_counter(); rule = abz.DeviceRule(s.device(), npt, bz.syms, L.WANT_H); ctx.sync(); t1 = time.perf_counter()
        ctx.prof_enable(True); ctx.prof_reset()
        for _ in range(5): rule.rebuild()
        ctx.sync(); ms, n = ctx.prof_read(L.K_EVAL); cms, cn = ctx.prof_read(L.K_CONTRACT); ctx.prof_enable(False)
        print(f"   rule npt={npt}: nk={rule.nk_local} build {1e3*(t1-t0):.2f} ms; rebuild eval {ms/n:.4f} ms contract {cms/max(cn,1):.4f} ms x{cn//max(n,1)}")
        rule.close()
