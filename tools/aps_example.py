"""The reference's demo (aps_example/aps_example.jl) end to end on the MI355X: SrVO3 3-band DOS on the
cubic IBZ, eta = 10 meV, IAI(abstol 1e-3) and PTR(npt=100), each wrapped in an adaptive Chebyshev
interpolation over omega in [10, 15] eV at atol 1e-2, then evaluated on 10:eta/100:15 (50 001 points).
The reference README quotes "about 5 minutes" on a laptop for this script (incl. Julia compilation and
plotting; unspecified hardware) -- not a like-for-like number, recorded here for orientation only."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import autobzcore.jl_amd as abz

t_all = time.perf_counter()
h = abz.load_w90_series(os.path.join(ROOT, "tests", "golden", "svo_hr.dat.gz"))
bz = abz.load_bz(abz.CubicSymIBZ(), os.path.join(ROOT, "tests", "golden", "svo.wout.gz"))
eta = 1e-2
integrand = abz.FourierIntegrand(abz.DOSIntegrand(), h, eta)
out = {}
grid = np.arange(10.0, 15.0 + 1e-12, eta / 100)
for name, alg in (("iai", abz.IAI()), ("ptr", abz.PTR(npt=100))):
    solver = abz.IntegralSolver(integrand, bz, alg, abstol=1e-3)
    solver(12.0)  # warm-up: context, allocations, first rule
    t0 = time.perf_counter()
    itp = abz.hchebinterp(solver, 10, 15, atol=1e-2, batch=lambda xs: abz.batchsolve(solver, xs))
    t1 = time.perf_counter()
    dos = itp(grid)
    t2 = time.perf_counter()
    out[name] = {"solver_evals": itp.numevals, "panels": len(itp.panels), "interp_seconds": t1 - t0,
                 "eval_50001_points_seconds": t2 - t1, "dos_min": float(dos.min()), "dos_max": float(dos.max()),
                 "dos_integral_over_10_15": float(np.trapezoid(dos, grid))}
    np.save(os.path.join(ROOT, "gpurun_out", f"aps_dos_{name}.npy"), dos)
d = np.load(os.path.join(ROOT, "gpurun_out", "aps_dos_iai.npy")) - np.load(os.path.join(ROOT, "gpurun_out", "aps_dos_ptr.npy"))
out["max_abs_diff_iai_vs_ptr"] = float(np.abs(d).max())
out["total_seconds_incl_io"] = time.perf_counter() - t_all
out["det_B_times_6"] = float(abs(np.linalg.det(bz.B)) * 6)
print(json.dumps(out, indent=1))
