"""GGR build (eigenvalues + band velocities, abz_ptr_rule_build(WANT_EIG | WANT_VEL)) and energy scan on the SVO model:
time per rebuild from the library's own HIP events, checksums of (e, v) and of the scanned DOS so that two builds of the
library (ABZ_GGR_FUSED=0: the unfused round-2 path) can be compared value for value.
Usage: time_ggr.py [npt ...]      env: ABZ_GGR_FUSED, ABZ_GGR_FUSE2, SYMS=1 (cubic IBZ too)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
from autobzcore.jl_amd import _lib as L

s = abz.load_w90_series(os.path.join(ROOT, "tests", "golden", "svo_hr.dat.gz"))
dev = s.device(); ctx = dev.ctx
Es = np.linspace(10, 15, 256)
npts = [int(v) for v in sys.argv[1:]] or [50, 100, 150]
kinds = [("FBZ", None)]
if os.environ.get("SYMS") == "1":
    kinds.append(("cubic", abz.load_bz(abz.CubicSymIBZ(), 3.85856 * np.eye(3)).syms))
for npt in npts:
    for name, syms in kinds:
        t0 = time.perf_counter(); r = abz.DeviceRule(dev, npt, syms, L.WANT_EIG | L.WANT_VEL); ctx.sync(); t1 = time.perf_counter()
        for _ in range(5): r.rebuild()
        ctx.sync()
        ids = [L.K_CONTRACT, L.K_EVAL, L.K_EIG, L.K_GGRBUILD]
        ctx.prof_enable(True, kernels=ids); ctx.prof_reset()
        reps = 50 if npt <= 200 else 10
        t2 = time.perf_counter()
        for _ in range(reps): r.rebuild()
        ctx.sync(); t3 = time.perf_counter()
        parts = {k: ctx.prof_read(k) for k in ids}; ctx.prof_enable(False)
        g = r.ggr(Es); ctx.sync()
        t4 = time.perf_counter()
        for _ in range(10): g = r.ggr(Es)
        t5 = time.perf_counter()
        nk = r.nk_local
        ev = "  ".join(f"k{k}: {ms / reps:.4f} ms/{n // reps}" for k, (ms, n) in parts.items() if n)
        line = (f"GGR {name} npt={npt} nk={nk}: cold build {1e3 * (t1 - t0):7.2f} ms  rebuild {1e3 * (t3 - t2) / reps:7.4f} ms wall "
                f"[{ev}]  {nk * reps / (t3 - t2) / 1e9:.2f} G k/s  {nk * 8 * 3 * 4 * reps / (t3 - t2) / 1e12:.3f} TB/s(alg 96 B/k)  "
                f"scan 256 E {1e3 * (t5 - t4) / 10:.3f} ms")
        if npt <= 100:
            out = r.export(x=False, w=False, eig=True, vel=True)
            line += f"  sum|e|={np.abs(out['eig']).sum():.12e} sum|v|={np.abs(out['vel']).sum():.12e}"
            if os.environ.get("DUMP"):
                np.savez(os.path.join(ROOT, "gpurun_out", f"ggr_{os.environ['DUMP']}_{name}_{npt}.npz"), eig=out["eig"], vel=out["vel"], dos=g)
        line += f"  dos[128]={g[128]:.12f} sum dos={g.sum():.12f}"
        print(line, flush=True)
        r.close()
