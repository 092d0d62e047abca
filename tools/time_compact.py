"""Rule builds with H(k) in the reference's full SMatrix layout against the Hermitian-compact layout (upper triangle,
ABZ_WANT_H_COMPACT): Fourier-eval kernel and whole rebuild on several buffers of one process, and the 32-omega DOS scan
of each.  Usage: time_compact.py [npt ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
L = abz._lib
s = abz.load_w90_series(os.path.join(ROOT, "tests", "golden", "svo_hr.dat.gz"))
dev = s.device(); ctx = dev.ctx
om = np.linspace(10, 15, 32)
for npt in [int(v) for v in sys.argv[1:]] or [150]:
    nk = npt ** 3
    for rep in range(2):
        keep = []
        for name, want in (("full   ", L.WANT_H | L.WANT_EIG), ("compact", L.WANT_H | L.WANT_EIG | L.WANT_H_COMPACT),
                           ("eig    ", L.WANT_EIG)):
            rule = abz.DeviceRule(dev, npt, None, want)
            nb = rule.values_ptr()[1]
            for _ in range(30): rule.rebuild()
            ctx.sync()
            ctx.prof_enable(True, kernels=[L.K_EVAL]); ctx.prof_reset()
            t0 = time.perf_counter()
            reps = 200 if npt < 300 else 40
            for _ in range(reps): rule.rebuild()
            ctx.sync()
            wall = (time.perf_counter() - t0) / reps
            ms, n = ctx.prof_read(L.K_EVAL); ctx.prof_enable(False)
            line = f"npt {npt} {name}: {nb/1e6:7.1f} MB  eval kernel {ms/n:.4f} ms = {nb/(ms/n*1e-3)/1e12:.2f} TB/s written, rebuild {1e3*wall:.4f} ms = {nk/wall/1e9:.1f} G k/s"
            if want & L.WANT_H:
                rule.reduce(L.F_DOS, [0.1], om)
                ctx.prof_enable(True, kernels=[L.K_REDUCE]); ctx.prof_reset()
                for _ in range(20): out = rule.reduce(L.F_DOS, [0.1], om)
                ms2, n2 = ctx.prof_read(L.K_REDUCE); ctx.prof_enable(False)
                line += f"; 32-omega scan {ms2/n2:.4f} ms checksum {np.abs(out).sum():.15e}"
            print(line, flush=True)
            keep.append(rule)
        for r in keep: r.close()
