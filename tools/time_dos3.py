"""3-band DOS sweeps over a cached rule: the sweep kernel (dos3_scan_kernel) against the generic scan (ABZ_DOS3_SCAN=0),
per number of sweep rows (ABZ_REDUCE_ROWS), on one rule buffer.  Kernel time by the
library's HIP events (reduce + final reduce)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
from autobzcore.jl_amd import _lib as L

npt = int(sys.argv[1]) if len(sys.argv) > 1 else 150
s = abz.load_w90_series(os.path.join(ROOT, "tests", "golden", "svo_hr.dat.gz"))
dev = s.device(); ctx = dev.ctx
rule = dev.rule(npt, None, L.WANT_H | L.WANT_EIG)


def t(fid, om, reps=20):
    rule.reduce(fid, [0.1], om)
    ctx.prof_enable(True, kernels=[L.K_REDUCE]); ctx.prof_reset()
    for _ in range(reps): out = rule.reduce(fid, [0.1], om)
    ms, n = ctx.prof_read(L.K_REDUCE); ctx.prof_enable(False)
    return ms / n, out


for fid, name in ((L.F_DOS, "DOS (matrix-cached)"), (L.F_DOS_EIG, "DOS (eig-cached)")):
    for nw in (1, 8, 32, 256):
        om = np.linspace(10, 15, nw)
        os.environ["ABZ_DOS3_SCAN"] = "0"
        t0, ref = t(fid, om)
        del os.environ["ABZ_DOS3_SCAN"]
        t1, out = t(fid, om)
        line = f"{name:22s} npt={npt} n_omega={nw:4d}: generic {t0:7.4f} ms  sweep kernel {t1:7.4f} ms ({npt**3*nw/(t1*1e-3)/1e9:7.1f} G (k,w)/s)  rel.diff {np.abs(out-ref).max()/np.abs(ref).max():.1e} |"
        if nw >= 32:
            for rows in (1, 2, 4):
                os.environ["ABZ_REDUCE_ROWS"] = str(rows)
                tk, _ = t(fid, om, 10)
                line += f" rows{rows} {tk:.4f}"
            del os.environ["ABZ_REDUCE_ROWS"]
        print(line, flush=True)
