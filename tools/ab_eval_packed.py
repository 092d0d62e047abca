"""A/B of the packed (Hermitian +-f folded) and the plain grid-evaluation chain on the SAME rule buffer of one process
(placement of the buffer moves the kernel by up to 10 % between processes).  Usage: ab_eval_packed.py [npt ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
L = abz._lib
s = abz.load_w90_series(os.path.join(ROOT, "tests", "golden", "svo_hr.dat.gz"))
dev = s.device(); ctx = dev.ctx
for npt in [int(v) for v in (sys.argv[1:] or ["100", "150", "160", "200", "250"])]:
    for want, name in ((L.WANT_H | L.WANT_EIG, "H+eig"), (L.WANT_EIG, "eig")):
        rule = abz.DeviceRule(dev, npt, None, want)
        row = []
        for rnd in range(3):
            for pk in ("1", "0"):
                os.environ["ABZ_EVAL_PACKED"] = pk
                for _ in range(10): rule.rebuild()
                ctx.sync()
                ctx.prof_enable(True, kernels=[L.K_EVAL, L.K_CONTRACT]); ctx.prof_reset()
                for _ in range(100): rule.rebuild()
                ctx.sync()
                (ms, n), (mc, nc) = ctx.prof_read(L.K_EVAL), ctx.prof_read(L.K_CONTRACT); ctx.prof_enable(False)
                row.append((pk, ms / n, mc / 100))
        os.environ.pop("ABZ_EVAL_PACKED")
        pk_e = min(t for p, t, c in row if p == "1"); pl_e = min(t for p, t, c in row if p == "0")
        pk_c = min(c for p, t, c in row if p == "1"); pl_c = min(c for p, t, c in row if p == "0")
        print(f"npt {npt} {name:6s}: eval packed {pk_e:.4f} plain {pl_e:.4f} ms ({100*(pl_e-pk_e)/pl_e:+.1f} %)   contractions packed {pk_c:.4f} plain {pl_c:.4f} ms   "
              f"rebuild packed {pk_e+pk_c:.4f} plain {pl_e+pl_c:.4f}", flush=True)
        rule.close()
