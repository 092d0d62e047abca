"""Rebuild pass of the SVO rule with the last contraction as its own launch against ABZ_FUSE2=1 (the wave contracts its
lines from the block's LDS copy of the level-2 set; packed Hermitian sets in both), per grid size, on one rule buffer."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
L = abz._lib
s = abz.load_w90_series(os.path.join(ROOT, "tests", "golden", "svo_hr.dat.gz"))
dev = s.device(); ctx = dev.ctx
om = np.linspace(10, 15, 8)
for npt in [int(v) for v in sys.argv[1:]] or [150]:
    rule = dev.rule(npt, None, L.WANT_H | L.WANT_EIG)
    line = f"npt {npt}:"
    for fuse in ("0", "1", "0", "1"):
        os.environ["ABZ_FUSE2"] = fuse
        for _ in range(100): rule.rebuild()
        ctx.sync()
        reps = 1500 if npt < 200 else 300
        t0 = time.perf_counter()
        for _ in range(reps): rule.rebuild()
        ctx.sync()
        dt = (time.perf_counter() - t0) / reps
        chk = np.abs(rule.reduce(L.F_DOS, [0.1], om)).sum()
        line += f"  FUSE2={fuse} {1e3*dt:.4f} ms ({npt**3/dt/1e9:.1f} G k/s, checksum {chk:.12e})"
    print(line, flush=True)
