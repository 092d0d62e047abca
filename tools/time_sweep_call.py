"""Phase B per CALL: wall time of abz_rule_reduce (host sweep values in, sums out) against the HIP-event time of its
kernels, for the bench's 32-omega sweep and others.  The gap is launch + copy + synchronisation overhead."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
from autobzcore.jl_amd import _lib as L
s = abz.load_w90_series(os.path.join(ROOT, "tests", "golden", "svo_hr.dat.gz"))
dev = s.device(); ctx = dev.ctx
npt = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rule = dev.rule(npt, None, L.WANT_H | L.WANT_EIG)
for fid, name in ((L.F_DOS, "DOS (matrix-cached)"), (L.F_DOS_EIG, "DOS (eig-cached)")):
    for nw in (1, 32, 256):
        om = np.linspace(10, 15, nw)
        for _ in range(10): out = rule.reduce(fid, [0.1], om)
        reps = 200
        t0 = time.perf_counter()
        for _ in range(reps): out = rule.reduce(fid, [0.1], om)
        wall = (time.perf_counter() - t0) / reps
        ctx.prof_enable(True, kernels=[L.K_REDUCE]); ctx.prof_reset()
        for _ in range(50): rule.reduce(fid, [0.1], om)
        ms, n = ctx.prof_read(L.K_REDUCE); ctx.prof_enable(False)
        print(f"{name:22s} npt={npt} n_omega={nw:4d}: call {1e3*wall:7.4f} ms  kernels {ms/n:7.4f} ms  overhead {1e3*wall-ms/n:7.4f} ms  "
              f"{nw/wall:10.0f} DOS pts/s  checksum {np.abs(out).sum():.12e}", flush=True)
