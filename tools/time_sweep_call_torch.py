"""abz_rule_reduce per call on a library-owned stream against a context on a stream borrowed from torch (what bench.py uses)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import autobzcore.jl_amd as abz
from autobzcore.jl_amd import _lib as L
s = abz.load_w90_series(os.path.join(ROOT, "tests", "golden", "svo_hr.dat.gz"))
om = np.linspace(10, 15, 32)
st = torch.cuda.Stream(device=0)
for name, ctx in (("own stream", abz.Context(0)), ("torch stream", abz.Context(0, stream=st.cuda_stream))):
    dev = s.device(ctx)
    rule = dev.rule(150, None, L.WANT_H | L.WANT_EIG)
    for _ in range(20): out = rule.reduce(L.F_DOS, [0.1], om)
    reps = 300
    t0 = time.perf_counter()
    for _ in range(reps): out = rule.reduce(L.F_DOS, [0.1], om)
    wall = (time.perf_counter() - t0) / reps
    ctx.prof_enable(True, kernels=[L.K_REDUCE]); ctx.prof_reset()
    t0 = time.perf_counter()
    for _ in range(reps): rule.reduce(L.F_DOS, [0.1], om)
    wallp = (time.perf_counter() - t0) / reps
    ms, n = ctx.prof_read(L.K_REDUCE); ctx.prof_enable(False)
    print(f"{name:13s}: call {1e3*wall:.4f} ms ({32/wall:.0f} DOS pts/s), with event profiling {1e3*wallp:.4f} ms, kernels {ms/n:.4f} ms", flush=True)
