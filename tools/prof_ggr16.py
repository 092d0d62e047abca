import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import autobzcore.jl_amd as abz
s = abz.synthetic_wannier(n=16, rmax=3, seed=7)
dev = s.device()
rule = dev.rule(24, None, want=2 | 4)
dev.ctx.sync()
for _ in range(3): rule.rebuild()
dev.ctx.sync()
t0 = time.perf_counter(); rule.rebuild(); dev.ctx.sync(); print("rebuild ms", 1e3*(time.perf_counter()-t0))
