"""Minimal driver for profiling: K rebuilds of the SVO npt^3 rule (+ optionally a sweep).
Usage: python3 tools/prof_eval.py [npt] [steps] [sweep]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import autobzcore.jl_amd as abz
from autobzcore.jl_amd import _lib as L

npt = int(sys.argv[1]) if len(sys.argv) > 1 else 150
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
sweep = int(sys.argv[3]) if len(sys.argv) > 3 else 0
s = abz.load_w90_series(os.path.join(ROOT, "tests", "golden", "svo_hr.dat.gz"))
dev = s.device()
rule = dev.rule(npt, None, L.WANT_H | L.WANT_EIG)
for _ in range(steps):
    rule.rebuild()
dev.ctx.sync()
if sweep:
    om = np.linspace(10, 15, sweep)
    for _ in range(steps):
        rule.reduce(L.F_DOS, [0.1], om)
        rule.reduce(L.F_DOS_EIG, [0.1], om)
print("done")
