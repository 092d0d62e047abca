"""Second randomised parity sweep (5...32 bands, IAI panel kernels, symmetric rules) with a seed of your choice; the cases
are tests/test_gpu_fuzz.py::fuzz_many_band_rules_iai_and_symmetric_rules, which the GPU suite runs with seed 1."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import autobzcore.jl_amd as abz
from test_gpu_fuzz import fuzz_many_band_rules_iai_and_symmetric_rules

t0 = time.time()
worst, bad = fuzz_many_band_rules_iai_and_symmetric_rules(abz, int(sys.argv[1]) if len(sys.argv) > 1 else 1,
                                                          emit=lambda s: print(f"[{time.time() - t0:6.1f} s] {s}", flush=True))
print(f"worst relative error {worst:.2e}, {len(bad)} failing case(s), {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
