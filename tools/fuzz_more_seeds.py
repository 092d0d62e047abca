"""The round-5 fuzz (tests/test_gpu_fuzz.py::fuzz_round5_kernels) over more seeds than the suite runs: python tools/fuzz_more_seeds.py [first [count]]"""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import autobzcore.jl_amd as abz
import test_gpu_fuzz as F
first = int(sys.argv[1]) if len(sys.argv) > 1 else 503
count = int(sys.argv[2]) if len(sys.argv) > 2 else 12
which = sys.argv[3] if len(sys.argv) > 3 else "round5"  # round5 | small | many | all
for seed in range(first, first + count):
    if which in ("round5", "all"):
        worst, bad = F.fuzz_round5_kernels(abz, seed)
        print("round5", seed, worst, bad[:5], flush=True)
    if which in ("small", "all"):
        worst, bad = F.fuzz_small_band_rules(abz, seed)
        print("small ", seed, worst, bad[:5], flush=True)
    if which in ("many", "all"):
        worst, bad = F.fuzz_many_band_rules_iai_and_symmetric_rules(abz, seed, quick=True)
        print("many  ", seed, worst, bad[:5], flush=True)
