"""abz_eval_nodes on 4096 random nodes (H + eigenvalues) for rocprofv3: python tools/prof_eval_nodes.py <bands>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
s = abz.synthetic_wannier(n=n, rmax=2, seed=7)
dev = s.device()
k = np.random.default_rng(1).uniform(0, 1, size=(4096, 3))
for _ in range(3):
    t0 = time.perf_counter(); H, E = dev.eval_nodes(k, want=3); dt = time.perf_counter() - t0
print(f"n={n}: eval_nodes(4096, H + eig) {1e3*dt:.3f} ms")
