import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
s = abz.load_w90_series(os.path.join(ROOT, "tests", "golden", "svo_hr.dat.gz"))
dev = s.device()
rng = np.random.default_rng(0)
for nk in (1000, 100000, 2000000):
    k = rng.random((nk, 3))
    dev.eval_nodes(k[:10])
    t0 = time.perf_counter(); H = dev.eval_nodes(k); t1 = time.perf_counter()
    print(f"abz_eval_nodes nk={nk}: {1e3*(t1-t0):8.2f} ms  {nk/(t1-t0)/1e6:7.2f} M nodes/s  ({(24+144)*nk/(t1-t0)/1e9:.2f} GB/s over PCIe)")
for nk in (15, 30, 1000):
    k = rng.random((nk, 3)); k[:, 1:] = k[0, 1:]  # one GK panel: shared outer coordinates
    dev.eval_nodes(k)
    t0 = time.perf_counter()
    for _ in range(20): dev.eval_nodes(k)
    t1 = time.perf_counter()
    print(f"panel-like batch nk={nk}: {1e3*(t1-t0)/20:8.3f} ms per call")
ctx = dev.ctx
from autobzcore.jl_amd import _lib as L
k = rng.random((1000, 3))
ctx.prof_enable(True); ctx.prof_reset()
t0 = time.perf_counter(); dev.eval_nodes(k); t1 = time.perf_counter()
for name, kid in (("contract", L.K_CONTRACT), ("eval", L.K_EVAL)):
    ms, n = ctx.prof_read(kid); print(f"  {name}: {n} launches {ms:.3f} ms")
print(f"  wall {1e3*(t1-t0):.3f} ms")
