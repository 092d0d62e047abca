"""Device-memory leak check: many series / rules / solves (IAI-heavy: the per-series staging buffers) created and
dropped; free memory must come back.  Run with the allocator cache off (ABZ_POOL_MB=0, set below unless given) so that
a leak of a few tens of MB is visible."""
import gc, os, sys, time
os.environ.setdefault("ABZ_POOL_MB", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import autobzcore.jl_amd as abz
L = abz._lib
rng = np.random.default_rng(0)
def free_gb():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2**30
svo = abz.load_w90_series(os.path.join(ROOT, "tests", "golden", "svo_hr.dat.gz"))
svo.device().rule(16, None, L.WANT_H)  # context up
f0 = free_gb()
t0 = time.time()
marks = []
for it in range(120):
    n = int(rng.integers(1, 5)); d = int(rng.integers(1, 4))
    dims = tuple(int(rng.choice([3, 5])) for _ in range(d))
    c = rng.standard_normal(dims + (n, n)) + 1j * rng.standard_normal(dims + (n, n))
    flip = c[tuple(slice(None, None, -1) for _ in dims)]
    c = 0.5 * (c + np.conj(np.swapaxes(flip, -1, -2)))
    s = abz.FourierSeries(c, period=1.0, first=tuple(-(m // 2) for m in dims), ndim=d)
    bz = abz.load_bz(abz.FBZ() if it % 2 else abz.InversionSymIBZ(), np.eye(d))
    for alg in (abz.PTR(npt=int(rng.integers(8, 60))), abz.IAI()):
        abz.IntegralSolver(abz.FourierIntegrand(abz.DOSIntegrand(), s, 0.4), bz, alg, abstol=1e-2)(0.1)
    if it % 10 == 0:
        r = abz.DeviceRule(svo.device(), 150, None, 3); r.rebuild(); r.close()
    del s
    if it % 30 == 29:
        gc.collect()
        marks.append(free_gb())
        print(f"iter {it+1}: free {marks[-1]:.3f} GiB (start {f0:.3f})", flush=True)
gc.collect()
f1 = free_gb()
pool = int(os.environ["ABZ_POOL_MB"])
print(f"done in {time.time()-t0:.1f} s: free at start {f0:.3f} GiB, after 30 iterations {marks[0]:.3f}, at end {f1:.3f} GiB; "
      f"drift over the last 90 iterations {(marks[0]-f1)*1024:.1f} MiB (pool cap {pool} MiB; the context's grow-only scratch "
      f"accounts for the first {(f0-marks[0])*1024:.0f} MiB)")
assert (marks[0] - f1) * 1024 < pool + 32  # steady state: nothing may accumulate once the scratch buffers have their size
