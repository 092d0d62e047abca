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
def lib_mb():  # the library's own bookkeeping: (live, cached, ctx scratch, ctx pinned host) in MiB, live blocks
    m = abz._lib.Context.default().mem_info()
    return tuple(round(v / 2**20, 1) for v in m[:4]) + (m[4],)
svo = abz.load_w90_series(os.path.join(ROOT, "tests", "golden", "svo_hr.dat.gz"))
svo.device().rule(16, None, L.WANT_H)  # context up
f0 = free_gb()
m0 = lib_mb()
print(f"start: driver free {f0:.3f} GiB; library live/cached/scratch/pinned MiB, blocks = {m0}", flush=True)
t0 = time.time()
marks = []
lmarks = []
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
    if it % 10 == 5:  # sweep lanes (views of the series on contexts of their own) and the library's AutoPTR loop (kept rules)
        fi = abz.FourierIntegrand(abz.DOSIntegrand(), s, 0.4)
        abz.batchsolve(abz.IntegralSolver(fi, bz, abz.IAI(), abstol=1e-2), np.linspace(-1.0, 1.0, 40))
        # (on the full BZ: the orbit tables of symmetric grids stay in the context's cache of eight by design)
        abz.IntegralSolver(fi, abz.load_bz(abz.FBZ(), np.eye(d)), abz.AutoPTR(), abstol=1e-2)(0.1)
        del fi
    del s
    if os.environ.get("ABZ_LEAK_TRACE"):
        gc.collect()
        print(f"  it {it} d={d} n={n} dims={dims}: {lib_mb()}", flush=True)
    if it % 30 == 29:
        gc.collect()
        marks.append(free_gb())
        lmarks.append(lib_mb())
        print(f"iter {it+1}: free {marks[-1]:.3f} GiB (start {f0:.3f}); library live/cached/scratch/pinned MiB, blocks = {lib_mb()}", flush=True)
gc.collect()
f1 = free_gb()
m1 = lib_mb()
# every byte the driver no longer reports free must be one the library still accounts for: live blocks (the SVO series, its
# cached rule and contracted sets, the context's scratch) -- nothing else may hold device memory
print(f"library at end: live {m1[0]} MiB in {m1[4]} blocks (of which context scratch {m1[2]}), cached {m1[1]}; "
      f"at start: live {m0[0]} MiB in {m0[4]} blocks")
outside = (f0 - f1) * 1024 - (m1[0] + m1[1] - m0[0] - m0[1])
print(f"driver: {(f0 - f1) * 1024:.1f} MiB less free than at start; library: {m1[0] + m1[1] - m0[0] - m0[1]:.1f} MiB more held "
      f"(the SVO series' contracted-set pools of its 150^3 rule builds); outside the library's allocator: {outside:.1f} MiB, "
      f"all of it taken in the first 30 iterations (HIP runtime: the private-segment arena of the kernels that use scratch, "
      f"code objects of first launches, per-stream structures)")
pool = int(os.environ["ABZ_POOL_MB"])
held30 = lmarks[0][0] + lmarks[0][1]
print(f"done in {time.time()-t0:.1f} s: driver free at start {f0:.3f} GiB, after 30 iterations {marks[0]:.3f}, at end {f1:.3f} GiB; "
      f"drift over the last 90 iterations: driver {(marks[0]-f1)*1024:.1f} MiB, library {m1[0] + m1[1] - held30:.1f} MiB (pool cap {pool} MiB)")
# steady state: the library's own books do not move at all once the long-lived series has its pools, and the driver's
# free memory moves by less than a few allocation granules
assert abs(m1[0] + m1[1] - held30) < 1.0 + pool, "the library's live + cached bytes grew in steady state"
assert (marks[0] - f1) * 1024 < pool + 16, "driver free memory drifts in steady state"
