"""One table against the number of bands (24^3 grid, 5^3 coefficients): where a band count costs more than its neighbours.
Rule builds (H, H + eig, eig, GGR), scans of the cached rule (G, tr G, DOS from H, DOS from eigenvalues; 8 omega), a store-free
8-omega sum, the rule of the cubic IBZ node list (H + eig), and abz_eval_nodes on 4096 nodes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
from autobzcore.jl_amd import _lib as L
npt = 24
bands = [int(a) for a in sys.argv[1:]] or [3, 4, 5, 8, 9, 16, 17, 24, 32, 33, 48, 64]
om = np.linspace(-1, 1, 8)


def best(f, reps=3):
    f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return 1e3 * min(ts)


print("n    H      H+eig  eig    GGR    | scan G   trG    DOS    DOSeig | sum8   | cubicIBZ H+eig | eval_nodes(4096, H+eig)")
for n in bands:
    s = abz.synthetic_wannier(n=n, rmax=2, seed=7)
    dev = s.device(); ctx = dev.ctx
    row = [f"{n:2d}"]
    for want in (L.WANT_H, L.WANT_H | L.WANT_EIG, L.WANT_EIG, L.WANT_EIG | L.WANT_VEL):
        try:
            r = abz.DeviceRule(dev, npt, None, want); ctx.sync()
            row.append(f"{best(lambda: (r.rebuild(), ctx.sync())):6.3f}")
            r.close()
        except Exception as e:
            row.append("  n/a ")
    row.append("|")
    r = abz.DeviceRule(dev, npt, None, L.WANT_H | L.WANT_EIG); ctx.sync()
    for fid in (L.F_GLOC, L.F_TRGLOC, L.F_DOS, L.F_DOS_EIG):
        try:
            row.append(f"{best(lambda: r.reduce(fid, [0.05], om)):6.3f}")
        except Exception as e:
            row.append("  n/a ")
    r.close()
    row.append("|")
    try:
        row.append(f"{best(lambda: dev.ptr_sum(npt, L.F_DOS, [0.05], om)):6.3f}")
    except Exception:
        row.append("  n/a ")
    row.append("|")
    try:
        bz = abz.load_bz(abz.CubicSymIBZ(), np.eye(3))
        rs = dev.rule(npt, bz.syms, want=L.WANT_H | L.WANT_EIG); ctx.sync()
        row.append(f"{best(lambda: (rs.rebuild(), ctx.sync())):6.3f} ({rs.nk} nodes)")
        rs.close()
    except Exception as e:
        row.append("  n/a " + repr(e)[:40])
    row.append("|")
    k = np.random.default_rng(1).uniform(0, 1, size=(4096, 3))
    try:
        row.append(f"{best(lambda: dev.eval_nodes(k, want=3)):6.3f}")
    except Exception as e:
        row.append("  n/a " + repr(e)[:40])
    print("  ".join(row), flush=True)
