"""Soak of the IAI sweep lanes: random series / widths / zones, sweeps on 4 lanes against the same sweep on one lane
(values, errors and evaluation counts must be identical)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
nseed = int(sys.argv[1]) if len(sys.argv) > 1 else 12
t0 = time.time()
for seed in range(nseed):
    rng = np.random.default_rng(1000 + seed)
    d = int(rng.choice([2, 3]))
    n = int(rng.choice([1, 2, 3, 4, 6]))
    dims = (3,) * d
    c = rng.standard_normal(dims + (n, n)) + 1j * rng.standard_normal(dims + (n, n))
    flip = c[tuple(slice(None, None, -1) for _ in dims)]
    c = 0.5 * (c + np.conj(np.swapaxes(flip, -1, -2))) / np.sqrt(n)
    s = abz.FourierSeries(c, period=1.0, first=(-1,) * d, ndim=d)
    bz = abz.load_bz([abz.FBZ(), abz.InversionSymIBZ(), abz.CubicSymIBZ()][seed % 3], np.eye(d))
    eta = float(rng.choice([0.05, 0.1, 0.3]))
    integ = abz.DOSIntegrand() if seed % 2 else abz.TrGlocIntegrand()
    solver = abz.IntegralSolver(abz.FourierIntegrand(integ, s, eta), bz, abz.EvalCounter(abz.IAI()), abstol=10 ** float(rng.uniform(-3, -1)))
    om = np.sort(rng.uniform(-2.5, 2.5, int(rng.integers(40, 120))))
    res = {}
    for lanes in ("1", "4"):
        os.environ["ABZ_IAI_LANES"] = lanes
        meta = []
        vals = abz.batchsolve(solver, om, callback=lambda sv, i, k, p, sol, t: meta.append((i, sol.resid, sol.numevals)))
        res[lanes] = (np.asarray(vals), sorted(meta))
    ok = np.array_equal(res["1"][0], res["4"][0]) and res["1"][1] == res["4"][1]
    print(f"seed {seed}: d={d} n={n} {type(integ).__name__} eta={eta} {len(om)} solves, {sum(m[2] for m in res['1'][1])} evals: {'identical' if ok else 'DIFFERENT'}", flush=True)
    assert ok
print(f"soak done in {time.time() - t0:.1f} s")
