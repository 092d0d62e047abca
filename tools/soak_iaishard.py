"""Soak of one-IAI-solve-over-several-ranks (run under torch.distributed.run; ranks may share one GPU, collectives through
gloo then): random series, zones, integrands, tolerances and world sizes 2..4 -- the sharded solve must equal the
unsharded one bit for bit on every rank.
    python -m torch.distributed.run --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29533 tools/soak_iaishard.py 8"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist
import autobzcore.jl_amd as abz

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
nseed = int(sys.argv[1]) if len(sys.argv) > 1 else 6
t0 = time.time()
for seed in range(nseed):
    rng = np.random.default_rng(4000 + seed)  # the same on every rank
    d = int(rng.choice([2, 3]))
    n = int(rng.choice([1, 3, 4, 6, 16]))
    dims = (3,) * d
    c = rng.standard_normal(dims + (n, n)) + 1j * rng.standard_normal(dims + (n, n))
    flip = c[tuple(slice(None, None, -1) for _ in dims)]
    c = 0.5 * (c + np.conj(np.swapaxes(flip, -1, -2))) / max(1.0, n / 2)
    s = abz.FourierSeries(c, period=1.0, first=(-1,) * d, ndim=d)
    bz = abz.load_bz([abz.FBZ(), abz.InversionSymIBZ(), abz.CubicSymIBZ()][seed % 3], np.eye(d))
    integ = abz.DOSIntegrand() if seed % 2 else abz.TrGlocIntegrand()
    eta = float(rng.choice([0.1, 0.3]))
    abstol = (0.3 if n == 16 else 10 ** float(rng.uniform(-3, -1.5)))
    prob = abz.IntegralProblem(abz.FourierIntegrand(integ, s, eta), bz, abz.MixedParameters(float(rng.uniform(-1, 1))))
    alone = abz.solve(prob, abz.EvalCounter(abz.IAI()), abstol=abstol, reltol=0.0)
    with abz.iaishard(s) as sh:
        both = abz.solve(prob, abz.EvalCounter(abz.IAI()), abstol=abstol, reltol=0.0)
    same = both.u == alone.u and both.resid == alone.resid and both.numevals == alone.numevals
    got = [None] * world
    dist.all_gather_object(got, (complex(both.u), both.numevals, bool(same)))
    ok = all(g == got[0] for g in got) and got[0][2]
    if rank == 0:
        print(f"seed {seed}: d={d} n={n} {type(integ).__name__} abstol={abstol:.1e}: numevals {both.numevals}, {sh.rounds} exchanges, "
              f"{'identical' if ok else 'DIFFERENT'}", flush=True)
    assert ok, (rank, both, alone)
if rank == 0:
    print(f"soak done on {world} ranks in {time.time() - t0:.1f} s", flush=True)
dist.barrier()
dist.destroy_process_group()
