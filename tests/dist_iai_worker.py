"""Worker of test_single_iai_solve_sharded_over_two_ranks (run under torch.distributed.run, 2 ranks sharing the one GPU of
the test box, collectives through gloo): a 3-D IAI solve sharded with `iaishard` equals the unsharded solve bit for bit
on both ranks, and the exchange was really used."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist

import autobzcore.jl_amd as abz

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
rng = np.random.default_rng(77)
for n, dims, eta, abstol in ((3, (5, 5, 5), 0.1, 1e-3), (16, (3, 3, 3), 0.3, 0.5)):
    c = rng.standard_normal(dims + (n, n)) + 1j * rng.standard_normal(dims + (n, n))
    c = 0.5 * (c + np.conj(np.swapaxes(c[::-1, ::-1, ::-1], -1, -2))) / max(1.0, n / 2)
    s = abz.FourierSeries(c, period=1.0, first=tuple(-(m // 2) for m in dims), ndim=3)
    f = abz.FourierIntegrand(abz.DOSIntegrand(), s, eta)
    prob = abz.IntegralProblem(f, abz.load_bz(abz.FBZ(), np.eye(3)), abz.MixedParameters(0.1))
    alone = abz.solve(prob, abz.EvalCounter(abz.IAI()), abstol=abstol, reltol=0.0)
    with abz.iaishard(s) as sh:
        both = abz.solve(prob, abz.EvalCounter(abz.IAI()), abstol=abstol, reltol=0.0)
    assert sh.rounds > 0, "the exchange was never called"
    assert both.u == alone.u and both.resid == alone.resid and both.numevals == alone.numevals, (rank, both, alone)
    got = [None] * world
    dist.all_gather_object(got, (both.u, both.numevals))
    assert all(g == got[0] for g in got)
    if rank == 0:
        print(f"sharded IAI ok: n = {n}, u = {both.u!r}, numevals = {both.numevals}, {sh.rounds} exchanges", flush=True)
dist.barrier()
dist.destroy_process_group()
