"""Config 5 as ONE IAI solve sharded over the ranks of this job (run under torch.distributed.run; on a one-GPU box the
ranks share the card and the collectives go through gloo), with the library's host statistics on (ABZ_IAI_STATS=1): what
every rank spends on the host per phase -- the describe / deliver work that is divided by the number of ranks against the
bookkeeping every rank repeats.
    ABZ_IAI_STATS=1 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 tools/c5_shard_rehearsal.py [abstol]"""
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
abstol = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-3
s16 = abz.synthetic_wannier()
prob = abz.IntegralProblem(abz.FourierIntegrand(abz.DOSIntegrand(), s16, 0.05), abz.load_bz(abz.FBZ(), np.eye(3)), abz.MixedParameters(0.2))
abz.solve(prob, abz.IAI(), abstol=10.0, reltol=0.0)
dist.barrier()
with abz.iaishard(s16) as sh:
    t0 = time.perf_counter()
    sol = abz.solve(prob, abz.EvalCounter(abz.IAI()), abstol=abstol, reltol=0.0)
    dt = time.perf_counter() - t0
print(f"rank {rank}/{world}: u = {sol.u!r} numevals {sol.numevals} in {dt:.3f} s, {sh.rounds} exchanges", flush=True)
dist.barrier()
dist.destroy_process_group()
