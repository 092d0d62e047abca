"""Multi-GPU parameter sweeps: one process per GPU, omega/parameter shards with no data-path
collective, one tiny all_gather of the results (RCCL over xGMI on GPUs, gloo in CPU tests).

ref: batchsolve / batchparam (src/interfaces.jl:199-243): threads take parameters round-robin
(group j gets ps[j], ps[j+n], ...) and every non-primary worker deep-copies the solver.  Here every
rank holds its own replica of the coefficients and of the cached rule (its "deep copy") and solves
ps[rank::world]; messages are <= a few hundred bytes per rank, so the collective is latency-bound
and the per-link xGMI bandwidth never matters.
"""
import numpy as np


def _dist():
    import torch.distributed as dist
    return dist


def world_info(group=None):
    dist = _dist()
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def shard_indices(n, world, rank):
    """Round-robin shard of range(n), identical to batchparam's grouping (src/interfaces.jl:199-208)."""
    return list(range(rank, n, world))


def sharded_map(fn, ps, group=None, device=None):
    """Evaluate fn(list_of_params) -> array [len, ...] on this rank's shard and all_gather.

    Every rank returns the full result array ordered like `ps`.  Values travel as complex128.
    `device`: torch device of the collective buffers ("cuda" for nccl/RCCL, None/"cpu" for gloo)."""
    import torch
    world, rank = world_info(group)
    ps = list(ps)
    n = len(ps)
    idx = shard_indices(n, world, rank)
    local = np.asarray(fn([ps[i] for i in idx])) if idx else np.zeros((0,))
    if world == 1:
        return local
    dist = _dist()
    # agree on the value shape (ranks with an empty shard do not know it)
    tail = list(local.shape[1:]) if len(idx) else []
    meta = [None] * world
    dist.all_gather_object(meta, (len(idx), tail, bool(np.iscomplexobj(local))), group=group)
    tail = next((m[1] for m in meta if m[0] > 0), [])
    is_complex = any(m[2] for m in meta if m[0] > 0)
    per = int(np.prod(tail)) if tail else 1
    maxlen = max(m[0] for m in meta)
    buf = np.zeros((maxlen, per), dtype=np.complex128)
    if len(idx):
        buf[: len(idx)] = np.asarray(local, dtype=np.complex128).reshape(len(idx), per)
    dev = device if device is not None else ("cuda" if dist.get_backend(group) == "nccl" else "cpu")
    t = torch.from_numpy(buf.view(np.float64)).to(dev)
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t, group=group)  # the one collective of the sweep (C1)
    out = np.zeros((n, per), dtype=np.complex128)
    for r, p in enumerate(parts):
        ridx = shard_indices(n, world, r)
        if ridx:
            out[ridx] = p.cpu().numpy().view(np.complex128)[: len(ridx)]
    out = out.reshape([n] + tail)
    return out if is_complex else out.real


def batchsolve_sharded(solver, ps, group=None, device=None, **kw):
    """batchsolve over all ranks of the process group: rank r solves ps[r::world] on its own GPU."""
    from .solver import batchsolve
    return sharded_map(lambda chunk: batchsolve(solver, chunk, **kw), ps, group=group, device=device)


# ------------------------------------------------------------------------ k-sharding of one solve
class kshard:
    """Context manager: shard the *nodes* of every PTR rule of a series over the ranks of a process
    group, for a single solve that is too big (or too urgent) for one GPU.

    ref: SURVEY 8e (2) -- the recursion of fourier_ptr! is independent per outer index
    (src/fourier.jl:148-164), so rank r builds and scans only its slab of the outermost variable of a
    full grid (or its block of the irreducible nodes of a symmetric rule); every rule value
    (abz_rule_reduce, abz_rule_ggr) is then one all-reduce(sum) of a few doubles.  The convergence
    test of AutoPTR runs redundantly on every rank on the same summed numbers, so all ranks take the
    same decisions.  IAI is not k-sharded (its panels are sharded by omega instead).

        with kshard(h, group):                      # h: FourierSeries (or its DeviceSeries)
            u = solver(omega)                       # PTR / AutoPTR / GGR as usual, same value on all ranks
    """

    def __init__(self, series, group=None, device=None, force=False):
        self.dev = series.device() if hasattr(series, "device") else series
        self.group = group
        self.device = device
        self.force = force  # install the hook at world size 1 too (rehearsal of the transport on one GPU)

    def _allreduce(self, a):
        import torch
        dist = _dist()
        a = np.ascontiguousarray(a, dtype=np.float64)
        dev = self.device if self.device is not None else ("cuda" if dist.get_backend(self.group) == "nccl" else "cpu")
        t = torch.from_numpy(a.reshape(-1).copy()).to(dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)  # the one collective per rule value
        return t.cpu().numpy().reshape(a.shape)

    def __enter__(self):
        world, rank = world_info(self.group)
        self._saved = (self.dev.kshard, self.dev.allreduce)
        if world > 1:
            self.dev.kshard = (rank, world)
            self.dev.allreduce = self._allreduce
        return self

    def __exit__(self, *exc):
        self.dev.kshard, self.dev.allreduce = self._saved
        return False


# ------------------------------------------------------------------------ one IAI solve on several GPUs
class iaishard:
    """Context manager: ONE IAI / NestedQuad solve of a series sharded over the ranks of a process group.

    ref: SURVEY 8e (2) -- "disjoint sets of (inner) panels in IAI, one collective per refinement round".  Every rank calls
    the solver with the same arguments; the driver deals the innermost integrals of every round to the ranks (blocks of 64
    nodes), each rank integrates its share on its own GPU and one all-gather per round (this object's callback, handed to
    the library through abz_iai_set_exchange) gives every rank all values.  All ranks return the same result, bit-identical
    to the single-GPU solve.

        with iaishard(h, group):                    # h: FourierSeries (or its DeviceSeries)
            u = solver(omega)                       # IAI() as usual
    """

    def __init__(self, series, group=None, device=None, force=False):
        self.dev = series.device() if hasattr(series, "device") else series
        self.group = group
        self.device = device
        self.force = force  # install the hook at world size 1 too (rehearsal of the transport on one GPU)
        self._cb = None
        self.rounds = 0

    def _exchange(self, user, buf, per):
        try:
            import ctypes as C
            import torch
            dist = _dist()
            world, rank = world_info(self.group)
            arr = np.ctypeslib.as_array(buf, shape=(world * per,))
            t = torch.from_numpy(arr)  # shares the library's buffer
            mine = t[rank * per:(rank + 1) * per]
            dev = self.device if self.device is not None else ("cuda" if dist.get_backend(self.group) == "nccl" else "cpu")
            if str(dev).startswith("cpu"):
                parts = [t[r * per:(r + 1) * per] for r in range(world)]
                dist.all_gather(parts, mine.clone(), group=self.group)
            else:
                out = torch.empty(world * per, dtype=torch.float64, device=dev)
                dist.all_gather_into_tensor(out, mine.to(dev), group=self.group)
                t.copy_(out)  # the library's buffer is pinned: a direct DMA
            self.rounds += 1
            return 0
        except Exception:  # never let an exception cross the C boundary
            import traceback
            traceback.print_exc()
            return 1

    def __enter__(self):
        from . import _lib as L
        world, rank = world_info(self.group)
        if world > 1 or (self.force and _dist().is_initialized()):
            self._cb = L.EXCHANGE_FN(self._exchange)
            L.check(L.lib().abz_iai_set_exchange(self.dev.h, self._cb, None, rank, world))
            self.dev.iai_exchange = True  # sweeps on this series stay on one lane (the hook lives on this handle)
        return self

    def __exit__(self, *exc):
        from . import _lib as L
        if self._cb is not None:
            L.check(L.lib().abz_iai_set_exchange(self.dev.h, None, None, 0, 1))
            self.dev.iai_exchange = False
            self._cb = None
        return False
