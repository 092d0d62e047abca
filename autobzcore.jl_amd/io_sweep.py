"""Sweep archive: the data sets of the reference's HDF5 writer with partial-result persistence.

ref: ext/HDF5Ext.jl:123-158 -- `batchsolve(h5, solver, ps)` creates `I, E, t, retcode, numevals` and the
parameter groups `args/<j>`, `kwargs/<name>`, fills them from the solver callback and flushes after every
point.  There is no HDF5 library in this pipeline; the same names and shapes go into a NumPy `.npz`
archive that is rewritten atomically (temp file + rename) after every chunk of the sweep, so an
interrupted run keeps everything solved so far (`done` marks the filled entries)."""
import os
import time

import numpy as np

from .solver import MixedParameters, batchsolve, _to_params


class SweepArchive:
    def __init__(self, path, shape):
        self.path = str(path)
        self.shape = tuple(shape)
        self.I = None  # allocated at the first record (value type / shape of the integral)
        self.E = np.full(self.shape, np.nan)
        self.t = np.full(self.shape, np.nan)
        self.retcode = np.zeros(self.shape, dtype=np.int32)
        self.numevals = np.full(self.shape, -1, dtype=np.int64)
        self.done = np.zeros(self.shape, dtype=bool)
        self.args = {}
        self.kwargs = {}

    def record(self, i, p, sol, t):
        """One solved point: index i (tuple), parameters p, IntegralSolution sol, seconds t."""
        i = tuple(np.atleast_1d(i).tolist()) if not isinstance(i, tuple) else i
        u = np.asarray(sol.u)
        if self.I is None:
            self.I = np.full(self.shape + u.shape, np.nan, dtype=np.complex128 if np.iscomplexobj(u) else np.float64)
        self.I[i] = u
        self.E[i] = np.nan if sol.resid is None else float(np.real(sol.resid))
        self.t[i] = t
        self.retcode[i] = int(bool(sol.retcode))
        self.numevals[i] = int(sol.numevals)
        self.done[i] = True
        mp = _to_params(p)
        for j, e in enumerate(mp.args):
            self.args.setdefault(str(j + 1), np.full(self.shape, np.nan))[i] = e
        for k, v in mp.kwargs.items():
            self.kwargs.setdefault(str(k), np.full(self.shape, np.nan))[i] = v

    def flush(self):
        data = {"E": self.E, "t": self.t, "retcode": self.retcode, "numevals": self.numevals, "done": self.done}
        if self.I is not None:
            data["I"] = self.I
        for k, v in self.args.items():
            data["args/" + k] = v
        for k, v in self.kwargs.items():
            data["kwargs/" + k] = v
        tmp = self.path + ".tmp.npz"
        np.savez(tmp, **data)
        os.replace(tmp, self.path)

    @staticmethod
    def load(path):
        with np.load(path) as z:
            return {k: z[k] for k in z.files}


def batchsolve_archive(path, solver, ps, chunk=64, flush=True, verb=False, solve=batchsolve):
    """batchsolve(h5, solver, ps): solve the sweep chunk by chunk (each chunk is one fused device pass),
    recording every point and rewriting the archive after each chunk.  Returns the array of values."""
    lst = list(ps)
    arch = SweepArchive(path, (len(lst),))
    out = []
    t0 = time.time()
    for c0 in range(0, len(lst), max(1, int(chunk))):
        part = lst[c0:c0 + max(1, int(chunk))]
        vals = solve(solver, part, callback=lambda sv, i, n, p, sol, t, c0=c0: arch.record((c0 + i[0],), p, sol, t))
        out.extend(list(vals))
        if flush:
            arch.flush()
        if verb:
            print(f"{min(c0 + len(part), len(lst)):5d} / {len(lst)} done in {time.time() - t0:e} (s)")
    arch.flush()
    return np.array(out)
