"""Sweep archive: the data sets of the reference's HDF5 writer with partial-result persistence.

ref: ext/HDF5Ext.jl:123-158 -- `batchsolve(h5, solver, ps)` creates `I, E, t, retcode, numevals` and the
parameter groups `args/<j>`, `kwargs/<name>`, fills them from the solver callback and flushes after every
point.

Two on-disk formats, chosen by the file name:
  * `*.h5` / `*.hdf5`: a real HDF5 file through the C library (`h5lite.py`, a ctypes binding; no h5py in this
    interpreter).  Data sets are created up front, every solved point is written in place (a one-element hyperslab,
    the reference's `set_value`), the file is flushed after every chunk of the sweep.  Shapes are the reference's read
    in C order: `I` is `size(ps)..., size(T)...` here = `size(T)..., size(ps)...` in Julia.  Raises if no libhdf5 can be
    loaded.
  * anything else: a NumPy `.npz` archive with the same names, rewritten atomically (temp file + rename) after every chunk.
Either way an interrupted run keeps everything solved so far; the extra data set `done` marks the filled entries."""
import os
import time

import numpy as np

from .solver import MixedParameters, batchsolve, _to_params


def _is_h5(path):
    return str(path).lower().endswith((".h5", ".hdf5", ".hdf"))


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
        self.h5 = None
        if _is_h5(self.path):
            from . import h5lite
            self.h5 = h5lite.File(self.path, "w")  # raises H5Error without a library: no silent change of format
            self._d = {k: self.h5.write_dataset(k, getattr(self, k)) for k in ("E", "t", "retcode", "numevals", "done")}
            self._g = {}
            self.h5.flush()

    def _h5_param(self, group, key, i, e):
        g = self._g.get(group)
        if g is None:
            g = self._g[group] = self.h5.create_group(group)
        name = group + "/" + key
        if name not in self._d:
            init = np.full(self.shape, np.nan) if np.issubdtype(np.asarray(e).dtype, np.floating) else np.zeros(self.shape, np.asarray(e).dtype)
            self._d[name] = g.write_dataset(key, init)
        self._d[name][i] = e

    def record(self, i, p, sol, t):
        """One solved point: index i (tuple), parameters p, IntegralSolution sol, seconds t."""
        i = tuple(np.atleast_1d(i).tolist()) if not isinstance(i, tuple) else i
        u = np.asarray(sol.u)
        if self.I is None:
            self.I = np.full(self.shape + u.shape, np.nan, dtype=np.complex128 if np.iscomplexobj(u) else np.float64)
            if self.h5 is not None:
                self._d["I"] = self.h5.write_dataset("I", self.I)
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
        if self.h5 is not None:
            for k in ("I", "E", "t", "retcode", "numevals", "done"):
                self._d[k][i] = getattr(self, k)[i]
            for j, e in enumerate(mp.args):
                self._h5_param("args", str(j + 1), i, e)
            for k, v in mp.kwargs.items():
                self._h5_param("kwargs", str(k), i, v)

    def flush(self):
        if self.h5 is not None:
            self.h5.flush()
            return
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

    def close(self):
        if self.h5 is not None:
            self.h5.close()
            self.h5 = None

    @staticmethod
    def load(path):
        """Flat dict `name -> array` ("args/1", "kwargs/eta", ...) of either format."""
        if _is_h5(path):
            from . import h5lite
            flat = {}

            def walk(d, prefix):
                for k, v in d.items():
                    if isinstance(v, dict):
                        walk(v, prefix + k + "/")
                    else:
                        flat[prefix + k] = v.astype(bool) if prefix + k == "done" else v
            walk(h5lite.read_h5_to_nt(path), "")
            return flat
        with np.load(path) as z:
            return {k: z[k] for k in z.files}


def batchsolve_archive(path, solver, ps, chunk=64, flush=True, verb=False, solve=batchsolve):
    """batchsolve(h5, solver, ps): solve the sweep chunk by chunk (each chunk is one fused device pass),
    recording every point and rewriting the archive after each chunk.  Returns the array of values."""
    shape = ps.shape if isinstance(ps, np.ndarray) and ps.ndim > 1 else None  # `ps::AbstractArray`: the data sets take its shape
    lst = list(ps.reshape(-1)) if shape is not None else list(ps)
    arch = SweepArchive(path, shape if shape is not None else (len(lst),))
    out = []
    t0 = time.time()
    try:
        res = _run_archive(arch, solver, lst, chunk, flush, verb, solve, out, t0)
        return res.reshape(arch.shape + res.shape[1:])
    finally:
        arch.close()


def _run_archive(arch, solver, lst, chunk, flush, verb, solve, out, t0):
    for c0 in range(0, len(lst), max(1, int(chunk))):
        part = lst[c0:c0 + max(1, int(chunk))]
        vals = solve(solver, part, callback=lambda sv, i, n, p, sol, t, c0=c0: arch.record(
            tuple(int(j) for j in np.unravel_index(c0 + i[0], arch.shape)), p, sol, t))
        out.extend(list(vals))
        if flush:
            arch.flush()
        if verb:
            print(f"{min(c0 + len(part), len(lst)):5d} / {len(lst)} done in {time.time() - t0:e} (s)")
    arch.flush()
    return np.array(out)
