"""Sweep archive: the data sets of the reference's HDF5 writer with partial-result persistence.

ref: ext/HDF5Ext.jl:123-158 -- `batchsolve(h5, solver, ps)` creates `I, E, t, retcode, numevals` and the parameter data
sets -- `p` for plain parameters, `params/<j>` for tuples, `args/<j>` + `kwargs/<name>` for MixedParameters
(ext/HDF5Ext.jl:73-93) -- fills them from the solver callback and flushes after every point.  AuxValue results are split
into the group `I` with `val` and `aux` (ext/HDF5Ext.jl:48-64), array-valued results are flattened into trailing axes.

Two on-disk formats, chosen by the target:
  * a path `*.h5` / `*.hdf5`, or an open `h5lite.Group` (the reference writes into `create_group(io, ...)` objects,
    test/hdf5ext.jl:9-55): a real HDF5 file through the C library (`h5lite.py`, a ctypes binding; no h5py in this
    interpreter).  Data sets are created at the first record, every solved point is written in place (a one-element
    hyperslab, the reference's `set_value`), the file is flushed after every chunk of the sweep.  Shapes are the
    reference's read in C order: `I` is `size(ps)..., size(T)...` here = `size(T)..., size(ps)...` in Julia.  Raises if no
    libhdf5 can be loaded.
  * any other path: a NumPy `.npz` archive with the same names, rewritten atomically (temp file + rename) after every chunk.
Either way an interrupted run keeps everything solved so far; the extra data set `done` marks the filled entries."""
import os
import time

import numpy as np

from .solver import AuxValue, MixedParameters, batchsolve


def _is_h5(path):
    return str(path).lower().endswith((".h5", ".hdf5", ".hdf"))


def _param_items(p):
    """[(data-set name, value)] of one parameter, named as ext/HDF5Ext.jl:73-112 names them."""
    if isinstance(p, MixedParameters):
        return [(f"args/{j + 1}", e) for j, e in enumerate(p.args)] + [(f"kwargs/{k}", v) for k, v in p.kwargs.items()]
    if isinstance(p, tuple):
        return [(f"params/{j + 1}", e) for j, e in enumerate(p)]
    return [("p", p)]


class SweepArchive:
    def __init__(self, target, shape):
        self.shape = tuple(shape)
        self.data = {}  # name -> array of shape `shape + value shape`, allocated at the first record
        self.done = np.zeros(self.shape, dtype=bool)
        self.h5 = None
        self.root = None
        self._own = False
        self._d = {}
        self._g = {}
        self.path = None
        from . import h5lite
        if isinstance(target, h5lite.Group):
            self.root, self.h5 = target, target
        else:
            self.path = str(target)
            if _is_h5(self.path):
                self.h5 = self.root = h5lite.File(self.path, "w")  # raises H5Error without a library: no silent change of format
                self._own = True
        if self.h5 is not None:
            self._d["done"] = self.root.write_dataset("done", self.done)

    # ---- one named array (memory copy + its HDF5 data set)
    def _put(self, name, i, value, fill):
        v = np.asarray(value)
        a = self.data.get(name)
        if a is None:
            dt = np.complex128 if np.iscomplexobj(v) else (v.dtype if v.dtype.kind in "iu" and fill is not np.nan else np.float64)
            if v.dtype == np.bool_:
                dt = np.uint8
            a = self.data[name] = np.full(self.shape + v.shape, fill, dtype=dt)
            if self.h5 is not None:
                parts = name.split("/")
                g, prefix = self.root, ""
                for part in parts[:-1]:
                    prefix += part + "/"
                    if prefix not in self._g:
                        self._g[prefix] = g.create_group(part)
                    g = self._g[prefix]
                self._d[name] = g.write_dataset(parts[-1], a)
        a[i] = v
        if self.h5 is not None:
            self._d[name][i] = a[i]

    def record(self, i, p, sol, t):
        """One solved point: index i (tuple), parameters p, IntegralSolution sol, seconds t."""
        i = tuple(np.atleast_1d(i).tolist()) if not isinstance(i, tuple) else i
        if isinstance(sol.u, AuxValue):
            self._put("I/val", i, sol.u.val, np.nan)
            self._put("I/aux", i, sol.u.aux, np.nan)
            resid = None if sol.resid is None else (sol.resid.val if isinstance(sol.resid, AuxValue) else sol.resid)
        else:
            self._put("I", i, sol.u, np.nan)
            resid = sol.resid
        self._put("E", i, np.float64(np.nan if resid is None else float(np.real(resid))), np.nan)
        self._put("t", i, np.float64(t), np.nan)
        self._put("retcode", i, np.int32(bool(sol.retcode)), 0)
        self._put("numevals", i, np.int64(sol.numevals), -1)
        for name, e in _param_items(p):
            self._put(name, i, e, np.nan if np.asarray(e).dtype.kind in "fc" else 0)
        self.done[i] = True
        if self.h5 is not None:
            self._d["done"][i] = np.uint8(1)

    def flush(self):
        if self.h5 is not None:
            top = self.root
            while not hasattr(top, "flush"):
                top = top.parent
            top.flush()
            return
        data = dict(self.data, done=self.done)
        tmp = self.path + ".tmp.npz"
        np.savez(tmp, **data)
        os.replace(tmp, self.path)

    def close(self):
        if self.h5 is not None and self._own:
            self.h5.close()
        self.h5 = None

    @staticmethod
    def load(path):
        """Flat dict `name -> array` ("I", "args/1", "kwargs/eta", "I/val", ...) of either format."""
        if _is_h5(path):
            from . import h5lite
            flat = {}

            def walk(d, prefix):
                for k, v in d.items():
                    if isinstance(v, dict):
                        walk(v, prefix + k + "/")
                    else:
                        flat[prefix + k] = v.astype(bool) if k == "done" else v
            walk(h5lite.read_h5_to_nt(path), "")
            return flat
        with np.load(path) as z:
            return {k: z[k] for k in z.files}


def batchsolve_archive(target, solver, ps, chunk=64, flush=True, verb=False, solve=batchsolve):
    """batchsolve(h5, solver, ps): solve the sweep chunk by chunk (each chunk is one fused device pass), recording every
    point and flushing the archive after each chunk.  `target`: a file name or an open h5lite.Group; `ps`: a sequence, or
    an N-d (0-d included) array of parameters whose shape every data set takes.  Returns the array of values."""
    if isinstance(ps, np.ndarray) and ps.ndim != 1:
        shape, lst = ps.shape, list(ps.reshape(-1))
    else:
        lst = list(ps)
        shape = (len(lst),)
    arch = SweepArchive(target, shape)
    out = []
    t0 = time.time()
    step = max(1, int(chunk))
    try:
        for c0 in range(0, len(lst), step):
            part = lst[c0:c0 + step]
            vals = solve(solver, part, callback=lambda sv, i, n, p, sol, t, c0=c0: arch.record(
                tuple(int(j) for j in np.unravel_index(c0 + i[0], shape)) if shape else (), p, sol, t))
            out.extend(list(vals))
            if flush:
                arch.flush()
            if verb:
                print(f"{min(c0 + len(part), len(lst)):5d} / {len(lst)} done in {time.time() - t0:e} (s)")
        arch.flush()
    finally:
        arch.close()
    if out and isinstance(out[0], AuxValue):
        res = np.empty(len(out), dtype=object)
        res[:] = out
        return res.reshape(shape)
    res = np.array(out)
    return res.reshape(shape + res.shape[1:])
