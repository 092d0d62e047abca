"""A thin ctypes binding of the HDF5 C library: just what the sweep archive needs.

ref: ext/HDF5Ext.jl:1-50 (`h5open`, `create_group`, `create_dataset`, `write_dataset`, `read_dataset`, element
assignment `dset[i...] = v`, `flush`).  No h5py in this interpreter; the image ships libhdf5 (1.10) under /opt/conda/lib,
and a maintainer's machine has one wherever HDF5.jl / h5py put it.  The library is looked up, in this order, at
`$ABZ_HDF5_LIB`, through the loader (`ctypes.util.find_library("hdf5")`), and at /opt/conda/lib/libhdf5.so*.
`available()` says whether one was found; everything else raises `H5Error` without one -- there is no silent fall-back
to another format.

Layout conventions (so that HDF5.jl and h5py read the files the same way they read the reference's):
  * NumPy arrays go in C order with their NumPy shape.  HDF5.jl shows the dimensions reversed, which is exactly a Julia
    array `(size(T)..., size(ps)...)` -- the layout `autobz_create_dataset` makes (ext/HDF5Ext.jl:45-49).
  * complex128 is the compound `{r: f64, i: f64}` both HDF5.jl and h5py use for complex numbers.
  * bool is stored as uint8.
"""
import ctypes as C
import ctypes.util
import glob
import os

import numpy as np

hid_t = C.c_int64
herr_t = C.c_int
hsize_t = C.c_uint64

H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC = 0, 1, 2
H5P_DEFAULT = 0
H5S_ALL = 0
H5S_SCALAR = 0
H5S_SELECT_SET = 0
H5F_SCOPE_GLOBAL = 1
H5T_INTEGER, H5T_FLOAT, H5T_COMPOUND = 0, 1, 6
H5T_SGN_NONE = 0
H5I_GROUP, H5I_DATASET = 2, 5
H5_INDEX_NAME, H5_ITER_INC = 0, 0


class H5Error(RuntimeError):
    pass


class _GInfo(C.Structure):
    _fields_ = [("storage_type", C.c_int), ("nlinks", hsize_t), ("max_corder", C.c_int64), ("mounted", C.c_int)]


_lib = None
_lib_err = None


def _candidates():
    env = os.environ.get("ABZ_HDF5_LIB")
    if env:
        yield env
    found = ctypes.util.find_library("hdf5")
    if found:
        yield found
    for pat in ("/opt/conda/lib/libhdf5.so", "/opt/conda/lib/libhdf5.so.*", "/usr/lib/x86_64-linux-gnu/libhdf5*.so*",
                "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*"):
        for p in sorted(glob.glob(pat)):
            yield p


def _proto(lib):
    P = C.POINTER
    sig = {
        "H5open": (herr_t, []),
        "H5get_libversion": (herr_t, [P(C.c_uint)] * 3),
        "H5Eset_auto2": (herr_t, [hid_t, C.c_void_p, C.c_void_p]),
        "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]),
        "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]),
        "H5Fflush": (herr_t, [hid_t, C.c_int]),
        "H5Fclose": (herr_t, [hid_t]),
        "H5Gcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]),
        "H5Gopen2": (hid_t, [hid_t, C.c_char_p, hid_t]),
        "H5Gget_info": (herr_t, [hid_t, P(_GInfo)]),
        "H5Gclose": (herr_t, [hid_t]),
        "H5Lget_name_by_idx": (C.c_ssize_t, [hid_t, C.c_char_p, C.c_int, C.c_int, hsize_t, C.c_char_p, C.c_size_t, hid_t]),
        "H5Lexists": (C.c_int, [hid_t, C.c_char_p, hid_t]),
        "H5Ldelete": (herr_t, [hid_t, C.c_char_p, hid_t]),
        "H5Oopen": (hid_t, [hid_t, C.c_char_p, hid_t]),
        "H5Oclose": (herr_t, [hid_t]),
        "H5Iget_type": (C.c_int, [hid_t]),
        "H5Screate": (hid_t, [C.c_int]),
        "H5Screate_simple": (hid_t, [C.c_int, P(hsize_t), P(hsize_t)]),
        "H5Sselect_hyperslab": (herr_t, [hid_t, C.c_int, P(hsize_t), P(hsize_t), P(hsize_t), P(hsize_t)]),
        "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
        "H5Sget_simple_extent_dims": (C.c_int, [hid_t, P(hsize_t), P(hsize_t)]),
        "H5Sclose": (herr_t, [hid_t]),
        "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
        "H5Dopen2": (hid_t, [hid_t, C.c_char_p, hid_t]),
        "H5Dget_space": (hid_t, [hid_t]),
        "H5Dget_type": (hid_t, [hid_t]),
        "H5Dwrite": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Dread": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Dclose": (herr_t, [hid_t]),
        "H5Tcreate": (hid_t, [C.c_int, C.c_size_t]),
        "H5Tinsert": (herr_t, [hid_t, C.c_char_p, C.c_size_t, hid_t]),
        "H5Tget_class": (C.c_int, [hid_t]),
        "H5Tget_size": (C.c_size_t, [hid_t]),
        "H5Tget_sign": (C.c_int, [hid_t]),
        "H5Tget_nmembers": (C.c_int, [hid_t]),
        "H5Tclose": (herr_t, [hid_t]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args


def lib():
    global _lib, _lib_err
    if _lib is not None:
        return _lib
    if _lib_err is not None:
        raise H5Error(_lib_err)
    tried = []
    for path in _candidates():
        try:
            handle = C.CDLL(path)
            _proto(handle)
            if handle.H5open() < 0:
                raise OSError("H5open failed")
            # the binding assumes the 64-bit hid_t of HDF5 >= 1.10 (1.8: 32-bit ids, other H5T_NATIVE_*_g widths)
            a, b, c = C.c_uint(), C.c_uint(), C.c_uint()
            if handle.H5get_libversion(C.byref(a), C.byref(b), C.byref(c)) < 0 or (a.value, b.value) < (1, 10):
                raise OSError(f"HDF5 {a.value}.{b.value}.{c.value} is older than 1.10 (32-bit hid_t)")
            handle.H5Eset_auto2(0, None, None)  # errors come back as negative ids; no stack dump on stderr
            _lib = handle
            return _lib
        except (OSError, AttributeError) as e:
            tried.append(f"{path}: {e}")
    _lib_err = ("no HDF5 C library found (set ABZ_HDF5_LIB to libhdf5.so; tried: " + ("; ".join(tried) or "nothing on the loader path") +
                ").  Use an .npz archive path instead.")
    raise H5Error(_lib_err)


def available():
    try:
        lib()
        return True
    except H5Error:
        return False


def version():
    a, b, c = C.c_uint(), C.c_uint(), C.c_uint()
    lib().H5get_libversion(C.byref(a), C.byref(b), C.byref(c))
    return (a.value, b.value, c.value)


def _native(name):
    return hid_t.in_dll(lib(), name).value


def _ck(v, what):
    if v < 0:
        raise H5Error(f"HDF5: {what} failed")
    return v


def _mem_type(dt):
    """(hid of the memory/file type, owned?) for a NumPy dtype."""
    dt = np.dtype(dt)
    if dt == np.complex128:
        t = _ck(lib().H5Tcreate(H5T_COMPOUND, 16), "H5Tcreate")
        f64 = _native("H5T_NATIVE_DOUBLE_g")
        _ck(lib().H5Tinsert(t, b"r", 0, f64), "H5Tinsert")
        _ck(lib().H5Tinsert(t, b"i", 8, f64), "H5Tinsert")
        return t, True
    table = {np.dtype(np.float64): "H5T_NATIVE_DOUBLE_g", np.dtype(np.float32): "H5T_NATIVE_FLOAT_g",
             np.dtype(np.int64): "H5T_NATIVE_INT64_g", np.dtype(np.int32): "H5T_NATIVE_INT32_g",
             np.dtype(np.uint8): "H5T_NATIVE_UINT8_g", np.dtype(np.int8): "H5T_NATIVE_INT8_g",
             np.dtype(np.uint32): "H5T_NATIVE_UINT32_g", np.dtype(np.uint64): "H5T_NATIVE_UINT64_g"}
    if dt not in table:
        raise H5Error(f"HDF5: dtype {dt} is not supported by this binding")
    return _native(table[dt]), False


def _storage_dtype(a):
    a = np.asarray(a)
    if a.dtype == np.bool_:
        return a.astype(np.uint8)
    if np.issubdtype(a.dtype, np.complexfloating):
        return a.astype(np.complex128)
    if np.issubdtype(a.dtype, np.floating) and a.dtype not in (np.float32, np.float64):
        return a.astype(np.float64)
    return a


class Dataset:
    def __init__(self, hid, name):
        self.id = hid
        self.name = name
        sp = _ck(lib().H5Dget_space(hid), "H5Dget_space")
        nd = lib().H5Sget_simple_extent_ndims(sp)
        dims = (hsize_t * max(nd, 1))()
        if nd > 0:
            lib().H5Sget_simple_extent_dims(sp, dims, None)
        lib().H5Sclose(sp)
        self.shape = tuple(int(dims[i]) for i in range(nd))
        t = _ck(lib().H5Dget_type(hid), "H5Dget_type")
        cls, size = lib().H5Tget_class(t), lib().H5Tget_size(t)
        if cls == H5T_FLOAT:
            self.dtype = np.dtype({4: np.float32, 8: np.float64}[size])
        elif cls == H5T_INTEGER:
            signed = lib().H5Tget_sign(t) != H5T_SGN_NONE
            self.dtype = np.dtype(f"{'i' if signed else 'u'}{size}")
        elif cls == H5T_COMPOUND and size == 16 and lib().H5Tget_nmembers(t) == 2:
            self.dtype = np.dtype(np.complex128)
        else:
            lib().H5Tclose(t)
            raise H5Error(f"HDF5: data set {name!r} has a type this binding does not read (class {cls}, {size} bytes)")
        lib().H5Tclose(t)

    def read(self):
        out = np.empty(self.shape, dtype=self.dtype)
        mt, own = _mem_type(self.dtype)
        try:
            _ck(lib().H5Dread(self.id, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data_as(C.c_void_p)), f"read of {self.name!r}")
        finally:
            if own:
                lib().H5Tclose(mt)
        return out if out.ndim else out[()]

    def write(self, a):
        a = np.require(np.asarray(a, dtype=self.dtype), requirements="C")
        if a.shape != self.shape:
            raise H5Error(f"HDF5: {self.name!r} has shape {self.shape}, got {a.shape}")
        mt, own = _mem_type(self.dtype)
        try:
            _ck(lib().H5Dwrite(self.id, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data_as(C.c_void_p)), f"write of {self.name!r}")
        finally:
            if own:
                lib().H5Tclose(mt)

    def __setitem__(self, idx, val):
        """dset[i, j, ...] = value: leading indices are integers; the value fills the remaining axes (a hyperslab, the
        `parent[ax..., i.I...] = sol` of ext/HDF5Ext.jl:57 in C order)."""
        if not isinstance(idx, tuple):
            idx = (idx,)
        idx = tuple(int(i) for i in idx)
        rest = self.shape[len(idx):]
        v = np.require(np.array(np.broadcast_to(np.asarray(val, dtype=self.dtype), rest)), requirements="C")
        nd = len(self.shape)
        if nd == 0:
            return self.write(v)
        for i, n in zip(idx, self.shape):
            if not 0 <= i < n:
                raise IndexError(f"index {idx} out of range for shape {self.shape}")
        start = (hsize_t * nd)(*(idx + (0,) * len(rest)))
        count = (hsize_t * nd)(*((1,) * len(idx) + rest))
        fs = _ck(lib().H5Dget_space(self.id), "H5Dget_space")
        ms = _ck(lib().H5Screate_simple(nd, count, None), "H5Screate_simple")
        mt, own = _mem_type(self.dtype)
        try:
            _ck(lib().H5Sselect_hyperslab(fs, H5S_SELECT_SET, start, None, count, None), "H5Sselect_hyperslab")
            _ck(lib().H5Dwrite(self.id, mt, ms, fs, H5P_DEFAULT, v.ctypes.data_as(C.c_void_p)), f"write into {self.name!r}")
        finally:
            if own:
                lib().H5Tclose(mt)
            lib().H5Sclose(ms)
            lib().H5Sclose(fs)

    def close(self):
        if self.id:
            lib().H5Dclose(self.id)
            self.id = 0


class Group:
    def __init__(self, hid, name="/", is_file=False, parent=None):
        self.id = hid
        self.name = name
        self.parent = parent  # the group it was opened from (flush walks up to the file)
        self._is_file = is_file
        self._open = []

    # ---- creation
    def create_group(self, name):
        g = Group(_ck(lib().H5Gcreate2(self.id, name.encode(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), f"create_group({name!r})"),
                  self.name.rstrip("/") + "/" + name, parent=self)
        self._open.append(g)
        return g

    def create_dataset(self, name, dtype, shape):
        shape = tuple(int(s) for s in shape)
        ft, own = _mem_type(np.uint8 if np.dtype(dtype) == np.bool_ else dtype)
        if shape:
            dims = (hsize_t * len(shape))(*shape)
            sp = _ck(lib().H5Screate_simple(len(shape), dims, None), "H5Screate_simple")
        else:
            sp = _ck(lib().H5Screate(H5S_SCALAR), "H5Screate")
        try:
            hid = _ck(lib().H5Dcreate2(self.id, name.encode(), ft, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), f"create_dataset({name!r})")
        finally:
            lib().H5Sclose(sp)
            if own:
                lib().H5Tclose(ft)
        d = Dataset(hid, self.name.rstrip("/") + "/" + name)
        self._open.append(d)
        return d

    def write_dataset(self, name, a):
        a = _storage_dtype(a)
        d = self.create_dataset(name, a.dtype, a.shape)
        d.write(a)
        return d

    # ---- access
    def keys(self):
        info = _GInfo()
        _ck(lib().H5Gget_info(self.id, C.byref(info)), "H5Gget_info")
        out = []
        for i in range(int(info.nlinks)):
            n = lib().H5Lget_name_by_idx(self.id, b".", H5_INDEX_NAME, H5_ITER_INC, i, None, 0, H5P_DEFAULT)
            buf = C.create_string_buffer(int(_ck(n, "H5Lget_name_by_idx")) + 1)
            lib().H5Lget_name_by_idx(self.id, b".", H5_INDEX_NAME, H5_ITER_INC, i, buf, len(buf), H5P_DEFAULT)
            out.append(buf.value.decode())
        return out

    def __contains__(self, name):
        return lib().H5Lexists(self.id, name.encode(), H5P_DEFAULT) > 0

    def __getitem__(self, name):
        o = _ck(lib().H5Oopen(self.id, name.encode(), H5P_DEFAULT), f"open of {name!r} in {self.name!r}")
        kind = lib().H5Iget_type(o)
        lib().H5Oclose(o)
        if kind == H5I_GROUP:
            g = Group(_ck(lib().H5Gopen2(self.id, name.encode(), H5P_DEFAULT), "H5Gopen2"), self.name.rstrip("/") + "/" + name, parent=self)
            self._open.append(g)
            return g
        if kind == H5I_DATASET:
            d = Dataset(_ck(lib().H5Dopen2(self.id, name.encode(), H5P_DEFAULT), "H5Dopen2"), self.name.rstrip("/") + "/" + name)
            self._open.append(d)
            return d
        raise H5Error(f"HDF5: {name!r} is neither a group nor a data set")

    def delete(self, name):
        _ck(lib().H5Ldelete(self.id, name.encode(), H5P_DEFAULT), f"delete of {name!r}")

    def close(self):
        for o in reversed(self._open):
            o.close()
        self._open = []
        if self.id:
            (lib().H5Fclose if self._is_file else lib().H5Gclose)(self.id)
            self.id = 0


class File(Group):
    """h5open(filename, mode): mode "r", "r+" or "w"."""

    def __init__(self, path, mode="r"):
        path = os.fspath(path)
        if mode == "w":
            hid = lib().H5Fcreate(path.encode(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
        elif mode in ("r", "r+"):
            hid = lib().H5Fopen(path.encode(), H5F_ACC_RDONLY if mode == "r" else H5F_ACC_RDWR, H5P_DEFAULT)
        else:
            raise ValueError("mode must be 'r', 'r+' or 'w'")
        super().__init__(_ck(hid, f"open of {path!r} ({mode})"), "/", is_file=True)
        self.path = path

    def flush(self):
        _ck(lib().H5Fflush(self.id, H5F_SCOPE_GLOBAL), "H5Fflush")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


def read_h5_to_nt(filename):
    """ref: ext/HDF5Ext.jl:12-19 -- data sets into a dict, groups into dicts recursively."""
    def walk(g):
        out = {}
        for k in g.keys():
            v = g[k]
            out[k] = walk(v) if isinstance(v, Group) else v.read()
        return out
    with File(filename, "r") as f:
        return walk(f)


def write_nt_to_h5(nt, filename):
    """ref: ext/HDF5Ext.jl:21-40 -- a (nested) dict of arrays into data sets / groups of the same names."""
    def walk(d, g):
        for k, v in d.items():
            if isinstance(v, dict):
                walk(v, g.create_group(str(k)))
            else:
                g.write_dataset(str(k), v)
    with File(filename, "w") as f:
        walk(nt, f)
