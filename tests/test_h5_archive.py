"""The HDF5 sweep archive (ext/HDF5Ext.jl) through the ctypes binding of libhdf5, checked by independent readers:
the HDF5 command-line tools and, where the image has one, another interpreter's h5py."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

import autobzcore.jl_amd as abz
from autobzcore.jl_amd import h5lite
from autobzcore.jl_amd.solver import IntegralSolution

pytestmark = pytest.mark.skipif(not h5lite.available(), reason="no HDF5 C library on this machine (set ABZ_HDF5_LIB)")


def _tool(name):
    return shutil.which(name) or (os.path.join("/opt/conda/bin", name) if os.path.exists(os.path.join("/opt/conda/bin", name)) else None)


def _fake_batchsolve(vector_valued=False):
    def run(solver, part, callback=None):  # stands in for the fused device sweep (the GPU test uses the real one)
        vals = []
        for i, p in enumerate(part):
            u = p.args[0] ** 2 + 1j * p.kwargs["b"]
            if vector_valued:
                u = np.array([u, 2 * u, -u])
            sol = IntegralSolution(u, 1e-4, True, 15 + i)
            callback(solver, (i,), i + 1, p, sol, 0.01)
            vals.append(sol.u)
        return np.array(vals)
    return run


def test_named_tuple_round_trip_with_groups_and_types(tmp_path):
    """ref: ext/HDF5Ext.jl:12-40 (`write_nt_to_h5` / `read_h5_to_nt`)."""
    rng = np.random.default_rng(0)
    nt = {"a": rng.standard_normal((3, 4)), "z": rng.standard_normal(5) + 1j * rng.standard_normal(5),
          "n": np.arange(6, dtype=np.int64).reshape(2, 3), "r": np.array([1, 0, 1], dtype=np.int32), "s": np.float64(2.5),
          "grp": {"inner": np.arange(3.0), "deeper": {"x": np.array([7], dtype=np.int64)}}}
    path = tmp_path / "nt.h5"
    h5lite.write_nt_to_h5(nt, path)
    back = h5lite.read_h5_to_nt(path)
    assert set(back) == set(nt) and set(back["grp"]) == {"inner", "deeper"}
    for k in ("a", "z", "n", "r"):
        assert back[k].dtype == nt[k].dtype and np.array_equal(back[k], nt[k])
    assert back["s"] == 2.5 and np.array_equal(back["grp"]["deeper"]["x"], [7])
    assert h5lite.version()[0] == 1


def test_sweep_archive_hdf5_layout_partial_flush_and_foreign_readers(tmp_path):
    """ref: ext/HDF5Ext.jl:123-158 -- I, E, t, retcode (Int32), numevals (Int), args/<j>, kwargs/<name>; flushed as the sweep
    goes.  The file must be a real HDF5 file: h5ls lists the data sets with the reference's types, and h5py (when another
    interpreter on the box has it) reads I as complex through the {r, i} compound."""
    path = tmp_path / "sweep.h5"
    seen = []
    inner = _fake_batchsolve()

    def solve(solver, part, callback=None):
        if path.exists() and seen is not None:
            z = abz.SweepArchive.load(path)  # a second reader while the writer holds the file: everything flushed so far
            seen.append(int(z["done"].sum()))
        return inner(solver, part, callback=callback)

    ps = [abz.MixedParameters(float(w), b=0.5 * k) for k, w in enumerate(np.linspace(0, 1, 10))]
    out = abz.batchsolve_archive(path, None, ps, chunk=4, solve=solve)
    z = abz.SweepArchive.load(path)
    assert seen == [0, 4, 8]
    assert z["done"].all() and z["I"].dtype == np.complex128 and np.array_equal(z["I"], out)
    assert z["numevals"].dtype == np.int64 and np.array_equal(z["numevals"], 15 + np.arange(10) % 4)
    assert np.allclose(z["args/1"], np.linspace(0, 1, 10)) and np.allclose(z["kwargs/b"], 0.5 * np.arange(10))
    assert np.allclose(z["E"], 1e-4) and z["retcode"].dtype == np.int32 and np.all(z["retcode"] == 1) and np.allclose(z["t"], 0.01)

    h5ls = _tool("h5ls")
    if h5ls:
        txt = subprocess.run([h5ls, "-r", "-v", str(path)], capture_output=True, text=True, check=True).stdout
        for name in ("/I", "/E", "/t", "/retcode", "/numevals", "/args/1", "/kwargs/b"):
            assert name + " " in txt.replace("\t", " "), txt
        assert "native int" in txt and "native double" in txt and "struct" in txt  # Int32 retcode, Float64, {r,i} compound
    h5dump = _tool("h5dump")
    if h5dump:
        txt = subprocess.run([h5dump, "-d", "/numevals", "-d", "/I", str(path)], capture_output=True, text=True, check=True).stdout
        assert 'H5T_IEEE_F64LE "r";' in txt and 'H5T_IEEE_F64LE "i";' in txt and "H5T_STD_I64LE" in txt
        data = txt[txt.index('DATASET "/numevals"'):]
        data = data[data.index("DATA {"):data.index("}", data.index("DATA {"))]
        nums = [int(tok) for line in data.splitlines()[1:] for tok in line.split(":")[-1].replace(",", " ").split()]
        assert nums == z["numevals"].tolist()
    py = "/opt/conda/bin/python"
    if os.path.exists(py) and subprocess.run([py, "-c", "import h5py"], capture_output=True).returncode == 0:
        code = ("import h5py, json, sys\n"
                "f = h5py.File(sys.argv[1], 'r')\n"
                "I = f['I'][...]\n"
                "print(json.dumps({'dtype': str(I.dtype), 're': I.real.tolist(), 'im': I.imag.tolist(), 'n': f['numevals'][...].tolist(),"
                " 'rc': str(f['retcode'].dtype), 'b': f['kwargs/b'][...].tolist(), 'keys': sorted(f.keys())}))\n")
        r = subprocess.run([py, "-c", code, str(path)], capture_output=True, text=True, check=True)
        got = json.loads(r.stdout.strip().splitlines()[-1])
        assert got["dtype"] == "complex128" and got["rc"] == "int32"
        assert np.array_equal(np.array(got["re"]) + 1j * np.array(got["im"]), out) and got["n"] == z["numevals"].tolist()
        assert got["keys"] == ["E", "I", "args", "done", "kwargs", "numevals", "retcode", "t"]


def test_array_valued_results_and_parameter_grids(tmp_path):
    """ref: ext/HDF5Ext.jl:43-49 -- array-valued T: the data set is `size(T)..., size(ps)...` in Julia = `ps.shape + T.shape`
    in C order; `ps` may be an N-d array and every data set takes its shape."""
    path = tmp_path / "grid.hdf5"
    ps = np.empty((2, 3), dtype=object)
    for i in range(2):
        for j in range(3):
            ps[i, j] = abz.MixedParameters(float(i + 0.1 * j), b=float(j))
    out = abz.batchsolve_archive(path, None, ps, chunk=4, solve=_fake_batchsolve(vector_valued=True))
    z = abz.SweepArchive.load(path)
    assert out.shape == (2, 3, 3) and z["I"].shape == (2, 3, 3) and np.array_equal(z["I"], out)
    assert z["E"].shape == (2, 3) and z["args/1"].shape == (2, 3) and z["args/1"][1, 2] == pytest.approx(1.2)
    assert np.array_equal(z["I"][1, 2], (1.2 ** 2 + 2j) * np.array([1, 2, -1]))


def test_missing_library_is_an_error_not_a_change_of_format(tmp_path, monkeypatch):
    monkeypatch.setattr(h5lite, "_lib", None)
    monkeypatch.setattr(h5lite, "_lib_err", "no HDF5 C library found (test)")
    with pytest.raises(h5lite.H5Error):
        abz.SweepArchive(tmp_path / "x.h5", (3,))
    assert not (tmp_path / "x.h5").exists() and not (tmp_path / "x.h5.npz").exists()
