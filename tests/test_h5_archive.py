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


def test_reference_hdf5ext_cases_number_sarray_auxvalue_0d_3d(tmp_path):
    """The cases of the reference's own test of this extension (test/hdf5ext.jl:7-55), solved by the host algorithms and
    written into groups of ONE file: Number (`p` data set), array-valued results flattened into trailing axes, AuxValue
    split into `I/val` + `I/aux`, 0-d parameter arrays (scalar data sets) and a 3-d `paramproduct`."""
    path = tmp_path / "ext.h5"
    with h5lite.File(path, "w") as io:
        # Number (test/hdf5ext.jl:9-16)
        g1 = io.create_group("Number")
        solver = abz.IntegralSolver(abz.IntegralProblem(lambda x, p: p, (0.0, 1.0)), abz.QuadGKJL())
        params = np.arange(1.0, 11.0)
        values = abz.batchsolve_archive(g1, solver, params)
        assert np.allclose(values, params, rtol=1e-14)
        # SArray (test/hdf5ext.jl:17-26): SHermitianCompact{2}([s, c, -s]) = [s c; c -s]
        g2 = io.create_group("SArray")

        def fm(x, p):
            s, c = np.sin(p * x), np.cos(p * x)
            return np.array([[s, c], [c, -s]])
        solver2 = abz.IntegralSolver(abz.IntegralProblem(fm, (0.0, np.pi)), abz.QuadGKJL())
        values2 = abz.batchsolve_archive(g2, solver2, [0.8, 0.9, 1.0])
        # AuxValue (test/hdf5ext.jl:27-36)
        g3 = io.create_group("AuxValue")

        def fa(x, p):
            z = 1.0 / complex(np.cos(x), p)
            return abz.AuxValue(z.real, z.imag)
        solver3 = abz.IntegralSolver(abz.IntegralProblem(fa, (0.0, 2 * np.pi)), abz.QuadGKJL(), abstol=1e-3)
        pa = [2.0, 1.0, 0.5]
        values3 = abz.batchsolve_archive(g3, solver3, pa)
        # parameter dimensions (test/hdf5ext.jl:37-54)
        solver4 = abz.IntegralSolver(abz.IntegralProblem(lambda x, p: p[0] + p[1] + p[2], (0.0, 1.0)), abz.QuadGKJL())
        g4 = io.create_group("0d")
        values4 = abz.batchsolve_archive(g4, solver4, abz.paramzip(0, 1, 2))
        g5 = io.create_group("3d")
        values5 = abz.batchsolve_archive(g5, solver4, abz.paramproduct([1, 2], [1, 2], [1, 2]))
    nt = abz.read_h5_to_nt(path)
    assert set(nt) == {"Number", "SArray", "AuxValue", "0d", "3d"}
    n = nt["Number"]
    assert np.array_equal(n["I"], values) and np.array_equal(n["p"], params) and n["retcode"].dtype == np.int32
    assert np.all(n["numevals"] == -1) and np.all(n["E"] < 1e-12) and n["done"].all()  # no EvalCounter: -1 like the reference
    sa = nt["SArray"]
    assert sa["I"].shape == (3, 2, 2) and np.array_equal(sa["I"], values2)
    for k, pk in enumerate([0.8, 0.9, 1.0]):  # closed forms of the integrals of sin / cos
        s_, c_ = (1 - np.cos(pk * np.pi)) / pk, np.sin(pk * np.pi) / pk
        assert np.allclose(sa["I"][k], [[s_, c_], [c_, -s_]], atol=1e-9)
    av = nt["AuxValue"]
    assert set(av["I"]) == {"val", "aux"} and av["I"]["val"].shape == (3,)
    assert np.array_equal(av["I"]["val"], [v.val for v in values3]) and np.array_equal(av["I"]["aux"], [v.aux for v in values3])
    assert np.allclose(av["I"]["val"], 0.0, atol=1e-3) and np.allclose(av["I"]["aux"], [-2 * np.pi / np.sqrt(1 + q * q) for q in pa], atol=1e-3)
    assert np.all(np.isfinite(av["E"])) and np.all(av["E"] <= 1e-3)  # E holds the `val` error (ext/HDF5Ext.jl:140)
    z0 = nt["0d"]
    assert z0["I"].shape == () and float(z0["I"]) == pytest.approx(3.0) and float(values4) == pytest.approx(3.0)
    assert [int(z0["args"][str(j)]) for j in (1, 2, 3)] == [0, 1, 2]
    z3 = nt["3d"]
    assert z3["I"].shape == (2, 2, 2) and z3["I"][0, 0, 0] == pytest.approx(3.0) and z3["I"][1, 1, 1] == pytest.approx(6.0)
    assert values5.shape == (2, 2, 2) and np.array_equal(z3["I"], values5)
    assert np.array_equal(z3["args"]["1"][:, 0, 0], [1, 2]) and np.array_equal(z3["args"]["3"][0, 0, :], [1, 2])


def test_auxvalue_refines_the_auxiliary_quantity_after_the_value():
    """AuxQuadGK's ordering: `val` converges first, then the same panels are refined on the `aux` error.  An integrand
    whose value is smooth and whose auxiliary part is sharply peaked must spend its evaluations on the peak, and a plain
    integrand of the auxiliary part alone must give the same number."""
    eta = 1e-2

    def f(x, p):
        return abz.AuxValue(np.cos(x), eta / ((x - 0.3) ** 2 + eta * eta))
    prob = abz.IntegralProblem(f, (-1.0, 1.0))
    sol = abz.solve(prob, abz.EvalCounter(abz.AuxQuadGKJL()), abstol=1e-8)
    exact = np.arctan((1 - 0.3) / eta) + np.arctan((1 + 0.3) / eta)
    assert abs(sol.u.val - 2 * np.sin(1.0)) < 1e-12 and abs(sol.u.aux - exact) < 1e-8
    assert sol.resid.val <= 1e-8 and sol.resid.aux <= 1e-8 and sol.numevals > 15 * 10
    only_val = abz.solve(abz.IntegralProblem(lambda x, p: np.cos(x), (-1.0, 1.0)), abz.EvalCounter(abz.AuxQuadGKJL()), abstol=1e-8)
    assert only_val.numevals == 15  # the value alone never needs a second panel
    # separate tolerances: a loose one on aux stops the second stage early
    loose = abz.solve(prob, abz.EvalCounter(abz.AuxQuadGKJL()), abstol=abz.AuxValue(1e-8, 1e-2))
    assert loose.numevals < sol.numevals and abs(loose.u.aux - exact) < 1e-2
