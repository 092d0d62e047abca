"""CPU-side checks: the C-ABI library loads, exports every symbol include/abzhip.h declares, fails
loudly without a GPU, and its host-only entry points (integer tables, GK rule) match the oracle."""
import os
import re

import numpy as np
import pytest

import abz_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def abz():
    import autobzcore.jl_amd as m
    return m


def test_header_symbols_exported_and_bound(abz):
    from autobzcore.jl_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "abzhip.h")).read()
    declared = set(re.findall(r"^(?:const char\*|int) (abz_\w+)\(", hdr, flags=re.M))
    assert len(declared) >= 34
    assert declared == set(_lib.PROTOTYPES)
    h = _lib.lib()
    for name in declared:
        assert hasattr(h, name)
    assert h.abz_version() == 502  # 502: ABZ_WANT_H_ROW_MAJOR (abz_eval_nodes); round 5: ABZ_MAX_BANDS 64, ABZ_ERR_INTERNAL (every entry point catches C++ exceptions); round 4: abz_autoptr_solve(_many), abz_series_drop_rules; round 3: abz_mem_info, fused GGR build (ABZ_K_GGRBUILD), status word in the IAI exchange; 301: ABZ_WANT_H_COMPACT


def test_header_constants_match_the_bindings(abz):
    """The `want` bits, integrand ids and limits kinds of include/abzhip.h against the ctypes mirror and the Julia shim."""
    from autobzcore.jl_amd import _lib as L
    hdr = open(os.path.join(ROOT, "include", "abzhip.h")).read()
    defs = {k: int(v) for k, v in re.findall(r"^#define (ABZ_\w+) (-?\d+)\b", hdr, flags=re.M)}
    assert (defs["ABZ_WANT_H"], defs["ABZ_WANT_EIG"], defs["ABZ_WANT_VEL"], defs["ABZ_WANT_H_COMPACT"]) == \
        (L.WANT_H, L.WANT_EIG, L.WANT_VEL, L.WANT_H_COMPACT) == (1, 2, 4, 8)
    ids = [defs[k] for k in ("ABZ_F_ONE", "ABZ_F_LINEAR", "ABZ_F_LINEAR_X", "ABZ_F_DOS", "ABZ_F_TRGLOC", "ABZ_F_GLOC", "ABZ_F_DOS_EIG")]
    assert ids == [L.F_ONE, L.F_LINEAR, L.F_LINEAR_X, L.F_DOS, L.F_TRGLOC, L.F_GLOC, L.F_DOS_EIG] == list(range(7))
    jl = open(os.path.join(ROOT, "julia", "AutoBZCoreHIP.jl")).read()
    assert "const WANT_H, WANT_EIG, WANT_VEL = Cint(1), Cint(2), Cint(4)" in jl and "const WANT_H_COMPACT = Cint(8)" in jl
    assert defs["ABZ_VERSION"] == 502


def test_fails_loudly_without_gpu(abz):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(abz.AbzError):
        abz.Context()
    s = abz.FourierSeries([0.5, 0.0, 0.5], offset=-2)
    with pytest.raises(abz.AbzError):
        s(0.1)  # no CPU fallback for series evaluation


@pytest.mark.parametrize("kind,d,npt", [("InversionSymIBZ", 1, 9), ("InversionSymIBZ", 3, 7), ("CubicSymIBZ", 2, 9),
                                        ("CubicSymIBZ", 3, 8), ("CubicSymIBZ", 3, 50)])
def test_symptr_rule_bit_exact_vs_oracle(abz, kind, d, npt):
    bzo = orc.load_bz(kind, np.eye(d))
    idx, w = abz.symptr_rule(npt, d, bzo.syms)
    wsym, flags, nsym = orc.symptr_rule(npt, d, bzo.syms)
    assert len(w) == nsym and w.sum() == npt**d
    lin = np.flatnonzero(wsym.reshape(-1, order="F"))
    strides = npt ** np.arange(d)
    assert np.array_equal(idx @ strides, lin)
    assert np.array_equal(w, wsym.reshape(-1, order="F")[lin])
    # product-side symmetry generators equal the oracle's (same order)
    bz = abz.load_bz({"InversionSymIBZ": abz.InversionSymIBZ(), "CubicSymIBZ": abz.CubicSymIBZ()}[kind], np.eye(d))
    assert all(np.array_equal(a, b) for a, b in zip(bz.syms, bzo.syms))


def test_load_bz_pins(abz):
    # ref: test/brillouin.jl:7-31
    A = np.eye(3)
    fbz = abz.load_bz(abz.FBZ(), A)
    assert np.allclose(fbz.B, 2 * np.pi * np.eye(3)) and abz.nsyms(fbz) == 1
    assert fbz.lims == abz.CubicLimits(np.zeros(3), np.ones(3))
    ibz = abz.load_bz(abz.InversionSymIBZ(), A)
    assert abz.nsyms(ibz) == 8 and ibz.lims == abz.CubicLimits(np.zeros(3), 0.5 * np.ones(3))
    cbz = abz.load_bz(abz.CubicSymIBZ(), A)
    assert abz.nsyms(cbz) == 48 and cbz.lims == abz.TetrahedralLimits(np.full(3, 0.5))
    with pytest.raises(ValueError):
        abz.load_bz(abz.FBZ(), A, 2 * A)


def test_gk15_host_entry_points_match_oracle(abz):
    from autobzcore.jl_amd import _lib as L
    x = np.empty(15)
    L.check(L.lib().abz_gk15_nodes(0.25, 1.5, x.ctypes.data_as(L.c_f64p)))
    assert np.array_equal(x, orc.gk_nodes(0.25, 1.5))
    rng = np.random.default_rng(0)
    vals = rng.standard_normal((3, 15, 2)) + 1j * rng.standard_normal((3, 15, 2))
    ab = np.array([[0.0, 1.0], [0.5, 0.75], [-1.0, 2.0]])
    I = np.empty((3, 2, 2))
    E = np.empty(3)
    L.check(L.lib().abz_gk15_batch(ab.ctypes.data_as(L.c_f64p), np.ascontiguousarray(vals).view(np.float64).ctypes.data_as(L.c_f64p),
                                   3, 2, I.ctypes.data_as(L.c_f64p), E.ctypes.data_as(L.c_f64p)))
    for p in range(3):
        Io, Eo = orc.gk_evalrule(vals[p], ab[p, 0], ab[p, 1])
        assert np.allclose(I[p].view(np.complex128).reshape(2), Io, rtol=1e-15, atol=1e-15)
        assert abs(E[p] - Eo) <= 1e-15 * max(1.0, Eo)


def test_parameters_and_batchparam(abz):
    # ref: test/brillouin.jl:46-60 (MixedParameters merge rules), src/interfaces.jl:199-208
    p = abz.MixedParameters(1, 2)
    q = abz.MixedParameters(a="a", b="b")
    for pq in (p.merge(q), p.merge(dict(a="a", b="b")), q.merge((1, 2))):
        assert pq[0] == 1 and pq[1] == 2 and pq.a == "a" and pq.b == "b"
    assert p.merge(3)[2] == 3 and q.merge(3)[0] == 3
    assert p.merge(dict(a="c")).a == "c" and q.merge(dict(a="c")).a == "c"
    groups = abz.batchparam(list(range(7)), 3)
    assert [[i[0] for i, _ in g] for g in groups] == orc.batchparam(7, 3)
    ps = abz.paramzip([1, 2, 3], b=[4, 5, 6])
    assert ps[1][0] == 2 and ps[1].b == 5
    pp = abz.paramproduct([1, 2], b=[4, 5, 6])
    assert pp.shape == (2, 3) and pp[1, 2][0] == 2 and pp[1, 2].b == 6


def test_autoptr_sequence_defaults(abz):
    assert abz.AutoPTR().npt_sequence() == orc.npt_sequence_params() == (50, 50)


def test_w90_reader(abz):
    s = abz.load_w90_series(os.path.join(ROOT, "tests", "golden", "svo_hr.dat.gz"))
    assert s.c.shape == (11, 11, 11, 3, 3) and s.first == (-5, -5, -5)
    c = s.c
    flip = c[::-1, ::-1, ::-1]
    assert np.abs(c - np.conj(np.swapaxes(flip, -1, -2))).max() == 0.0  # H(-R) = H(R)^dagger exactly
    assert np.allclose(np.diag(c[6, 5, 5]).real, [-0.255871, -0.026000, -0.255871])
    # known eigenvalues through the ORACLE: pins the fixture + reader
    so = orc.FourierSeries(c, period=1.0, first=s.first, ndim=3)
    e = np.linalg.eigvalsh(orc.evaluate(so, [0.1, 0.2, 0.3]))
    assert np.abs(e - [12.351303, 12.824980, 12.909626]).max() < 2e-6


def test_w90_wout_reader(abz):
    # ref: ext/WannierIOExt.jl:12-17 + aps_example/aps_example.jl:25 (load_bz(CubicSymIBZ(), "svo.wout"))
    path = os.path.join(ROOT, "tests", "golden", "svo.wout.gz")
    A, B = abz.read_w90_wout(path)
    assert np.allclose(A, 3.858560 * np.eye(3)) and np.allclose(B, 1.628376 * np.eye(3))
    bz = abz.load_bz(abz.CubicSymIBZ(), path)
    assert abz.nsyms(bz) == 48 and abs(abs(np.linalg.det(bz.B)) - 4.31781) < 1e-4


def test_hchebinterp_driver(abz):
    """The adaptive-in-omega driver of aps_example/aps_example.jl:36-39 on an analytic stand-in:
    a Lorentzian comb (what a DOS with eta = 0.05 looks like), atol 1e-4; every new panel level is one
    batch call."""
    calls = []
    f = lambda w: sum(0.05 / ((w - e) ** 2 + 0.05**2) for e in (10.7, 12.2, 12.25, 14.1)) / np.pi

    def batch(xs):
        calls.append(len(xs))
        return np.array([f(x) for x in xs])
    itp = abz.hchebinterp(None, 10.0, 15.0, atol=1e-4, batch=batch)
    x = np.linspace(10, 15, 2001)
    assert np.abs(itp(x) - np.array([f(t) for t in x])).max() < 5e-4
    assert itp.numevals == sum(calls) and calls[0] == 16 and all(c % 16 == 0 for c in calls)
    assert abs(itp(12.2) - f(12.2)) < 5e-4 and np.ndim(itp(12.2)) == 0


def test_synthetic_models_match_the_oracle_recipe():
    """SURVEY 8d synthetic inputs: the product-side generators (bench / configs 2 and 5) draw the same
    splitmix64 stream in the same order as the oracle's restatement -> identical coefficients."""
    import itertools
    import autobzcore.jl_amd as abz
    g = orc.splitmix64(20240601)
    assert np.array_equal(abz.splitmix64_uniform(20240601, 64), np.array(list(itertools.islice(g, 64))))
    for kw in (dict(n=5, rmax=2, seed=7), dict(n=16, rmax=6, seed=20240601)):
        a, b = orc.synthetic_wannier(**kw), abz.synthetic_wannier(**kw)
        assert np.array_equal(a.c, b.c) and tuple(np.atleast_1d(b.first)) == tuple(np.atleast_1d(a.first))
    assert np.array_equal(orc.tb_integer(3, 1.5).c, abz.tb_integer(3, 1.5).c)


def test_sweep_archive_layout_and_partial_flush(tmp_path):
    """ref: ext/HDF5Ext.jl:123-158 -- data sets I, E, t, retcode, numevals, args/<j>, kwargs/<name>; the
    archive on disk always holds every chunk finished so far."""
    import autobzcore.jl_amd as abz
    from autobzcore.jl_amd.solver import IntegralSolution
    path = tmp_path / "sweep.npz"
    seen = []

    def fake_batchsolve(solver, part, callback=None):  # stands in for the fused device sweep
        vals = []
        for i, p in enumerate(part):
            sol = IntegralSolution(p.args[0] ** 2 + 1j * p.kwargs["b"], 1e-4, True, 15)
            callback(solver, (i,), i + 1, p, sol, 0.01)
            vals.append(sol.u)
        if path.exists():
            seen.append(int(abz.SweepArchive.load(path)["done"].sum()))
        return np.array(vals)

    ps = [abz.MixedParameters(float(w), b=0.5 * k) for k, w in enumerate(np.linspace(0, 1, 10))]
    out = abz.batchsolve_archive(path, None, ps, chunk=4, solve=fake_batchsolve)
    z = abz.SweepArchive.load(path)
    assert seen == [4, 8]  # after chunks 1 and 2, before the last one
    assert z["done"].all() and np.array_equal(z["I"], out) and np.array_equal(z["numevals"], np.full(10, 15))
    assert np.allclose(z["args/1"], np.linspace(0, 1, 10)) and np.allclose(z["kwargs/b"], 0.5 * np.arange(10))
    assert np.allclose(z["E"], 1e-4) and z["retcode"].dtype == np.int32 and np.all(z["retcode"] == 1)


def _build_c_client(tmp_path):
    """gcc -std=c11 on tests/c/abi_smoke.c against include/abzhip.h and the in-tree libabzhip.so."""
    import subprocess
    exe = tmp_path / "abi_smoke"
    libdir = os.path.join(ROOT, "autobzcore.jl_amd")
    cmd = ["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-o", str(exe), "-L", libdir, "-labzhip", "-lm",
           "-Wl,-rpath," + libdir]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    return exe


def test_header_is_plain_c_and_a_c_client_links(tmp_path):
    """The boundary is a C ABI: the header compiles as C11 with -Wall -Wextra -Werror and a plain-C program
    links against the library.  Without a GPU it must stop at abz_ctx_create with ABZ_ERR_NOGPU (exit 77)."""
    import subprocess
    exe = _build_c_client(tmp_path)
    run = subprocess.run([str(exe)], capture_output=True, text=True)
    assert run.returncode in (0, 77), run.stdout + run.stderr
    if run.returncode == 77:
        assert "no CPU fallback" in run.stderr
