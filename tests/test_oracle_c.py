"""The C restatement (oracle/abz_oracle.c, the cpu_baseline 'port') against the numpy oracle."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import abz_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def clib():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    return ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))


@pytest.mark.parametrize("d,n,npt", [(1, 2, 9), (2, 3, 8), (3, 3, 7), (3, 5, 4)])
def test_c_oracle_ptr_and_dos(clib, d, n, npt):
    rng = np.random.default_rng(d * 10 + n)
    dims = (3, 5, 3)[:d]
    c = rng.standard_normal(dims + (n, n)) + 1j * rng.standard_normal(dims + (n, n))
    flip = c[tuple(slice(None, None, -1) for _ in dims)]
    c = 0.5 * (c + np.conj(np.swapaxes(flip, -1, -2)))
    first = tuple(-(m // 2) for m in dims)
    so = orc.FourierSeries(c, period=1.0, first=first, ndim=d)
    axes = tuple(range(d - 1, -1, -1)) + (d + 1, d)
    coef = np.ascontiguousarray(np.transpose(c, axes)).reshape(-1)
    nk = npt**d
    vals = np.empty(nk * n * n, dtype=np.complex128)
    eig = np.empty(nk * n)
    P = ctypes.c_void_p
    dm = np.array(dims, dtype=np.int32)
    fr = np.array(first, dtype=np.int32)
    clib.orc_fourier_ptr(coef.ctypes.data_as(P), d, dm.ctypes.data_as(P), fr.ctypes.data_as(P), n, npt,
                         vals.ctypes.data_as(P), eig.ctypes.data_as(P))
    ref = orc.fourier_ptr(so, npt)
    perm = tuple(range(d - 1, -1, -1))
    ref = np.transpose(ref, perm + (d, d + 1)).reshape(nk, n, n)
    got = vals.reshape(nk, n, n).transpose(0, 2, 1)
    assert np.abs(got - ref).max() < 1e-12 * np.abs(ref).max()
    assert np.abs(eig.reshape(nk, n) - np.linalg.eigvalsh(ref, UPLO="U")).max() < 1e-11 * np.abs(ref).max()
    omegas = np.array([-0.7, 0.4])
    out = np.empty(2)
    clib.orc_dos_scan.argtypes = [P, ctypes.c_int64, ctypes.c_int, ctypes.c_double, P, ctypes.c_int, P]
    clib.orc_dos_scan(vals.ctypes.data_as(P), nk, n, 0.3, omegas.ctypes.data_as(P), 2, out.ctypes.data_as(P))
    for w, o in zip(omegas, out):
        r, _ = orc._ptr_rule_sum(so, npt, None, orc.f_dos(0.3, w))
        assert abs(o - r) < 1e-11 * abs(r)


def test_c_oracle_fixed_size_3band_paths(clib):
    """orc_fourier_ptr3 / orc_dos_scan3 / orc_eig3_closed (the bench's cpu_baseline: StaticArrays-style closed
    forms) against the generic C loops, LAPACK and the numpy oracle, incl. degenerate spectra."""
    rng = np.random.default_rng(7)
    d, n, npt = 3, 3, 6
    dims = (3, 5, 3)
    c = rng.standard_normal(dims + (n, n)) + 1j * rng.standard_normal(dims + (n, n))
    flip = c[::-1, ::-1, ::-1]
    c = 0.5 * (c + np.conj(np.swapaxes(flip, -1, -2)))
    first = tuple(-(m // 2) for m in dims)
    coef = np.ascontiguousarray(np.transpose(c, (2, 1, 0, 4, 3))).reshape(-1)
    nk = npt**d
    P = ctypes.c_void_p
    dm = np.array(dims, dtype=np.int32)
    fr = np.array(first, dtype=np.int32)
    res = {}
    for name in ("orc_fourier_ptr", "orc_fourier_ptr3"):
        vals = np.empty(nk * n * n, dtype=np.complex128)
        eig = np.empty(nk * n)
        getattr(clib, name)(coef.ctypes.data_as(P), d, dm.ctypes.data_as(P), fr.ctypes.data_as(P), n, npt,
                            vals.ctypes.data_as(P), eig.ctypes.data_as(P))
        res[name] = (vals, eig)
    assert np.abs(res["orc_fourier_ptr3"][0] - res["orc_fourier_ptr"][0]).max() < 1e-13 * np.abs(res["orc_fourier_ptr"][0]).max()
    H = res["orc_fourier_ptr3"][0].reshape(nk, n, n).transpose(0, 2, 1)
    scale = np.abs(H).max()
    assert np.abs(res["orc_fourier_ptr3"][1].reshape(nk, n) - np.linalg.eigvalsh(H, UPLO="U")).max() < 1e-12 * scale
    omegas = np.array([-0.7, 0.4, 1.9])
    outs = []
    for name in ("orc_dos_scan", "orc_dos_scan3"):
        fn = getattr(clib, name)
        fn.argtypes = [P, ctypes.c_int64, ctypes.c_int, ctypes.c_double, P, ctypes.c_int, P]
        out = np.empty(3)
        fn(res["orc_fourier_ptr3"][0].ctypes.data_as(P), nk, n, 0.3, omegas.ctypes.data_as(P), 3, out.ctypes.data_as(P))
        outs.append(out)
    assert np.abs(outs[0] - outs[1]).max() < 1e-13 * np.abs(outs[0]).max()
    so = orc.FourierSeries(c, period=1.0, first=first, ndim=d)
    for w, o in zip(omegas, outs[1]):
        r, _ = orc._ptr_rule_sum(so, npt, None, orc.f_dos(0.3, w))
        assert abs(o - r) < 1e-11 * abs(r)
    # closed-form eigenvalues on degenerate / clustered spectra (the accuracy of the trigonometric form
    # degrades like sqrt(eps) relative to the spread only for the clustered pair: StaticArrays has the same property)
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)) + 1j * rng.standard_normal((3, 3)))
    for lam in ([1.0, 1.0, 1.0], [-1.0, 2.0, 2.0], [0.0, 1e-3, 1.0], [-3.0, 0.5, 4.0]):
        A = (q * np.array(lam)) @ q.conj().T
        A = 0.5 * (A + A.conj().T)
        a = np.ascontiguousarray(A.T).reshape(-1)  # column-major
        e = np.empty(3)
        clib.orc_eig3_closed(a.ctypes.data_as(P), e.ctypes.data_as(P))
        assert np.abs(e - np.sort(lam)).max() < 2e-7


def _herm_series(rng, dims, n, scale=1.0):
    c = rng.standard_normal(dims + (n, n)) + 1j * rng.standard_normal(dims + (n, n))
    flip = c[tuple(slice(None, None, -1) for _ in dims)]
    c = 0.5 * scale * (c + np.conj(np.swapaxes(flip, -1, -2)))
    first = tuple(-(m // 2) for m in dims)
    d = len(dims)
    coef = np.ascontiguousarray(np.transpose(c, tuple(range(d - 1, -1, -1)) + (d + 1, d))).reshape(-1)
    return c, first, coef


@pytest.mark.parametrize("n,threads", [(3, 1), (3, 3), (5, 2)])
def test_c_oracle_iai_is_the_numpy_oracles_panel_tree(clib, n, threads):
    """orc_iai_dos3 (the nested GK(7,15) loop timed as the CPU baseline of the IAI legs; ref src/fourier.jl:432-510) against
    oracle/abz_oracle.py::solve_iai: the same value to rounding and the SAME number of integrand evaluations, whatever
    the number of threads the outermost nodes are dealt to."""
    rng = np.random.default_rng(40 + n)
    dims = (3, 3, 3)
    c, first, coef = _herm_series(rng, dims, n, scale=0.5)
    so = orc.FourierSeries(c, period=1.0, first=first, ndim=3)
    eta, omega, abstol = 0.5, 0.3, 5e-4
    bz = orc.load_bz("FBZ", 2 * np.pi * np.eye(3))  # B = I: |det B| = 1, the nested tolerance is abstol itself
    assert abs(abs(np.linalg.det(bz.B)) - 1.0) < 1e-12
    ref = orc.solve_iai(so, bz, orc.f_dos(eta, omega), abstol=abstol)
    P = ctypes.c_void_p
    dm, fr = np.array(dims, dtype=np.int32), np.array(first, dtype=np.int32)
    lo, hi = np.zeros(3), np.ones(3)
    err, nev = ctypes.c_double(0), ctypes.c_int64(0)
    clib.orc_iai_dos3.restype = ctypes.c_double
    clib.orc_iai_dos3.argtypes = [P, P, P, ctypes.c_int, P, P, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                  ctypes.c_int64, P, P]
    clib.orc_set_threads(threads)
    got = clib.orc_iai_dos3(coef.ctypes.data_as(P), dm.ctypes.data_as(P), fr.ctypes.data_as(P), n, lo.ctypes.data_as(P),
                            hi.ctypes.data_as(P), eta, omega, abstol, -1.0, 2**62, ctypes.byref(err), ctypes.byref(nev))
    assert nev.value == ref.numevals and nev.value > 15**3
    assert abs(got - ref.u) <= 1e-12 * abs(ref.u)
    assert abs(err.value - ref.resid) <= 1e-9 * abs(ref.resid) + 1e-18


@pytest.mark.parametrize("n", [3, 4])
def test_c_oracle_ggr(clib, n):
    """orc_ggr_data + orc_sum_ggr3 (the CPU baseline of the GGR leg; ref src/dos_ggr.jl:14-65,90-104) against the numpy
    oracle's get_ggr_data / sum_ggr, with periods other than one."""
    rng = np.random.default_rng(50 + n)
    dims = (3, 5, 3)
    c, first, coef = _herm_series(rng, dims, n)
    period = (1.0, 2.0, 0.5)
    so = orc.FourierSeries(c, period=period, first=first, ndim=3)
    npt = 6
    w, e, v = orc.get_ggr_data(so, npt, None)
    nk = npt**3
    eig, vel = np.empty(nk * n), np.empty(nk * 3 * n)
    P = ctypes.c_void_p
    dm, fr, per = np.array(dims, dtype=np.int32), np.array(first, dtype=np.int32), np.array(period)
    clib.orc_ggr_data(coef.ctypes.data_as(P), dm.ctypes.data_as(P), fr.ctypes.data_as(P), n, npt, per.ctypes.data_as(P),
                      eig.ctypes.data_as(P), vel.ctypes.data_as(P))
    eig, vel = eig.reshape(nk, n), vel.reshape(nk, 3, n)
    scale, vscale = np.abs(e).max(), np.abs(v).max()
    assert np.abs(eig - e).max() <= 1e-11 * scale
    ok = np.min(np.diff(e, axis=1), axis=1) > 1e-5 * scale
    assert ok.mean() > 0.9
    assert np.abs(vel[ok] - v[ok]).max() <= 1e-8 * vscale
    assert np.abs(vel.sum(axis=2) - v.sum(axis=2)).max() <= 1e-9 * vscale * n
    Es = np.linspace(-3.0, 3.0, 7)
    out = np.empty(len(Es))
    clib.orc_sum_ggr3.argtypes = [ctypes.c_int, P, ctypes.c_int, ctypes.c_int64, ctypes.c_int, P, P, P]
    clib.orc_sum_ggr3(npt, Es.ctypes.data_as(P), len(Es), nk, n, eig.ctypes.data_as(P), vel.ctypes.data_as(P), out.ctypes.data_as(P))
    ref = np.array([orc.sum_ggr(3, npt, E, w, e, v) for E in Es])
    assert np.abs(ref).max() > 0
    assert np.abs(out - ref).max() <= 1e-8 * np.abs(ref).max()
