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
