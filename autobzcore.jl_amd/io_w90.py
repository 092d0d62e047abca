"""Wannier90 `_hr.dat` reader -> dense H_R / degeneracy array (the step before the hot path).
ref: aps_example/aps_example.jl:5-21 (WannierIO.read_w90_hrdat + OffsetArray fill)."""
import gzip

import numpy as np

from .series import FourierSeries


def read_w90_hrdat(path):
    """Returns (H_R [M1,M2,M3,n,n] complex with H/deg, first (Rmin per dim))."""
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rt") as fh:
        fh.readline()
        n = int(fh.readline().split()[0])
        nr = int(fh.readline().split()[0])
        degs = []
        while len(degs) < nr:
            degs.extend(int(t) for t in fh.readline().split())
        body = np.loadtxt(fh)
    if body.shape != (nr * n * n, 7):
        raise ValueError(f"{path}: expected {nr * n * n} hopping lines, found {body.shape}")
    R = body[:, :3].astype(np.int64)
    mi = body[:, 3].astype(np.int64) - 1
    ni = body[:, 4].astype(np.int64) - 1
    val = body[:, 5] + 1j * body[:, 6]
    Rmin, Rmax = R.min(axis=0), R.max(axis=0)
    shape = tuple(Rmax - Rmin + 1)
    H = np.zeros(shape + (n, n), dtype=np.complex128)
    deg = np.repeat(np.asarray(degs, dtype=np.float64), n * n)
    idx = R - Rmin
    H[idx[:, 0], idx[:, 1], idx[:, 2], mi, ni] = val / deg
    return H, tuple(int(v) for v in Rmin)


def load_w90_series(path, period=1.0):
    H, first = read_w90_hrdat(path)
    return FourierSeries(H, period=period, first=first, ndim=3)


def read_w90_wout(path):
    """Lattice (columns a_1..a_3, Angstrom) and reciprocal lattice (columns b_1..b_3, 1/Angstrom) from a
    Wannier90 `seedname.wout`.  ref: WannierIO.read_wout as used by ext/WannierIOExt.jl:12-16."""
    opener = gzip.open if str(path).endswith(".gz") else open
    A = np.zeros((3, 3))
    B = np.zeros((3, 3))
    got = set()
    with opener(path, "rt") as fh:
        for line in fh:
            t = line.split()
            if len(t) == 4 and t[0] in ("a_1", "a_2", "a_3", "b_1", "b_2", "b_3"):
                tgt = A if t[0][0] == "a" else B
                tgt[:, int(t[0][2]) - 1] = [float(v) for v in t[1:]]
                got.add(t[0])
            if len(got) == 6:
                break
    if len(got) != 6:
        raise ValueError(f"{path}: lattice vectors not found")
    return A, B
