"""Synthetic tight-binding / Wannier models of the benchmark configurations (BASELINE.json configs,
SURVEY 8d "Synthetic inputs").  Language-neutral RNG: splitmix64(seed) -> u = (x >> 11) 2^-53, value
2u - 1, drawn in a fixed order, so that any host language generates the same coefficients."""
import math

import numpy as np

from .series import FourierSeries

_G = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64_uniform(seed: int, count: int) -> np.ndarray:
    """First `count` outputs u_i in [0, 1) of splitmix64(seed), vectorised: the i-th state is
    seed + (i + 1) * gamma (mod 2^64)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed & ((1 << 64) - 1)) + _G * np.arange(1, count + 1, dtype=np.uint64)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * 2.0**-53


def tb_integer(d: int, t: float = 1.0) -> FourierSeries:
    """Config 2: one-band nearest-neighbour model H(k) = 2 t sum_i cos(2 pi k_i) on the integer
    lattice (the model of the reference's DOS tests, test/dos.jl:34-41)."""
    c = np.zeros((3,) * d + (1, 1))
    for i in range(d):
        for end in (0, 2):
            idx = [1] * d
            idx[i] = end
            c[tuple(idx) + (0, 0)] = t
    return FourierSeries(c, period=1.0, first=-1, ndim=d)


def synthetic_wannier(n: int = 16, rmax: int = 6, seed: int = 20240601, scale: float = 0.25,
                      decay: float = 1.5) -> FourierSeries:
    """Config 5: n-band 3-D model on R in [-rmax, rmax]^3 ((2 rmax + 1)^3 = 2197 vectors).  For R = 0 and
    every R in the lexicographic upper half space, in lexicographic order, draw A_R[m, k] =
    (2u - 1) + i (2u' - 1) row by row, scale by `scale * exp(-|R|_2 / decay)`; H_R = A_R,
    H_{-R} = A_R^dagger; H_0 = (A_0 + A_0^dagger) / 2 + diag(linspace(-1, 1, n))."""
    m = 2 * rmax + 1
    ax = np.arange(-rmax, rmax + 1)
    R = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), axis=-1).reshape(-1, 3)  # lexicographic order
    flat = (R[:, 0] * m + R[:, 1]) * m + R[:, 2]
    keep = R[flat >= 0]  # R = 0 and the upper half space, still in lexicographic order
    u = splitmix64_uniform(seed, len(keep) * n * n * 2).reshape(len(keep), n, n, 2)
    damp = np.array([scale * math.exp(-math.sqrt(float(r @ r)) / decay) for r in keep])  # libm, one value per R
    A = ((2 * u[..., 0] - 1) + 1j * (2 * u[..., 1] - 1)) * damp[:, None, None]
    c = np.zeros((m, m, m, n, n), dtype=np.complex128)
    i = keep + rmax
    j = -keep + rmax
    c[j[:, 0], j[:, 1], j[:, 2]] = A.conj().transpose(0, 2, 1)
    c[i[:, 0], i[:, 1], i[:, 2]] = A
    z = len(keep) - 1 - np.argmax((keep[::-1] == 0).all(1))  # position of R = 0
    c[rmax, rmax, rmax] = (A[z] + A[z].conj().T) / 2 + np.diag(np.linspace(-1, 1, n))
    return FourierSeries(c, period=1.0, first=-rmax, ndim=3)
