"""Brillouin zones, symmetries and iterated limits.  ref: src/brillouin.jl, src/domains.jl."""
import itertools
import warnings
from dataclasses import dataclass
from typing import Optional

import numpy as np


@dataclass
class CubicLimits:
    """ref: IteratedIntegration.CubicLimits(a, b) as built at src/brillouin.jl:2-5,267."""

    a: np.ndarray
    b: np.ndarray

    def __post_init__(self):
        self.a = np.atleast_1d(np.asarray(self.a, dtype=np.float64))
        self.b = np.atleast_1d(np.asarray(self.b, dtype=np.float64))

    @property
    def ndim(self):
        return len(self.a)

    def __eq__(self, o):
        return isinstance(o, CubicLimits) and np.array_equal(self.a, o.a) and np.array_equal(self.b, o.b)


@dataclass
class TetrahedralLimits:
    """0 <= x_1 <= ... <= x_d <= a_d (scaled).  ref: src/brillouin.jl:304."""

    a: np.ndarray

    def __post_init__(self):
        self.a = np.atleast_1d(np.asarray(self.a, dtype=np.float64))

    @property
    def ndim(self):
        return len(self.a)

    def __eq__(self, o):
        return isinstance(o, TetrahedralLimits) and np.array_equal(self.a, o.a)


@dataclass
class Basis:
    """PTR domain: columns of B span the cell.  ref: AutoSymPTR.Basis, src/brillouin.jl:10."""

    B: np.ndarray

    def __post_init__(self):
        self.B = np.atleast_2d(np.asarray(self.B, dtype=np.float64))

    @property
    def ndim(self):
        return self.B.shape[0]


@dataclass
class PuncturedInterval:
    """ref: src/domains.jl (segments of a 1-D domain)."""

    segs: tuple


@dataclass
class HyperCube:
    a: np.ndarray
    b: np.ndarray


class SymmetricBZ:
    """ref: src/brillouin.jl:33-46.  All limits/symmetries are in the reciprocal lattice basis."""

    def __init__(self, A, B, lims, syms):
        self.A = np.atleast_2d(np.asarray(A, dtype=np.float64))
        self.B = np.atleast_2d(np.asarray(B, dtype=np.float64))
        self.lims = lims
        self.syms = None if syms is None else [np.asarray(S) for S in syms]

    @property
    def ndim(self):
        return self.A.shape[0]

    def __repr__(self):
        return f"{self.ndim}-dimensional Brillouin zone with " + ("trivial" if self.syms is None else str(nsyms(self))) + " symmetries"


def nsyms(bz: SymmetricBZ) -> int:
    """ref: src/brillouin.jl:43-46."""
    return 1 if bz.syms is None else len(bz.syms)


class AbstractBZ:
    def __init__(self, n=None):
        self.n = n


class FBZ(AbstractBZ):
    """ref: src/brillouin.jl:205-212."""


class InversionSymIBZ(AbstractBZ):
    """ref: src/brillouin.jl:260-270."""


class CubicSymIBZ(AbstractBZ):
    """ref: src/brillouin.jl:297-307."""


class IBZ(AbstractBZ):
    """ref: src/brillouin.jl:220-244 -- needs SymmetryReduceBZ (out of the hot-path scope)."""


def canonical_reciprocal_basis(A):
    """B = A' \\ 2 pi I.  ref: src/brillouin.jl:9."""
    A = np.atleast_2d(np.asarray(A, dtype=np.float64))
    return np.linalg.solve(A.T, 2 * np.pi * np.eye(A.shape[0]))


def sign_flip_matrices(d):
    """ref: src/brillouin.jl:248-249 (Iterators.product: first factor fastest)."""
    return [np.diag(rev[::-1]).astype(np.int64) for rev in itertools.product((1, -1), repeat=d)]


def permutation_matrices(d):
    """ref: src/brillouin.jl:272-277."""
    out = []
    for p in itertools.permutations(range(d)):
        P = np.zeros((d, d), dtype=np.int64)
        for i in range(d):
            P[i, p[i]] = 1
        out.append(P)
    return out


def cube_automorphisms(d):
    """ref: src/brillouin.jl:286 (S*P for S in sign flips, P in permutations)."""
    return [S @ P for P in permutation_matrices(d) for S in sign_flip_matrices(d)]


def load_bz(bz: AbstractBZ, A=None, B=None, atol=None) -> SymmetricBZ:
    """ref: src/brillouin.jl:179-212,264-307; `A` may be the path of a Wannier90 `seedname.wout`
    (ext/WannierIOExt.jl:12-17, default atol 1e-5 for the printed 6-digit lattice)."""
    if isinstance(A, (str, bytes)) or hasattr(A, "__fspath__"):
        from .io_w90 import read_w90_wout
        A, B = read_w90_wout(A)
        atol = 1e-5 if atol is None else atol
    if A is None:
        if bz.n is None:
            raise ValueError("BZ dimension must be integer")
        A = np.eye(bz.n)
    A = np.atleast_2d(np.asarray(A, dtype=np.float64))
    d = A.shape[0]
    if A.shape[0] != A.shape[1]:
        raise ValueError("Bravais lattice must be square")
    if bz.n is not None and bz.n != d:
        raise ValueError(f"BZ dimension {bz.n} does not match the lattice ({d})")
    if B is None:
        B = canonical_reciprocal_basis(A)
    B = np.atleast_2d(np.asarray(B, dtype=np.float64))
    if B.shape != A.shape:
        raise ValueError(f"Bravais lattices {A} and {B} must have the same shape")
    tol = np.sqrt(np.finfo(float).eps) if atol is None else atol
    if np.linalg.norm(A.T @ B - 2 * np.pi * np.eye(d)) >= tol:
        raise ValueError(f"Real and reciprocal Bravais lattice bases non-orthogonal to tolerance {tol}")
    if isinstance(bz, FBZ):
        return SymmetricBZ(A, B, CubicLimits(np.zeros(d), np.ones(d)), None)
    G = A.T @ A
    orthog = np.allclose(G, np.diag(np.diag(G)))
    if isinstance(bz, InversionSymIBZ):
        if not orthog:
            warnings.warn("Non-orthogonal lattice vectors detected with InversionSymIBZ. Unexpected behavior may occur")
        return SymmetricBZ(A, B, CubicLimits(np.zeros(d), np.full(d, 0.5)), sign_flip_matrices(d))
    if isinstance(bz, CubicSymIBZ):
        if not orthog:
            warnings.warn("Non-orthogonal lattice vectors detected with CubicSymIBZ. Unexpected behavior may occur")
        return SymmetricBZ(A, B, TetrahedralLimits(np.full(d, 0.5)), cube_automorphisms(d))
    if isinstance(bz, IBZ):
        raise NotImplementedError("SymmetryReduceBZ extension not loaded (IBZ is outside the hot-path scope)")
    raise TypeError(f"unknown BZ kind {bz!r}")
