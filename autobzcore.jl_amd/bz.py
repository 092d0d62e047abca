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

    # iterated-limits protocol (IteratedIntegration.segments / fixandeliminate): break points of the
    # outermost variable, and the limits of the remaining ones once it is fixed
    def segs(self):
        return (float(self.a[-1]), float(self.b[-1]))

    def fix(self, x):
        return CubicLimits(self.a[:-1], self.b[:-1])


@dataclass
class TetrahedralLimits:
    """0 <= x_1 <= ... <= x_d <= a_d (scaled).  ref: src/brillouin.jl:304."""

    a: np.ndarray
    s: float = 1.0  # scale left by the variables fixed so far

    def __post_init__(self):
        self.a = np.atleast_1d(np.asarray(self.a, dtype=np.float64))

    @property
    def ndim(self):
        return len(self.a)

    def __eq__(self, o):
        return isinstance(o, TetrahedralLimits) and np.array_equal(self.a, o.a) and self.s == o.s

    def segs(self):
        return (0.0, float(self.a[-1]) * self.s)

    def fix(self, x):
        return TetrahedralLimits(self.a[:-1], x / float(self.a[-1]))


def _unique_sorted(vals):
    """Distinct coordinates (sqrt(eps) tolerances, first occurrence kept), ascending.
    ref: get_segs, ext/SymmetryReduceBZExt.jl:15-31."""
    tol = float(np.sqrt(np.finfo(float).eps))
    uniq = []
    for v in vals:
        if not any(abs(v - u) <= max(tol, tol * max(abs(v), abs(u))) for u in uniq):
            uniq.append(float(v))
    return tuple(sorted(uniq))


class PolygonLimits:
    """Convex polygon in (x, y): verts [nv, 2] in order around the boundary (IAI over a 2-D zone, and the
    z-slices of PolyhedralLimits).  ref: Polygon2, ext/SymmetryReduceBZExt.jl:43-59."""

    def __init__(self, verts):
        self.verts = np.ascontiguousarray(np.asarray(verts, dtype=np.float64).reshape(-1, 2))

    @property
    def ndim(self):
        return 2

    def segs(self):
        return _unique_sorted(self.verts[:, 1])

    def packed(self):
        return self.verts.reshape(-1).copy()

    def fix(self, y):
        v, nv = self.verts, len(self.verts)
        hits = []
        for j in range(nv):
            y1, y2 = v[j, 1], v[(j + 1) % nv, 1]
            if (y1 < y and y2 > y) or (y1 > y and y2 < y) or y1 == y:
                t = (y - y1) / (y2 - y1) if y2 != y1 else 0.0
                hits.append(t * v[(j + 1) % nv, 0] + (1 - t) * v[j, 0])
                if len(hits) == 2:
                    break
        if not hits:
            raise ValueError("could not find intersection with polygon")
        return CubicLimits(np.array([min(hits)]), np.array([max(hits)]))


class PolyhedralLimits:
    """Convex polyhedron in (x, y, z) by its faces (each [nv, 3], vertices in order around the face, shared
    vertices bit-identical) -- the irreducible Brillouin zone of a general lattice, in the coordinates of
    the reciprocal basis.  ref: Polyhedron3, ext/SymmetryReduceBZExt.jl:33-58, ext/ibzlims.jl:198-243."""

    def __init__(self, faces):
        self.faces = [np.ascontiguousarray(np.asarray(f, dtype=np.float64).reshape(-1, 3)) for f in faces]

    @classmethod
    def from_vertices(cls, vertices):
        """Faces of the convex hull of `vertices` (coplanar hull triangles merged, vertices ordered around
        each face), like get_uniquefacets of the reference's SymmetryReduceBZ extension."""
        from scipy.spatial import ConvexHull
        V = np.asarray(vertices, dtype=np.float64)
        hull = ConvexHull(V)
        groups = []  # (unit normal with offset, set of vertex indices)
        for eq, simp in zip(hull.equations, hull.simplices):
            for g in groups:
                if np.allclose(g[0], eq, atol=1e-9):
                    g[1].update(int(i) for i in simp)
                    break
            else:
                groups.append((eq, set(int(i) for i in simp)))
        faces = []
        for eq, idx in groups:
            idx = sorted(idx)
            P = V[idx]
            nrm = eq[:3]
            c = P.mean(axis=0)
            u = P[0] - c
            u = u / np.linalg.norm(u)
            w = np.cross(nrm, u)
            ang = np.arctan2((P - c) @ w, (P - c) @ u)
            faces.append(P[np.argsort(ang, kind="stable")])
        return cls(faces)

    @property
    def ndim(self):
        return 3

    def segs(self):
        return _unique_sorted(np.concatenate([f[:, 2] for f in self.faces]))

    def packed(self):
        return np.concatenate([np.concatenate([[float(len(f))], f.reshape(-1)]) for f in self.faces])

    def fix(self, z):
        pts = []
        for face in self.faces:
            nv = len(face)
            for j in range(nv):
                p1, p2 = face[j], face[(j + 1) % nv]
                z1, z2 = p1[2], p2[2]
                if ((z1 <= z and z2 >= z) or (z1 >= z and z2 <= z)) and z2 != z1:
                    t = (z - z1) / (z2 - z1)
                    q = (t * p2[0] + (1 - t) * p1[0], t * p2[1] + (1 - t) * p1[1])
                    if q not in pts:
                        pts.append(q)
        P = np.array(pts)
        c = P.mean(axis=0)
        return PolygonLimits(P[np.argsort(np.arctan2(P[:, 1] - c[1], P[:, 0] - c[0]), kind="stable")])

    def volume(self):
        from scipy.spatial import ConvexHull
        return float(ConvexHull(np.concatenate(self.faces)).volume)


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
    """ref: src/brillouin.jl:220-244.  The reference computes the zone with SymmetryReduceBZ.jl from the
    crystal structure (ext/SymmetryReduceBZExt.jl:85-123); that geometry package is outside this build, so
    the irreducible zone is given explicitly: `load_bz(IBZ(), A, hull=vertices, syms=point_group)` with the
    vertices of the convex zone and the point-group operations, both in the reciprocal-lattice basis."""


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


def load_bz(bz: AbstractBZ, A=None, B=None, atol=None, hull=None, syms=None) -> SymmetricBZ:
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
        if hull is None or syms is None:
            raise NotImplementedError("IBZ(): SymmetryReduceBZ is not part of this build -- pass the zone explicitly: "
                                      "load_bz(IBZ(), A, hull=vertices, syms=point_group)")
        V = np.asarray(hull, dtype=np.float64)
        if V.ndim != 2 or V.shape[1] != d or d not in (2, 3):
            raise ValueError("hull: [nv, d] vertices of the irreducible zone, d = 2 or 3")
        if d == 3:
            lims = PolyhedralLimits.from_vertices(V)
        else:
            from scipy.spatial import ConvexHull
            lims = PolygonLimits(V[ConvexHull(V).vertices])  # counter-clockwise
        return SymmetricBZ(A, B, lims, [np.asarray(S) for S in syms])
    raise TypeError(f"unknown BZ kind {bz!r}")
