"""CPU oracle for the AutoBZCore.jl hot path -- TEST INFRASTRUCTURE ONLY.

This is a plain numpy / pure-Python restatement of the reference algorithm for the path
named by BASELINE.json's north_star (batched Fourier/Wannier interpolation + quadrature
panel evaluation).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import it; the product (autobzcore.jl_amd) never does.

Parity status: **pinned analytically**.  The reference (Julia, v0.3.8) cannot run in this
pipeline and holds no golden vectors; every pin used by tests/test_oracle_pins.py is a
known answer from the reference's own tests/docs:
  test/fourier.jl:40-56, test/brillouin.jl:15-44, test/interface_tests.jl:90-158,
  test/dos.jl:88-132, docs/src/examples.md:60,105, src/AutoBZCore.jl:14-17.
The polytope limits (PolyhedralLimits / PolygonLimits) restate ext/SymmetryReduceBZExt.jl:15-58 and
ext/ibzlims.jl:198-289, which ARE part of the reference tree; they are pinned by volumes, by the
tetrahedral special case and by exact slices (tests/test_oracle_pins.py).

The arithmetic of the path lives in un-vendored Julia packages (pins from Project.toml:36-53
and aps_example/Manifest.toml): FourierSeriesEvaluators 1.x, AutoSymPTR 0.4.x,
IteratedIntegration 0.5.x (AuxQuadGK = QuadGK 2.x's adaptive Gauss-Kronrod), QuadGK 2.6+,
DataStructures (binary heap).  Their published algorithms are restated here and anchored on
the reference's own call sites, cited per function as `ref: file:line` into /root/reference.
"""
from __future__ import annotations

import itertools
import math
from dataclasses import dataclass, field
from typing import Callable, Optional, Sequence

import numpy as np

# ----------------------------------------------------------------------------------------
# Fourier series (FourierSeriesEvaluators v1 semantics as used at src/fourier.jl:122,133-158)
# ----------------------------------------------------------------------------------------


class FourierSeries:
    """s(x) = sum_i c[i] * exp(2 pi i * sum_j (i_j + o_j) x_j / t_j), 1-based i_j (Julia).

    ref: docs/src/examples.md:26-42 (`FourierSeries([0.5,0,0.5]; period=1, offset=-2)` is
    cos(2 pi k)), test/dos.jl:114-129 (fields c, t, o), test/utils.jl:3-9 (OffsetArray axes).

    `c` has shape (M_1, ..., M_d) + (n, n) (matrix valued) or (M_1, ..., M_d) (scalar).
    `first[j]` is the integer frequency of c[0] along dim j, i.e. 1 + offset_j.
    """

    def __init__(self, c, period=1.0, offset=0, *, ndim=None, first=None):
        c = np.asarray(c)
        if ndim is None:
            # matrix valued iff the trailing two axes are square and there is >= 1 leading axis
            ndim = c.ndim - 2 if (c.ndim >= 3 and c.shape[-1] == c.shape[-2]) else c.ndim
        self.d = int(ndim)
        self.scalar = c.ndim == self.d
        if self.scalar:
            c = c.reshape(c.shape + (1, 1))
        assert c.ndim == self.d + 2 and c.shape[-1] == c.shape[-2]
        self.c = np.ascontiguousarray(c, dtype=np.complex128)
        self.n = c.shape[-1]
        self.t = tuple(float(p) for p in (period if np.ndim(period) else (period,) * self.d))
        if first is not None:
            self.first = tuple(int(v) for v in (first if np.ndim(first) else (first,) * self.d))
        else:
            off = offset if np.ndim(offset) else (offset,) * self.d
            self.first = tuple(int(o) + 1 for o in off)

    @property
    def dims(self):
        return self.c.shape[: self.d]

    def freqs(self, j):
        return self.first[j] + np.arange(self.dims[j])

    def _out(self, v):
        return v[..., 0, 0] if self.scalar else v


def phases(s: FourierSeries, j: int, x) -> np.ndarray:
    """e^{2 pi i m x / t_j} for every frequency m of dim j; x scalar or array -> (..., M_j)."""
    x = np.asarray(x, dtype=np.float64)
    return np.exp(2j * np.pi * (x[..., None] / s.t[j]) * s.freqs(j))


def contract(s: FourierSeries, x: float) -> FourierSeries:
    """Fix the LAST (outermost) variable: the (d-1)-dim series  c'[...] = sum_m c[..., m] ph_m.

    ref: workspace_contract!(w, x) as called at src/fourier.jl:152,158,242,252,468,478.
    """
    d = s.d
    ph = phases(s, d - 1, x)  # (M_d,)
    c = np.tensordot(ph, np.moveaxis(s.c, d - 1, 0), axes=(0, 0))
    out = FourierSeries(c, period=s.t[: d - 1] or 1.0, first=s.first[: d - 1] or 0, ndim=d - 1)
    out.scalar = s.scalar
    return out


def evaluate(s: FourierSeries, x) -> np.ndarray:
    """Full evaluation at one point x (len d), contracting the last variable first.

    ref: fallback evaluator src/fourier.jl:120-122 (`f.w(x)`), workspace_evaluate at :170,:269.
    """
    x = np.atleast_1d(np.asarray(x, dtype=np.float64))
    assert x.shape == (s.d,)
    cur = s
    for j in range(s.d - 1, 0, -1):
        cur = contract(cur, x[j])
    ph = phases(cur, 0, x[0])
    v = np.tensordot(ph, cur.c, axes=(0, 0))
    return s._out(v)


def evaluate_direct(s: FourierSeries, x) -> np.ndarray:
    """Naive O(M^d) sum over all R vectors; independent check of `evaluate`."""
    x = np.atleast_1d(np.asarray(x, dtype=np.float64))
    v = np.zeros((s.n, s.n), dtype=np.complex128)
    for idx in itertools.product(*[range(m) for m in s.dims]):
        ph = 1.0
        for j, i in enumerate(idx):
            ph = ph * np.exp(2j * np.pi * (s.first[j] + i) * x[j] / s.t[j])
        v += s.c[idx] * ph
    return s._out(v)


def evaluate_many(s: FourierSeries, xs) -> np.ndarray:
    """Values at arbitrary points xs (N, d) -> (N, n, n) (or (N,) for scalar series)."""
    xs = np.asarray(xs, dtype=np.float64).reshape(-1, s.d)
    cur = None
    for j in range(s.d - 1, -1, -1):
        ph = phases(s, j, xs[:, j])  # (N, M_j); always contracts the last remaining M axis
        if cur is None:
            cur = np.einsum("...mab,nm->n...ab", s.c, ph)
        else:
            cur = np.einsum("n...mab,nm->n...ab", cur, ph)
    return s._out(cur)


def derivative_series(s: FourierSeries, j: int) -> FourierSeries:
    """d/dx_j of the series: coefficient times 2 pi i (i_j+o_j)/t_j.

    ref: JacobianSeries(h) consumed as `h, V = x.s` at src/dos_ggr.jl:6-7,18,33.
    """
    fac = 2j * np.pi * s.freqs(j) / s.t[j]
    shape = [1] * (s.d + 2)
    shape[j] = -1
    out = FourierSeries(s.c * fac.reshape(shape), period=s.t, first=s.first, ndim=s.d)
    out.scalar = s.scalar
    return out


# ----------------------------------------------------------------------------------------
# PTR rules
# ----------------------------------------------------------------------------------------


def ptrpoints(npt: int) -> np.ndarray:
    """x_j = (j-1)/npt, j = 1..npt.  ref: AutoSymPTR.ptrpoints used at src/fourier.jl:268."""
    return np.arange(npt, dtype=np.float64) / npt


def fourier_ptr(s: FourierSeries, npt: int) -> np.ndarray:
    """vals[i1,...,id] = s(x_{i1},...,x_{id}) by contracting dim d, ..., 2 then evaluating dim 1.

    ref: fourier_ptr! src/fourier.jl:132-164 and FourierPTR ctor :166-174.  Returned array is
    indexed [i1, ..., id] (+ (n, n)); the reference's column-major order makes i1 fastest.
    """
    x = ptrpoints(npt)
    cur = s.c  # axes (M1..Md, n, n)
    d = s.d
    # contract outermost first: after step for dim j the axis j holds grid index i_j
    for j in range(d - 1, -1, -1):
        ph = np.exp(2j * np.pi * np.outer(x, s.freqs(j)))  # (npt, M_j); period*x/period
        cur = np.moveaxis(np.tensordot(ph, cur, axes=(1, j)), 0, j)
    return s._out(cur)


def symptr_rule(npt: int, d: int, syms: Sequence[np.ndarray]):
    """Integer symmetric-PTR tables (wsym, flags, nsym).

    ref: AutoSymPTR.symptr_rule called at src/fourier.jl:271 and consumed at :216-263 (SURVEY
    A.2).  wsym[i1..id] = size of the orbit of grid point i under `syms` (acting on fractional
    coordinates mod 1) stored at the orbit's first member in column-major order (i1 fastest),
    0 elsewhere.  flags[j] (j = 0..d-1) has shape (npt,)*(d-1-j)... here returned as a list of
    arrays where flags[j][i_{j+2}, ..., i_d] is the 1-based output offset of the first
    irreducible point of the slab with the trailing indices fixed (0 if the slab has none);
    flags[d-1] is 0-dim.  Output order = column-major order over irreducible points.
    """
    syms = [np.asarray(S) for S in syms]
    for S in syms:
        assert np.allclose(S, np.rint(S)), "grid symmetries must be integer matrices"
    symsi = np.stack([np.rint(S).astype(np.int64) for S in syms])  # (ns, d, d)
    N = npt**d
    strides = npt ** np.arange(d, dtype=np.int64)  # column-major: i1 fastest
    wflat = np.zeros(N, dtype=np.int64)
    chunk = 1 << 20
    symsT = np.ascontiguousarray(np.transpose(symsi, (0, 2, 1)))
    for lo in range(0, N, chunk):
        lin = np.arange(lo, min(N, lo + chunk), dtype=np.int64)
        v = (lin[:, None] // strides[None, :]) % npt  # (n, d) grid indices i1..id
        # first member of the orbit in column-major order = min over the images' linear indices
        rep = None
        for ST in symsT:
            il = ((v @ ST) % npt) @ strides
            rep = il if rep is None else np.minimum(rep, il)
        isrep = rep == lin
        vr = v[isrep]
        if len(vr):
            img = np.einsum("sab,nb->sna", symsi, vr) % npt  # (ns, nrep, d): orbits of the representatives only
            ilin = img @ strides
            ilin.sort(axis=0)
            wflat[lin[isrep]] = 1 + np.count_nonzero(np.diff(ilin, axis=0), axis=0)
    wsym = wflat.reshape((npt,) * d, order="F")
    nsym = int(np.count_nonzero(wflat))
    order_flat = np.where(wflat > 0, np.cumsum(wflat > 0), 0)  # 1-based output position
    order = order_flat.reshape((npt,) * d, order="F")
    flags = []
    big = np.iinfo(np.int64).max
    for j in range(d):
        # flags[j][i_{j+2}..i_d]: first output offset in the slab with those indices fixed
        if j == d - 1:
            f = np.array(1 if nsym else 0, dtype=np.int64)
        else:
            f = np.where(order > 0, order, big).min(axis=tuple(range(j + 1)))
            f = np.where(f == big, 0, f)
        flags.append(f)
    return wsym, flags, nsym


def fourier_symptr(s: FourierSeries, npt: int, syms):
    """Symmetry-reduced rule: list order/weights/nodes/values exactly as the reference fills wxs.

    ref: _fourier_symptr! src/fourier.jl:216-258 (skip slabs whose flag is 0, contract only the
    needed slabs, write at precomputed offsets), ctor :265-277.
    Returns (w[nsym] int64, x[nsym,d] float64, vals[nsym,n,n], idx[nsym,d] int64 0-based).
    """
    d = s.d
    wsym, flags, nsym = symptr_rule(npt, d, syms)
    u = ptrpoints(npt)
    w = np.zeros(nsym, dtype=np.int64)
    xs = np.zeros((nsym, d))
    idxs = np.zeros((nsym, d), dtype=np.int64)
    vals = np.zeros((nsym, s.n, s.n), dtype=np.complex128)

    def rec(cur: FourierSeries, level: int, idx_tail: tuple, offset: int):
        # level = number of variables still free (cur.d)
        if level == 1:
            n = 0
            o = offset - 1
            for i in range(npt):
                wi = wsym[(i,) + idx_tail]
                if wi == 0:
                    continue
                k = o + n
                n += 1
                w[k] = wi
                idxs[k] = (i,) + idx_tail
                xs[k] = [u[t] for t in idxs[k]]
                ph = phases(cur, 0, cur.t[0] * u[i])
                vals[k] = np.tensordot(ph, cur.c, axes=(0, 0))
            return
        f = flags[level - 2]
        for i in range(npt):
            fi = int(f[(i,) + idx_tail])
            if fi == 0:
                continue
            rec(contract(cur, cur.t[level - 1] * u[i]), level - 1, (i,) + idx_tail, fi)

    if nsym:
        rec(s, d, (), int(flags[d - 1]))
    return w, xs, (vals[:, 0, 0] if s.scalar else vals), idxs


def fourier_symptr_fast(s: FourierSeries, npt: int, syms):
    """Same output as fourier_symptr (order, weights, nodes, values) without the Python recursion:
    irreducible nodes from symptr_rule in column-major order, values by evaluate_many.  Used for the
    large grids of the GGR pins; equality with the traversal is tested on small grids."""
    d = s.d
    wsym, _, nsym = symptr_rule(npt, d, syms)
    wf = wsym.reshape(-1, order="F")
    lin = np.flatnonzero(wf)
    strides = npt ** np.arange(d, dtype=np.int64)
    idx = (lin[:, None] // strides[None, :]) % npt
    u = ptrpoints(npt)
    xs = u[idx]
    xe = xs * np.asarray(s.t)[None, :]  # the rule evaluates the series at period * x (ref: src/fourier.jl:133,149), like fourier_symptr
    vals = np.concatenate([np.asarray(evaluate_many(s, xe[i:i + 65536])) for i in range(0, len(xs), 65536)]) if len(xs) else np.zeros((0,))
    return wf[lin], xs, vals, idx


def npt_sequence_params(a=1.0, nmin=50, nmax=1000, n0=6.0, dn=math.log(10)):
    """Integer (n0', dn') of AutoSymPTR.MonkhorstPackRule as read at src/fourier.jl:301-321.

    SURVEY A.2 (unverified against the dep's source): clamp(round(x/a), nmin, nmax) for both.
    Defaults (src/brillouin.jl:415) give npt = 50, 100, 150, ...
    """
    n0i = int(min(max(round(n0 / a), nmin), nmax))
    dni = int(min(max(round(dn / a), nmin), nmax))
    return n0i, dni


# ----------------------------------------------------------------------------------------
# Brillouin zones and iterated limits
# ----------------------------------------------------------------------------------------


@dataclass
class CubicLimits:
    """ref: IteratedIntegration.CubicLimits(a, b); src/brillouin.jl:2-5,267."""

    a: np.ndarray
    b: np.ndarray

    @property
    def ndim(self):
        return len(self.a)

    def segs(self):  # endpoints of the outermost (last) variable
        return (float(self.a[-1]), float(self.b[-1]))

    def fix(self, x):  # limits of the remaining variables once the last one is fixed
        return CubicLimits(self.a[:-1], self.b[:-1])


@dataclass
class TetrahedralLimits:
    """0 <= x_1 <= ... <= x_d <= a_d (scaled).  ref: src/brillouin.jl:304; SURVEY A.3."""

    a: np.ndarray
    s: float = 1.0

    @property
    def ndim(self):
        return len(self.a)

    def segs(self):
        return (0.0, float(self.a[-1]) * self.s)

    def fix(self, x):
        return TetrahedralLimits(self.a[:-1], x / float(self.a[-1]) * 1.0)

    # NB: after fixing x_d = x the next variable runs over [0, a_{d-1} * x / a_d]


def _unique_sorted(vals):
    """Break points of a polytope along one coordinate: the distinct vertex coordinates (sqrt(eps)
    tolerances, first occurrence kept), ascending.  ref: get_segs, ext/SymmetryReduceBZExt.jl:15-31."""
    tol = math.sqrt(np.finfo(float).eps)
    uniq = []
    for v in vals:
        if not any(abs(v - u) <= max(tol, tol * max(abs(v), abs(u))) for u in uniq):
            uniq.append(float(v))
    assert len(uniq) >= 2, uniq
    return tuple(sorted(uniq))


class PolygonLimits:
    """Convex polygon in (x, y): verts [nv, 2] ordered around the boundary.
    ref: Polygon2 + xlim_from_yslice, ext/SymmetryReduceBZExt.jl:43-59, ext/ibzlims.jl:245-289."""

    def __init__(self, verts):
        self.verts = np.asarray(verts, dtype=np.float64)

    @property
    def ndim(self):
        return 2

    def segs(self):
        return _unique_sorted(self.verts[:, 1])

    def fix(self, y):
        v = self.verts
        nv = len(v)
        lb, k = None, 0
        for j in range(nv):
            y1, y2 = v[j, 1], v[(j + 1) % nv, 1]
            if (y1 < y and y2 > y) or (y1 > y and y2 < y) or y1 == y:
                t = (y - y1) / (y2 - y1) if y2 != y1 else 0.0
                lim = t * v[(j + 1) % nv, 0] + (1 - t) * v[j, 0]
                k += 1
                if k == 1:
                    lb = lim
                elif k == 2:
                    return CubicLimits(np.array([min(lb, lim)]), np.array([max(lb, lim)]))
        assert k == 1, "could not find intersection with polygon"
        return CubicLimits(np.array([lb]), np.array([lb]))


class PolyhedralLimits:
    """Convex polyhedron in (x, y, z) given by its faces (each [nv, 3], vertices in order around the
    face; triangles of a hull triangulation qualify).  Vertices shared between faces must be bit-identical.
    ref: Polyhedron3 + pg_vert_from_zslice, ext/SymmetryReduceBZExt.jl:33-58, ext/ibzlims.jl:198-243."""

    def __init__(self, faces):
        self.faces = [np.asarray(f, dtype=np.float64) for f in faces]

    @property
    def ndim(self):
        return 3

    def segs(self):
        return _unique_sorted(np.concatenate([f[:, 2] for f in self.faces]))

    def fix(self, z):
        pts = []
        for face in self.faces:
            nv = len(face)
            for j in range(nv):
                p1, p2 = face[j], face[(j + 1) % nv]
                z1, z2 = p1[2], p2[2]
                if (z1 <= z and z2 >= z) or (z1 >= z and z2 <= z):
                    if z2 == z1:
                        continue  # edge in the plane: its end points come from the neighbouring edges
                    t = (z - z1) / (z2 - z1)
                    pts.append((t * p2[0] + (1 - t) * p1[0], t * p2[1] + (1 - t) * p1[1]))
        uniq = []
        for q in pts:  # exact duplicates from shared edges
            if q not in uniq:
                uniq.append(q)
        P = np.array(uniq)
        c = P.mean(axis=0)
        order = np.argsort(np.arctan2(P[:, 1] - c[1], P[:, 0] - c[0]), kind="stable")
        return PolygonLimits(P[order])


@dataclass
class SymmetricBZ:
    """ref: src/brillouin.jl:33-46."""

    A: np.ndarray
    B: np.ndarray
    lims: object
    syms: Optional[list]

    @property
    def nsyms(self):
        return 1 if self.syms is None else len(self.syms)

    @property
    def d(self):
        return self.A.shape[0]


def sign_flip_matrices(d):
    """ref: src/brillouin.jl:248-249 (Iterators.product order: first factor fastest)."""
    out = []
    for rev in itertools.product((1, -1), repeat=d):
        out.append(np.diag(rev[::-1]).astype(np.int64))
    return out


def permutation_matrices(d):
    """ref: src/brillouin.jl:272-277."""
    out = []
    for p in itertools.permutations(range(d)):
        P = np.zeros((d, d), dtype=np.int64)
        for i in range(d):
            P[i, p[i]] = 1
        out.append(P)
    return out


def load_bz(kind: str, A) -> SymmetricBZ:
    """kind in {"FBZ","InversionSymIBZ","CubicSymIBZ"}.  ref: src/brillouin.jl:179-212,264-307.

    B = A' \\ 2 pi I (canonical_reciprocal_basis, :9).
    """
    A = np.atleast_2d(np.asarray(A, dtype=np.float64))
    d = A.shape[0]
    B = np.linalg.solve(A.T, 2 * np.pi * np.eye(d))
    if kind == "FBZ":
        return SymmetricBZ(A, B, CubicLimits(np.zeros(d), np.ones(d)), None)
    if kind == "InversionSymIBZ":
        return SymmetricBZ(A, B, CubicLimits(np.zeros(d), np.full(d, 0.5)), sign_flip_matrices(d))
    if kind == "CubicSymIBZ":
        syms = [S @ P for P in permutation_matrices(d) for S in sign_flip_matrices(d)]
        return SymmetricBZ(A, B, TetrahedralLimits(np.full(d, 0.5)), syms)
    raise ValueError(kind)


# ----------------------------------------------------------------------------------------
# Gauss-Kronrod (QuadGK order 7 -> GK(7,15)) and the globally adaptive loop
# ----------------------------------------------------------------------------------------

# Kronrod abscissae on [-1, 0] (QuadGK.kronrod(7) ordering: most negative first, 0 last) and
# weights; Gauss-7 weights for the odd-indexed abscissae.  Table = QUADPACK qk15, verified
# with mpmath to be exact through degree 22 (tests/test_oracle_pins.py).
GK_X = -np.array([
    0.991455371120812639206854697526329, 0.949107912342758524526189684047851,
    0.864864423359769072789712788640926, 0.741531185599394439863864773280788,
    0.586087235467691130294144838258730, 0.405845151377397166906606412076961,
    0.207784955007898467600689403773245, 0.0])
GK_W = np.array([
    0.022935322010529224963732008058970, 0.063092092629978553290700663189204,
    0.104790010322250183839876322541518, 0.140653259715525918745189590510238,
    0.169004726639267902826583426598550, 0.190350578064785409913256402421014,
    0.204432940075298892414161999234649, 0.209482141084727828012999174891714])
GK_GW = np.array([
    0.129484966168869693270611432679082, 0.279705391489276667901467771423780,
    0.381830050505118944950369775488975, 0.417959183673469387755102040816327])


def gk_nodes(a: float, b: float) -> np.ndarray:
    """The 15 evaluation points of one segment, in QuadGK.evalrule's batch order:
    for i=1..7: a+(1+x_i)s, a+(1-x_i)s ; then the midpoint a+s."""
    s = 0.5 * (b - a)
    out = np.empty(15)
    for i in range(7):
        out[2 * i] = a + (1 + GK_X[i]) * s
        out[2 * i + 1] = a + (1 - GK_X[i]) * s
    out[14] = a + s
    return out


def gk_evalrule(fv, a: float, b: float, norm=None):
    """Kronrod/Gauss sums of one segment from its 15 values (order of gk_nodes).

    ref: QuadGK.evalrule (order 7 is odd: the centre node is in both rules); reached through
    auxquadgk at src/algorithms.jl:215-239.  Returns (I, E) with I = Ik*s, E = norm(Ik*s-Ig*s).
    """
    norm = norm or _norm
    s = 0.5 * (b - a)
    fg = fv[2] + fv[3]
    fk = fv[0] + fv[1]
    Ig = fg * GK_GW[0]
    Ik = fg * GK_W[1] + fk * GK_W[0]
    for i in range(2, 4):  # i = 2..length(gw)-1 (1-based)
        fg = fv[2 * (2 * i - 1)] + fv[2 * (2 * i - 1) + 1]
        fk = fv[2 * (2 * i - 2)] + fv[2 * (2 * i - 2) + 1]
        Ig = Ig + fg * GK_GW[i - 1]
        Ik = Ik + fg * GK_W[2 * i - 1] + fk * GK_W[2 * i - 2]
    f0 = fv[14]
    Ig = Ig + f0 * GK_GW[3]
    Ik = Ik + f0 * GK_W[7] + (fv[12] + fv[13]) * GK_W[6]
    Ik_s, Ig_s = Ik * s, Ig * s
    E = norm(Ik_s - Ig_s)
    if not np.isfinite(E):
        raise FloatingPointError(f"integrand produced {E} in the interval ({a}, {b})")
    return Ik_s, E


def _norm(v):
    return float(np.linalg.norm(np.atleast_1d(v)))


class _MaxHeap:
    """Binary max-heap on key E with DataStructures.jl percolate semantics (Base.Order.Reverse)."""

    def __init__(self):
        self.xs = []

    @staticmethod
    def _lt(a, b):  # lt(Reverse, a, b) == isless(b.E, a.E)
        return b[3] < a[3]

    def _down(self, i, x, length):
        xs = self.xs
        while True:
            l = 2 * i + 1
            if l >= length:
                break
            r = l + 1
            j = l if (r >= length or self._lt(xs[l], xs[r])) else r
            if not self._lt(xs[j], x):
                break
            xs[i] = xs[j]
            i = j
        xs[i] = x

    def _up(self, i, x):
        xs = self.xs
        while i > 0:
            j = (i - 1) // 2
            if not self._lt(x, xs[j]):
                break
            xs[i] = xs[j]
            i = j
        xs[i] = x

    def heapify(self):
        n = len(self.xs)
        for i in range(n // 2 - 1, -1, -1):
            self._down(i, self.xs[i], n)

    def push(self, x):
        self.xs.append(x)
        self._up(len(self.xs) - 1, x)

    def pop(self):
        xs = self.xs
        x = xs[0]
        y = xs.pop()
        if xs:
            self._down(0, y, len(xs))
        return x


def auxquadgk(f, segs, *, atol=None, rtol=None, maxevals=2**62, norm=None, batch=False,
              max_batch=2**62, record=None):
    """Globally adaptive GK(7,15) over the breakpoints `segs`.

    ref: auxquadgk called at src/algorithms.jl:231-238 (QuadGK do_quadgk/adapt semantics, SURVEY
    A.3).  Scalar mode pops the worst segment, bisects, evaluates 30 nodes.  batch=True follows
    the BatchIntegrand `refine`: pop segments while the error of the remaining ones exceeds the
    tolerance, evaluate all their children in ONE call f(xs)->values.
    `f` maps an array of points to a sequence of values.  Segments are tuples (a, b, I, E).
    `record`, if a list, receives the final segments (a, b) for panel-tree parity checks.
    Returns (I, E, numevals).
    """
    norm = norm or _norm
    atol_ = 0.0 if atol is None else atol
    rtol_ = (0.0 if atol_ > 0 else math.sqrt(np.finfo(float).eps)) if rtol is None else rtol
    heap = _MaxHeap()
    nseg = len(segs) - 1
    xs = np.concatenate([gk_nodes(segs[i], segs[i + 1]) for i in range(nseg)])
    fv = f(xs)
    for i in range(nseg):
        I, E = gk_evalrule(fv[15 * i:15 * i + 15], segs[i], segs[i + 1], norm)
        heap.xs.append((segs[i], segs[i + 1], I, E))
    I = heap.xs[0][2]
    E = heap.xs[0][3]
    for sgm in heap.xs[1:]:
        I = I + sgm[2]
        E = E + sgm[3]
    numevals = 15 * nseg
    if not (E <= max(atol_, rtol_ * norm(I)) or numevals >= maxevals):
        heap.heapify()
        while E > max(atol_, rtol_ * norm(I)) and numevals < maxevals:
            if not batch:
                popped = [heap.pop()]
                numevals += 30
            else:
                tol = max(atol_, rtol_ * norm(I))
                popped = []
                while heap.xs and 30 * (len(popped) + 1) <= max_batch and E > tol and numevals < maxevals:
                    s = heap.pop()
                    popped.append(s)
                    tol += s[3]
                    numevals += 30
            pts = []
            for (a, b, _, _) in popped:
                mid = (a + b) / 2
                pts.append(gk_nodes(a, mid))
                pts.append(gk_nodes(mid, b))
            fv = f(np.concatenate(pts))
            for k, (a, b, Is, Es) in enumerate(popped):
                mid = (a + b) / 2
                I1, E1 = gk_evalrule(fv[30 * k:30 * k + 15], a, mid, norm)
                I2, E2 = gk_evalrule(fv[30 * k + 15:30 * k + 30], mid, b, norm)
                I = (I - Is) + I1 + I2
                E = (E - Es) + E1 + E2
                heap.push((a, mid, I1, E1))
                heap.push((mid, b, I2, E2))
        # re-sum to limit accumulated roundoff (QuadGK does this after adapt)
        I = heap.xs[0][2]
        E = heap.xs[0][3]
        for sgm in heap.xs[1:]:
            I = I + sgm[2]
            E = E + sgm[3]
    if record is not None:
        record.extend(sorted((s[0], s[1]) for s in heap.xs))
    return I, E, numevals


# ----------------------------------------------------------------------------------------
# Integrand menu (the integrands that appear in the reference's tests / docs / example)
# ----------------------------------------------------------------------------------------


# All oracle integrands are vectorised over the leading (node) axis:
#   f(x[N,d], s[N,n,n] or s[N]) -> values[N, ...]


def f_linear(a, b):
    """a*s + b.  ref: test/fourier.jl:41."""
    return lambda x, s: a * s + b


def f_linear_x(a, b):
    """a*s*x .+ b (vector valued, scalar series).  ref: test/fourier.jl:16."""
    return lambda x, s: a * np.asarray(s)[:, None] * np.asarray(x) + b


def f_dos(eta, omega):
    """-Im tr inv((omega + i eta) I - H) / pi.  ref: aps_example/aps_example.jl:30."""
    def f(x, s):
        h = np.asarray(s)
        if h.ndim == 1:
            h = h[:, None, None]
        n = h.shape[-1]
        g = np.linalg.inv((omega + 1j * eta) * np.eye(n) - h)
        return -np.imag(np.trace(g, axis1=-2, axis2=-1)) / np.pi
    return f


def f_gloc(eta, omega):
    """inv((omega + i eta) I - H).  ref: docs/src/examples.md:21,90."""
    def f(x, s):
        h = np.asarray(s)
        scalar = h.ndim == 1
        if scalar:
            h = h[:, None, None]
        n = h.shape[-1]
        g = np.linalg.inv(complex(omega, eta) * np.eye(n) - h)
        return g[:, 0, 0] if scalar else g
    return f


def f_one():
    """Unit measure.  ref: test/brillouin.jl:38."""
    return lambda x, s: np.ones(len(x))


# ----------------------------------------------------------------------------------------
# Solvers: PTR / AutoPTR / IAI on a SymmetricBZ with a FourierSeries integrand
# ----------------------------------------------------------------------------------------


@dataclass
class Solution:
    u: object
    resid: object
    retcode: bool
    numevals: int
    extra: dict = field(default_factory=dict)


def _ptr_rule_sum(s, npt, syms, f):
    """rule(f, Basis(I)) = quadsum(...).  ref: src/fourier.jl:204-207 (full grid, weight 1,
    dvol = 1/npt^d) and :289-292 (symmetric, integer weights, dvol = 1/(npt^d nsyms)).
    Nodes are visited in column-major order (i1 fastest) like iterate(p.s)/iterate(p.p)."""
    d = s.d
    if syms is None:
        vals = fourier_ptr(s, npt)
        x = ptrpoints(npt)
        perm = tuple(range(d - 1, -1, -1))
        tail = tuple(range(d, vals.ndim))
        vl = np.transpose(vals, perm + tail).reshape((npt**d,) + vals.shape[d:])
        grids = np.meshgrid(*([x] * d), indexing="ij")
        xl = np.stack([np.transpose(g, perm).reshape(-1) for g in grids], axis=1)
        fv = np.asarray(f(xl, vl))
        return fv.sum(axis=0) * (1.0 / npt**d), npt**d
    w, xs, vals, _ = fourier_symptr(s, npt, syms)
    fv = np.asarray(f(xs, vals))
    acc = np.tensordot(w.astype(np.float64), fv, axes=(0, 0))
    return acc * (1.0 / (npt**d * len(syms))), len(w)


def solve_ptr(s, bz: SymmetricBZ, f, npt=50, abstol=None):
    """PTR on a SymmetricBZ.  ref: src/brillouin.jl:337-355,392-394; src/fourier.jl:381-384;
    src/algorithms.jl:368-380.  Result = |det B| * symmetrize(u) with TrivialRep => nsyms*u."""
    j = abs(np.linalg.det(bz.B))
    u, nev = _ptr_rule_sum(s, npt, bz.syms, f)
    return Solution(j * bz.nsyms * u, None, True, nev)


def solve_autoptr(s, bz: SymmetricBZ, f, abstol=None, reltol=None, maxiters=2**62, a=1.0,
                  nmin=50, nmax=1000, n0=6.0, dn=math.log(10), norm=None):
    """AutoPTR.  ref: src/brillouin.jl:418-444 (abstol /= |det B| only; symmetrisation inside
    each rule via SymmetricRule :127-130), src/algorithms.jl:418-432 and autosymptr:
    I1 = rule(n0), I2 = rule(n0+dn), err = norm(I2-I1); loop until err <= max(abstol, reltol*norm(I2))."""
    norm = norm or _norm
    j = abs(np.linalg.det(bz.B))
    atol = None if abstol is None else abstol / j
    if atol is None and reltol is None:
        rtol, atol_ = math.sqrt(np.finfo(float).eps), 0.0
    else:
        rtol = 0.0 if reltol is None else reltol
        atol_ = 0.0 if atol is None else atol
    n0i, dni = npt_sequence_params(a, nmin, nmax, n0, dn)
    npt = n0i
    numevals = 0
    grids = []

    def rule(npt):
        u, nev = _ptr_rule_sum(s, npt, bz.syms, f)
        return bz.nsyms * u, nev  # symmetrize inside the rule (TrivialRep)

    I1, nev = rule(npt)
    numevals += nev
    grids.append(npt)
    npt += dni
    I2, nev = rule(npt)
    numevals += nev
    grids.append(npt)
    err = norm(I2 - I1)
    while not (err <= max(atol_, rtol * norm(I2)) or numevals >= maxiters or not np.isfinite(err)):
        I1 = I2
        npt += dni
        I2, nev = rule(npt)
        numevals += nev
        grids.append(npt)
        err = norm(I2 - I1)
    return Solution(I2 * j, err * j, True, numevals, {"grids": grids})


def nested_quad(s, lims, f, abstol=None, reltol=None, maxiters=2**62, norm=None, record=None,
                batch=False):
    """NestedQuad(AuxQuadGKJL()) over iterated limits for a Fourier integrand.

    ref: src/fourier.jl:432-510: outer levels contract the series at the GK node and recurse with
    abstol/len (len = length of the next variable's interval, :479-480; reltol unchanged), the
    innermost level evaluates the 1-D series and calls f (:452-456).  Depth-first, scalar GK
    refinement (batch=True: the BatchIntegrand refinement at every level).
    Returns (I, E, numevals of f); `record` receives the panels of the OUTERMOST integral.
    """
    norm = norm or _norm
    count = [0]
    last_err = [0.0]

    def level(cur: FourierSeries, lims, tail: tuple, atol_l, rec=None):
        segs = tuple(lims.segs())  # break points of the outermost variable (2 for boxes, more for polytopes)
        if cur.d == 1:
            def g(xs):
                ph = phases(cur, 0, xs)  # (N, M)
                vs = np.tensordot(ph, cur.c, axes=(1, 0))  # (N, n, n)
                vs = vs[:, 0, 0] if cur.scalar else vs
                count[0] += len(xs)
                X = np.column_stack([xs] + [np.full(len(xs), t) for t in tail])
                return np.asarray(f(X, vs))
        else:
            def g(xs):
                out = []
                for x in xs:
                    inner = lims.fix(x)
                    isegs = inner.segs()
                    ln = isegs[-1] - isegs[0]  # ref: len = segs[end] - segs[1], src/fourier.jl:466,480
                    at = None if atol_l is None else atol_l / ln
                    out.append(level(contract(cur, x), inner, (x,) + tail, at))
                return out
        I, E, _ = auxquadgk(g, segs, atol=atol_l, rtol=reltol, maxevals=maxiters, norm=norm,
                            record=rec, batch=batch)
        last_err[0] = E
        return I

    u = level(s, lims, (), abstol, record)
    return u, last_err[0], count[0]


def solve_iai(s, bz: SymmetricBZ, f, abstol=None, reltol=None, maxiters=2**62, norm=None,
              record=None, batch=False):
    """IAI on a SymmetricBZ.  ref: src/brillouin.jl:337-355,375-377 (abstol /= |det B| nsyms;
    result * |det B| nsyms for TrivialRep)."""
    j = abs(np.linalg.det(bz.B))
    atol = None if abstol is None else abstol / (j * bz.nsyms)
    u, err, cnt = nested_quad(s, bz.lims, f, atol, reltol, maxiters, norm, record, batch)
    return Solution(j * bz.nsyms * u, j * bz.nsyms * err, True, cnt)


def batchparam(n_params: int, nthreads: int):
    """Round-robin groups: group j gets indices j, j+nthreads, ...  ref: src/interfaces.jl:199-208."""
    groups = [[] for _ in range(min(nthreads, n_params))]
    for i in range(n_params):
        groups[i % nthreads].append(i)
    return groups


# ----------------------------------------------------------------------------------------
# GGR density of states
# ----------------------------------------------------------------------------------------


def eigh_upper(h):
    """eigen(Hermitian(h)): ascending eigenvalues, orthonormal U, upper triangle is used.
    ref: src/dos_ggr.jl:19,34 (LAPACK for n > 3, closed forms below)."""
    return np.linalg.eigh(np.atleast_2d(h), UPLO="U")


def get_ggr_data(s: FourierSeries, npt: int, syms):
    """Per node: eigenvalues and band velocities v_j = Re diag(U' dH/dx_j U) * t_j, weights.
    ref: src/dos_ggr.jl:1-44."""
    d = s.d
    ders = [derivative_series(s, j) for j in range(d)]
    if syms is None:
        H = fourier_ptr(s, npt)
        Vs = [fourier_ptr(ds, npt) for ds in ders]
        if s.scalar:
            H = H[..., None, None]
            Vs = [v[..., None, None] for v in Vs]
        # column-major node order
        perm = tuple(range(d - 1, -1, -1))
        Hl = np.transpose(H, perm + (d, d + 1)).reshape(-1, s.n, s.n)
        Vl = [np.transpose(v, perm + (d, d + 1)).reshape(-1, s.n, s.n) for v in Vs]
        w = np.ones(len(Hl))
    else:
        w, _, Hl, idx = fourier_symptr_fast(s, npt, syms)
        Vl = []
        u = ptrpoints(npt)
        for ds in ders:
            Vl.append(np.asarray(evaluate_many(ds, u[idx] * np.asarray(s.t)[None, :])).reshape(len(w), s.n, s.n))
        Hl = np.asarray(Hl).reshape(len(w), s.n, s.n)
    e, U = np.linalg.eigh(Hl, UPLO="U")
    vel = np.empty((len(Hl), d, s.n))
    for j in range(d):
        vel[:, j, :] = np.real(np.einsum("kan,kab,kbn->kn", U.conj(), Vl[j], U)) * s.t[j]
    return np.asarray(w, dtype=np.float64), e, vel


def ggr_formula(b, E, e, *v):
    """Area of the plane E = e + v.kappa inside the cube of half-width b over |v|.
    ref: src/dos_ggr.jl:75-104 (1-D, 2-D, 3-D)."""
    if len(v) == 1:
        v1 = abs(v[0])
        dw = abs(E - e)
        w1 = b * v1
        return 1 / v1 if 0 <= dw <= w1 else 0.0
    if len(v) == 2:
        v2, v1 = sorted((abs(v[0]), abs(v[1])))
        dw = abs(E - e)
        w1 = b * abs(v1 - v2)
        w3 = b * (v1 + v2)
        if 0 <= dw <= w1:
            return 2 * b / v1
        if w1 <= dw <= w3:
            return (b * (v1 + v2) - dw) / (v1 * v2)
        return 0.0
    if len(v) == 3:
        v3, v2, v1 = sorted((abs(v[0]), abs(v[1]), abs(v[2])))
        dw = abs(E - e)
        w1 = b * abs(v1 - v2 - v3)
        w2 = b * (v1 - v2 + v3)
        w3 = b * (v1 + v2 - v3)
        w4 = b * (v1 + v2 + v3)
        vn = math.sqrt(v1 * v1 + v2 * v2 + v3 * v3)
        with np.errstate(divide="ignore", invalid="ignore"):
            if v1 >= v2 + v3 and 0 <= dw <= w1:
                return float(np.float64(4 * b * b) / v1)
            if v1 <= v2 + v3 and 0 <= dw <= w1:
                return float(np.float64(2 * b * b * (v1 * v2 + v2 * v3 + v3 * v1) - (dw * dw + (vn * b) ** 2)) / (v1 * v2 * v3))
            if w1 <= dw <= w2:
                return float(np.float64(b * b * (v1 * v2 + 3 * v2 * v3 + v3 * v1) - b * dw * (-v1 + v2 + v3) - (dw * dw + (vn * b) ** 2) / 2) / (v1 * v2 * v3))
            if w2 <= dw <= w3:
                return float(np.float64(2 * b * (b * (v1 + v2) - dw)) / (v1 * v2))
            if w3 <= dw <= w4:
                return float(np.float64((b * (v1 + v2 + v3) - dw) ** 2) / (2 * v1 * v2 * v3))
        return 0.0
    raise ValueError("GGR implemented for up to 3d BZ")


def ggr_formula_vec(b, E, e, v):
    """Vectorised ggr_formula: e (N,), v (N, d) -> (N,).  Same branches as src/dos_ggr.jl:75-104."""
    v = np.sort(np.abs(np.asarray(v, dtype=np.float64)), axis=1)[:, ::-1]  # v1 >= v2 >= v3
    d = v.shape[1]
    dw = np.abs(E - e)
    with np.errstate(divide="ignore", invalid="ignore"):
        if d == 1:
            v1 = v[:, 0]
            return np.where(dw <= b * v1, 1 / v1, 0.0)
        if d == 2:
            v1, v2 = v[:, 0], v[:, 1]
            w1 = b * np.abs(v1 - v2)
            w3 = b * (v1 + v2)
            return np.select([dw <= w1, dw <= w3], [2 * b / v1, (b * (v1 + v2) - dw) / (v1 * v2)], 0.0)
        v1, v2, v3 = v[:, 0], v[:, 1], v[:, 2]
        w1 = b * np.abs(v1 - v2 - v3)
        w2 = b * (v1 - v2 + v3)
        w3 = b * (v1 + v2 - v3)
        w4 = b * (v1 + v2 + v3)
        vn2 = v1 * v1 + v2 * v2 + v3 * v3  # hypot(v1,v2,v3)^2
        p = v1 * v2 * v3
        return np.select(
            [(v1 >= v2 + v3) & (dw <= w1), (v1 <= v2 + v3) & (dw <= w1),
             (w1 <= dw) & (dw <= w2), (w2 <= dw) & (dw <= w3), (w3 <= dw) & (dw <= w4)],
            [4 * b * b / v1,
             (2 * b * b * (v1 * v2 + v2 * v3 + v3 * v1) - (dw * dw + vn2 * b * b)) / p,
             (b * b * (v1 * v2 + 3 * v2 * v3 + v3 * v1) - b * dw * (-v1 + v2 + v3) - (dw * dw + vn2 * b * b) / 2) / p,
             2 * b * (b * (v1 + v2) - dw) / (v1 * v2),
             (b * (v1 + v2 + v3) - dw) ** 2 / (2 * p)], 0.0)


def sum_ggr(d, npt, E, weights, energies, velocities):
    """sum_k w_k sum_bands ggr_formula(1/(2 npt), E, e, v...).  ref: src/dos_ggr.jl:58-65.
    energies (N, n), velocities (N, d, n)."""
    b = 1 / (2 * npt)
    tot = 0.0
    for n in range(energies.shape[1]):
        tot += float(np.dot(weights, ggr_formula_vec(b, E, energies[:, n], velocities[:, :, n])))
    return tot


def dos_ggr(s: FourierSeries, bz: SymmetricBZ, Es, npt=50):
    """DOSProblem + GGR.  ref: src/dos_ggr.jl:1-12,46-65 (no Jacobian/symmetry rescale at all:
    the reference returns the bare sum, test/dos.jl:88-111 compares it with the exact DOS)."""
    w, e, v = get_ggr_data(s, npt, bz.syms)
    return np.array([sum_ggr(s.d, npt, E, w, e, v) for E in np.atleast_1d(Es)])


# ----------------------------------------------------------------------------------------
# Synthetic inputs -- language neutral RNG
# ----------------------------------------------------------------------------------------


def splitmix64(seed: int):
    """Generator of u in [0,1): u = (x >> 11) * 2^-53 with x from splitmix64(seed)."""
    mask = (1 << 64) - 1
    state = seed & mask
    while True:
        state = (state + 0x9E3779B97F4A7C15) & mask
        z = state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & mask
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & mask
        z = z ^ (z >> 31)
        yield (z >> 11) * 2.0**-53


def integer_lattice(d: int) -> FourierSeries:
    """s(x) = (1/d) sum_i cos(2 pi x_i).  ref: test/utils.jl:3-9."""
    C = np.zeros((3,) * d)
    for i in range(d):
        for jj in (0, 2):
            idx = [1] * d
            idx[i] = jj
            C[tuple(idx)] = 1 / (2 * d)
    return FourierSeries(C, period=1.0, first=-1, ndim=d)


def tb_integer(d: int, t: float = 1.0) -> FourierSeries:
    """1-band nearest-neighbour model H(k) = 2 t sum cos(2 pi k_i).  ref: test/dos.jl:34-41."""
    C = np.zeros((3,) * d + (1, 1))
    for i in range(d):
        for jj in (0, 2):
            idx = [1] * d
            idx[i] = jj
            C[tuple(idx) + (0, 0)] = t
    return FourierSeries(C, period=1.0, first=-1, ndim=d)


def tb_graphene(t: float = 1.0) -> FourierSeries:
    """2-band graphene on a 5x5 coefficient grid (axes -2:2).  ref: test/dos.jl:8-15."""
    C = np.zeros((5, 5, 2, 2))
    o = 2  # index offset: axis value v lives at v + 2
    for (i, j) in ((1, 1), (1, -2), (-2, 1)):
        C[i + o, j + o, 0, 1] = t
    for (i, j) in ((-1, -1), (-1, 2), (2, -1)):
        C[i + o, j + o, 1, 0] = t
    return FourierSeries(C, period=1.0, first=-2, ndim=2)


def synthetic_wannier(n=16, rmax=6, seed=20240601, scale=0.25, decay=1.5) -> FourierSeries:
    """Config C5: Hermitian-symmetrised random H_R with exponential decay."""
    M = 2 * rmax + 1
    rng = splitmix64(seed)
    C = np.zeros((M, M, M, n, n), dtype=np.complex128)
    for R in itertools.product(range(-rmax, rmax + 1), repeat=3):
        if R <= (0, 0, 0) and R != (0, 0, 0):
            continue  # lexicographic half space R > 0 plus R = 0
        A = np.empty((n, n), dtype=np.complex128)
        for m in range(n):
            for k in range(n):
                re = 2 * next(rng) - 1
                im = 2 * next(rng) - 1
                A[m, k] = re + 1j * im
        A *= scale * math.exp(-math.sqrt(sum(r * r for r in R)) / decay)
        i = tuple(r + rmax for r in R)
        if R == (0, 0, 0):
            C[i] = (A + A.conj().T) / 2 + np.diag(np.linspace(-1, 1, n))
        else:
            C[i] = A
            C[tuple(-r + rmax for r in R)] = A.conj().T
    return FourierSeries(C, period=1.0, first=-rmax, ndim=3)
