"""Nested adaptive GK(7,15) for USER closures (Python callables): the adaptive loop and the closure
run on the host, every innermost batch of Fourier evaluations runs on the GPU (abz_eval_nodes).
ref: src/fourier.jl:432-510 (init_nest / do_solve for NestedQuad), src/algorithms.jl:215-239."""
import math

import numpy as np

from . import _lib as L
from .bz import CubicLimits, TetrahedralLimits


def _gk_nodes(a, b):
    x = np.empty(15)
    L.check(L.lib().abz_gk15_nodes(float(a), float(b), x.ctypes.data_as(L.c_f64p)))
    return x


def _gk_eval(vals, a, b):
    """vals: list of 15 values (scalars or arrays) -> (I, E) via abz_gk15_batch."""
    arr = np.asarray(vals)
    shape = arr.shape[1:]
    flat = np.ascontiguousarray(arr.reshape(15, -1).astype(np.complex128))
    ncomp = flat.shape[1]
    ab = np.array([a, b], dtype=np.float64)
    I = np.empty((ncomp, 2))
    E = np.empty(1)
    L.check(L.lib().abz_gk15_batch(ab.ctypes.data_as(L.c_f64p), flat.view(np.float64).ctypes.data_as(L.c_f64p), 1,
                                   ncomp, I.ctypes.data_as(L.c_f64p), E.ctypes.data_as(L.c_f64p)))
    Ic = I.view(np.complex128).reshape(shape)
    if not np.iscomplexobj(arr):
        Ic = Ic.real
    return (Ic if shape else Ic[()]), float(E[0])


class _Heap:
    """max-heap on E with DataStructures.jl percolate semantics (Base.Order.Reverse)."""

    def __init__(self):
        self.xs = []

    @staticmethod
    def lt(a, b):
        return b[3] < a[3]

    def down(self, i, x, n):
        xs = self.xs
        while True:
            l = 2 * i + 1
            if l >= n:
                break
            r = l + 1
            j = l if (r >= n or self.lt(xs[l], xs[r])) else r
            if not self.lt(xs[j], x):
                break
            xs[i] = xs[j]
            i = j
        xs[i] = x

    def up(self, i, x):
        xs = self.xs
        while i > 0:
            j = (i - 1) // 2
            if not self.lt(x, xs[j]):
                break
            xs[i] = xs[j]
            i = j
        xs[i] = x

    def heapify(self):
        n = len(self.xs)
        for i in range(n // 2 - 1, -1, -1):
            self.down(i, self.xs[i], n)

    def push(self, x):
        self.xs.append(x)
        self.up(len(self.xs) - 1, x)

    def pop(self):
        x = self.xs[0]
        y = self.xs.pop()
        if self.xs:
            self.down(0, y, len(self.xs))
        return x


def _norm(v):
    return float(np.linalg.norm(np.atleast_1d(np.asarray(v)).reshape(-1)))


def auxquadgk(g, segs, atol, rtol, maxevals):
    """Scalar-mode globally adaptive GK(7,15) over the break points `segs`: g maps points to a list of values."""
    atol_ = 0.0 if atol is None else atol
    rtol_ = (0.0 if atol_ > 0 else math.sqrt(np.finfo(float).eps)) if rtol is None else rtol
    nseg = len(segs) - 1
    fv = g(np.concatenate([_gk_nodes(segs[i], segs[i + 1]) for i in range(nseg)]))
    heap = _Heap()
    for i in range(nseg):
        Ii, Ei = _gk_eval(fv[15 * i:15 * i + 15], segs[i], segs[i + 1])
        heap.xs.append((segs[i], segs[i + 1], Ii, Ei))
    I, E = heap.xs[0][2], heap.xs[0][3]
    for sg in heap.xs[1:]:
        I = I + sg[2]
        E = E + sg[3]
    numevals = 15 * nseg
    if not (E <= max(atol_, rtol_ * _norm(I)) or numevals >= maxevals):
        heap.heapify()
        while E > max(atol_, rtol_ * _norm(I)) and numevals < maxevals:
            (sa, sb, sI, sE) = heap.pop()
            mid = (sa + sb) / 2
            fv = g(np.concatenate([_gk_nodes(sa, mid), _gk_nodes(mid, sb)]))
            I1, E1 = _gk_eval(fv[:15], sa, mid)
            I2, E2 = _gk_eval(fv[15:], mid, sb)
            I = (I - sI) + I1 + I2
            E = (E - sE) + E1 + E2
            numevals += 30
            heap.push((sa, mid, I1, E1))
            heap.push((mid, sb, I2, E2))
        I, E = heap.xs[0][2], heap.xs[0][3]
        for s in heap.xs[1:]:
            I = I + s[2]
            E = E + s[3]
    return I, E


def nested_quad_host(f, dev, lims, p, abstol, reltol, maxiters):
    """Any iterated limits with the protocol `segs()` / `fix(x)` (bz.py: Cubic, Tetrahedral, Polyhedral, Polygon)."""
    from .solver import FourierValue  # local import: solver imports this module lazily
    d = f.w.d
    if lims.ndim != d:
        raise ValueError("variables in Fourier series don't match domain")
    user = f.f.f
    count = [0]

    def level_solve(level, lim, tail, atol):
        if level == 1:
            def g(xs):
                pts = np.column_stack([xs] + [np.full(len(xs), t) for t in tail])
                vals = dev.eval_nodes(pts)
                count[0] += len(xs)
                return [user(FourierValue(pts[i], vals[i]), *p.args, **p.kwargs) for i in range(len(xs))]
        else:
            def g(xs):
                out = []
                for x in xs:
                    inner = lim.fix(x)
                    sg = inner.segs()
                    at = None if atol is None else atol / (sg[-1] - sg[0])  # ref: src/fourier.jl:479-480
                    out.append(level_solve(level - 1, inner, (x,) + tail, at)[0])
                return out
        return auxquadgk(g, tuple(lim.segs()), atol, reltol, maxiters)

    I, E = level_solve(d, lims, (), abstol)
    return I, E, count[0]
