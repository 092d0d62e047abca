"""Adaptive piecewise-Chebyshev interpolation of a solver in its parameter -- the driver the
reference's example wraps around its DOS solvers (`hchebinterp(dos_solver, 10, 15; atol=1e-2)`,
aps_example/aps_example.jl:36-39).  HChebInterp.jl is not vendored with the reference; this is a
plain h-adaptive scheme of the same shape: order-p Chebyshev panels, error estimated from the tail of
the panel's Chebyshev coefficients, bisection until every panel meets max(atol, rtol*|f|), all new
panels of a refinement level evaluated in ONE batch (so PTR sweeps fuse on the GPU)."""
import numpy as np


def _cheb_nodes(order, a, b):
    j = np.arange(order + 1)
    return 0.5 * (a + b) + 0.5 * (b - a) * np.cos(np.pi * j / order)


def _cheb_coeffs(vals):
    """Chebyshev coefficients of the interpolant through 2nd-kind points (values ordered x: b -> a)."""
    p = len(vals) - 1
    v = np.asarray(vals, dtype=np.complex128 if np.iscomplexobj(vals) else np.float64)
    ext = np.concatenate([v, v[-2:0:-1]])
    c = np.fft.fft(ext, axis=0)[: p + 1] / p
    c = c.real if not np.iscomplexobj(v) else c
    c[0] *= 0.5
    c[p] *= 0.5
    return c


class ChebInterp:
    """Piecewise Chebyshev interpolant: panels [(a, b, coeffs)] sorted by a."""

    def __init__(self, panels, numevals):
        self.panels = sorted(panels, key=lambda t: t[0])
        self.edges = np.array([p[0] for p in self.panels] + [self.panels[-1][1]])
        self.numevals = numevals

    def __call__(self, x):
        x = np.asarray(x, dtype=np.float64)
        out = np.empty(x.shape, dtype=self.panels[0][2].dtype)
        flat, of = x.reshape(-1), out.reshape(-1)
        k = np.clip(np.searchsorted(self.edges, flat, side="right") - 1, 0, len(self.panels) - 1)
        for i, (xi, ki) in enumerate(zip(flat, k)):
            a, b, c = self.panels[ki]
            t = (2 * xi - a - b) / (b - a)
            b1 = b2 = 0.0
            for cj in c[:0:-1]:  # Clenshaw
                b1, b2 = 2 * t * b1 - b2 + cj, b1
            of[i] = t * b1 - b2 + c[0]
        return out if out.shape else out[()]


def hchebinterp(f, a, b, atol=0.0, rtol=None, order=15, ntail=3, maxevals=10**6, batch=None):
    """Adaptively interpolate f on [a, b].  `f` maps one parameter to a value; `batch(list_of_x)` (e.g.
    lambda xs: batchsolve(solver, xs)) evaluates many at once.  Returns a ChebInterp."""
    if rtol is None:
        rtol = 0.0 if atol > 0 else np.sqrt(np.finfo(float).eps)
    ev = batch if batch is not None else (lambda xs: np.array([f(x) for x in xs]))
    todo = [(float(a), float(b))]
    done = []
    numevals = 0
    while todo and numevals < maxevals:
        xs = np.concatenate([_cheb_nodes(order, pa, pb) for pa, pb in todo])
        vals = np.asarray(ev(list(xs)))
        numevals += len(xs)
        nxt = []
        for i, (pa, pb) in enumerate(todo):
            v = vals[i * (order + 1):(i + 1) * (order + 1)]
            c = _cheb_coeffs(v)
            err = np.abs(c[-ntail:]).sum()
            if err <= max(atol, rtol * np.abs(v).max()) or (pb - pa) < 1e-12 * max(1.0, abs(pa), abs(pb)):
                done.append((pa, pb, c))
            else:
                mid = 0.5 * (pa + pb)
                nxt += [(pa, mid), (mid, pb)]
        todo = nxt
    for pa, pb in todo:  # maxevals hit: keep the last level's panels as they are
        v = np.asarray(ev(list(_cheb_nodes(order, pa, pb))))
        done.append((pa, pb, _cheb_coeffs(v)))
    return ChebInterp(done, numevals)
