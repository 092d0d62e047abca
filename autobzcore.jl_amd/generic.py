"""Generic (non-Fourier) integrands behind the same dispatch: plain callables `f(x, p)`, `BatchIntegrand(f!, y, x;
max_batch)` and `NestedBatchIntegrand`, under AuxQuadGKJL (1-D), MonkhorstPack / AutoSymPTRJL (Basis domains) and
NestedQuad (iterated limits).  The adaptive loops and the user's function run on the host; the GK(7,15) panel sums go
through the library's shared rule (abz_gk15_batch, the same table the device loops use).

`BatchIntegrand` is the reference's stated GPU hook ("f!(y, x, p) ... can evaluate the integrand at multiple quadrature
nodes using, for example, threads, the GPU", src/batch.jl:1-9): `fourier_batch(g, series)` builds the batch body that
evaluates H(k) for all nodes of a batch in one abz_eval_nodes call and hands the user `FourierValue` batches.

ref: src/batch.jl:10-38 (types), src/algorithms.jl:215-239 (AuxQuadGKJL incl. the BatchIntegrand branch :227-233),
:360-380 (MonkhorstPack, :370-372), :409-432 (AutoSymPTRJL, :421-423), :450-612 (NestedQuad: `init_nest` batches the
innermost integral only for a BatchIntegrand :469-471,517-518, every level for a NestedBatchIntegrand :483-488,
:519-529,549-560; inner tolerance abstol/len :532-533,556-557).
"""
import math

import numpy as np

from . import _lib as L
from .hostquad import _Heap, _gk_eval, _gk_nodes, _norm


class PuncturedInterval:
    """Break points of a 1-D domain.  ref: src/domains.jl (segments)."""

    def __init__(self, segs):
        self.segs = tuple(float(s) for s in segs)
        if len(self.segs) < 2:
            raise ValueError("a 1-D domain needs at least two break points")


def _segments(dom):
    if isinstance(dom, PuncturedInterval):
        return dom.segs
    if hasattr(dom, "segs") and not callable(dom.segs):  # bz.PuncturedInterval dataclass
        return tuple(float(s) for s in dom.segs)
    if isinstance(dom, (tuple, list)) and len(dom) >= 2 and all(np.ndim(v) == 0 for v in dom):
        return tuple(float(v) for v in dom)
    raise ValueError("1-D quadrature needs an interval (a, b) or a PuncturedInterval")


class _Evaluator:
    """values = ev(points): one call of a BatchIntegrand body per <= max_batch points, or a loop over a plain
    callable.  Counts evaluations (EvalCounter)."""

    def __init__(self, f, p, point=None):
        from .solver import BatchIntegrand
        self.f, self.p, self.point = f, p, point or (lambda x: x)
        self.batch = isinstance(f, BatchIntegrand)
        self.numevals = 0

    def __call__(self, xs):
        pts = [self.point(x) for x in xs]
        self.numevals += len(pts)
        if not self.batch:
            return [self.f(x, self.p) for x in pts]
        out = []
        mb = self.f.max_batch
        for i in range(0, len(pts), mb):
            chunk = pts[i:i + mb]
            y = [None] * len(chunk)  # the resize!-able output buffer of the reference
            self.f.f(y, chunk, self.p)
            if any(v is None for v in y):
                raise ValueError("BatchIntegrand body left entries of y unset")
            out.extend(y)
        return out


class _GKRule:
    """Gauss-Kronrod (n, 2n+1) on an interval: nodes(a, b) and eval(values, a, b) -> (I, E = |I_K - I_G|).  Order 7 is
    the library's table (the one the device kernels use, bit for bit); any other order -- `QuadGKJL(order = 9)` in
    test/interface_tests.jl:151-156 -- is computed once on the host: the Kronrod points are the zeros of the Stieltjes
    polynomial E_{n+1}, orthogonal on [-1, 1] to all lower degrees with weight P_n, the weights follow from exactness."""

    def __init__(self, order):
        self.order, self.npts = int(order), 2 * int(order) + 1
        if self.order == 7:
            return
        from numpy.polynomial import legendre as Lg
        n = self.order
        xq, wq = Lg.leggauss(2 * n + 4)  # exact for every product below
        Pn = Lg.legval(xq, [0] * n + [1])
        basis = [Lg.legval(xq, [0] * k + [1]) for k in range(n + 2)]
        # E = P_{n+1} + sum_{k<=n} c_k P_k with  int P_n E P_j = 0  for j = 0..n
        A = np.array([[np.sum(wq * Pn * basis[k] * basis[j]) for k in range(n + 1)] for j in range(n + 1)])
        rhs = -np.array([np.sum(wq * Pn * basis[n + 1] * basis[j]) for j in range(n + 1)])
        c = np.linalg.lstsq(A, rhs, rcond=None)[0]
        kron = np.real(Lg.legroots(np.concatenate([c, [1.0]])))
        gauss, gw = Lg.leggauss(n)
        x = np.sort(np.concatenate([kron, gauss]))
        x = 0.5 * (x - x[::-1])  # symmetrise
        V = np.array([Lg.legval(x, [0] * k + [1]) for k in range(2 * n + 1)])
        mom = np.zeros(2 * n + 1)
        mom[0] = 2.0
        self.x, self.w = x, np.linalg.solve(V, mom)
        self.gidx = np.array([int(np.argmin(np.abs(x - g))) for g in gauss])
        self.gw = gw

    def nodes(self, a, b):
        if self.order == 7:
            return _gk_nodes(a, b)
        return 0.5 * (a + b) + 0.5 * (b - a) * self.x

    def eval(self, vals, a, b):
        if self.order == 7:
            return _gk_eval(vals, a, b)
        arr = np.asarray(vals)
        h = 0.5 * (b - a)
        Ik = h * np.tensordot(self.w, arr, axes=(0, 0))
        Ig = h * np.tensordot(self.gw, arr[self.gidx], axes=(0, 0))
        return (Ik if np.ndim(Ik) else Ik[()]), _norm(Ik - Ig)


_RULES = {}


def gk_rule(order):
    if order not in _RULES:
        _RULES[order] = _GKRule(order)
    return _RULES[order]


def auxquadgk(g, segs, atol, rtol, maxevals, batch=False, max_batch=2**62, order=7):
    """Globally adaptive GK(7,15).  Scalar refinement: pop the worst panel, bisect, 30 new nodes.  Batch refinement
    (AuxQuadGK.BatchIntegrand, reached from src/algorithms.jl:227-233): pop panels while the error of the REMAINING
    ones still exceeds the tolerance and 30 * popped <= max_batch, evaluate all children in ONE call of g.
    -> (I, E, numevals)."""
    from .solver import AuxValue
    atol_ = 0.0 if atol is None else atol
    anyatol = (atol_.val > 0 or atol_.aux > 0) if isinstance(atol_, AuxValue) else atol_ > 0
    rtol_ = (0.0 if anyatol else math.sqrt(np.finfo(float).eps)) if rtol is None else rtol
    nseg = len(segs) - 1
    rule = gk_rule(order)
    _gk_nodes, _gk_eval, npt = rule.nodes, rule.eval, rule.npts
    fv = g(np.concatenate([_gk_nodes(segs[i], segs[i + 1]) for i in range(nseg)]))
    if len(fv) and isinstance(fv[0], AuxValue):
        return _auxquadgk_auxvalue(g, segs, fv, atol_, rtol_, maxevals, batch, max_batch, rule)
    heap = _Heap()
    for i in range(nseg):
        Ii, Ei = _gk_eval(fv[npt * i:npt * (i + 1)], segs[i], segs[i + 1])
        heap.xs.append((segs[i], segs[i + 1], Ii, Ei))

    def total():
        I, E = heap.xs[0][2], heap.xs[0][3]
        for sg in heap.xs[1:]:
            I = I + sg[2]
            E = E + sg[3]
        return I, E

    I, E = total()
    numevals = npt * nseg
    if not (E <= max(atol_, rtol_ * _norm(I)) or numevals >= maxevals):
        heap.heapify()
        while E > max(atol_, rtol_ * _norm(I)) and numevals < maxevals:
            if not batch:
                popped = [heap.pop()]
                numevals += 2 * npt
            else:
                tol = max(atol_, rtol_ * _norm(I))
                popped = []
                while heap.xs and 2 * npt * (len(popped) + 1) <= max_batch and E > tol and numevals < maxevals:
                    sg = heap.pop()
                    popped.append(sg)
                    tol += sg[3]
                    numevals += 2 * npt
            pts = []
            for (sa, sb, _, _) in popped:
                mid = (sa + sb) / 2
                pts += [_gk_nodes(sa, mid), _gk_nodes(mid, sb)]
            fv = g(np.concatenate(pts))
            for k, (sa, sb, sI, sE) in enumerate(popped):
                mid = (sa + sb) / 2
                I1, E1 = _gk_eval(fv[2 * npt * k:2 * npt * k + npt], sa, mid)
                I2, E2 = _gk_eval(fv[2 * npt * k + npt:2 * npt * (k + 1)], mid, sb)
                I = (I - sI) + I1 + I2
                E = (E - sE) + E1 + E2
                heap.push((sa, mid, I1, E1))
                heap.push((mid, sb, I2, E2))
        I, E = total()
    return I, E, numevals


class _KeyHeap(_Heap):
    """The same heap ordered by one component of an (E_val, E_aux) error pair."""

    def __init__(self, key):
        super().__init__()
        self.key = key

    def lt(self, a, b):
        return b[3][self.key] < a[3][self.key]


def _auxquadgk_auxvalue(g, segs, fv, atol, rtol, maxevals, batch, max_batch, rule):
    """AuxValue integrands: the quadrature sums `val` and `aux` side by side with separate error estimates; refinement
    runs once per component in order -- first a heap ordered by the `val` errors until `val` meets its tolerance, then
    the surviving panels are re-heapified by their `aux` errors and refined until `aux` meets it too (a tolerance given
    as one number applies to both, an AuxValue tolerance to each).  ref: IteratedIntegration.AuxQuadGK (auxquadgk's
    `eachorder` loop), reached from src/algorithms.jl:215-239."""
    from .solver import AuxValue
    _gk_nodes, _gk_eval, npt = rule.nodes, rule.eval, rule.npts

    def pair(t):
        return (t.val, t.aux) if isinstance(t, AuxValue) else (t, t)

    def gk(vals, a, b):
        Iv, Ev = _gk_eval([v.val for v in vals], a, b)
        Ia, Ea = _gk_eval([v.aux for v in vals], a, b)
        return (Iv, Ia), (Ev, Ea)

    nseg = len(segs) - 1
    xs = []
    for i in range(nseg):
        Ii, Ei = gk(fv[npt * i:npt * (i + 1)], segs[i], segs[i + 1])
        xs.append((segs[i], segs[i + 1], Ii, Ei))

    def total(lst):
        I = [lst[0][2][0], lst[0][2][1]]
        E = [lst[0][3][0], lst[0][3][1]]
        for sg in lst[1:]:
            for c in (0, 1):
                I[c] = I[c] + sg[2][c]
                E[c] = E[c] + sg[3][c]
        return I, E

    I, E = total(xs)
    numevals = npt * nseg
    at = pair(atol)
    for key in (0, 1):
        def tol():
            return max(at[key], rtol * _norm(I[key]))
        if E[key] <= tol() or numevals >= maxevals:
            continue
        heap = _KeyHeap(key)
        heap.xs = xs
        heap.heapify()
        while E[key] > tol() and numevals < maxevals:
            popped = []
            if not batch:
                popped.append(heap.pop())
                numevals += 2 * npt
            else:
                t = tol()
                while heap.xs and 2 * npt * (len(popped) + 1) <= max_batch and E[key] > t and numevals < maxevals:
                    sg = heap.pop()
                    popped.append(sg)
                    t += sg[3][key]
                    numevals += 2 * npt
            pts = []
            for (sa, sb, _, _) in popped:
                mid = (sa + sb) / 2
                pts += [_gk_nodes(sa, mid), _gk_nodes(mid, sb)]
            fv = g(np.concatenate(pts))
            for k, (sa, sb, sI, sE) in enumerate(popped):
                mid = (sa + sb) / 2
                I1, E1 = gk(fv[2 * npt * k:2 * npt * k + npt], sa, mid)
                I2, E2 = gk(fv[2 * npt * k + npt:2 * npt * (k + 1)], mid, sb)
                for c in (0, 1):
                    I[c] = (I[c] - sI[c]) + I1[c] + I2[c]
                    E[c] = (E[c] - sE[c]) + E1[c] + E2[c]
                heap.push((sa, mid, I1, E1))
                heap.push((mid, sb, I2, E2))
        xs = heap.xs
        I, E = total(xs)
    return AuxValue(I[0], I[1]), AuxValue(E[0], E[1]), numevals


def solve_auxquadgk(f, dom, p, abstol, reltol, maxiters, order=7):
    """AuxQuadGKJL on an interval.  ref: src/algorithms.jl:215-239."""
    from .solver import BatchIntegrand, NestedBatchIntegrand
    if isinstance(f, NestedBatchIntegrand):
        raise ValueError("AuxQuadGKJL doesn't support nested batching")  # ref: src/algorithms.jl:211
    ev = _Evaluator(f, p, point=float)
    isb = isinstance(f, BatchIntegrand)
    I, E, _ = auxquadgk(ev, _segments(dom), abstol, reltol, maxiters, batch=isb, max_batch=f.max_batch if isb else 2**62, order=order)
    return I, E, ev.numevals


def _ptr_nodes(B, npt, syms):
    """Nodes B (i / npt) and weights of the PTR / Monkhorst-Pack rule on Basis(B).  ref: AutoSymPTR.PTR as used at
    src/algorithms.jl:347-352; symmetric rules take the library's integer tables (abz_symptr_rule)."""
    d = B.shape[0]
    if syms is None:
        idx = np.stack(np.meshgrid(*[np.arange(npt)] * d, indexing="ij"), -1).reshape(-1, d)[:, ::-1]  # i_1 fastest
        w = np.ones(len(idx))
        nsym = 1
    else:
        from .series import symptr_rule
        idx, w = symptr_rule(npt, d, syms)
        nsym = len(syms)
    x = (idx / float(npt)) @ B.T
    return x, w, abs(np.linalg.det(B)) / (npt**d * nsym)


def ptr_rule_value(f, B, p, npt, syms):
    """rule(f, Basis(B)) = sum_k w_k f(x_k) vol / (npt^d nsyms).  ref: src/algorithms.jl:368-380."""
    from .solver import NestedBatchIntegrand
    if isinstance(f, NestedBatchIntegrand):
        raise ValueError("MonkhorstPack doesn't support nested batching")  # ref: src/algorithms.jl:362,411
    x, w, scale = _ptr_nodes(B, npt, syms)
    ev = _Evaluator(f, p, point=(lambda v: float(v[0])) if B.shape[0] == 1 else (lambda v: v))
    vals = ev(list(x))
    acc = None
    for wk, v in zip(w, vals):
        t = wk * np.asarray(v)
        acc = t if acc is None else acc + t
    acc = acc * scale
    return (acc if np.ndim(acc) else acc[()]), ev.numevals


def solve_autosymptr(f, B, p, alg, abstol, reltol, maxiters):
    """autosymptr on a Basis.  ref: src/algorithms.jl:418-432 (+ SURVEY A.2): I1 = rule(n0), I2 = rule(n0 + dn),
    err = norm(I2 - I1), until err <= max(abstol, reltol norm(I2)) or numevals >= maxevals."""
    if abstol is None and reltol is None:
        rtol, atol = math.sqrt(np.finfo(float).eps), 0.0
    else:
        rtol, atol = (0.0 if reltol is None else reltol), (0.0 if abstol is None else abstol)
    n0, dn = alg.inner.npt_sequence()
    npt = n0
    I1, nev = ptr_rule_value(f, B, p, npt, alg.syms)
    while True:
        npt += dn
        I2, n2 = ptr_rule_value(f, B, p, npt, alg.syms)
        nev += n2
        err = _norm(np.asarray(I2) - np.asarray(I1))
        if err <= max(atol, rtol * _norm(I2)) or nev >= maxiters or not np.isfinite(err):
            return I2, err, nev
        I1 = I2


def _fixed_rule(alg, g, segs):
    """QuadratureFunction on the segments of a 1-D domain: sum_seg s sum_j w_j g(a + s (1 + x_j)).  -> (I, None, numevals)"""
    x, w = alg.fun(alg.npt)
    x, w = np.asarray(x, dtype=np.float64), np.asarray(w, dtype=np.float64)
    pts = np.concatenate([segs[i] + 0.5 * (segs[i + 1] - segs[i]) * (1.0 + x) for i in range(len(segs) - 1)])
    vals = g(pts)
    I = None
    for i in range(len(segs) - 1):
        sc = 0.5 * (segs[i + 1] - segs[i])
        for j in range(len(x)):
            t = vals[i * len(x) + j] * (w[j] * sc)
            I = t if I is None else I + t
    return I, None, len(pts)


def solve_quadrature_function(f, dom, p, alg):
    """ref: src/algorithms.jl:167-191."""
    from .solver import NestedBatchIntegrand
    if isinstance(f, NestedBatchIntegrand):
        raise ValueError("QuadratureFunction doesn't support nested batching")
    ev = _Evaluator(f, p, point=float)
    I, E, _ = _fixed_rule(alg, ev, _segments(dom))
    return I, E, ev.numevals


def nested_quad(f, lims, p, abstol, reltol, maxiters, algs=None):
    """NestedQuad(algs...) -- AuxQuadGKJL / QuadGKJL (adaptive) or QuadratureFunction (fixed rule) per level, the last
    one repeated inwards like the reference's single-algorithm form -- over iterated limits (bz.CubicLimits / TetrahedralLimits / ...: `segs()`, `fix(x)`)
    for a plain callable (depth-first, scalar refinement), a BatchIntegrand (the innermost integral only is batched)
    or a NestedBatchIntegrand (every level batched).  The integrand sees full points x = (x_1 .. x_d)."""
    from .solver import BatchIntegrand, NestedBatchIntegrand
    nest = isinstance(f, NestedBatchIntegrand)
    worker = f.f[0] if nest else f
    while isinstance(worker, NestedBatchIntegrand):
        worker = worker.f[0]
    inner_batch = nest or isinstance(f, BatchIntegrand)
    mb = f.max_batch if (nest or isinstance(f, BatchIntegrand)) else 2**62
    count = [0]
    from .solver import QuadratureFunction
    algs = tuple(algs) if algs else ()
    ntot = lims.ndim

    def alg_of(d):  # level d = number of variables still free; algs[0] is the OUTERMOST level's algorithm
        if not algs:
            return None
        return algs[min(ntot - d, len(algs) - 1)]

    def integrate(alg, g, segs, atol, batch):
        if isinstance(alg, QuadratureFunction):
            I, E, _ = _fixed_rule(alg, g, segs)
            return I, (0.0 if E is None else E)
        I, E, _ = auxquadgk(g, segs, atol, reltol, maxiters, batch=batch, max_batch=mb, order=getattr(alg, "order", 7))
        return I, E

    def level(lim, tail, atol):
        d = lim.ndim
        if d == 1:
            ev = _Evaluator(worker if nest else f, p, point=lambda x: np.array((float(x),) + tail))
            I, E = integrate(alg_of(1), ev, tuple(lim.segs()), atol, inner_batch)
            count[0] += ev.numevals
            return I, E

        def g(xs):
            out = []
            for x in xs:
                inner = lim.fix(float(x))
                sg = inner.segs()
                at = None if atol is None else atol / (sg[-1] - sg[0])  # ref: src/algorithms.jl:532-533,556-557
                out.append(level(inner, (float(x),) + tail, at)[0])
            return out
        return integrate(alg_of(d), g, tuple(lim.segs()), atol, nest)

    I, E = level(lims, (), abstol)
    return I, E, count[0]


def fourier_batch(g, series, max_batch=2**62, want_eig=False):
    """BatchIntegrand whose body evaluates the series at ALL nodes of a batch on the GPU (one abz_eval_nodes call) and
    applies the user closure g(FourierValue(k, H(k)), p) per node -- the reference's GPU hook (src/batch.jl:4-6) for
    MonkhorstPack, AutoSymPTRJL, AuxQuadGKJL (1-D series) and NestedQuad.  With `want_eig` the FourierValue carries
    (H(k), eigenvalues)."""
    from .solver import BatchIntegrand, FourierValue

    def body(y, x, p):
        k = np.asarray(x, dtype=np.float64).reshape(len(x), series.d)
        dev = series.device()
        if want_eig:
            H, E = dev.eval_nodes(k, want=L.WANT_H | L.WANT_EIG)
            for i in range(len(x)):
                y[i] = g(FourierValue(k[i] if series.d > 1 else float(k[i, 0]), (H[i], E[i])), p)
        else:
            H = dev.eval_nodes(k)
            for i in range(len(x)):
                y[i] = g(FourierValue(k[i] if series.d > 1 else float(k[i, 0]), H[i]), p)

    return BatchIntegrand(body, max_batch=max_batch)


# ---------------------------------------------------------------------------- HCubatureJL / TAI (tree-adaptive cubature)
def _genz_malik(d):
    """Points (unit cube [-1, 1]^d) and weights of the Genz-Malik degree-7 rule with its embedded degree-5 rule
    (A. Genz, A. Malik, J. Comput. Appl. Math. 6 (1980) 295), the rule HCubature.jl uses for d >= 2: 2^d + 2 d^2 + 2 d + 1
    points.  Returns (points [npts, d], w7, w5, index blocks)."""
    l2, l3, l4, l5 = math.sqrt(9 / 70), math.sqrt(9 / 10), math.sqrt(9 / 10), math.sqrt(9 / 19)
    pts = [np.zeros(d)]
    for lam in (l2, l3):
        for i in range(d):
            for sgn in (1.0, -1.0):
                x = np.zeros(d)
                x[i] = sgn * lam
                pts.append(x)
    for i in range(d):
        for j in range(i + 1, d):
            for si in (1.0, -1.0):
                for sj in (1.0, -1.0):
                    x = np.zeros(d)
                    x[i], x[j] = si * l4, sj * l4
                    pts.append(x)
    for bits in range(2 ** d):
        pts.append(np.array([l5 if (bits >> i) & 1 == 0 else -l5 for i in range(d)]))
    pts = np.array(pts)
    n2 = 2 * d
    n4 = 2 * d * (d - 1)
    n5 = 2 ** d
    w7 = np.concatenate([[(12824 - 9120 * d + 400 * d * d) / 19683], np.full(n2, 980 / 6561), np.full(n2, (1820 - 400 * d) / 19683),
                         np.full(n4, 200 / 19683), np.full(n5, 6859 / 19683 / 2 ** d)])
    w5 = np.concatenate([[(729 - 950 * d + 50 * d * d) / 729], np.full(n2, 245 / 486), np.full(n2, (265 - 100 * d) / 1458),
                         np.full(n4, 25 / 729), np.zeros(n5)])
    return pts, w7, w5


_GM = {}


def hcubature(fbatch, a, b, atol=0.0, rtol=0.0, maxevals=2**62, initdiv=1):
    """Globally adaptive cubature over the box [a, b] like HCubature.jl's `hcubature` / `hquadrature` (the routine behind
    HCubatureJL and TAI, src/algorithms.jl:94-124, src/brillouin.jl:446-463): Genz-Malik (7, 5) rule per box in d >= 2,
    GK(7, 15) in d = 1; the box with the largest error is halved along the axis of its largest fourth divided difference
    (ties within the error density: the widest axis); stop at E <= max(atol, rtol |I|).  `fbatch(points [m, d])` returns the
    m values (numbers or arrays).  -> (I, E, numevals)."""
    a = np.atleast_1d(np.asarray(a, dtype=np.float64))
    b = np.atleast_1d(np.asarray(b, dtype=np.float64))
    d = len(a)
    if rtol == 0 and atol == 0:
        rtol = math.sqrt(np.finfo(float).eps)
    if d == 1:
        g = lambda xs: fbatch(np.asarray(xs, dtype=np.float64).reshape(-1, 1))
        segs = tuple(np.linspace(a[0], b[0], max(1, int(initdiv)) + 1))
        return auxquadgk(g, segs, atol, rtol, maxevals)
    if d not in _GM:
        _GM[d] = _genz_malik(d)
    pts, w7, w5 = _GM[d]
    npts = len(pts)

    def rule(lo, hi):
        c, h = 0.5 * (lo + hi), 0.5 * (hi - lo)
        vals = fbatch(c + pts * h)
        vol = float(np.prod(hi - lo))
        f = [np.asarray(v) for v in vals]
        I7 = sum(w * v for w, v in zip(w7, f)) * vol
        I5 = sum(w * v for w, v in zip(w5, f)) * vol
        E = _norm(I7 - I5)
        # fourth divided differences along the axes: |f(+l2) + f(-l2) - 2 f0 - (l2/l3)^2 (f(+l3) + f(-l3) - 2 f0)|
        div = np.empty(d)
        for i in range(d):
            f2 = f[1 + 2 * i] + f[2 + 2 * i]
            f3 = f[1 + 2 * d + 2 * i] + f[2 + 2 * d + 2 * i]
            div[i] = _norm(f2 - 2 * f[0] - (f3 - 2 * f[0]) / 7.0)
        delta = E / (10.0 ** d * vol) if vol > 0 else 0.0
        k = int(np.argmax(div))
        width = hi - lo
        for i in range(d):
            if i != k and abs(div[i] - div[k]) <= delta and width[i] > width[k]:
                k = i
        return (I7 if np.ndim(I7) else I7[()]), E, k

    boxes = []
    grid = [np.linspace(a[i], b[i], max(1, int(initdiv)) + 1) for i in range(d)]
    import itertools as _it
    for idx in _it.product(range(max(1, int(initdiv))), repeat=d):
        lo = np.array([grid[i][idx[i]] for i in range(d)])
        hi = np.array([grid[i][idx[i] + 1] for i in range(d)])
        boxes.append((lo, hi) + rule(lo, hi))
    numevals = npts * len(boxes)
    I = sum(bx[2] for bx in boxes)
    E = sum(bx[3] for bx in boxes)
    heap = _KeyHeap(0)
    heap.lt = lambda x, y: y[3] < x[3]
    heap.xs = boxes
    heap.heapify()
    while E > max(atol, rtol * _norm(I)) and numevals < maxevals:
        lo, hi, bI, bE, k = heap.pop()
        mid = 0.5 * (lo[k] + hi[k])
        hi1, lo2 = hi.copy(), lo.copy()
        hi1[k], lo2[k] = mid, mid
        c1 = (lo, hi1) + rule(lo, hi1)
        c2 = (lo2, hi) + rule(lo2, hi)
        numevals += 2 * npts
        I = (I - bI) + c1[2] + c2[2]
        E = (E - bE) + c1[3] + c2[3]
        heap.push(c1)
        heap.push(c2)
    I = sum(bx[2] for bx in heap.xs)  # re-sum: the running totals carry the roundoff of every update
    E = sum(bx[3] for bx in heap.xs)
    return I, E, numevals


def solve_hcubature(f, dom, p, alg, abstol, reltol, maxiters):
    """HCubatureJL on a HyperCube (or an interval).  ref: src/algorithms.jl:104-124."""
    from .solver import BatchIntegrand, NestedBatchIntegrand
    if isinstance(f, NestedBatchIntegrand):
        raise ValueError("HCubatureJL doesn't support nested batching")
    if isinstance(f, BatchIntegrand):
        raise ValueError("HCubatureJL doesn't support batching")
    if hasattr(dom, "a") and hasattr(dom, "b"):
        a, b = np.atleast_1d(dom.a), np.atleast_1d(dom.b)
    else:
        segs = _segments(dom)
        a, b = np.array([segs[0]]), np.array([segs[-1]])
    scalar = len(a) == 1
    ev = _Evaluator(f, p, point=(lambda x: float(x[0])) if scalar else (lambda x: np.asarray(x, dtype=np.float64)))
    I, E, _ = hcubature(ev, a, b, 0.0 if abstol is None else abstol, 0.0 if reltol is None else reltol, maxiters, alg.initdiv)
    return I, E, ev.numevals


# ---------------------------------------------------------------------------- ContQuadGKJL / MeroQuadGKJL
def _near_poles(ts, fvals, rho):
    """Zeros of the degree-14 interpolant of 1/f through the 15 Kronrod points (standard coordinate t in [-1, 1]) that lie
    inside the Bernstein ellipse with semi-axes cosh(rho), sinh(rho): [(t0, d(1/f)/dt at t0)].  The reference finds them by
    Newton deflation on the same kind of interpolant (IteratedIntegration.ContQuadGK / MeroQuadGK, `rootmeth`); here the
    companion matrix of the Chebyshev series gives all of them at once."""
    from numpy.polynomial import chebyshev as Ch
    f = np.asarray(fvals, dtype=np.complex128)
    if not np.all(np.isfinite(f)) or np.any(f == 0):
        return []
    g = 1.0 / f
    scale = np.abs(g).max()
    if not scale > 0:
        return []
    V = Ch.chebvander(np.asarray(ts, dtype=np.float64), len(ts) - 1)
    try:
        c = np.linalg.solve(V.astype(np.complex128), g / scale)
    except np.linalg.LinAlgError:
        return []
    while len(c) > 1 and abs(c[-1]) < 1e-14 * np.abs(c).max():
        c = c[:-1]
    if len(c) < 2:
        return []
    roots = Ch.chebroots(c)
    dc = Ch.chebder(c)
    out = []
    for z in roots:
        w = z + np.sqrt(z * z - 1 + 0j)
        if abs(w) < 1:
            w = 1 / w
        if abs(w) < math.exp(rho):
            out.append((complex(z), complex(Ch.chebval(z, dc)) * scale))
    return out


def _cquad(f, p, dom, alg, abstol, reltol, maxiters, mode):
    """The shared adaptive loop of ContQuadGKJL (mode "cont") and MeroQuadGKJL (mode "mero"): globally adaptive GK(7,15)
    over segments with real or complex end points; a REAL segment whose integrand has poles within the Bernstein ellipse of
    parameter rho is treated specially.  ref: src/algorithms.jl:242-328 (behaviour as documented there: plain quadgk unless a
    root of 1/f is found nearby)."""
    from .solver import BatchIntegrand, InplaceIntegrand, NestedBatchIntegrand
    name = "ContQuadGK" if mode == "cont" else "MeroQuadGK"
    if isinstance(f, NestedBatchIntegrand):
        raise ValueError(f"{name} doesn't support nested batching")
    if isinstance(f, BatchIntegrand):
        raise ValueError(f"{name} doesn't support batching")
    if isinstance(f, InplaceIntegrand):
        raise ValueError(f"{name} doesn't support inplace integrands")
    rule = gk_rule(alg.order)
    tk = np.asarray(rule.nodes(-1.0, 1.0), dtype=np.float64)
    atol = 0.0 if abstol is None else abstol
    rtol = (0.0 if atol > 0 else math.sqrt(np.finfo(float).eps)) if reltol is None else reltol
    nev = [0]

    def call(x):
        nev[0] += 1
        return complex(f(x, p))

    def plain(a, b):
        """GK on the straight segment a -> b (complex end points allowed): (I, E)."""
        c, h = 0.5 * (a + b), 0.5 * (b - a)
        xs = [c + h * t for t in tk]
        if isinstance(a, float) and isinstance(b, float):
            vals = [call(float(x)) for x in xs]
        else:
            vals = [call(complex(x)) for x in xs]
        Ik, Ek = rule.eval(vals, -1.0, 1.0)  # on the standard interval; the (complex) half length h scales it
        return Ik * h, Ek * abs(h), vals

    def segment(a, b):
        """-> list of heap entries (a, b, I, E) for the piece a -> b of the path"""
        I, E, vals = plain(a, b)
        if not (isinstance(a, float) and isinstance(b, float)):
            return [(a, b, I, E)]
        poles = _near_poles(tk, vals, alg.rho)
        poles = [(t0, dg) for t0, dg in poles if abs(t0.imag) > 1e-14 and dg != 0]
        if not poles:
            return [(a, b, I, E)]
        c, h = 0.5 * (a + b), 0.5 * (b - a)
        if mode == "mero":
            # subtract r / (x - x0) for every nearby simple pole, add r log((b - x0) / (a - x0)) back
            xs = [c + h * t for t in tk]
            sub = np.array(vals, dtype=np.complex128)
            extra = 0.0j
            for t0, dg in poles:
                x0 = c + h * t0
                r = h / dg  # residue of f = 1 / (d(1/f)/dx) = h / (d(1/f)/dt)
                sub = sub - r / (np.array(xs) - x0)
                extra += r * (np.log(b - x0) - np.log(a - x0))
            Ik, Ek = rule.eval(list(sub), -1.0, 1.0)
            if Ek * abs(h) < E:  # keep the subtraction only where it helps
                return [(a, b, Ik * h + extra, Ek * abs(h))]
            return [(a, b, I, E)]
        ups = [t0 for t0, _ in poles if t0.imag > 0]
        downs = [t0 for t0, _ in poles if t0.imag < 0]
        if ups and downs:
            return [(a, b, I, E)]  # poles on both sides: no dent helps, refine on the real axis
        side = -1.0 if ups else 1.0  # go away from the poles
        m = complex(c, side * abs(h) * math.sinh(alg.rho))
        out = []
        for (u, v) in ((a, m), (m, b)):
            I2, E2, _ = plain(u, v)
            out.append((u, v, I2, E2))
        return out

    segs = _segments(dom)
    heap = _KeyHeap(0)
    heap.lt = lambda x, y: y[3] < x[3]
    for i in range(len(segs) - 1):
        heap.xs.extend(segment(float(segs[i]), float(segs[i + 1])))
    heap.heapify()
    I = sum(sg[2] for sg in heap.xs)
    E = sum(sg[3] for sg in heap.xs)
    while E > max(atol, rtol * abs(I)) and nev[0] < maxiters:
        a, b, sI, sE = heap.pop()
        mid = 0.5 * (a + b)
        if isinstance(a, float) and isinstance(b, float):
            mid = float(mid)
        new = segment(a, mid) + segment(mid, b)
        I = I - sI + sum(sg[2] for sg in new)
        E = E - sE + sum(sg[3] for sg in new)
        for sg in new:
            heap.push(sg)
    I = sum(sg[2] for sg in heap.xs)
    E = sum(sg[3] for sg in heap.xs)
    return I, E, nev[0]


def solve_contquadgk(f, dom, p, alg, abstol, reltol, maxiters):
    return _cquad(f, p, dom, alg, abstol, reltol, maxiters, "cont")


def solve_meroquadgk(f, dom, p, alg, abstol, reltol, maxiters):
    return _cquad(f, p, dom, alg, abstol, reltol, maxiters, "mero")
