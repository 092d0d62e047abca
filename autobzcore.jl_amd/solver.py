"""Problem / solver API and the algorithm dispatch for Fourier integrands.

Mirrors src/interfaces.jl (IntegralProblem, init/solve/solve!, IntegralSolver, batchsolve),
src/parameters.jl (MixedParameters, ParameterIntegrand, paramzip/paramproduct), src/batch.jl
(BatchIntegrand), src/fourier.jl (FourierIntegrand / FourierValue and their PTR / AutoPTR / IAI
dispatch) and src/brillouin.jl (IAI / PTR / AutoPTR, do_solve_autobz).  The numerics run in
libabzhip.so; the adaptive logic (AutoPTR grid sequence, error tests, parameter sweeps) lives here.
"""
import ctypes as C
import itertools
import math
import time
import warnings
from dataclasses import dataclass, field
from typing import Any, Callable, Optional

import numpy as np

from . import _lib as L
from .bz import (Basis, CubicLimits, HyperCube, PolygonLimits, PolyhedralLimits, PuncturedInterval, SymmetricBZ,
                 TetrahedralLimits, nsyms)
from .series import FourierSeries


# ---------------------------------------------------------------------------- parameters
class NullParameters:
    """ref: src/interfaces.jl:23."""


class MixedParameters:
    """Positional + keyword parameters of an integrand.  ref: src/parameters.jl:11-35."""

    def __init__(self, *args, **kwargs):
        object.__setattr__(self, "args", tuple(args))
        object.__setattr__(self, "kwargs", dict(kwargs))

    def __getitem__(self, i):
        return self.args[i]

    def __getattr__(self, name):
        try:
            return object.__getattribute__(self, "kwargs")[name]
        except KeyError:
            raise AttributeError(name)

    def merge(self, q):
        """merge(p, q): tuples/MixedParameters append args, dicts overwrite keywords, anything
        else is appended as one positional argument.  ref: src/parameters.jl:23-35."""
        if isinstance(q, MixedParameters):
            return MixedParameters(*(self.args + q.args), **{**self.kwargs, **q.kwargs})
        if isinstance(q, dict):
            return MixedParameters(*self.args, **{**self.kwargs, **q})
        if isinstance(q, tuple):
            return MixedParameters(*(self.args + q), **self.kwargs)
        if isinstance(q, NullParameters) or q is None:
            return self
        return MixedParameters(*(self.args + (q,)), **self.kwargs)

    def __eq__(self, o):
        return isinstance(o, MixedParameters) and self.args == o.args and self.kwargs == o.kwargs

    def __repr__(self):
        return f"MixedParameters{self.args}{self.kwargs}"


def paramzip(*args, **kwargs):
    """ref: src/parameters.jl:52-60.  Scalars zip to a 0-d array (Julia numbers iterate as 0-d containers)."""
    if (args or kwargs) and all(np.ndim(a) == 0 for a in list(args) + list(kwargs.values())):
        out = np.empty((), dtype=object)
        out[()] = MixedParameters(*args, **kwargs)
        return out
    n = len(args[0]) if args else len(next(iter(kwargs.values())))
    return [MixedParameters(*(a[i] for a in args), **{k: v[i] for k, v in kwargs.items()}) for i in range(n)]


def paramproduct(*args, **kwargs):
    """ref: src/parameters.jl:62-73 (first factor fastest, like Iterators.product)."""
    seqs = list(args) + list(kwargs.values())
    keys = list(kwargs.keys())
    out = []
    for rev in itertools.product(*[range(len(s)) for s in reversed(seqs)]):
        idx = rev[::-1]
        vals = [s[i] for s, i in zip(seqs, idx)]
        out.append(MixedParameters(*vals[: len(args)], **dict(zip(keys, vals[len(args):]))))
    arr = np.empty(len(out), dtype=object)
    arr[:] = out
    return arr.reshape([len(s) for s in seqs], order="F")


class AuxValue:
    """A value that carries an auxiliary quantity integrated along with it: both are summed by the quadrature, the adaptive
    loop converges `val` first and `aux` after it (each with its own error estimate).
    ref: IteratedIntegration.AuxQuadGK.AuxValue, used at src/algorithms.jl:198, src/brillouin.jl:113, ext/HDF5Ext.jl:48-64."""
    __array_ufunc__ = None  # numpy scalars defer to __rmul__ / __radd__

    def __init__(self, val, aux):
        self.val, self.aux = val, aux

    def __add__(self, o):
        return AuxValue(self.val + o.val, self.aux + o.aux)

    def __sub__(self, o):
        return AuxValue(self.val - o.val, self.aux - o.aux)

    def __neg__(self):
        return AuxValue(-self.val, -self.aux)

    def __mul__(self, c):
        return AuxValue(self.val * c, self.aux * c)

    __rmul__ = __mul__

    def __truediv__(self, c):
        return AuxValue(self.val / c, self.aux / c)

    def __eq__(self, o):
        return isinstance(o, AuxValue) and np.array_equal(self.val, o.val) and np.array_equal(self.aux, o.aux)

    def __iter__(self):
        return iter((self.val, self.aux))

    def __repr__(self):
        return f"AuxValue({self.val!r}, {self.aux!r})"


class ParameterIntegrand:
    """f(x, args...; kwargs...) with a partial set of parameters.  ref: src/parameters.jl:75-97."""

    def __init__(self, f, *args, **kwargs):
        self.f = f
        self.p = MixedParameters(*args, **kwargs)

    def __call__(self, x, q=None):
        p = self.p.merge(q)
        return self.f(x, *p.args, **p.kwargs)


# ---------------------------------------------------------------------------- integrands
@dataclass
class FourierValue:
    """(x, s): point and series value handed to integrands.  ref: src/fourier.jl:111-118."""

    x: Any
    s: Any


class BatchIntegrand:
    """f!(y, x, p) fills y[i] for nodes x[i].  ref: src/batch.jl:1-38."""

    def __init__(self, f, y=None, x=None, max_batch=2**62):
        if max_batch <= 0:
            raise ValueError("maximum batch size must be positive")
        self.f = f
        self.y = y
        self.x = x
        self.max_batch = int(max_batch)


class InplaceIntegrand:
    """InplaceIntegrand(f!, result): f!(y, x, p) writes the (array) value at one point into y; `result` gives the shape and
    type.  ref: src/inplace.jl:1-15.  Host-side: every evaluation gets a fresh y, the solution is an array like `result`."""

    def __init__(self, f, I):
        self.f = f
        self.I = np.asarray(I)

    def __call__(self, x, p):
        y = np.zeros_like(self.I)
        self.f(y, x, p)
        return y


class NestedBatchIntegrand:
    """Tuple of worker integrands + resizable buffers for (thread-)parallel nested quadrature.
    ref: src/batch.jl:41-77.  On the GPU path the workers are not needed (every batch is one kernel
    launch); what the type selects is the BatchIntegrand refinement rule of auxquadgk at every level
    of NestedQuad (several panels popped per round, src/fourier.jl:441-473) and its max_batch."""

    def __init__(self, f, y=None, x=None, max_batch=2**62):
        if max_batch <= 0:
            raise ValueError("maximum batch size must be positive")
        self.f = tuple(f) if isinstance(f, (tuple, list)) else (f,)
        self.y, self.x = y, x
        self.max_batch = int(max_batch)


class DeviceIntegrand:
    """A built-in integrand evaluated on the GPU, fused with the Fourier evaluation / reduction.
    Positional parameter names follow the reference functions they stand for."""

    fid = None
    argnames = ()
    nparams = 0      # leading args that are fixed parameters
    swept = False    # last arg is the swept parameter (omega)
    matrix = False   # value is a matrix/vector (UnknownRep under symmetries unless `symrep` is set)
    symrep = None    # an AbstractSymRep: how the integral maps from the IBZ to the full BZ (SymRep(f))

    def with_symrep(self, rep):
        """The same integrand with a symmetry representation attached (the reference's `SymRep(f)` trait)."""
        import copy
        g = copy.copy(self)
        g.symrep = rep
        return g

    def bind(self, p: MixedParameters):
        vals = list(p.args)
        kw = dict(p.kwargs)
        names = list(self.argnames)
        for i, nm in enumerate(names):
            if i < len(vals):
                if nm in kw:
                    raise ValueError(f"parameter {nm} given twice")
            elif nm in kw:
                vals.append(kw.pop(nm))
            else:
                raise ValueError(f"{type(self).__name__}: missing parameter {nm}")
        if kw or len(vals) != len(names):
            raise ValueError(f"{type(self).__name__}: expected parameters {names}")
        vals = [float(v) for v in vals]
        if self.swept:
            return vals[:-1], vals[-1]
        return vals, None

    def host(self, v: FourierValue, *args):
        raise NotImplementedError


class UnitIntegrand(DeviceIntegrand):
    """(x, p) -> 1.  ref: test/brillouin.jl:38."""
    fid = L.F_ONE

    def host(self, v):
        return 1.0


class LinearIntegrand(DeviceIntegrand):
    """f(x, a; b) = a*x.s + b.  ref: test/fourier.jl:41."""
    fid = L.F_LINEAR
    argnames = ("a", "b")

    def host(self, v, a, b):
        return a * v.s + b


class LinearXIntegrand(DeviceIntegrand):
    """f(x, a; b) = a*x.s*x.x .+ b.  ref: test/fourier.jl:16."""
    fid = L.F_LINEAR_X
    argnames = ("a", "b")
    matrix = True

    def host(self, v, a, b):
        return a * v.s * np.asarray(v.x) + b


class DOSIntegrand(DeviceIntegrand):
    """dos_integrand(h_k, eta, omega) = -imag(tr(inv((omega+im*eta)*I - h_k.s)))/pi.
    ref: aps_example/aps_example.jl:30.  form="eig" evaluates the same value from cached
    eigenvalues: (eta/pi) sum_b 1/((omega-e_b)^2 + eta^2)."""
    argnames = ("eta", "omega")
    swept = True

    def __init__(self, form="inv"):
        if form not in ("inv", "eig"):
            raise ValueError("form must be 'inv' or 'eig'")
        self.form = form
        self.fid = L.F_DOS if form == "inv" else L.F_DOS_EIG

    def host(self, v, eta, omega):
        h = np.atleast_2d(v.s)
        return -np.imag(np.trace(np.linalg.inv((omega + 1j * eta) * np.eye(len(h)) - h))) / np.pi


class TrGlocIntegrand(DeviceIntegrand):
    """tr inv((omega + i eta) I - H).  ref: docs/src/examples.md:12-15."""
    fid = L.F_TRGLOC
    argnames = ("eta", "omega")
    swept = True

    def host(self, v, eta, omega):
        h = np.atleast_2d(v.s)
        return np.trace(np.linalg.inv(complex(omega, eta) * np.eye(len(h)) - h))


class GlocIntegrand(DeviceIntegrand):
    """gloc_integrand(h_k; eta, omega) = inv(complex(omega,eta)*I - h_k.s).  ref: docs/src/examples.md:90."""
    fid = L.F_GLOC
    argnames = ("eta", "omega")
    swept = True
    matrix = True

    def host(self, v, eta, omega):
        h = np.atleast_2d(v.s)
        g = np.linalg.inv(complex(omega, eta) * np.eye(len(h)) - h)
        return g[0, 0] if np.ndim(v.s) == 0 else g


class FourierIntegrand:
    """f(FourierValue(x, w(x)), args...; kwargs...) with the series evaluated on the GPU one
    dimension at a time.  ref: src/fourier.jl:22-58.  `f` is a DeviceIntegrand (fused on the GPU)
    or any Python callable (H(k) batches come back to the host)."""

    def __init__(self, f, w, *args, **kwargs):
        self.nest = None
        if args and isinstance(args[0], NestedBatchIntegrand):  # FourierIntegrand(p, w, nest), src/fourier.jl:29-31
            self.nest = args[0]
            args = args[1:]
        if isinstance(f, ParameterIntegrand):
            self.f = f
        else:
            self.f = ParameterIntegrand(f, *args, **kwargs)
        if not isinstance(w, FourierSeries):
            raise TypeError("FourierIntegrand needs a FourierSeries")
        self.w = w

    def __call__(self, x, p=None):
        if isinstance(x, FourierValue):
            return _call_user(self.f.f, x, self.f.p.merge(p))
        return self(FourierValue(np.asarray(x), self.w(x)), p)  # ref: src/fourier.jl:120-122


def _call_user(f, v, p: MixedParameters):
    if isinstance(f, DeviceIntegrand):
        params, sw = f.bind(p)
        return f.host(v, *(params + ([sw] if sw is not None else [])))
    return f(v, *p.args, **p.kwargs)


# ---------------------------------------------------------------------------- algorithms
class IntegralAlgorithm:
    pass


class AutoBZAlgorithm(IntegralAlgorithm):
    pass


class AuxQuadGKJL(IntegralAlgorithm):
    """ref: src/algorithms.jl:202-208.  Only order 7 (GK(7,15)) is built."""

    def __init__(self, order=7, norm=None):
        if order != 7:
            raise ValueError("AuxQuadGKJL: only order = 7 is supported")
        self.order = order
        self.norm = norm


class ContQuadGKJL(IntegralAlgorithm):
    """1-D contour deformation for scalar complex integrands: quadgk on the real axis unless 1/f has a root within the
    Bernstein ellipse (semi-axes cosh rho, sinh rho) of a segment on one side of the axis -- the segment is then dented
    away from it (the integrand must accept complex arguments).  ref: src/algorithms.jl:242-290; host-side, generic.py."""

    def __init__(self, order=7, norm=None, rho=1.0, rootmeth=None):
        self.order, self.norm, self.rho, self.rootmeth = int(order), norm, float(rho), rootmeth


class MeroQuadGKJL(IntegralAlgorithm):
    """1-D pole subtraction for meromorphic scalar integrands: quadgk on the real axis, with the simple poles found
    within the Bernstein ellipse of a segment subtracted and integrated analytically.  ref: src/algorithms.jl:292-328."""

    def __init__(self, order=7, norm=None, rho=1.0, rootmeth=None):
        self.order, self.norm, self.rho, self.rootmeth = int(order), norm, float(rho), rootmeth


def trapz(n):
    """Nodes and weights of the trapezoidal rule on [-1, 1].  ref: src/algorithms.jl:132-140."""
    if n <= 1:
        raise ValueError("trapz needs at least two points")
    x = np.linspace(-1.0, 1.0, n)
    h = x[1] - x[0]
    w = np.full(n, h)
    w[0] = w[-1] = h / 2
    return x, w


class QuadratureFunction(IntegralAlgorithm):
    """A fixed rule `x, w = fun(npt)` for [-1, 1] applied to every segment of the domain (default: trapz, 50 points); no
    error estimate.  ref: src/algorithms.jl:142-191."""

    def __init__(self, fun=trapz, npt=50, nthreads=1):
        self.fun, self.npt, self.nthreads = fun, int(npt), nthreads


class QuadGKJL(AuxQuadGKJL):
    """ref: src/algorithms.jl:9-20,67-96 (the duplicate of Integrals.jl's QuadGKJL): the same globally adaptive
    Gauss-Kronrod as AuxQuadGKJL without the AuxValue ordering -- which only differs for AuxValue integrands.  Host-side
    only, so any `order` is allowed (the rule is computed on first use, generic._GKRule); the device paths are GK(7,15)."""

    def __init__(self, order=7, norm=None):
        if int(order) < 1:
            raise ValueError("QuadGKJL: order must be positive")
        self.order = int(order)
        self.norm = norm


class IAI(AutoBZAlgorithm):
    """Iterated adaptive integration.  ref: src/brillouin.jl:368-377."""

    def __init__(self, *algs):
        self.algs = algs or (AuxQuadGKJL(),)
        for a in self.algs:
            if not isinstance(a, AuxQuadGKJL):
                raise ValueError("IAI: only AuxQuadGKJL() inner algorithms are supported")


class PTR(AutoBZAlgorithm):
    """ref: src/brillouin.jl:386-394."""

    def __init__(self, npt=50, nthreads=1):
        self.npt = int(npt)
        self.nthreads = nthreads


class AutoPTR(AutoBZAlgorithm):
    """ref: src/brillouin.jl:405-420."""

    def __init__(self, norm=None, a=1.0, nmin=50, nmax=1000, n0=6.0, dn=math.log(10), keepmost=2, nthreads=1):
        self.norm = norm
        self.a, self.nmin, self.nmax, self.n0, self.dn = float(a), int(nmin), int(nmax), float(n0), float(dn)
        self.keepmost = keepmost
        self.nthreads = nthreads

    def npt_sequence(self):
        """Integer (n0', dn') of AutoSymPTR.MonkhorstPackRule read at src/fourier.jl:301-321:
        clamp(round(x/a), nmin, nmax); defaults give npt = 50, 100, 150, ...."""
        n0 = int(min(max(round(self.n0 / self.a), self.nmin), self.nmax))
        dn = int(min(max(round(self.dn / self.a), self.nmin), self.nmax))
        return n0, dn


class HCubatureJL(IntegralAlgorithm):
    """ref: src/algorithms.jl:94-124 (h-adaptive cubature: Genz-Malik boxes, GK(7,15) in one dimension).  Host-side: the
    tree lives in generic.hcubature; a FourierIntegrand's series values at every box's points come from ONE
    abz_eval_nodes call (the fallback evaluator, src/fourier.jl:120-122)."""

    def __init__(self, norm=None, initdiv=1):
        self.norm, self.initdiv = norm, int(initdiv)


class TAI(AutoBZAlgorithm):
    """Tree-adaptive integration over the (cubic) limits of the zone: HCubatureJL on HyperCube(lims.a, lims.b); zones whose
    limits are not cubic are integrated over the full BZ without symmetries.  ref: src/brillouin.jl:446-463."""

    def __init__(self, norm=None, initdiv=1):
        self.norm, self.initdiv = norm, int(initdiv)


class MonkhorstPack(IntegralAlgorithm):
    """ref: src/algorithms.jl:342-347."""

    def __init__(self, npt=50, syms=None, nthreads=1):
        self.npt, self.syms, self.nthreads = int(npt), syms, nthreads


class AutoSymPTRJL(IntegralAlgorithm):
    """ref: src/algorithms.jl:393-406."""

    def __init__(self, norm=None, a=1.0, nmin=50, nmax=1000, n0=6.0, dn=math.log(10), keepmost=2, syms=None, nthreads=1):
        self.inner = AutoPTR(norm, a, nmin, nmax, n0, dn, keepmost, nthreads)
        self.syms = syms


class NestedQuad(IntegralAlgorithm):
    """ref: src/algorithms.jl:450-455."""

    def __init__(self, *algs):
        self.algs = algs or (AuxQuadGKJL(),)


class AbsoluteEstimate(IntegralAlgorithm):
    """AbsoluteEstimate(est_alg, abs_alg; norm, kws...): a rough solve with `est_alg` (given `kws`) sizes the integral, then
    `abs_alg` runs with abstol = max(abstol, reltol * norm(I_est)) and reltol = 0.  ref: src/algorithms.jl:614-653."""

    def __init__(self, est_alg, abs_alg, norm=None, **kws):
        checkkwargs(kws)
        self.est_alg, self.abs_alg, self.norm, self.kws = est_alg, abs_alg, norm, kws


def PTR_IAI(ptr=None, iai=None, **kws):
    """IAI with abstol = reltol * |PTR estimate|.  ref: src/brillouin.jl:464-474."""
    return AbsoluteEstimate(ptr or PTR(), iai or IAI(), **kws)


def AutoPTR_IAI(reltol=1.0, ptr=None, iai=None, **kws):
    """IAI with abstol = rtol * |AutoPTR estimate computed to `reltol`|.  ref: src/brillouin.jl:477-488."""
    return AbsoluteEstimate(ptr or AutoPTR(), iai or IAI(), reltol=reltol, **kws)


class EvalCounter(IntegralAlgorithm):
    """Counts integrand evaluations into sol.numevals.  ref: src/algorithms.jl:656-691, src/fourier.jl:512-530."""

    def __init__(self, alg):
        self.alg = alg


# ---------------------------------------------------------------------------- problem / solution
@dataclass
class IntegralProblem:
    """ref: src/interfaces.jl:34-47."""
    f: Any
    dom: Any
    p: Any = field(default_factory=NullParameters)


@dataclass
class IntegralSolution:
    """ref: src/interfaces.jl:120-126 (numevals < 0 means unknown)."""
    u: Any
    resid: Any
    retcode: bool
    numevals: int


def checkkwargs(kwargs):
    """ref: src/interfaces.jl:64-69."""
    for k in kwargs:
        if k not in ("abstol", "reltol", "maxiters"):
            raise ValueError(f"keyword {k} unrecognized")


@dataclass
class IntegralCache:
    f: Any
    dom: Any
    p: Any
    alg: Any
    cacheval: Any
    kwargs: dict


def init(prob: IntegralProblem, alg: IntegralAlgorithm, **kwargs):
    """ref: src/interfaces.jl:78-82."""
    checkkwargs(kwargs)
    return IntegralCache(prob.f, prob.dom, prob.p, alg, init_cacheval(prob.f, prob.dom, prob.p, alg), kwargs)


def solve(prob: IntegralProblem, alg: IntegralAlgorithm, **kwargs):
    """ref: src/interfaces.jl:105-108."""
    return solve_(init(prob, alg, **kwargs))


def solve_(c: IntegralCache):
    """solve!(cache).  ref: src/interfaces.jl:116-118."""
    return do_solve(c.f, c.dom, c.p, c.alg, c.cacheval, **c.kwargs)


def init_cacheval(f, dom, p, alg):
    """The device-resident series is the cache; rules are cached inside it by (npt, syms, want)
    so that, unlike the reference (src/interfaces.jl:174-179), nothing is rebuilt per call."""
    if isinstance(f, FourierIntegrand):
        return f.w.device()
    return None


# ---------------------------------------------------------------------------- the dispatch
def _as_params(f: FourierIntegrand, p):
    return f.f.p.merge(p)


def _norm(v):
    return float(np.linalg.norm(np.atleast_1d(np.asarray(v)).reshape(-1)))


def _shape_value(f: FourierIntegrand, vals):
    """complex [ncomp] from the device -> the value type the reference integrand returns."""
    fi = f.f.f
    s = f.w
    if isinstance(fi, GlocIntegrand):
        g = vals.reshape(s.n, s.n).T  # column-major block
        return g[0, 0] if s.scalar else g
    if isinstance(fi, LinearXIntegrand):
        return vals.copy()
    v = vals[0]
    if isinstance(fi, (DOSIntegrand, UnitIntegrand)):
        return float(v.real)
    return complex(v)


def _ncomp(fi, n, d):
    """Components of a device integrand's value (integrand_ncomp of the library)."""
    return {L.F_GLOC: n * n, L.F_LINEAR_X: d}.get(fi.fid, 1)


def _want_for(fi):
    return L.WANT_EIG if isinstance(fi, DOSIntegrand) and fi.form == "eig" else L.WANT_H


def _ptr_rule_values(f: FourierIntegrand, dev, npt, syms, plist):
    """rule(f, B) for every parameter set in plist -> (list of values, numevals per solve).
    Device integrands that share their fixed parameters are reduced in ONE fused pass."""
    fi = f.f.f
    if isinstance(fi, DeviceIntegrand):
        bound = [fi.bind(p) for p in plist]
        out = [None] * len(plist)
        groups = {}
        for i, (params, sw) in enumerate(bound):
            groups.setdefault(tuple(params), []).append(i)
        # A full grid whose values would not be worth (or not fit) keeping: sum it on the fly (abz_ptr_sum)
        want = _want_for(fi)
        n, d = f.w.n, f.w.d
        nk = int(npt) ** d
        rule_bytes = nk * (16 * n * n if want & L.WANT_H else 8 * n)
        # (more than 4 bands: the store-free kernels beat building + scanning a rule at any size -- except the matrix-valued G
        # of a Hermitian series up to 16 bands, whose cached rule is scanned by the 16-lane row kernel at half the cost of the
        # store-free inverse: kept as a rule unless it is too large)
        prefer_stream = n > 4 and not (fi.fid == L.F_GLOC and n <= 16 and dev.hermitian())
        stream = (syms is None and (rule_bytes > dev.stream_above_bytes or prefer_stream) and
                  not dev.has_rule(npt, syms, want) and dev.ptr_sum_supported(npt, fi.fid))
        rule = None
        for params, idxs in groups.items():
            sweeps = [bound[i][1] for i in idxs] if fi.swept else None
            vals = None
            if stream:
                try:
                    vals = dev.ptr_sum(npt, fi.fid, params, sweeps)
                except L.AbzError:
                    stream = False  # the library declined (e.g. its own Hermiticity test): use a rule
            if vals is None:
                rule = rule or dev.rule(npt, syms, want)
                vals = rule.reduce(fi.fid, params, sweeps)
            for j, i in enumerate(idxs):
                out[i] = _shape_value(f, vals[j if fi.swept else 0])
        return out, (rule.nk if rule is not None else nk)
    # host path: H(k) batch back to the host, user closure per node (ref: quadsum)
    rule = dev.rule(npt, syms, L.WANT_H)
    data = rule.export(x=True, w=True, H=True)
    d = f.w.d
    scale = 1.0 / (npt**d * (1 if syms is None else len(syms)))
    out = []
    for p in plist:
        acc = None
        for k in range(rule.nk_local):
            v = data["w"][k] * np.asarray(f.f.f(FourierValue(data["x"][k], data["H"][k]), *p.args, **p.kwargs))
            acc = v if acc is None else acc + v
        if rule.shard and rule.shard[1] > 1:  # k-sharded: sum the ranks' partial sums
            if acc is None:
                raise ValueError("k-sharded rule with an empty share on this rank (more ranks than nodes)")
            part = np.asarray(acc)
            tot = rule._sum_over_ranks(np.ascontiguousarray(part, dtype=np.complex128).reshape(-1).view(np.float64))
            tot = tot.view(np.complex128).reshape(part.shape)
            tot = tot if np.iscomplexobj(part) else tot.real
            acc = tot if part.shape else tot[()]
        out.append(acc * scale)
    return out, rule.nk


class AbstractSymRep:
    """ref: src/brillouin.jl:48-58.  A representation maps the integral over the irreducible domain to
    the integral over the full BZ: `symmetrize_(bz, x)`."""


class TrivialRep(AbstractSymRep):
    """ref: src/brillouin.jl:67-71,106."""

    def symmetrize_(self, bz, x):
        return nsyms(bz) * x


class UnknownRep(AbstractSymRep):
    """ref: src/brillouin.jl:60-65,107 -- the solve is repeated on the full BZ."""

    def symmetrize_(self, bz, x):
        return x


class MatrixRep(AbstractSymRep):
    """A user-defined representation (the reference's extension point `SymRep(f)` + `symmetrize_`,
    src/brillouin.jl:73-108) for matrix-valued integrals whose integrand transforms as
    f(S k) = D_S f(k) D_S^dagger: x_FBZ = sum_S D_S x_IBZ D_S^dagger, `reps[i]` belonging to `bz.syms[i]`."""

    def __init__(self, reps):
        self.reps = [np.asarray(D) for D in reps]

    def symmetrize_(self, bz, x):
        if len(self.reps) != nsyms(bz):
            raise ValueError("MatrixRep needs one matrix per symmetry of the BZ")
        x = np.asarray(x)
        return sum(D @ x @ D.conj().T for D in self.reps)


def SymRep(f):
    """Representation of the integral of f: its `symrep` attribute, else UnknownRep.  ref: src/brillouin.jl:85."""
    return getattr(f, "symrep", None) or UnknownRep()


def _is_trivial(u):
    return np.ndim(u) == 0


def symmetrize(f, bz, x):
    """ref: src/brillouin.jl:96-114: numbers are TrivialRep, a full BZ maps x to itself."""
    if bz.syms is None:
        return x
    if isinstance(x, AuxValue):  # ref: src/brillouin.jl:113
        return AuxValue(symmetrize(f, bz, x.val), symmetrize(f, bz, x.aux))
    if _is_trivial(x):
        return nsyms(bz) * x
    return (f if isinstance(f, AbstractSymRep) else SymRep(f)).symmetrize_(bz, x)


_symmetrize_value = symmetrize  # (keyword arguments named `symmetrize` shadow the function below)


def _needs_fbz(f, bz, u):
    """An array-valued integral on a symmetric BZ without a known representation."""
    return bz.syms is not None and not _is_trivial(u) and isinstance(SymRep(f.f.f), UnknownRep)


def _redo_on_fbz(f, bz, p, alg, kws):
    warnings.warn("A symmetric BZ was used with an integrand whose symmetry representation is unknown. "
                  "For correctness, the calculation will be repeated on the full BZ.")  # ref: src/brillouin.jl:332-351
    fbz = SymmetricBZ(bz.A, bz.B, CubicLimits(np.zeros(bz.ndim), np.ones(bz.ndim)), None)
    return do_solve(f, fbz, p, alg, init_cacheval(f, fbz, p, alg), **kws)


def _iai_device_many(f: FourierIntegrand, dev, lims, plist, abstol, reltol, maxiters, want_panels=False):
    """IAI for several parameter sets at once (abz_iai_solve_many): solves that differ only in the
    swept parameter share every launch, and sweeps of >= 32 of them are dealt to lanes inside the library (host thread +
    stream each: ABZ_IAI_LANES, ABZ_IAI_LANE_MIN).  Returns [(value, err, numevals, extra)] in plist order."""
    fi = f.f.f
    max_batch = 0 if f.nest is None else min(f.nest.max_batch, 2**62)
    d, n = f.w.d, f.w.n
    ncomp = {L.F_GLOC: n * n, L.F_LINEAR_X: d}.get(fi.fid, 1)
    if isinstance(lims, CubicLimits):
        kind, a, b = L.LIMS_CUBIC, lims.a, lims.b
    elif isinstance(lims, TetrahedralLimits):
        kind, a, b = L.LIMS_TETRAHEDRAL, lims.a, None
    elif isinstance(lims, (PolyhedralLimits, PolygonLimits)):
        kind = L.LIMS_POLYHEDRAL if isinstance(lims, PolyhedralLimits) else L.LIMS_POLYGON
        a = lims.packed()
        b = np.array([float(len(a))])
    else:
        raise ValueError("IAI needs CubicLimits, TetrahedralLimits, PolyhedralLimits or PolygonLimits")
    if lims.ndim != d:
        raise ValueError("variables in Fourier series don't match domain")  # ref: src/fourier.jl:506
    _, pa = L.f64(a)
    pb = L.f64(b)[1] if b is not None else None
    groups = {}  # fixed parameters -> [(position, sweep value)]
    for i, p in enumerate(plist):
        params, sw = fi.bind(p)
        groups.setdefault(tuple(float(v) for v in params), []).append((i, 0.0 if sw is None else float(sw)))
    res = [None] * len(plist)
    maxp = 1 << 16
    for params, members in groups.items():
        par = np.ascontiguousarray(params, dtype=np.float64)
        sweeps = np.ascontiguousarray([sw for _, sw in members], dtype=np.float64)
        m = len(members)
        out = np.empty((m, ncomp, 2))
        err = np.zeros(m)
        nev = np.zeros(m, dtype=np.int64)
        npan = C.c_int64(0)
        panels = np.empty((maxp, 2)) if want_panels else None
        L.check(L.lib().abz_iai_solve_many(
            dev.h, kind, pa, pb, fi.fid, par.ctypes.data_as(L.c_f64p) if len(par) else None, len(par),
            sweeps.ctypes.data_as(L.c_f64p), m, -1.0 if abstol is None else float(abstol),
            -1.0 if reltol is None else float(reltol), int(min(maxiters, 2**62)), int(max_batch),
            out.ctypes.data_as(L.c_f64p), err.ctypes.data_as(L.c_f64p), nev.ctypes.data_as(L.c_i64p),
            panels.ctypes.data_as(L.c_f64p) if want_panels else None, maxp, C.byref(npan)))
        vals = out.view(np.complex128).reshape(m, ncomp)
        for r, (i, _) in enumerate(members):
            extra = {"panels": panels[: npan.value].copy()} if want_panels and r == 0 else {}
            res[i] = (_shape_value(f, vals[r].copy()), float(err[r]), int(nev[r]), extra)
    return res


def _iai_device(f: FourierIntegrand, dev, lims, p: MixedParameters, abstol, reltol, maxiters, want_panels=False):
    return _iai_device_many(f, dev, lims, [p], abstol, reltol, maxiters, want_panels)[0]


def _iai_host(f: FourierIntegrand, dev, lims, p: MixedParameters, abstol, reltol, maxiters):
    """User closure on the host: depth-first nested GK(7,15) whose innermost batches are evaluated
    by abz_eval_nodes.  ref: src/fourier.jl:432-510."""
    from .hostquad import nested_quad_host
    return nested_quad_host(f, dev, lims, p, abstol, reltol, maxiters)


def _hcubature_fourier(f: FourierIntegrand, dev, cube, pm: MixedParameters, alg, abstol, reltol, maxiters):
    """HCubatureJL for a FourierIntegrand: the tree on the host, H(k) at each box's Genz-Malik points in one device batch,
    the integrand (a DeviceIntegrand's host form or the user's closure) per point.  ref: src/fourier.jl:120-122."""
    from . import generic as G
    if f.w.d != len(np.atleast_1d(cube.a)):
        raise ValueError("variables in Fourier series don't match domain")
    fi = f.f.f
    nev = [0]
    if isinstance(fi, DeviceIntegrand):
        fixed, sw = fi.bind(pm)
        hargs = tuple(fixed) + ((sw,) if fi.swept else ())

    def batch(pts):
        pts = np.asarray(pts, dtype=np.float64).reshape(len(pts), -1)
        nev[0] += len(pts)
        H = dev.eval_nodes(pts, L.WANT_H)  # [m, n, n], or [m] for a scalar series
        out = []
        for x, h in zip(pts, H):
            v = FourierValue(x if f.w.d > 1 else float(x[0]), h)
            out.append(fi.host(v, *hargs) if isinstance(fi, DeviceIntegrand) else _call_user(fi, v, pm))
        return out
    I, E, _ = G.hcubature(batch, cube.a, cube.b, 0.0 if abstol is None else abstol, 0.0 if reltol is None else reltol, maxiters, alg.initdiv)
    return I, E, nev[0]


def do_solve(f, dom, p, alg, cacheval=None, abstol=None, reltol=None, maxiters=2**62, _panels=False):
    """do_solve(f, dom, p, alg, cacheval; abstol, reltol, maxiters) -> IntegralSolution.
    ref: src/interfaces.jl:116-125 and the methods at src/fourier.jl:381-389,493-510,
    src/brillouin.jl:328-355,429-444."""
    kws = dict(abstol=abstol, reltol=reltol, maxiters=maxiters)
    counter = isinstance(alg, EvalCounter)
    if counter:
        alg = alg.alg
    if isinstance(alg, AbsoluteEstimate):  # ref: src/algorithms.jl:644-653
        wrap = (lambda a_: EvalCounter(a_)) if counter else (lambda a_: a_)
        est = do_solve(f, dom, p, wrap(alg.est_alg), cacheval, **alg.kws)
        val = (alg.norm or _norm)(est.u)
        rtol = math.sqrt(np.finfo(float).eps) if reltol is None else reltol
        atol = max(0.0 if abstol is None else abstol, rtol * val)
        sol = do_solve(f, dom, p, wrap(alg.abs_alg), cacheval, abstol=atol, reltol=0.0, maxiters=maxiters)
        if counter:  # the reference's counter wraps the integrand, so it sees the evaluations of both stages
            sol.numevals += est.numevals
        return sol
    if not isinstance(f, FourierIntegrand):
        return _do_solve_generic(f, dom, p, alg, counter, abstol, reltol, maxiters)
    dev = cacheval if cacheval is not None else f.w.device()
    pm = _as_params(f, p)
    fi = f.f.f
    isdev = isinstance(fi, DeviceIntegrand)

    # ---- BZ algorithms: rescale like do_solve_autobz
    if isinstance(dom, SymmetricBZ):
        bz = dom
        if bz.ndim != f.w.d:
            raise ValueError("variables in Fourier series don't match domain")
        j = abs(np.linalg.det(bz.B))
        ns = nsyms(bz)
        if isinstance(alg, PTR):
            vals, nev = _ptr_rule_values(f, dev, alg.npt, bz.syms, [pm])
            u = vals[0]
            if _needs_fbz(f, bz, u):
                return _redo_on_fbz(f, bz, p, EvalCounter(alg) if counter else alg, kws)
            return IntegralSolution(j * symmetrize(fi, bz, u), None, True, nev if counter else -1)
        if isinstance(alg, AutoPTR):
            sols = _autoptr_many(f, dev, bz, [pm], alg, abstol, reltol, maxiters)
            s = sols[0]
            if s is None:
                return _redo_on_fbz(f, bz, p, EvalCounter(alg) if counter else alg, kws)
            if not counter:
                s.numevals = -1
            return s
        if isinstance(alg, IAI):
            at = None if abstol is None else abstol / (j * ns)  # ref: src/brillouin.jl:340-342
            if isdev:
                u, err, nev, extra = _iai_device(f, dev, bz.lims, pm, at, reltol, maxiters, _panels)
            else:
                u, err, nev = _iai_host(f, dev, bz.lims, pm, at, reltol, maxiters)
                extra = {}
            if _needs_fbz(f, bz, u):
                return _redo_on_fbz(f, bz, p, EvalCounter(alg) if counter else alg, kws)
            sol = IntegralSolution(j * symmetrize(fi, bz, u), j * ns * err, True, nev if counter else -1)
            sol.extra = extra
            return sol
        if isinstance(alg, TAI):  # ref: src/brillouin.jl:458-462 + do_solve_autobz :337-355
            bz_ = bz if isinstance(bz.lims, CubicLimits) else SymmetricBZ(bz.A, bz.B, CubicLimits(np.zeros(bz.ndim), np.ones(bz.ndim)), None)
            ns_ = nsyms(bz_)
            at = None if abstol is None else abstol / (j * ns_)
            u, err, nev = _hcubature_fourier(f, dev, HyperCube(np.asarray(bz_.lims.a, dtype=np.float64), np.asarray(bz_.lims.b, dtype=np.float64)),
                                             pm, HCubatureJL(alg.norm, alg.initdiv), at, reltol, maxiters)
            if _needs_fbz(f, bz_, u):
                return _redo_on_fbz(f, bz_, p, EvalCounter(alg) if counter else alg, kws)
            return IntegralSolution(j * symmetrize(fi, bz_, u), j * ns_ * err, True, nev if counter else -1)
        raise ValueError(f"unsupported BZ algorithm {type(alg).__name__}")

    # ---- generic algorithms on unit domains (ref: test/fourier.jl:25-37)
    if isinstance(alg, MonkhorstPack):
        if not isinstance(dom, Basis):
            raise ValueError("MonkhorstPack needs a Basis domain")
        vals, nev = _ptr_rule_values(f, dev, alg.npt, alg.syms, [pm])
        vol = abs(np.linalg.det(dom.B))
        return IntegralSolution(vals[0] * vol, None, True, nev if counter else -1)
    if isinstance(alg, AutoSymPTRJL):
        if not isinstance(dom, Basis):
            raise ValueError("AutoSymPTRJL needs a Basis domain")
        fake = SymmetricBZ(np.eye(dom.ndim), dom.B, None, alg.syms)
        sols = _autoptr_many(f, dev, fake, [pm], alg.inner, abstol, reltol, maxiters, symmetrize=False, jac=abs(np.linalg.det(dom.B)), scale_tol=False)
        s = sols[0]
        if not counter:
            s.numevals = -1
        return s
    if isinstance(alg, HCubatureJL):
        if not isinstance(dom, HyperCube):
            raise ValueError("HCubatureJL needs a HyperCube domain")
        u, err, nev = _hcubature_fourier(f, dev, dom, pm, alg, abstol, reltol, maxiters)
        return IntegralSolution(u, err, True, nev if counter else -1)
    if isinstance(alg, NestedQuad):
        if not isinstance(dom, (CubicLimits, TetrahedralLimits, PolyhedralLimits, PolygonLimits)):
            raise ValueError("NestedQuad needs iterated limits")
        if isdev:
            u, err, nev, _ = _iai_device(f, dev, dom, pm, abstol, reltol, maxiters)
        else:
            u, err, nev = _iai_host(f, dev, dom, pm, abstol, reltol, maxiters)
        return IntegralSolution(u, err, True, nev if counter else -1)
    raise ValueError(f"unsupported algorithm {type(alg).__name__}")


def _do_solve_generic(f, dom, p, alg, counter, abstol, reltol, maxiters):
    """Plain callables f(x, p), BatchIntegrand and NestedBatchIntegrand under the generic algorithms (generic.py).
    ref: src/algorithms.jl:215-239,360-432,450-612; known answers test/interface_tests.jl:90-130."""
    from . import generic as G
    if not (callable(f) or isinstance(f, (BatchIntegrand, NestedBatchIntegrand))):
        raise ValueError(f"unsupported integrand {type(f).__name__}")
    if isinstance(alg, AuxQuadGKJL):
        u, err, nev = G.solve_auxquadgk(f, dom, p, abstol, reltol, maxiters, order=alg.order)
    elif isinstance(alg, QuadratureFunction):
        u, err, nev = G.solve_quadrature_function(f, dom, p, alg)
    elif isinstance(alg, ContQuadGKJL):
        u, err, nev = G.solve_contquadgk(f, dom, p, alg, abstol, reltol, maxiters)
    elif isinstance(alg, MeroQuadGKJL):
        u, err, nev = G.solve_meroquadgk(f, dom, p, alg, abstol, reltol, maxiters)
    elif isinstance(alg, MonkhorstPack):
        if not isinstance(dom, Basis):
            raise ValueError("MonkhorstPack needs a Basis domain")
        (u, nev), err = G.ptr_rule_value(f, dom.B, p, alg.npt, alg.syms), None
    elif isinstance(alg, AutoSymPTRJL):
        if not isinstance(dom, Basis):
            raise ValueError("AutoSymPTRJL needs a Basis domain")
        u, err, nev = G.solve_autosymptr(f, dom.B, p, alg, abstol, reltol, maxiters)
    elif isinstance(alg, HCubatureJL):
        u, err, nev = G.solve_hcubature(f, dom, p, alg, abstol, reltol, maxiters)
    elif isinstance(alg, NestedQuad):
        if not (hasattr(dom, "segs") and hasattr(dom, "fix")):
            raise ValueError("NestedQuad needs iterated limits")
        for a_ in alg.algs:
            if not isinstance(a_, (AuxQuadGKJL, QuadratureFunction)):
                raise ValueError("NestedQuad: AuxQuadGKJL / QuadGKJL / QuadratureFunction levels are supported")
        u, err, nev = G.nested_quad(f, dom, p, abstol, reltol, maxiters, algs=alg.algs)
    else:
        raise ValueError(f"{type(alg).__name__} needs a FourierIntegrand on a SymmetricBZ (generic integrands: "
                         "AuxQuadGKJL, MonkhorstPack, AutoSymPTRJL, NestedQuad)")
    return IntegralSolution(u, err, True, nev if counter else -1)


def _autoptr_library(f, dev, bz, plist, alg, abstol, reltol, maxiters, symmetrize, j, scale_tol):
    """The whole AutoPTR loop inside the library (abz_autoptr_solve_many: grid sequence, rules kept by the series,
    store-free sums for grids used once, error test, numevals -- one C call per batch of parameter sets, one stream
    synchronisation per converged two-grid solve).  Applies to the library's own integrands with the default norm on an
    unsharded series whose symmetrisation is a number (TrivialRep: the value of every rule times nsyms,
    src/brillouin.jl:127-130); returns None otherwise and the host loop below drives the same rules one call at a time."""
    fi = f.f.f
    if not isinstance(fi, DeviceIntegrand) or alg.norm is not None or dev.kshard is not None:
        return None
    if symmetrize and bz.syms is not None and fi.matrix:
        return None  # matrix-valued integrals on a symmetric BZ: UnknownRep (redo on the FBZ) or a user representation
    bound = [fi.bind(p) for p in plist]
    if any(tuple(b[0]) != tuple(bound[0][0]) for b in bound):
        return None
    params = np.ascontiguousarray(bound[0][0], dtype=np.float64)
    if fi.swept:
        sweeps = np.ascontiguousarray([b[1] for b in bound], dtype=np.float64)
        nsolve = len(plist)
    else:
        sweeps = np.zeros(1)
        nsolve = 1  # no swept parameter: every entry of plist is the same integral
    n0, dn = alg.npt_sequence()
    atol = -1.0 if abstol is None else (abstol / j if scale_tol else abstol)
    rtol = -1.0 if reltol is None else float(reltol)
    syms = None if bz.syms is None else np.ascontiguousarray(np.asarray(bz.syms), dtype=np.int32)
    nsy = 0 if syms is None else len(bz.syms)
    factor = float(nsyms(bz)) if (symmetrize and bz.syms is not None) else 1.0
    ncomp = _ncomp(fi, f.w.n, f.w.d)
    out = np.zeros((nsolve, ncomp), dtype=np.complex128)
    err = np.zeros(nsolve)
    nev = np.zeros(nsolve, dtype=np.int64)
    npt = np.zeros(nsolve, dtype=np.int32)
    L.check(L.lib().abz_autoptr_solve_many(
        dev.h, None if syms is None else syms.ctypes.data_as(L.c_i32p), nsy, fi.fid, params.ctypes.data_as(L.c_f64p), len(params),
        sweeps.ctypes.data_as(L.c_f64p), nsolve, n0, dn, atol, rtol, int(min(maxiters, 2**62)), int(alg.keepmost), factor,
        out.view(np.float64).ctypes.data_as(L.c_f64p), err.ctypes.data_as(L.c_f64p), nev.ctypes.data_as(L.c_i64p),
        npt.ctypes.data_as(L.c_i32p)))
    sols = []
    for i in range(len(plist)):
        k = i if fi.swept else 0
        sol = IntegralSolution(_shape_value(f, out[k]) * j, float(err[k]) * j, True, int(nev[k]))
        sol.extra = {"npt": int(npt[k])}
        sols.append(sol)
    return sols


def _autoptr_many(f, dev, bz, plist, alg: AutoPTR, abstol, reltol, maxiters, symmetrize=True, jac=None,
                  scale_tol=True):
    """autosymptr for a batch of parameter sets in lockstep: every grid is built (or found in the
    device cache) once and reduced for all not-yet-converged parameter sets in one fused pass.
    ref: src/algorithms.jl:418-432 + autosymptr: I1 = rule(n0), I2 = rule(n0 + dn),
    err = norm(I2 - I1); iterate until err <= max(abstol, reltol norm(I2)) or numevals >= maxevals.
    Tolerance / Jacobian as in do_solve_autobz for AutoPTR: abstol /= |det B|, symmetrisation inside
    every rule evaluation (src/brillouin.jl:127-130,429-444).  Entries are None when the value has
    an unknown symmetry representation on a symmetric BZ."""
    norm = alg.norm or _norm
    j = abs(np.linalg.det(bz.B)) if jac is None else jac
    ns = nsyms(bz) if symmetrize else 1
    fast = _autoptr_library(f, dev, bz, plist, alg, abstol, reltol, maxiters, symmetrize, j, scale_tol)
    if fast is not None:
        return fast
    if abstol is None and reltol is None:
        rtol, atol = math.sqrt(np.finfo(float).eps), 0.0
    else:
        rtol = 0.0 if reltol is None else reltol
        atol = 0.0 if abstol is None else (abstol / j if scale_tol else abstol)
    n0, dn = alg.npt_sequence()
    npt = n0
    active = list(range(len(plist)))
    sym = (lambda v: _symmetrize_value(f.f.f, bz, v)) if symmetrize else (lambda v: v)
    I1, nev1 = _ptr_rule_values(f, dev, npt, bz.syms, plist)
    if symmetrize and _needs_fbz(f, bz, I1[0]):
        return [None] * len(plist)
    I1 = [sym(v) for v in I1]
    numevals = [nev1] * len(plist)
    out = [None] * len(plist)
    I2 = list(I1)
    err = [math.inf] * len(plist)
    first = True
    while active:
        npt += dn
        vals, nev = _ptr_rule_values(f, dev, npt, bz.syms, [plist[i] for i in active])
        nxt = []
        for v, i in zip(vals, active):
            if not first:
                I1[i] = I2[i]
            I2[i] = sym(v)
            numevals[i] += nev
            err[i] = norm(np.asarray(I2[i]) - np.asarray(I1[i]))
            if err[i] <= max(atol, rtol * norm(I2[i])) or numevals[i] >= maxiters or not np.isfinite(err[i]):
                out[i] = IntegralSolution(I2[i] * j, err[i] * j, True, numevals[i])
                out[i].extra = {"npt": npt}
            else:
                nxt.append(i)
        active = nxt
        first = False
    return out


# ---------------------------------------------------------------------------- solver functor + sweeps
class IntegralSolver:
    """solver(args...; kwargs...) -> solve(IntegralProblem(f, dom, merge(f.p, params)), alg).u
    ref: src/interfaces.jl:142-187, src/fourier.jl:89-93."""

    def __init__(self, f, dom=None, alg=None, *, abstol=None, reltol=None, maxiters=None, **bad):
        if isinstance(f, IntegralProblem):  # IntegralSolver(prob, alg; kws...)
            prob, alg = f, dom
            f, dom, self.p0 = prob.f, prob.dom, prob.p
        else:
            self.p0 = NullParameters()
        checkkwargs(bad)
        self.f, self.dom, self.alg = f, dom, alg
        self.kwargs = {k: v for k, v in dict(abstol=abstol, reltol=reltol, maxiters=maxiters).items() if v is not None}
        self.cacheval = None

    def solve_p(self, p):
        """ref: src/interfaces.jl:174-182."""
        if self.cacheval is None:
            self.cacheval = init_cacheval(self.f, self.dom, p, self.alg)
        return do_solve(self.f, self.dom, p, self.alg, self.cacheval, **self.kwargs)

    def __call__(self, *args, **kwargs):
        if isinstance(self.f, (FourierIntegrand, ParameterIntegrand)):
            base = self.p0 if isinstance(self.p0, MixedParameters) else MixedParameters()
            p = base.merge(MixedParameters(*args, **kwargs))
        else:
            p = args[0] if args else self.p0
        return self.solve_p(p).u


def batchparam(ps, nthreads):
    """Round-robin groups along the longest axis: group j gets ps[j], ps[j+nthreads], ...
    ref: src/interfaces.jl:199-208.  Returns lists of (index, parameter)."""
    arr = np.asarray(ps, dtype=object) if not isinstance(ps, np.ndarray) else ps
    if arr.shape == ():
        return [[((), arr.item())]]
    assert nthreads >= 1
    dim = int(np.argmax(arr.shape))
    ln = arr.shape[dim]
    batches = [[] for _ in range(min(nthreads, ln))]
    for idx in np.ndindex(*arr.shape[::-1]):
        i = idx[::-1]  # column-major traversal like CartesianIndices
        batches[i[dim] % nthreads].append((i, arr[i]))
    return batches


def _to_params(p):
    return p if isinstance(p, MixedParameters) else MixedParameters(p)


def batchsolve(solver: IntegralSolver, ps, nthreads=1, callback=None):
    """Evaluate the solver at every parameter in ps -> array like ps.  ref: src/interfaces.jl:234-243.
    For device integrands under PTR / AutoPTR the whole sweep is fused on the GPU (one pass over the
    cached rule for all parameters), and under IAI / NestedQuad the adaptive solves of all parameters
    advance in lock-step sharing their launches (abz_iai_solve_many; each solve makes the decisions it
    would make alone); otherwise parameters are solved one by one in batchparam order."""
    arr = ps if isinstance(ps, np.ndarray) and ps.dtype == object else None
    if arr is None:
        lst = list(ps)
        arr = np.empty(len(lst), dtype=object)
        arr[:] = lst
    out = np.empty(arr.shape, dtype=object)
    f, alg = solver.f, solver.alg
    inner = alg.alg if isinstance(alg, EvalCounter) else alg
    isdev = isinstance(f, FourierIntegrand) and isinstance(f.f.f, DeviceIntegrand)
    fused = isdev and isinstance(solver.dom, SymmetricBZ) and isinstance(inner, (PTR, AutoPTR, IAI))
    fused = fused or (isdev and isinstance(inner, NestedQuad) and
                      isinstance(solver.dom, (CubicLimits, TetrahedralLimits, PolyhedralLimits, PolygonLimits)))
    flat_idx = [i[::-1] for i in np.ndindex(*arr.shape[::-1])]
    t0 = time.time()
    if fused:
        base = solver.p0 if isinstance(solver.p0, MixedParameters) else MixedParameters()
        plist = [f.f.p.merge(base.merge(_to_params(arr[i]))) for i in flat_idx]
        dev = f.w.device()
        bz = solver.dom
        abstol, reltol = solver.kwargs.get("abstol"), solver.kwargs.get("reltol")
        maxiters = solver.kwargs.get("maxiters", 2**62)
        if isinstance(inner, NestedQuad):
            res = _iai_device_many(f, dev, bz, plist, abstol, reltol, maxiters)
            sols = [IntegralSolution(u, e, True, nev) for u, e, nev, _ in res]
        elif isinstance(inner, IAI):
            j = abs(np.linalg.det(bz.B))
            jn = j * nsyms(bz)
            res = _iai_device_many(f, dev, bz.lims, plist, None if abstol is None else abstol / jn, reltol, maxiters)
            sols = [IntegralSolution(j * symmetrize(f.f.f, bz, u), jn * e, True, nev) for u, e, nev, _ in res]
            if _needs_fbz(f, bz, res[0][0]):
                sols = None
        elif isinstance(inner, PTR):
            j = abs(np.linalg.det(bz.B))
            vals, nev = _ptr_rule_values(f, dev, inner.npt, bz.syms, plist)
            sols = [IntegralSolution(j * symmetrize(f.f.f, bz, v), None, True, nev) for v in vals]
            if _needs_fbz(f, bz, vals[0]):
                sols = None
        else:
            sols = _autoptr_many(f, dev, bz, plist, inner, abstol, reltol, maxiters)
            if sols[0] is None:
                sols = None
        if sols is not None:
            dt = (time.time() - t0) / max(len(sols), 1)
            for n, (i, sol) in enumerate(zip(flat_idx, sols)):
                if callback:
                    callback(solver, i, n + 1, arr[i], sol, dt)
                out[i] = sol.u
            return _densify(out)
    n = 0
    for batch in batchparam(arr, nthreads):
        for i, p in batch:
            t = time.time()
            sol = solver.solve_p(_to_params(p) if isinstance(f, (FourierIntegrand, ParameterIntegrand)) else p)
            n += 1
            if callback:
                callback(solver, i, n, p, sol, time.time() - t)
            out[i] = sol.u
    return _densify(out)


def _densify(out):
    try:
        first = out.flat[0]
        if np.ndim(first) == 0:
            return np.array(out.tolist())
    except Exception:
        pass
    return out
