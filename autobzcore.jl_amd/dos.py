"""Density of states: DOSProblem + GGR.  ref: src/dos_interfaces.jl, src/dos_algorithms.jl, src/dos_ggr.jl."""
from dataclasses import dataclass
from typing import Any

import numpy as np

from . import _lib as L
from .bz import SymmetricBZ
from .series import FourierSeries
from .solver import NullParameters, checkkwargs


class DOSAlgorithm:
    pass


class GGR(DOSAlgorithm):
    """Generalized Gilat-Raubenheimer method.  ref: src/dos_algorithms.jl:7-26."""

    def __init__(self, npt=50):
        self.npt = int(npt)


@dataclass
class DOSProblem:
    """ref: src/dos_interfaces.jl:33-38."""
    H: Any
    domain: Any
    p: Any = None


@dataclass
class DOSSolution:
    u: Any
    err: Any
    retcode: bool
    numevals: int


class DOSCache:
    """Mutable cache; assigning `H` marks it fresh so the eigen-data are rebuilt on the next solve.
    ref: src/dos_interfaces.jl:49-64."""

    def __init__(self, H, domain, p, alg, cacheval, kwargs):
        object.__setattr__(self, "H", H)
        self.domain, self.p, self.alg, self.cacheval, self.kwargs = domain, p, alg, cacheval, kwargs
        self.isfresh = False

    def __setattr__(self, name, value):
        if name == "H":
            object.__setattr__(self, "isfresh", True)
        object.__setattr__(self, name, value)


def _init_cacheval(h, domain, p, alg):
    """get_ggr_data on the GPU: eigenvalues + band velocities at every (irreducible) PTR node stay
    resident in HBM.  ref: src/dos_ggr.jl:1-44."""
    if not isinstance(alg, GGR):
        return None
    if not isinstance(h, FourierSeries):
        raise ValueError("GGR currently supports Fourier series Hamiltonians")
    if not isinstance(p, SymmetricBZ):
        raise ValueError("GGR supports BZ parameters from load_bz")
    if p.ndim != h.d:
        raise ValueError("GGR: BZ and series dimensions differ")
    h.invalidate()  # coefficients may have been mutated in place (test/dos.jl:123): re-upload, rules refill lazily
    return h.device().rule(alg.npt, p.syms, L.WANT_EIG | L.WANT_VEL)


def init(prob: DOSProblem, alg: DOSAlgorithm, **kwargs):
    """ref: src/dos_interfaces.jl:82-86."""
    checkkwargs(kwargs)
    return DOSCache(prob.H, prob.domain, prob.p, alg, _init_cacheval(prob.H, prob.domain, prob.p, alg), kwargs)


def solve_(c: DOSCache):
    """solve!(cache).  ref: src/dos_interfaces.jl:104-112, dos_solve src/dos_ggr.jl:46-56."""
    if c.isfresh:
        c.cacheval = _init_cacheval(c.H, c.domain, c.p, c.alg)
        c.isfresh = False
    if not isinstance(c.alg, GGR):
        raise ValueError("unknown DOS algorithm")
    scalar = np.ndim(c.domain) == 0
    if not scalar and not isinstance(c.domain, (list, tuple, np.ndarray)):
        raise ValueError("GGR supports domains of individual eigenvalues")
    u = c.cacheval.ggr(np.atleast_1d(np.asarray(c.domain, dtype=np.float64)))
    return DOSSolution(float(u[0]) if scalar else u, None, True, -1)


def solve(prob: DOSProblem, alg: DOSAlgorithm, **kwargs):
    return solve_(init(prob, alg, **kwargs))
