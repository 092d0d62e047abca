"""autobzcore.jl_amd -- MI355X-native hot path of AutoBZCore.jl (v0.3.8 API shapes).

Fourier/Wannier interpolation H(k), Hermitian eigensolves and integrand scans run in
libabzhip.so (hand-written HIP for gfx950, C ABI in include/abzhip.h); this package is the
host-side mirror of the reference's integrand/algorithm interface for that path.  There is no CPU
fallback: without the built library and a gfx950 device the compute calls raise.
"""
from . import _lib
from ._lib import AbzError, Context
from .bz import (FBZ, IBZ, Basis, CubicLimits, CubicSymIBZ, HyperCube, InversionSymIBZ, PolygonLimits, PolyhedralLimits, PuncturedInterval,
                 SymmetricBZ, TetrahedralLimits, canonical_reciprocal_basis, load_bz, nsyms)
from .dos import DOSProblem, DOSSolution, GGR
from . import dos
from .interp import ChebInterp, hchebinterp
from .io_w90 import load_w90_series, read_w90_hrdat, read_w90_wout
from . import dist
from .dist import batchsolve_sharded, sharded_map, kshard, iaishard
from .io_sweep import SweepArchive, batchsolve_archive
from .h5lite import read_h5_to_nt, write_nt_to_h5
from .generic import fourier_batch
from .synthetic import synthetic_wannier, tb_integer, splitmix64_uniform
from .series import DeviceRule, DeviceSeries, FourierSeries, symptr_rule
from .solver import (IAI, PTR, TAI, PTR_IAI, AutoPTR_IAI, HCubatureJL, ContQuadGKJL, MeroQuadGKJL, InplaceIntegrand, QuadratureFunction, trapz, AbsoluteEstimate, AutoPTR, AutoSymPTRJL, AuxQuadGKJL, AuxValue, QuadGKJL, BatchIntegrand, DOSIntegrand, DeviceIntegrand,
                     EvalCounter, FourierIntegrand, FourierValue, GlocIntegrand, IntegralProblem, IntegralSolution,
                     IntegralSolver, LinearIntegrand, LinearXIntegrand, MixedParameters, MonkhorstPack, NestedBatchIntegrand, NestedQuad,
                     NullParameters, ParameterIntegrand, TrGlocIntegrand, UnitIntegrand, batchparam, batchsolve,
                     do_solve, init, paramproduct, paramzip, solve, solve_,
                     AbstractSymRep, TrivialRep, UnknownRep, MatrixRep, SymRep, symmetrize)

__all__ = [n for n in dir() if not n.startswith("_")]
