"""ctypes binding of libabzhip.so (include/abzhip.h).  There is no fallback: if the HIP library is
missing or no MI355X is visible, every compute entry point raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ABZ_LIB: another build of the same library (kernel A/B experiments, tools/); never a fallback
LIB_PATH = os.environ.get("ABZ_LIB") or os.path.join(_HERE, "libabzhip.so")

c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
c_f64p = C.POINTER(C.c_double)
c_vpp = C.POINTER(C.c_void_p)
c_ip = C.POINTER(C.c_int)

# name -> (restype, argtypes); lists every symbol include/abzhip.h declares (tests check that)
PROTOTYPES = {
    "abz_last_error": (C.c_char_p, []),
    "abz_version": (C.c_int, []),
    "abz_device_count": (C.c_int, [c_ip]),
    "abz_ctx_create": (C.c_int, [C.c_int, c_vpp]),
    "abz_ctx_create_on_stream": (C.c_int, [C.c_int, C.c_void_p, c_vpp]),
    "abz_ctx_destroy": (C.c_int, [C.c_void_p]),
    "abz_ctx_sync": (C.c_int, [C.c_void_p]),
    "abz_prof_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "abz_prof_reset": (C.c_int, [C.c_void_p]),
    "abz_prof_read": (C.c_int, [C.c_void_p, C.c_int, c_f64p, c_i64p]),
    "abz_series_create": (C.c_int, [C.c_void_p, c_f64p, C.c_int, c_i32p, c_i32p, c_f64p, C.c_int, c_vpp]),
    "abz_series_destroy": (C.c_int, [C.c_void_p]),
    "abz_series_update": (C.c_int, [C.c_void_p, c_f64p]),
    "abz_eval_nodes": (C.c_int, [C.c_void_p, c_f64p, C.c_int64, C.c_int, c_f64p, c_f64p]),
    "abz_ptr_rule_build": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, c_i32p, c_i64p, C.c_int, c_vpp]),
    "abz_ptr_rule_build_sym": (C.c_int, [C.c_void_p, C.c_int, c_i32p, C.c_int, C.c_int, c_vpp]),
    "abz_ptr_sum": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, c_f64p, C.c_int, c_f64p, C.c_int, C.c_int, c_f64p]),
    "abz_autoptr_solve": (C.c_int, [C.c_void_p, c_i32p, C.c_int, C.c_int, c_f64p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_double,
                                    C.c_double, C.c_int64, C.c_int, C.c_double, c_f64p, c_f64p, c_i64p, c_i32p]),
    "abz_autoptr_solve_many": (C.c_int, [C.c_void_p, c_i32p, C.c_int, C.c_int, c_f64p, C.c_int, c_f64p, C.c_int, C.c_int, C.c_int,
                                         C.c_double, C.c_double, C.c_int64, C.c_int, C.c_double, c_f64p, c_f64p, c_i64p, c_i32p]),
    "abz_series_drop_rules": (C.c_int, [C.c_void_p]),
    "abz_ptr_rule_build_slab": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, c_vpp]),
    "abz_rule_destroy": (C.c_int, [C.c_void_p]),
    "abz_rule_rebuild": (C.c_int, [C.c_void_p]),
    "abz_rule_info": (C.c_int, [C.c_void_p, c_i64p, c_ip, c_ip, c_ip, c_ip]),
    "abz_rule_export": (C.c_int, [C.c_void_p, c_f64p, c_f64p, c_f64p, c_f64p, c_f64p]),
    "abz_rule_reduce": (C.c_int, [C.c_void_p, C.c_int, c_f64p, C.c_int, c_f64p, C.c_int, C.c_int, c_f64p]),
    "abz_rule_reduce_device": (C.c_int, [C.c_void_p, C.c_int, c_f64p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "abz_rule_values_ptr": (C.c_int, [C.c_void_p, c_vpp, c_i64p]),
    "abz_rule_ggr": (C.c_int, [C.c_void_p, c_f64p, C.c_int, c_f64p]),
    "abz_mem_info": (C.c_int, [C.c_void_p, c_i64p]),
    "abz_symptr_rule": (C.c_int, [C.c_int, C.c_int, c_i32p, C.c_int, c_i64p, c_i32p, c_i64p]),
    "abz_symptr_rule_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_i32p, C.c_int, c_i64p, c_i32p, c_i64p]),
    "abz_contract_nodes": (C.c_int, [C.c_void_p, C.c_int, c_i64p, c_f64p, C.c_int64, c_i64p]),
    "abz_eval_line_nodes": (C.c_int, [C.c_void_p, c_i64p, c_f64p, c_f64p, C.c_int64, C.c_int, c_f64p, C.c_int,
                                      C.c_double, c_f64p]),
    "abz_release_level": (C.c_int, [C.c_void_p, C.c_int]),
    "abz_iai_solve": (C.c_int, [C.c_void_p, C.c_int, c_f64p, c_f64p, C.c_int, c_f64p, C.c_int, C.c_double, C.c_double,
                                C.c_double, C.c_int64, C.c_int64, c_f64p, c_f64p, c_i64p, c_f64p, C.c_int64, c_i64p]),
    "abz_iai_solve_many": (C.c_int, [C.c_void_p, C.c_int, c_f64p, c_f64p, C.c_int, c_f64p, C.c_int, c_f64p, C.c_int,
                                     C.c_double, C.c_double, C.c_int64, C.c_int64, c_f64p, c_f64p, c_i64p, c_f64p,
                                     C.c_int64, c_i64p]),
    "abz_iai_set_exchange": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "abz_gk15_nodes": (C.c_int, [C.c_double, C.c_double, c_f64p]),
    "abz_gk15_batch": (C.c_int, [c_f64p, c_f64p, C.c_int64, C.c_int, c_f64p, c_f64p]),
}

# constants of include/abzhip.h
WANT_H, WANT_EIG, WANT_VEL = 1, 2, 4
WANT_H_COMPACT = 8  # with WANT_H on a Hermitian series of n <= 4 bands: upper-triangle planes only (abzhip.h)
WANT_H_ROW_MAJOR = 16  # abz_eval_nodes: matrices row-major in H_out (numpy's order; abzhip.h)
F_ONE, F_LINEAR, F_LINEAR_X, F_DOS, F_TRGLOC, F_GLOC, F_DOS_EIG = range(7)
LIMS_CUBIC, LIMS_TETRAHEDRAL, LIMS_POLYHEDRAL, LIMS_POLYGON = 0, 1, 2, 3
K_CONTRACT, K_EVAL, K_REDUCE, K_GGR, K_EIG, K_GGRBUILD = range(6)
ERR_ARG, ERR_HIP, ERR_NOGPU, ERR_UNSUPPORTED, ERR_NOMEM, ERR_INTERNAL = -1, -2, -3, -4, -5, -6


# all-gather hook of a sharded IAI solve: int fn(void* user, double* buf, int64_t per_rank)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int64)


class AbzError(RuntimeError):
    """A libabzhip call failed (ABZ_ERR_ARG is raised as ValueError = the reference's ArgumentError)."""


_lib = None


def lib():
    """Load libabzhip.so (built in-tree by `make -C autobzcore.jl_amd/csrc` / __graft_entry__.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AbzError(f"{LIB_PATH} is missing: build it with __graft_entry__.build(); "
                           "the product has no CPU fallback")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc):
    if rc == 0:
        return
    msg = lib().abz_last_error().decode("utf-8", "replace")
    if rc == ERR_ARG:
        raise ValueError(msg)
    raise AbzError(f"libabzhip error {rc}: {msg}")


def f64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(c_f64p)


def i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(c_i32p)


def i64(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(c_i64p)


class Context:
    """One device context (HIP stream) per process/thread.  ref: per-thread workspaces and
    deep-copied solvers, src/fourier.jl:60-86, src/interfaces.jl:213."""

    _default = None

    def __init__(self, device=None, stream=None):
        """`stream`: a raw hipStream_t the caller owns (e.g. `torch.cuda.Stream().cuda_stream`); the library's
        launches are then ordered with the caller's work and with RCCL collectives on that stream."""
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        n = C.c_int(0)
        check(lib().abz_device_count(C.byref(n)))
        self.device = device % max(n.value, 1)
        h = C.c_void_p()
        if stream is None:
            check(lib().abz_ctx_create(self.device, C.byref(h)))
        else:
            check(lib().abz_ctx_create_on_stream(self.device, C.c_void_p(int(stream)), C.byref(h)))
        self.h = h

    @classmethod
    def default(cls):
        if cls._default is None:
            cls._default = cls()
        return cls._default

    def sync(self):
        check(lib().abz_ctx_sync(self.h))

    def prof_enable(self, on=True, kernels=None):
        """Record HIP events around the library's launches; `kernels`: iterable of K_* ids to restrict to."""
        mask = 0 if not on else (1 if kernels is None else sum(1 << (k + 1) for k in kernels))
        check(lib().abz_prof_enable(self.h, mask))

    def mem_info(self):
        """(live device bytes, cached device bytes, this context's scratch bytes, its pinned host bytes, live blocks):
        the allocator's own bookkeeping (abz_mem_info)."""
        info = (C.c_int64 * 5)()
        check(lib().abz_mem_info(self.h, info))
        return tuple(int(v) for v in info)

    def prof_reset(self):
        check(lib().abz_prof_reset(self.h))

    def prof_read(self, kernel_id):
        ms = C.c_double(0)
        n = C.c_int64(0)
        check(lib().abz_prof_read(self.h, kernel_id, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def close(self):
        if self.h:
            lib().abz_ctx_destroy(self.h)
            self.h = None
