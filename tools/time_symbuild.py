import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
from autobzcore.jl_amd import _lib as L
from autobzcore.jl_amd.series import symptr_rule
s = abz.load_w90_series(os.path.join(ROOT, "tests", "golden", "svo_hr.dat.gz"))
dev = s.device(); ctx = dev.ctx
for kind, bzk in (("InversionSymIBZ", abz.InversionSymIBZ()), ("CubicSymIBZ", abz.CubicSymIBZ())):
    bz = abz.load_bz(bzk, np.eye(3))
    for rep in range(2):
        t0 = time.perf_counter(); idx, w = symptr_rule(150, 3, bz.syms, ctx=ctx); t1 = time.perf_counter()
        h = C.c_void_p()
        L.check(L.lib().abz_ptr_rule_build(dev.h, 150, len(w), idx.ctypes.data_as(L.c_i32p), w.ctypes.data_as(L.c_i64p), L.WANT_H, C.byref(h)))
        t2 = time.perf_counter()
        L.lib().abz_rule_destroy(h)
        print(f"{kind:16s} rep {rep}: symptr_rule (device) {1e3*(t1-t0):7.2f} ms   abz_ptr_rule_build {1e3*(t2-t1):7.2f} ms  nirr={len(w)}")
