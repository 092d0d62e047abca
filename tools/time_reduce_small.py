"""256-omega DOS scan of a cached SVO rule for grid sizes from one rank's slab of a k-sharded solve to the full 150^3."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
L = abz._lib
s = abz.load_w90_series(os.path.join(ROOT, "tests", "golden", "svo_hr.dat.gz"))
dev = s.device()
om = np.linspace(10, 15, 256)
for z in (19, 38, 75, 150):
    dev.kshard = (0, 150 // z) if z < 150 else None
    if z < 150:
        import ctypes as C
        h = C.c_void_p()
        L.check(L.lib().abz_ptr_rule_build_slab(dev.h, 150, 0, z, 3, C.byref(h)))
        class R: pass
        out = np.zeros((256, 1, 2))
        def scan():
            L.check(L.lib().abz_rule_reduce(h, L.F_DOS, np.array([0.1]).ctypes.data_as(L.c_f64p), 1, om.ctypes.data_as(L.c_f64p), 256, 1, out.ctypes.data_as(L.c_f64p)))
    else:
        rule = dev.rule(150, None, 3)
        def scan():
            rule.reduce(L.F_DOS, [0.1], om)
    for _ in range(3): scan()
    dev.ctx.prof_enable(True, kernels=[L.K_REDUCE]); dev.ctx.prof_reset()
    for _ in range(20): scan()
    ms, n = dev.ctx.prof_read(L.K_REDUCE); dev.ctx.prof_enable(False)
    print(f"slab of {z} planes ({z*150*150} nodes): 256-omega scan kernel {ms/n:.4f} ms = {z*150*150*256/(ms/n)*1e-6:.0f} G (k,omega)/s  KT={os.environ.get('ABZ_REDUCE_KT','auto')}", flush=True)
