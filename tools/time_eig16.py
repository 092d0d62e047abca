"""16-band 48^3 rule builds (H compact, H + eig, eig only): ms per build.  ABZ_LIB selects the library build."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
from autobzcore.jl_amd import _lib as L
s = abz.synthetic_wannier()
dev = s.device(); ctx = dev.ctx
npt = int(sys.argv[1]) if len(sys.argv) > 1 else 48
for want, name in ((L.WANT_H | L.WANT_H_COMPACT, "Hc"), (L.WANT_H | L.WANT_EIG | L.WANT_H_COMPACT, "Hc+EIG"), (L.WANT_EIG, "EIG")):
    r = abz.DeviceRule(dev, npt, None, want); ctx.sync()
    for _ in range(5):
        r.rebuild()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        r.rebuild()
    ctx.sync(); dt = (time.perf_counter() - t0) / 20
    print(f"{os.environ.get('ABZ_LIB', 'default'):40s} n=16 npt={npt} rebuild {name:6s}: {1e3*dt:7.3f} ms", flush=True)
    r.close()
