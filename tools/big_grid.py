"""Large-grid sanity (index widths): SVO on 400^3 = 64 M k-points (10.8 GB of rule values)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
from autobzcore.jl_amd import _lib as L
npt = int(sys.argv[1]) if len(sys.argv) > 1 else 400
s = abz.load_w90_series(os.path.join(ROOT, "tests", "golden", "svo_hr.dat.gz"))
dev = s.device()
t0 = time.perf_counter()
rule = dev.rule(npt, None, L.WANT_H | L.WANT_EIG)
dev.ctx.sync()
print(f"npt={npt}: nk={rule.nk} build {time.perf_counter()-t0:.3f} s ({rule.nbytes/1e9:.2f} GB)")
one = rule.reduce(L.F_ONE)[0, 0]
om = np.array([11.0, 12.5, 13.5])
dos = rule.reduce(L.F_DOS, [0.1], om)[:, 0].real
dose = rule.reduce(L.F_DOS_EIG, [0.1], om)[:, 0].real
ref = dev.rule(100, None, L.WANT_H).reduce(L.F_DOS, [0.1], om)[:, 0].real
print("sum(1) =", one, " dos", dos, " eig-form", dose, " npt=100 reference", ref)
assert abs(one - 1) < 1e-12 and np.allclose(dos, dose, rtol=1e-9) and np.allclose(dos, ref, rtol=1e-3)
# spot-check the last nodes against direct evaluation
k = np.array([[(npt - 1) / npt, (npt - 1) / npt, (npt - 1) / npt], [0.5, (npt - 2) / npt, (npt - 1) / npt]])
H = dev.eval_nodes(k)
# a slab holding the last planes must agree with direct evaluation
dev.kshard, dev.allreduce = (7, 8), (lambda a: a)
slab = dev.rule(npt, None, L.WANT_H)
ex = slab.export(x=True, w=False, H=True)
dev.kshard, dev.allreduce = None, None
i = np.where(np.all(np.isclose(ex["x"], k[0]), axis=1))[0]
assert len(i) == 1 and np.abs(ex["H"][i[0]] - H[0]).max() < 1e-12, i
print("last node of the last slab matches direct evaluation; OK")
