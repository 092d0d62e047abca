"""Does the Fourier-eval kernel time depend on where the rule buffer lands?  Build / time / free the
same rule several times in one process, then keep several alive at once."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import autobzcore.jl_amd as abz
from autobzcore.jl_amd import _lib as L

npt = int(sys.argv[1]) if len(sys.argv) > 1 else 150
s = abz.load_w90_series(os.path.join(ROOT, "tests", "golden", "svo_hr.dat.gz"))
dev = s.device()
ctx = dev.ctx


def timeit(rule, reps=20):
    for _ in range(3):
        rule.rebuild()
    ctx.sync()
    ctx.prof_enable(True)
    ctx.prof_reset()
    for _ in range(reps):
        rule.rebuild()
    ctx.sync()
    ms, n = ctx.prof_read(L.K_EVAL)
    ctx.prof_enable(False)
    return ms / n


print("build/free:", " ".join(f"{timeit(r):.4f}" for r in (abz.DeviceRule(dev, npt, None, 3) for _ in range(1)) ))
for i in range(5):
    r = abz.DeviceRule(dev, npt, None, 3)
    print(f"  fresh rule {i}: {timeit(r):.4f} ms")
    r.close()
keep = [abz.DeviceRule(dev, npt, None, 3) for _ in range(5)]
for i, r in enumerate(keep):
    print(f"  live rule {i}: {timeit(r):.4f} ms")
for rep in range(3):
    print("  again live 0..4:", " ".join(f"{timeit(r):.4f}" for r in keep))
