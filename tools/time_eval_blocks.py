"""Fourier-eval kernel time against the number of workgroups (ABZ_EVAL_BLOCKS), all candidates on the SAME rule buffer of
one process (placement of the buffer moves the kernel by up to 10 % between processes, so block counts must not be compared
across processes), repeated for a few re-allocations of the buffer.  Usage: time_eval_blocks.py npt [npt ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import autobzcore.jl_amd as abz
L = abz._lib
s = abz.load_w90_series(os.path.join(ROOT, "tests", "golden", "svo_hr.dat.gz"))
dev = s.device(); ctx = dev.ctx
cands = [int(v) for v in os.environ.get("CANDS", "2048,3072,4096,6144,8192").split(",")]
for npt in [int(v) for v in sys.argv[1:]]:
    keep = []
    for alloc in range(3):
        rule = abz.DeviceRule(dev, npt, None, L.WANT_H | L.WANT_EIG | (0 if os.environ.get("FULL_LAYOUT") else L.WANT_H_COMPACT))
        base, nb = rule.values_ptr()
        for _ in range(30): rule.rebuild()
        ctx.sync()
        row = []
        for b in cands + cands[:1]:
            os.environ["ABZ_EVAL_BLOCKS"] = str(abs(b))
            os.environ["ABZ_NT_STORES"] = "0" if b < 0 else "1"  # negative candidate: temporal stores
            if os.environ.get("OCCS"):  # candidates are then "occ * 100000 + blocks"
                os.environ["ABZ_EVAL_OCC"] = str(abs(b) // 100000)
                os.environ["ABZ_EVAL_BLOCKS"] = str(abs(b) % 100000)
            for _ in range(10): rule.rebuild()
            ctx.sync()
            ctx.prof_enable(True, kernels=[L.K_EVAL]); ctx.prof_reset()
            for _ in range(200 if npt < 300 else 40): rule.rebuild()
            ctx.sync()
            ms, n = ctx.prof_read(L.K_EVAL); ctx.prof_enable(False)
            row.append(ms / n)
        print(f"npt {npt:4d} buffer {base:#x}: " + "  ".join(f"{b}: {t:.4f}" for b, t in zip(cands + cands[:1], row)), flush=True)
        keep.append(rule)  # keep it allocated so that the next buffer lands elsewhere
    for r in keep: r.close()
