"""Condense rocprofv3 output of tools/collect_profiles.sh into small files fit for profiles/."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

out, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(os.path.dirname(out), f"profiles_{tag}_summary")
os.makedirs(dst, exist_ok=True)
for f in glob.glob(os.path.join(out, "stats", "*", "*_kernel_stats.csv")):
    shutil.copy(f, os.path.join(dst, f"{tag}_kernel_stats.csv"))
summary = {}
for name in ("pmc_sq", "pmc_write", "pmc_fetch"):
    for f in glob.glob(os.path.join(out, name, "*", "*_counter_collection.csv")):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[k]["_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        for k, v in agg.items():
            ent = summary.setdefault(k, {})
            for c, xs in v.items():
                if c == "_ns":
                    ent.setdefault("avg_ns_under_pmc", {})[name] = sum(xs) / len(xs)
                else:
                    ent[c] = sum(xs) / len(xs)
                    ent["launches_" + c] = len(xs)
# HBM traffic per launch of the Fourier-eval kernel (KB counters; FETCH_SIZE doubled on gfx950 for wide reads)
for k, ent in summary.items():
    if "WRITE_SIZE" in ent and "FETCH_SIZE" in ent:
        ent["hbm_bytes_per_launch"] = 1024.0 * (ent["WRITE_SIZE"] + 2.0 * ent["FETCH_SIZE"])
        ent["hbm_bytes_note"] = "1024*(WRITE_SIZE + 2*FETCH_SIZE): gfx950 FETCH_SIZE reports half of wide coalesced reads (MI355X_MICROARCH.md)"
json.dump(summary, open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w"), indent=1, sort_keys=True)
for k, ent in summary.items():
    if ("eval_grid_kernel" in k or "eval_grid_fused_kernel" in k) and "hbm_bytes_per_launch" in ent:
        json.dump({"kernel": k, "npt": 150, "tag": tag, "hbm_bytes_per_launch": ent["hbm_bytes_per_launch"],
                   "WRITE_SIZE_KB": ent["WRITE_SIZE"], "FETCH_SIZE_KB": ent["FETCH_SIZE"], "note": ent["hbm_bytes_note"],
                   "algorithmic_bytes_per_launch": 150**3 * 168},
                  open(os.path.join(dst, "r01_traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if "eval_grid" in k}, indent=1))
