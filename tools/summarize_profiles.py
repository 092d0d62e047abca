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
# the same statistics WITHOUT each kernel's first (cold: code object load, first-touch) launch of the process
for f in glob.glob(os.path.join(out, "stats", "*", "*_kernel_trace.csv")):
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        per.setdefault(r["Kernel_Name"], []).append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    rows = []
    for k, xs in per.items():
        warm = xs[1:] if len(xs) > 1 else xs
        rows.append((k, len(xs), sum(xs), sum(warm) / len(warm), min(warm), max(warm), xs[0]))
    tot = sum(r[2] for r in rows) or 1.0
    with open(os.path.join(dst, f"{tag}_kernel_stats_warm.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationNs", "WarmAverageNs", "WarmMinNs", "WarmMaxNs", "FirstLaunchNs", "Percentage"])
        for r in sorted(rows, key=lambda r: -r[2]):
            w.writerow([r[0], r[1], int(r[2]), round(r[3], 1), int(r[4]), int(r[5]), int(r[6]), round(100.0 * r[2] / tot, 3)])
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
        compact = os.environ.get("ABZ_RULE_COMPACT", "1") != "0"  # the layout tools/prof_eval.py built (the host mirror's default)
        json.dump({"kernel": k, "npt": 150, "tag": tag, "layout": "hermitian-compact (96 B/k)" if compact else "reference (168 B/k)",
                   "hbm_bytes_per_launch": ent["hbm_bytes_per_launch"],
                   "WRITE_SIZE_KB": ent["WRITE_SIZE"], "FETCH_SIZE_KB": ent["FETCH_SIZE"], "note": ent["hbm_bytes_note"],
                   "algorithmic_bytes_per_launch": 150**3 * (96 if compact else 168)},
                  open(os.path.join(dst, f"{tag[:3]}_traffic{'_compact' if compact else ''}.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if "eval_grid" in k}, indent=1))
