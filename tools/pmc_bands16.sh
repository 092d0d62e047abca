#!/bin/bash
# Run ON THE GPU BOX from the repo root:  bash tools/pmc_bands16.sh <tag>
# SQ / LDS counters of the 16-band kernels (separate --pmc passes; the program directly after `--`).
set -e
TAG=${1:-r02}
R=$(pwd)
OUT=$R/gpurun_out/pmc_bands16_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/sq -- python3 $R/tools/prof_bands16.py 0.1 > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT/lds -- python3 $R/tools/prof_bands16.py 0.1 > $OUT/lds.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/prof_bands16.py 0.1 > $OUT/stats.log 2>&1
cd $R
python3 - "$TAG" <<'PY'
import collections, csv, glob, json, sys
tag = sys.argv[1]
base = f"gpurun_out/pmc_bands16_{tag}"
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for name in ("sq", "lds"):
    for f in glob.glob(f"{base}/{name}/*/*_counter_collection.csv"):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:70]
            agg[k][f"{r['Counter_Name']}"] += float(r["Counter_Value"]) if name == "sq" or r["Counter_Name"] not in ("SQ_INSTS_VALU", "SQ_WAVE_CYCLES") else 0.0
            if name == "lds" and r["Counter_Name"] in ("SQ_INSTS_VALU", "SQ_WAVE_CYCLES"):
                agg[k][r["Counter_Name"] + "_ldspass"] += float(r["Counter_Value"])
            key = (name, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                agg[k]["ns_" + name] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                agg[k]["launches_" + name] += 1
out = {}
for k, v in agg.items():
    if not any(t in k for t in ("gen_inner_panel", "gen_grid_sum", "gen_grid_eig", "gen_rows_reduce", "gen_quad", "gen_duo")):
        continue
    d = dict(v)
    wc = d.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        d["frac_wave_cycles_valu_active"] = d.get("SQ_ACTIVE_INST_VALU", 0.0) / wc
        d["frac_wave_cycles_issue_stalled"] = d.get("SQ_WAIT_INST_ANY", 0.0) / wc
        d["frac_wave_cycles_waiting"] = d.get("SQ_WAIT_ANY", 0.0) / wc
    wl = d.get("SQ_WAVE_CYCLES_ldspass", 0.0)
    if wl:
        d["frac_wave_cycles_lds_active"] = d.get("SQ_ACTIVE_INST_LDS", 0.0) / wl
        d["frac_wave_cycles_lds_issue_stalled"] = d.get("SQ_WAIT_INST_LDS", 0.0) / wl
        d["lds_insts_per_valu_inst"] = d.get("SQ_INSTS_LDS", 0.0) / max(d.get("SQ_INSTS_VALU_ldspass", 1.0), 1.0)
    out[k] = d
json.dump({"note": "sums over all dispatches of the pass; SQ_* cycle counters are in quad-cycles (MI355X_MICROARCH.md); "
                   "fractions are of SQ_WAVE_CYCLES of the same pass", "kernels": out},
          open(f"{base}/{tag}_bands16_pmc_summary.json", "w"), indent=1, sort_keys=True)
for k, d in out.items():
    print(k, {x: round(y, 3) for x, y in d.items() if x.startswith("frac") or x.startswith("lds_insts")})
PY
for f in $OUT/stats/*/*_kernel_stats.csv; do cp $f $OUT/${TAG}_bands16_kernel_stats.csv; done
