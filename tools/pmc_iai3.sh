#!/bin/bash
# Run ON THE GPU BOX from the repo root:  bash tools/pmc_iai3.sh <tag>
# SQ / LDS counters and kernel statistics of the 3-band IAI solve (inner_adaptive_kernel<3>, contraction kernels).
set -e
TAG=${1:-r03}
R=$(pwd)
OUT=$R/gpurun_out/pmc_iai3_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/prof_iai3.py 3 > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/sq -- python3 $R/tools/prof_iai3.py 2 > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/lds -- python3 $R/tools/prof_iai3.py 2 > $OUT/lds.log 2>&1
cd $R
python3 - "$TAG" <<'PY'
import collections, csv, glob, json, sys
tag = sys.argv[1]
base = f"gpurun_out/pmc_iai3_{tag}"
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for name in ("sq", "lds"):
    for f in glob.glob(f"{base}/{name}/*/*_counter_collection.csv"):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:70]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = r["Dispatch_Id"]
            if key not in seen:
                seen.add(key)
                agg[k]["ns_" + name] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                agg[k]["launches_" + name] += 1
out = {}
for k, v in agg.items():
    if v.get("ns_sq", 0) <= 0:
        continue
    wc = max(v.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    v["frac_wave_cycles_valu_active"] = v.get("SQ_ACTIVE_INST_VALU", 0.0) / wc
    v["frac_wave_cycles_waiting"] = v.get("SQ_WAIT_ANY", 0.0) / wc
    v["frac_wave_cycles_issue_stalled"] = v.get("SQ_WAIT_INST_ANY", 0.0) / wc
    v["valu_wave_insts_per_launch"] = v.get("SQ_INSTS_VALU", 0.0) / max(v.get("launches_sq", 1.0), 1.0)
    out[k] = dict(v)
json.dump({"workload": "SVO 3-band IAI on the FBZ, eta = 0.01, abstol 1e-3 (2 solves per counter pass)", "kernels": out},
          open(f"{base}/{tag}_iai3_pmc_summary.json", "w"), indent=1, sort_keys=True)
import shutil
for f in glob.glob(f"{base}/stats/*/*_kernel_stats.csv"):
    shutil.copy(f, f"{base}/{tag}_iai3_kernel_stats.csv")
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["ns_sq"])[:4]:
    print(k, {a: round(b, 3) for a, b in v.items() if a.startswith("frac") or a.startswith("ns_") or a.startswith("launch") or a == "valu_wave_insts_per_launch"})
PY
