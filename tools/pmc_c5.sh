#!/bin/bash
# Run ON THE GPU BOX from the repo root: SQ / LDS counters of the config-5 (16-band IAI) kernels.
set -e
R=$(pwd)
OUT=$R/gpurun_out/pmc_c5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/sq -- python3 $R/tools/prof_c5.py 0.001 > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/lds -- python3 $R/tools/prof_c5.py 0.001 > $OUT/lds.log 2>&1
cd $R
python3 - <<'PY'
import collections, csv, glob, json
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for name in ("sq", "lds"):
    for f in glob.glob(f"gpurun_out/pmc_c5/{name}/*/*_counter_collection.csv"):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (name, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                agg[k]["ns_" + name] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                agg[k]["launches_" + name] += 1
json.dump(agg, open("gpurun_out/pmc_c5/summary.json", "w"), indent=1, sort_keys=True)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("ns_sq", 0))[:4]:
    print(k, json.dumps(v, sort_keys=True))
PY
