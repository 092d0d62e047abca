#!/bin/bash
# Run ON THE GPU BOX from the repo root:  bash tools/pmc_any.sh <tag> <kernel-substring> <nodes-per-launch> <python tool> [args...]
# rocprofv3 passes over any timing tool: kernel statistics, SQ and LDS counters (counters never together with trace domains
# other than --kernel-trace; the program directly after `--`).  Summary: gpurun_out/pmc_<tag>/<tag>_pmc_summary.json
set -e
TAG=$1; FILTER=$2; NODES=$3; TOOL=$(pwd)/$4; shift 4
R=$(pwd)
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $TOOL "$@" > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/sq -- python3 $TOOL "$@" > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/lds -- python3 $TOOL "$@" > $OUT/lds.log 2>&1
cd $R
python3 - "$TAG" "$FILTER" "$NODES" <<'PY'
import collections, csv, glob, json, sys
tag, flt, nodes = sys.argv[1], sys.argv[2], float(sys.argv[3])
base = f"gpurun_out/pmc_{tag}"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for name in ("sq", "lds"):
    for f in glob.glob(f"{base}/{name}/*/*_counter_collection.csv"):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:100]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            key = r["Dispatch_Id"]
            if key not in seen:
                seen.add(key)
                agg[k]["ns_" + name].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
                agg[k]["vgpr"].append(float(r.get("VGPR_Count", 0) or 0))
                agg[k]["scratch"].append(float(r.get("Scratch_Size", 0) or 0))
                agg[k]["lds_bytes"].append(float(r.get("LDS_Block_Size", 0) or 0))
out = {}
for k, v in agg.items():
    if flt not in k:
        continue
    d = {c: sum(xs) / len(xs) for c, xs in v.items()}  # per launch
    d["launches"] = {c: len(xs) for c, xs in v.items() if c.startswith("ns_")}
    wc = d.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        d["frac_wave_cycles_valu_active"] = d.get("SQ_ACTIVE_INST_VALU", 0.0) / wc
        d["frac_wave_cycles_issue_stalled"] = d.get("SQ_WAIT_INST_ANY", 0.0) / wc
        d["frac_wave_cycles_waiting"] = d.get("SQ_WAIT_ANY", 0.0) / wc
        d["valu_insts_per_node"] = 64.0 * d.get("SQ_INSTS_VALU", 0.0) / nodes
        d["valu_insts_per_wave"] = d.get("SQ_INSTS_VALU", 0.0) / max(d.get("SQ_WAVES", 1.0), 1.0)
    out[k] = d
json.dump({"note": "averages per launch; SQ_* cycle counters are in quad-cycles (MI355X_MICROARCH.md); nodes per launch = %g" % nodes,
           "kernels": out}, open(f"{base}/{tag}_pmc_summary.json", "w"), indent=1, sort_keys=True)
for k, d in out.items():
    print(k[:70], {x: (round(y, 4) if isinstance(y, float) else y) for x, y in d.items()
                   if x.startswith("frac") or x.startswith("valu_insts") or x.startswith("ns_") or x in ("vgpr", "scratch", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_WAVES")})
PY
for f in $OUT/stats/*/*_kernel_stats.csv; do cp $f $OUT/${TAG}_kernel_stats.csv; done
