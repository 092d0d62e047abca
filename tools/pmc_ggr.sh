#!/bin/bash
# Run ON THE GPU BOX from the repo root:  bash tools/pmc_ggr.sh <tag> [npt]
# rocprofv3 passes over the GGR build + scan kernels: kernel statistics, SQ / LDS counters, HBM traffic (WRITE_SIZE and
# FETCH_SIZE in separate passes; the program directly after `--`, counters never together with trace domains).
set -e
TAG=${1:-r03}
NPT=${2:-150}
R=$(pwd)
OUT=$R/gpurun_out/pmc_ggr_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/prof_ggr.py $NPT 20 5 > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/sq -- python3 $R/tools/prof_ggr.py $NPT 5 2 > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $OUT/lds -- python3 $R/tools/prof_ggr.py $NPT 5 2 > $OUT/lds.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/tools/prof_ggr.py $NPT 5 2 > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/tools/prof_ggr.py $NPT 5 2 > $OUT/fetch.log 2>&1
cd $R
python3 - "$TAG" "$NPT" <<'PY'
import collections, csv, glob, json, sys
tag, npt = sys.argv[1], int(sys.argv[2])
base = f"gpurun_out/pmc_ggr_{tag}"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for name in ("sq", "lds", "write", "fetch"):
    for f in glob.glob(f"{base}/{name}/*/*_counter_collection.csv"):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:90]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            key = r["Dispatch_Id"]
            if key not in seen:
                seen.add(key)
                agg[k]["ns_" + name].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
out = {}
for k, v in agg.items():
    if not any(t in k for t in ("ggr", "contract_grid", "velocity", "eval_grid")):
        continue
    d = {c: sum(xs) / len(xs) for c, xs in v.items()}  # per launch
    d["launches"] = {c: len(xs) for c, xs in v.items() if c.startswith("ns_")}
    wc = d.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        d["frac_wave_cycles_valu_active"] = d.get("SQ_ACTIVE_INST_VALU", 0.0) / wc
        d["frac_wave_cycles_issue_stalled"] = d.get("SQ_WAIT_INST_ANY", 0.0) / wc
        d["frac_wave_cycles_waiting"] = d.get("SQ_WAIT_ANY", 0.0) / wc
        d["valu_insts_per_node"] = 64.0 * d.get("SQ_INSTS_VALU", 0.0) / npt ** 3
    if "WRITE_SIZE" in d and "FETCH_SIZE" in d:
        d["hbm_bytes_per_launch"] = 1024.0 * (d["WRITE_SIZE"] + 2.0 * d["FETCH_SIZE"])
        d["hbm_bytes_note"] = "1024*(WRITE_SIZE + 2*FETCH_SIZE): gfx950 FETCH_SIZE reports half of wide coalesced reads (MI355X_MICROARCH.md)"
    out[k] = d
json.dump({"note": "averages per launch; SQ_* cycle counters are in quad-cycles (MI355X_MICROARCH.md); npt = %d, SVO 3 bands" % npt,
           "kernels": out}, open(f"{base}/{tag}_ggr_pmc_summary.json", "w"), indent=1, sort_keys=True)
for k, d in out.items():
    print(k[:60], {x: (round(y, 4) if isinstance(y, float) else y) for x, y in d.items() if x.startswith("frac") or x.startswith("valu_insts") or x.startswith("hbm_bytes_per") or x.startswith("ns_")})
PY
for f in $OUT/stats/*/*_kernel_stats.csv; do cp $f $OUT/${TAG}_ggr_kernel_stats.csv; done
