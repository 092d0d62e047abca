#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root:  bash tools/collect_profiles.sh r01b
# Writes rocprofv3 summaries under gpurun_out/profiles_<tag>/ ; copy what should be judged into profiles/.
# Counters are collected in their own passes (no --stats / trace domains beside --pmc), FETCH_SIZE and
# WRITE_SIZE separately (TCC slot limit), as MI355X_MICROARCH.md prescribes.
set -e
TAG=${1:-r01}
R=$(pwd)
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 20 --warmup 3 --passes-per-step 16 --no-cpu --no-ref-layout --no-scaling-model > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq -- python3 $R/tools/prof_eval.py 150 5 32 > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/tools/prof_eval.py 150 5 32 > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/tools/prof_eval.py 150 5 32 > $OUT/pmc_fetch.log 2>&1
cd $R
python3 tools/summarize_profiles.py $OUT $TAG
