mkdir -p gpurun_out/r3b
run() { echo "== $*"; env "$@" timeout -k 10 120 python tools/time_ggr.py ${NPTS:-150} 2>&1 | grep GGR | sed -E 's/scan.*//'; }
{
run X=0
run ABZ_GGR_DEBUG=1
run ABZ_GGR_DEBUG=2
run ABZ_GGR_DEBUG=3
run ABZ_GGR_DEBUG=4
run ABZ_GGR_KB=1
run ABZ_GGR_FUSE2=0
run ABZ_GGR_PAIRS_PER_BLOCK=4
run ABZ_GGR_PAIRS_PER_BLOCK=12
run ABZ_NT_STORES=0
} > gpurun_out/r3b/exp2.log 2>&1
cat gpurun_out/r3b/exp2.log
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "ggr or kshard or slab" 2>&1 | tail -3
