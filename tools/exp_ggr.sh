mkdir -p gpurun_out/r3b
run() { echo "== $*"; env "$@" timeout -k 10 120 python tools/time_ggr.py ${NPTS:-150} 2>&1 | grep GGR | sed -E 's/scan.*dos\[/dos[/'; }
{
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "ggr or kshard or slab" 2>&1 | tail -3
run X=1
run ABZ_GGR_PAIRS_PER_BLOCK=0
run ABZ_GGR_PAIRS_PER_BLOCK=4
run ABZ_GGR_PAIRS_PER_BLOCK=12
run ABZ_GGR_KB=1
run ABZ_GGR_KB=1 ABZ_GGR_PAIRS_PER_BLOCK=0
run ABZ_GGR_KB=1 ABZ_GGR_PAIRS_PER_BLOCK=4
run ABZ_GGR_KB=1 ABZ_GGR_PAIRS_PER_BLOCK=6
run ABZ_GGR_KB=1 ABZ_GGR_PAIRS_PER_BLOCK=12
run ABZ_GGR_KB=1 NPTS=100
run ABZ_GGR_KB=1 NPTS=200
run NPTS=100
run NPTS=200
} > gpurun_out/r3b/exp7.log 2>&1
cat gpurun_out/r3b/exp7.log
