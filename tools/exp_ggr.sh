mkdir -p gpurun_out/r3b
{
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "ggr or kshard or slab" 2>&1 | tail -3
SYMS=1 timeout -k 10 200 python tools/time_ggr.py 24 50 100 150
ABZ_GGR_SCAN=0 SYMS=1 timeout -k 10 200 python tools/time_ggr.py 24 50 100 150
} > gpurun_out/r3b/exp3.log 2>&1
cat gpurun_out/r3b/exp3.log
