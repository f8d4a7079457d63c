#!/bin/bash
exec < /dev/null
O=gpurun_out/r5dws1; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_paths.py -q -x -W ignore -k "streaming or one_sweep or (fast_paths and dw_)" 2>&1 | tail -15 > $O/tests.log; tail -8 $O/tests.log | cut -c1-600
b() { ( export "$@"; timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'])" ); }
for rep in 1 2; do
b FROST_DW_STREAM=0
b FROST_DW_STREAM=1
b FROST_DW_STREAM=2
b FROST_DW_STREAM=4
b FROST_DW_STREAM=7
b FROST_DW_STREAM=7 FROST_DWB_OCC=4
done 2>&1 | tee $O/ab.txt
