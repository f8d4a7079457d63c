#!/bin/bash
# one-sweep depthwise backward incl. stride 2: parity tests, then interleaved whole-step A/B
exec < /dev/null
O=gpurun_out/r5dwb2; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_paths.py -q -x -W ignore -k "one_sweep" 2>&1 | tail -15 > $O/tests.log; tail -8 $O/tests.log | cut -c1-600
b() { ( export "$@"; timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'])" ); }
for rep in 1 2; do
b FROST_DW_BWD_ONE=0
b FROST_DWB_S2=0
b FROST_DWB_S2=1
b FROST_DWB_MINW=28
b FROST_DWB_MINW=56
b FROST_DWB_MINW=112
done 2>&1 | tee $O/ab.txt
