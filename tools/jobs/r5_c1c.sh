#!/bin/bash
exec < /dev/null
O=gpurun_out/r5c1; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_paths.py -q -x -W ignore -s -k "conv1_reduce" 2>&1 | grep -v "^$" | tail -25 > $O/tests3.log; tail -6 $O/tests3.log | cut -c1-400
b() { ( export "$@"; timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'])" ); }
for rep in 1 2 3; do
b FROST_DWB_C1=0
b FROST_DWB_C1=1
b FROST_DWB_C1=2
b FROST_DWB_C1=3
done 2>&1 | tee $O/ab2.txt
