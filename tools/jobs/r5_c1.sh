#!/bin/bash
exec < /dev/null
O=gpurun_out/r5c1; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_paths.py -q -x -W ignore -k "conv1_reduce or one_sweep" 2>&1 | tail -25 > $O/tests.log; tail -12 $O/tests.log | cut -c1-500
timeout 900 python -m pytest tests/test_gpu_prod.py -q -x -W ignore -k "block_by_block" 2>&1 | tail -5 | cut -c1-300
b() { ( export "$@"; timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'])" ); }
for rep in 1 2 3; do
b FROST_DWB_C1=0
b FROST_DWB_C1=1
done 2>&1 | tee $O/ab.txt
