#!/bin/bash
exec < /dev/null
O=gpurun_out/r5ob; mkdir -p $O
b() { ( export "$@"; timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'])" ); }
for rep in 1 2 3; do
b FROST_X=0
b FROST_DWB_OVER_BLK=1 FROST_DWB_MINW=14
done 2>&1 | tee $O/ab.txt
