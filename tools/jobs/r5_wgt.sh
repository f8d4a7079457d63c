#!/bin/bash
exec < /dev/null
O=gpurun_out/r5wgt; mkdir -p $O
b() { ( export "$@"; timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'])" ); }
for rep in 1 2; do
b FROST_WG_TARGET=128
b FROST_WG_TARGET=64
b FROST_WG_TARGET=96
b FROST_WG_TARGET=192
b FROST_WG_TARGET=256
b FROST_WG_TARGET=384
done 2>&1 | tee $O/ab.txt
