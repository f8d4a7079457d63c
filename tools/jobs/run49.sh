#!/bin/bash
mkdir -p gpurun_out; : > gpurun_out/blk.log
for cfg in "FROST_WG_STREAM=1" "FROST_WG_STREAM=3" "FROST_WG_STREAM=0" "FROST_DW_WG_MAXW=14" "FROST_PWEW_CAP=512" "FROST_WG_TARGET=512"; do
echo "== $cfg" >> gpurun_out/blk.log
env $cfg timeout 900 python bench.py --steps 30 --warmup 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/blk.log
done
cat gpurun_out/blk.log
