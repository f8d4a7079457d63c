#!/bin/bash
mkdir -p gpurun_out; : > gpurun_out/blk.log
for cfg in "256 0" "128 0" "64 0" "256 2" "256 4" "256 6"; do
set -- $cfg
echo "== bench FROST_PWC=$1 CPW=$2" >> gpurun_out/blk.log
FROST_PWC=$1 FROST_PWC_CPW=$2 timeout 900 python bench.py --steps 30 --warmup 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/blk.log
done
cat gpurun_out/blk.log
