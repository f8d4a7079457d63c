#!/bin/bash
mkdir -p gpurun_out; : > gpurun_out/blk.log
for v in 256 128 64; do
echo "== FROST_PWC=$v" >> gpurun_out/blk.log
FROST_PWC=$v timeout 900 python bench.py --steps 30 --warmup 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/blk.log
done
FROST_PWC=96 timeout 900 python tests/devtools/layer_times.py 512 > gpurun_out/lt_pwc96.txt 2>&1
timeout 900 python tests/devtools/layer_times.py 512 > gpurun_out/lt_now.txt 2>&1
cat gpurun_out/blk.log
