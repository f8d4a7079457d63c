#!/bin/bash
mkdir -p gpurun_out; : > gpurun_out/blk.log
for cfg in "1 1" "2 1" "1 2" "2 2"; do
set -- $cfg
echo "== ROUNDS_A=$1 ROUNDS_R=$2" >> gpurun_out/blk.log
FROST_BLK_ROUNDS_A=$1 FROST_BLK_ROUNDS_R=$2 timeout 900 python tests/devtools/layer_times.py 512 > gpurun_out/lt_1.txt 2>&1
grep -E "blk_expand_dw" gpurun_out/lt_1.txt | grep "^ " | awk '{print $4}' | tr '\n' ';' >> gpurun_out/blk.log; echo >> gpurun_out/blk.log
grep -E "blk_dw_bred" gpurun_out/lt_1.txt | grep "^ " | awk '{print $4}' | tr '\n' ';' >> gpurun_out/blk.log; echo >> gpurun_out/blk.log
done
timeout 900 python bench.py --steps 30 --warmup 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/blk.log
cat gpurun_out/blk.log
