#!/bin/bash
mkdir -p gpurun_out; : > gpurun_out/blk.log
for r in 1 2; do
echo "== ROUNDS_C=$r" >> gpurun_out/blk.log
FROST_BLK_ROUNDS_C=$r timeout 900 python tests/devtools/layer_times.py 512 > gpurun_out/lt_1.txt 2>&1
grep -E "blk_dw_bwd" gpurun_out/lt_1.txt | grep "^ " | awk '{print $2,$4}' | tr '\n' ';' >> gpurun_out/blk.log; echo >> gpurun_out/blk.log
FROST_BLK_ROUNDS_C=$r timeout 900 python bench.py --steps 30 --warmup 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/blk.log
done
cat gpurun_out/blk.log
