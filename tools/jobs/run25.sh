#!/bin/bash
mkdir -p gpurun_out; : > gpurun_out/blk.log
for cfg in "0 0" "1 0" "1 1"; do
set -- $cfg
echo "== bench FROST_BLOCK_PAIR=$1 FROST_BLOCK_DWRED=$2" >> gpurun_out/blk.log
FROST_BLOCK_PAIR=$1 FROST_BLOCK_DWRED=$2 timeout 900 python bench.py --steps 30 --warmup 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/blk.log
done
for i in 2 4; do
echo "== IMGS_B=$i" >> gpurun_out/blk.log
FROST_BLK_IMGS_B=$i timeout 900 python bench.py --steps 30 --warmup 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/blk.log
done
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_fbgemm.py tests/test_gpu_prod.py -x -q 2>&1 | tail -3 >> gpurun_out/blk.log
cat gpurun_out/blk.log
