#!/bin/bash
mkdir -p gpurun_out; : > gpurun_out/blk.log
timeout 1200 python -m pytest tests/test_gpu_paths.py -x -q -k "dw_ and (14 or 7) or exact_without" 2>&1 | tail -5 >> gpurun_out/blk.log
for v in 0 1 2; do
echo "== bench FROST_BLOCK_DWBWD=$v" >> gpurun_out/blk.log
FROST_BLOCK_DWBWD=$v timeout 900 python bench.py --steps 30 --warmup 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/blk.log
done
cat gpurun_out/blk.log
