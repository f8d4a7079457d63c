#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_block.py -q -k "without_stochastic" 2>&1 | grep -E "assert|passed|failed|where" | head -20 > gpurun_out/blk.log
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_block.py -q -k "without_stochastic" 2>&1 | tail -1 >> gpurun_out/blk.log; done
timeout 900 python bench.py --steps 30 --warmup 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/blk.log
cat gpurun_out/blk.log
