#!/bin/bash
mkdir -p gpurun_out; : > gpurun_out/blk.log
for v in 0 150000 500000; do
echo "== bench FUSE_MINPIX=$v" >> gpurun_out/blk.log
FROST_PW_FUSE_MINPIX=$v timeout 900 python bench.py --steps 30 --warmup 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/blk.log
done
cat gpurun_out/blk.log
