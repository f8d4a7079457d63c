#!/bin/bash
mkdir -p gpurun_out; : > gpurun_out/blk.log
timeout 1500 python -m pytest tests/test_gpu_block.py -x -q 2>&1 | tail -12 >> gpurun_out/blk.log
timeout 900 python tests/devtools/layer_times.py 512 > gpurun_out/lt_1.txt 2>&1
grep -E "blk_dw_bred" gpurun_out/lt_1.txt | awk '{print $2,$3,$4}' | tr '\n' ';' >> gpurun_out/blk.log; echo >> gpurun_out/blk.log
tail -1 gpurun_out/lt_1.txt >> gpurun_out/blk.log
for v in 0 1; do
echo "== bench FROST_BLOCK_DGRED=$v" >> gpurun_out/blk.log
FROST_BLOCK_DGRED=$v timeout 900 python bench.py --steps 30 --warmup 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/blk.log
done
cat gpurun_out/blk.log
