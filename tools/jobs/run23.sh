#!/bin/bash
mkdir -p gpurun_out; : > gpurun_out/blk.log
timeout 600 python tests/devtools/blk_pair.py "240,1440,7,5" "192,1152,7,3" "288,1728,7,5" "104,624,14,5" "120,360,14,3" "160,960,14,5" --n 512 2>&1 | grep -E "fused|layerwise|Error|error" >> gpurun_out/blk.log
for v in 0 1; do
echo "== bench FROST_BLOCK_PAIR=$v" >> gpurun_out/blk.log
FROST_BLOCK_PAIR=$v timeout 900 python bench.py --steps 30 --warmup 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/blk.log
done
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_fbgemm.py -x -q 2>&1 | tail -3 >> gpurun_out/blk.log
cat gpurun_out/blk.log
