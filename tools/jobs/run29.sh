#!/bin/bash
mkdir -p gpurun_out; : > gpurun_out/blk.log
timeout 1200 python -m pytest tests/test_gpu_paths.py -x -q -k "dw_ and (14 or 7) or exact_without" 2>&1 | tail -15 >> gpurun_out/blk.log
for v in 0 1; do
echo "== FROST_BLOCK_DWBWD=$v" >> gpurun_out/blk.log
FROST_BLOCK_DWBWD=$v timeout 600 python tests/devtools/pw_micro.py "1440,1440,7,5,1" "1152,1152,7,3,1" "624,624,14,5,1" "360,360,14,3,1" "312,312,14,5,1" "960,960,14,5,1" --n 512 2>&1 | grep -v amdgpu >> gpurun_out/blk.log
done
cat gpurun_out/blk.log
