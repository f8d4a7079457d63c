#!/bin/bash
mkdir -p gpurun_out; : > gpurun_out/blk.log
timeout 1200 python -m pytest tests/test_gpu_paths.py -x -q -k "exact_without" 2>&1 | tail -5 >> gpurun_out/blk.log
for v in 0 1; do
FROST_BLOCK_DWBWD=$v timeout 900 python tests/devtools/layer_times.py 512 > gpurun_out/lt_$v.txt 2>&1
done
grep -E "layer4.[1-4].conv2|layer5.0.conv2" gpurun_out/lt_0.txt | grep -E "bwd|wgrad|dgrad" >> gpurun_out/blk.log
echo ==== >> gpurun_out/blk.log
grep -E "layer4.[1-4].conv2|layer5.0.conv2" gpurun_out/lt_1.txt | grep -E "bwd|wgrad|dgrad" >> gpurun_out/blk.log
tail -1 gpurun_out/lt_0.txt >> gpurun_out/blk.log; tail -1 gpurun_out/lt_1.txt >> gpurun_out/blk.log
cat gpurun_out/blk.log
