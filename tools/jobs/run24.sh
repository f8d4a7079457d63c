#!/bin/bash
mkdir -p gpurun_out; : > gpurun_out/blk.log
timeout 600 python tests/devtools/blk_pair.py "240,1440,7,5,192" "192,1152,7,3,192" "288,1728,7,5,320" "104,624,14,5,96" "120,360,14,3,96" "160,960,14,5,96" "100,312,14,5,80" --n 512 2>&1 | grep -E "fused|layer|Error|error|max" >> gpurun_out/blk.log
timeout 600 python tests/devtools/blk_pair.py "240,1440,7,5,192" "104,624,14,5,96" --n 37 2>&1 | grep -E "fused|layer|Error|error|max" >> gpurun_out/blk.log
cat gpurun_out/blk.log
