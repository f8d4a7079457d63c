#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tests/devtools/blk_pair.py "240,1440,7,5" "192,1152,7,3" "288,1728,7,5" "104,624,14,5" "120,360,14,3" "160,960,14,5" --n 512 > gpurun_out/blk.log 2>&1
timeout 300 python tests/devtools/blk_pair.py "240,1440,7,5" "104,624,14,5" --n 37 >> gpurun_out/blk.log 2>&1
tail -40 gpurun_out/blk.log
