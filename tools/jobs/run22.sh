#!/bin/bash
mkdir -p gpurun_out; : > gpurun_out/blk.log
timeout 600 python tests/devtools/blk_pair.py "240,1440,7,5" "192,1152,7,3" "288,1728,7,5" "104,624,14,5" "120,360,14,3" "160,960,14,5" --n 512 2>&1 | grep -E "fused|layerwise|Error|error" >> gpurun_out/blk.log
for cfg in "2 4" "3 3" "4 4" "2 2" "3 6"; do
  set -- $cfg
  echo "== CPW=$1 IMGS=$2" >> gpurun_out/blk.log
  FROST_BLK_CPW=$1 FROST_BLK_IMGS=$2 timeout 600 python tests/devtools/blk_pair.py "240,1440,7,5" "104,624,14,5" --n 512 2>&1 | grep -E "fused " >> gpurun_out/blk.log
done
cat gpurun_out/blk.log
