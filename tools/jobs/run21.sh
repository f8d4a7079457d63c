#!/bin/bash
mkdir -p gpurun_out; : > gpurun_out/blk.log
for a in 0 1 2 4 8 16 3 31; do
  echo "== ABL=$a" >> gpurun_out/blk.log
  FROST_BLK_ABL=$a timeout 600 python tests/devtools/blk_pair.py "240,1440,7,5" "104,624,14,5" --n 512 2>&1 | grep -E "fused|layerwise" >> gpurun_out/blk.log
done
for cs in 1 3 4; do
  echo "== CS=$cs" >> gpurun_out/blk.log
  FROST_BLK_CS=$cs timeout 600 python tests/devtools/blk_pair.py "240,1440,7,5" "104,624,14,5" --n 512 2>&1 | grep -E "fused|layerwise" >> gpurun_out/blk.log
done
cat gpurun_out/blk.log
