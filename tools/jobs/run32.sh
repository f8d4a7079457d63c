#!/bin/bash
mkdir -p gpurun_out; : > gpurun_out/blk.log
for wg in 256 512 768 1024; do
echo "== WGS_C=$wg" >> gpurun_out/blk.log
FROST_BLK_WGS_C=$wg FROST_BLOCK_DWBWD=2 timeout 900 python tests/devtools/layer_times.py 512 > gpurun_out/lt_1.txt 2>&1
grep -E "blk_dw_bwd" gpurun_out/lt_1.txt | awk '{print $2,$3,$4}' | tr '\n' ';' >> gpurun_out/blk.log; echo >> gpurun_out/blk.log
tail -1 gpurun_out/lt_1.txt >> gpurun_out/blk.log
done
cat gpurun_out/blk.log
