#!/bin/bash
mkdir -p gpurun_out; : > gpurun_out/blk.log
timeout 1500 python -m pytest tests/test_gpu_block.py tests/test_gpu_paths.py -x -q -k "block or pw_104_624 or pw_288_1728 or dw_624 or dw_1152" 2>&1 | tail -3 >> gpurun_out/blk.log
timeout 900 python bench.py --steps 30 --warmup 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/blk.log
timeout 900 python tests/devtools/layer_times.py 512 > gpurun_out/lt_now.txt 2>&1
grep -E "blk_|pw_bwd_dc" gpurun_out/lt_now.txt | grep "^ " | awk '{print $2,$3,$4}' | head -60 >> gpurun_out/blk.log
tail -1 gpurun_out/lt_now.txt >> gpurun_out/blk.log
cat gpurun_out/blk.log
