#!/bin/bash
mkdir -p gpurun_out; : > gpurun_out/blk.log
timeout 1500 python -m pytest tests/test_gpu_paths.py tests/test_gpu_block.py -x -q -k "pw_104_624 or pw_120_360 or pw_320_1280 or pw_288_1728 or pw_240_1440 or bit_identical" 2>&1 | tail -4 >> gpurun_out/blk.log
for v in 0 1; do
echo "== FROST_PWC_STATS=$v" >> gpurun_out/blk.log
FROST_PWC_STATS=$v timeout 600 python tests/devtools/pw_micro.py "240,1440,7" "192,1152,7" "288,1728,7" "104,624,14" "160,960,14" "320,1280,7" "56,304,28" --n 512 2>&1 | grep -v amdgpu | awk '{print $1,$2,$3,$4}' >> gpurun_out/blk.log
echo "bench" >> gpurun_out/blk.log
FROST_PWC_STATS=$v timeout 900 python bench.py --steps 30 --warmup 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/blk.log
done
cat gpurun_out/blk.log
