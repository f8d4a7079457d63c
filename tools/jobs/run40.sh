#!/bin/bash
mkdir -p gpurun_out; : > gpurun_out/blk.log
timeout 1500 python -m pytest tests/test_gpu_paths.py -x -q -k "pw_104_624 or pw_120_360 or pw_320_1280 or pw_288_1728 or pw_240_1440" 2>&1 | tail -4 >> gpurun_out/blk.log
timeout 600 python tests/devtools/pw_micro.py "240,1440,7" "288,1728,7" "104,624,14" "160,960,14" "56,336,28" "50,304,28" --n 512 2>&1 | grep -v amdgpu >> gpurun_out/blk.log
for wg in 1536 3072 6144; do
echo "== bench WGS=$wg" >> gpurun_out/blk.log
FROST_PWC_WGS=$wg timeout 900 python bench.py --steps 30 --warmup 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/blk.log
done
cat gpurun_out/blk.log
