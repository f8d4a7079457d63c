#!/bin/bash
mkdir -p gpurun_out; : > gpurun_out/blk.log
timeout 1500 python -m pytest tests/test_gpu_paths.py -x -q -k "pw_" 2>&1 | tail -3 >> gpurun_out/blk.log
timeout 600 python tests/devtools/pw_micro.py "240,1440,7" "320,1280,7" "104,624,14" "120,720,14" "56,304,28" --n 512 2>&1 | grep -v amdgpu | grep npix >> gpurun_out/blk.log
for v in 0 1; do
FROST_PWC_EMIT=$v timeout 900 python bench.py --steps 30 --warmup 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/blk.log
done
cat gpurun_out/blk.log
