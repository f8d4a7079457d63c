#!/bin/bash
mkdir -p gpurun_out; : > gpurun_out/blk.log
timeout 1500 python -m pytest tests/test_gpu_paths.py tests/test_gpu_block.py -x -q -k "pw_ or block_backward" 2>&1 | tail -3 >> gpurun_out/blk.log
timeout 600 python tests/devtools/pw_micro.py "240,1440,7" "104,624,14" "1152,192,7" "1440,240,7" "1728,320,7" "624,96,14" "312,80,14" --n 512 2>&1 | grep -v amdgpu | grep npix >> gpurun_out/blk.log
timeout 900 python bench.py --steps 30 --warmup 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/blk.log
cat gpurun_out/blk.log
