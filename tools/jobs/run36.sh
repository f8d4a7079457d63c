#!/bin/bash
mkdir -p gpurun_out; : > gpurun_out/blk.log
timeout 1500 python -m pytest tests/test_gpu_block.py -q 2>&1 | tail -25 >> gpurun_out/blk.log
cat gpurun_out/blk.log
