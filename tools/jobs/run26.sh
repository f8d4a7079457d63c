#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_block.py -x -q 2>&1 | tail -15 > gpurun_out/blk.log
cat gpurun_out/blk.log
