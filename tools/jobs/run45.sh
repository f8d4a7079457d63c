#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_round3.py -q -k "flag_summary" 2>&1 | tail -40 > gpurun_out/flag.log
cat gpurun_out/flag.log
