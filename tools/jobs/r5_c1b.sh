#!/bin/bash
exec < /dev/null
O=gpurun_out/r5c1; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_paths.py -q -x -W ignore -s -k "conv1_reduce" 2>&1 | grep -v "^$" | tail -25 > $O/tests2.log; tail -14 $O/tests2.log | cut -c1-600
