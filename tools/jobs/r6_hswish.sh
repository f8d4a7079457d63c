#!/bin/bash
# round 6: hard-swish through the float, bf16-inference and converted kernels -- its tests and the suites of the kernels it touched
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_round6.py -q -k hswish -s 2>&1 | grep -v "^$" | tail -60 > gpurun_out/r6_hswish_tests.txt
timeout 2400 python -m pytest tests/test_gpu_convert.py tests/test_gpu_round3.py tests/test_gpu_float.py tests/test_gpu_infer.py -x -q 2>&1 | tail -8 > gpurun_out/r6_hswish_suites.txt
cat gpurun_out/r6_hswish_tests.txt gpurun_out/r6_hswish_suites.txt
