#!/bin/bash
mkdir -p gpurun_out
timeout 900 python tests/devtools/layer_times.py 512 > gpurun_out/lt_now.txt 2>&1
tail -1 gpurun_out/lt_now.txt
