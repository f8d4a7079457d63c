#!/bin/bash
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/bench_r03.json 2> gpurun_out/bench_r03.err
tail -c 300 gpurun_out/bench_r03.json
