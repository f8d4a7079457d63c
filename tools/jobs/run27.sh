#!/bin/bash
mkdir -p gpurun_out
bash tools/collect_profiles.sh r03 512 > gpurun_out/collect_r03.log 2>&1
timeout 900 python bench.py > gpurun_out/bench_r03.json 2> gpurun_out/bench_r03.err
timeout 900 python tests/devtools/layer_times.py 512 > gpurun_out/layer_times_r03_b512.txt 2>&1
cp gpurun_out/prof_r03/stats/*/*kernel_stats.csv gpurun_out/ 2>/dev/null || find gpurun_out/prof_r03/stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/r03_kernel_stats.csv \;
# keep the merge small: drop raw traces
find gpurun_out/prof_r03 -name "*kernel_trace.csv" -size +20M -delete
find gpurun_out/prof_r03 -name "*counter_collection.csv" -size +20M -delete
tail -5 gpurun_out/collect_r03.log; tail -c 600 gpurun_out/bench_r03.json; tail -5 gpurun_out/layer_times_r03_b512.txt
