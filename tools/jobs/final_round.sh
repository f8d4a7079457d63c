#!/bin/bash
mkdir -p gpurun_out
bash tools/collect_profiles.sh r03 512 > gpurun_out/collect_r03.log 2>&1
cp gpurun_out/prof_r03/summary.json profiles/r03_kernels_b512.json 2>/dev/null
timeout 900 python bench.py > gpurun_out/bench_r03.json 2> gpurun_out/bench_r03.err
timeout 900 python tests/devtools/layer_times.py 512 > gpurun_out/layer_times_r03_b512.txt 2>&1
find gpurun_out/prof_r03/stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/r03_kernel_stats.csv \;
find gpurun_out/prof_r03 -name "*kernel_trace.csv" -size +20M -delete
find gpurun_out/prof_r03 -name "*counter_collection.csv" -size +20M -delete
: > gpurun_out/r03_side_workloads.jsonl
for wl in infer int8 detect float; do timeout 600 python bench.py --workload $wl 2>/dev/null | tail -1 >> gpurun_out/r03_side_workloads.jsonl; done
( timeout 3400 python -m pytest tests -q -m gpu 2>&1 | tail -8 ) > gpurun_out/gpu_suite_r03.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke_r03.log 2>&1
tail -c 400 gpurun_out/bench_r03.json; tail -4 gpurun_out/gpu_suite_r03.log; tail -1 gpurun_out/smoke_r03.log; tail -1 gpurun_out/layer_times_r03_b512.txt
