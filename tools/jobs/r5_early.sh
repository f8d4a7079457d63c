#!/bin/bash
# Early end-of-round-style job (profiles + bench line + per-layer table), so that r05 evidence exists before further kernel work; the full job is final_round5.sh.
exec < /dev/null
mkdir -p gpurun_out
timeout 1200 bash tools/collect_profiles.sh r05 512 > gpurun_out/collect_r05.log 2>&1
timeout 600 python bench.py > gpurun_out/bench_r05.json 2> gpurun_out/bench_r05.err
timeout 600 python tests/devtools/layer_times.py 512 > gpurun_out/layer_times_r05_b512.txt 2>&1
timeout 300 python tests/devtools/layer_times.py 8 > gpurun_out/layer_times_r05_b8.txt 2>&1
f=$(find gpurun_out/prof_r05/stats -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r05_kernel_stats.csv
find gpurun_out/prof_r05 -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
find gpurun_out/prof_r05 -name "*counter_collection.csv" -size +20M -delete 2>/dev/null
tail -c 400 gpurun_out/bench_r05.json; echo; tail -5 gpurun_out/collect_r05.log
