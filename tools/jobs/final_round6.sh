#!/bin/bash
# End-of-round job: profiles (rocprofv3 kernel stats + PMC passes), default bench line, per-layer table, side workloads, other batch sizes, full GPU suite, smoke.
# Every step runs under `timeout` and reads nothing from stdin.
exec < /dev/null
mkdir -p gpurun_out
timeout 1500 bash tools/collect_profiles.sh r06 512 > gpurun_out/collect_r06.log 2>&1
[ -f gpurun_out/prof_r06/summary.json ] && cp gpurun_out/prof_r06/summary.json profiles/r06_kernels_b512.json
timeout 1200 python bench.py > gpurun_out/bench_r06.json 2> gpurun_out/bench_r06.err
timeout 900 python tests/devtools/layer_times.py 512 > gpurun_out/layer_times_r06_b512.txt 2>&1
f=$(find gpurun_out/prof_r06/stats -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r06_kernel_stats.csv
find gpurun_out/prof_r06 -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
find gpurun_out/prof_r06 -name "*counter_collection.csv" -size +20M -delete 2>/dev/null
: > gpurun_out/r06_side_workloads.jsonl
for wl in infer int8 detect float; do timeout 600 python bench.py --workload $wl 2>/dev/null | tail -1 >> gpurun_out/r06_side_workloads.jsonl; done
FROST_FLOAT_PRECISION=fp32 timeout 600 python bench.py --workload float 2>/dev/null | tail -1 >> gpurun_out/r06_side_workloads.jsonl
( cd /tmp && export TMPDIR=/tmp && cd $OLDPWD && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r06_float -o s -- python bench.py --workload float --steps 10 --warmup 3 > gpurun_out/prof_r06_float.log 2>&1 )
f=$(find gpurun_out/prof_r06_float -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r06_float_b256_kernel_stats.csv
find gpurun_out/prof_r06_float -name "*kernel_trace.csv" -delete 2>/dev/null
: > gpurun_out/r06_other_batches.jsonl
for b in 64 200 256; do timeout 600 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 >> gpurun_out/r06_other_batches.jsonl; done
# the fp32-gradient mode next to the production bf16 backward: B = 64 eager (bf16, fp32 fast forms, fp32 plain round-4 kernels), B = 512 captured (fp32), and its kernel table
: > gpurun_out/r06_grad_modes.jsonl
timeout 900 python bench.py --batch 64 --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 >> gpurun_out/r06_grad_modes.jsonl
FROST_GRAD=fp32 timeout 900 python bench.py --batch 64 --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 >> gpurun_out/r06_grad_modes.jsonl
FROST_GRAD=fp32 FROST_G32_PLAIN=1 timeout 1500 python bench.py --batch 64 --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 >> gpurun_out/r06_grad_modes.jsonl
FROST_GRAD=fp32 timeout 900 python bench.py --batch 512 --steps 10 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 >> gpurun_out/r06_grad_modes.jsonl
FROST_GRAD=mixed timeout 900 python bench.py --batch 512 --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 >> gpurun_out/r06_grad_modes.jsonl
( cd /tmp && export TMPDIR=/tmp && cd $OLDPWD && FROST_GRAD=fp32 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r06_g32 -o s -- python bench.py --batch 512 --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-roofline > gpurun_out/prof_r06_g32.log 2>&1 )
f=$(find gpurun_out/prof_r06_g32 -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r06_g32_b512_kernel_stats.csv
find gpurun_out/prof_r06_g32 -name "*kernel_trace.csv" -delete 2>/dev/null
timeout 600 python tools/bench_iblock.py > gpurun_out/r06_infer_blocks.txt 2>&1
timeout 600 bash tools/jobs/r6_trace.sh > gpurun_out/r06_trace.log 2>&1; cp gpurun_out/r6trace/replay_nodes.txt gpurun_out/r06_replay_nodes.txt 2>/dev/null; cp gpurun_out/r6trace/replay_nodes.json gpurun_out/r06_replay_nodes.json 2>/dev/null
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/gpu_suite_r06_full.log 2>&1; tail -8 gpurun_out/gpu_suite_r06_full.log > gpurun_out/gpu_suite_r06.log; grep -n -B2 -A40 "^____\|^E  " gpurun_out/gpu_suite_r06_full.log | head -150 > gpurun_out/gpu_suite_r06_failures.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke_r06.log 2>&1
tail -c 300 gpurun_out/bench_r06.json; echo; tail -3 gpurun_out/gpu_suite_r06.log; tail -1 gpurun_out/smoke_r06.log; cut -c1-200 gpurun_out/r06_other_batches.jsonl
