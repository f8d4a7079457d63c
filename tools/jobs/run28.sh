#!/bin/bash
mkdir -p gpurun_out
( timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 ) > gpurun_out/gpu_suite_r03.log
: > gpurun_out/r03_side_workloads.jsonl
for wl in infer int8 detect float; do
  timeout 600 python bench.py --workload $wl 2>/dev/null | tail -1 >> gpurun_out/r03_side_workloads.jsonl
done
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_r03b.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke_r03.log 2>&1
tail -3 gpurun_out/gpu_suite_r03.log; cut -c1-300 gpurun_out/r03_side_workloads.jsonl; tail -2 gpurun_out/smoke_r03.log
