#!/bin/bash
# the full GPU suite with its complete log kept (gpurun_out/gpu_suite_r05_full.log) + smoke
exec < /dev/null
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/gpu_suite_r05_full.log 2>&1
tail -8 gpurun_out/gpu_suite_r05_full.log > gpurun_out/gpu_suite_r05.log
grep -n -A60 "^=* FAILURES\|^____" gpurun_out/gpu_suite_r05_full.log | head -300 > gpurun_out/gpu_suite_r05_failures.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke_r05.log 2>&1
tail -3 gpurun_out/gpu_suite_r05.log; tail -1 gpurun_out/smoke_r05.log
