#!/bin/bash
mkdir -p gpurun_out
( timeout 3400 python -m pytest tests -q -m gpu 2>&1 | tail -25 ) > gpurun_out/gpu_suite_r03.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke_r03.log 2>&1
tail -25 gpurun_out/gpu_suite_r03.log; tail -2 gpurun_out/smoke_r03.log
