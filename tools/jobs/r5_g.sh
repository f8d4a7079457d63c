#!/bin/bash
exec < /dev/null
O=gpurun_out/r5g; mkdir -p $O
timeout 120 build/probe_gridbar > $O/gridbar.txt 2>&1; cat $O/gridbar.txt
timeout 2400 python -m pytest tests/test_gpu_model.py tests/test_gpu_block.py tests/test_gpu_prod.py tests/test_gpu_dp.py tests/test_gpu_convert.py tests/test_gpu_detect.py -q -x -s -W ignore 2>&1 | grep -v "^\[W9\|Gloo\|amdgpu.ids" | tail -60 > $O/tests.log; grep -n "tail grad\|\[dp\]\|passed\|failed\|FAILED\|Error\|converted detector" $O/tests.log | cut -c1-500
b() { ( export "$@"; timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'])" ); }
for rep in 1 2; do
b FROST_SQ_BWD_CAT=1
b FROST_SQ_BWD_CAT=0
done
timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['graph_nodes'])"
