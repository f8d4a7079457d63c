#!/bin/bash
exec < /dev/null
O=gpurun_out/r5f; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_model.py tests/test_gpu_block.py tests/test_gpu_prod.py tests/test_gpu_dp.py -q -x -W ignore 2>&1 | tail -30 > $O/tests.log; tail -15 $O/tests.log | cut -c1-400
b() { ( export "$@"; timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'])" ); }
for rep in 1 2; do
b FROST_ADD_BWD_FUSE=1
b FROST_ADD_BWD_FUSE=0
b FROST_WG_DEFER=28
b FROST_WG_DEFER=56
b FROST_WG_DEFER=112
done
