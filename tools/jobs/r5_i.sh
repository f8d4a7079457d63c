#!/bin/bash
exec < /dev/null
O=gpurun_out/r5i; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_detect.py tests/test_gpu_dp.py::test_detector_segmented_step_equals_module_surface -q -x -s -W ignore 2>&1 | tail -40 > $O/tests.log; tail -25 $O/tests.log | cut -c1-500
for v in 1 0; do FROST_MBOX_HIP=$v timeout 600 python bench.py --workload detect --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('detect mbox_hip=$v', d['ms_per_step'], d['value'], d['config'].get('hip_graph'))"; done
for v in 1 0; do FROST_MBOX_HIP=$v timeout 600 python bench.py --workload detect --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('detect mbox_hip=$v', d['ms_per_step'], d['value'], d['config'].get('hip_graph'))"; done
