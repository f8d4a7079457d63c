#!/bin/bash
exec < /dev/null
O=gpurun_out/r5e; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_dp.py tests/test_gpu_convert.py -q -s -x --deselect tests/test_gpu_dp.py::test_eight_rank_rehearsal_detector_on_one_gpu -W ignore 2>&1 | tail -80 > $O/tests.log; tail -50 $O/tests.log | cut -c1-400
timeout 1500 python bench.py --workload detect --gpus 8 --share-gpu --batch 4 --res 256 --steps 3 --warmup 2 --check-allreduce --no-roofline --no-cpu-baseline > $O/d8.out 2> $O/d8.err
echo rc=$?; tail -c 600 $O/d8.out; grep -v "Warning\|amdgpu.ids\|socket.cpp\|Gloo" $O/d8.err | tail -30 | cut -c1-300
for v in 1 0; do FROST_WG_DEFER=$v timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('defer=$v', d['ms_per_step'], d['value'])"; done
for v in 1 0; do FROST_WG_DEFER=$v timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('defer=$v', d['ms_per_step'], d['value'])"; done
