#!/bin/bash
# fp32-gradient mode after a change: fast vs plain entry tests, then the step at B = 512 (graph) and B = 64 (eager)
exec < /dev/null
O=gpurun_out/r5g32c; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_round5.py -q -W ignore -k "fast_forms or trains_a_step" 2>&1 | tail -3 ) | tee $O/tests.log
( export FROST_GRAD=fp32; timeout 900 python bench.py --batch 512 --steps 10 --warmup 2 --no-cpu-baseline --no-roofline 2>$O/err32.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32 B=512 graph', d['ms_per_step'], d['value'])" ) | tee $O/modes.txt
( export FROST_GRAD=fp32; timeout 900 python bench.py --batch 64 --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-roofline 2>$O/err32.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32 B=64 eager', d['ms_per_step'], d['value'])" ) | tee -a $O/modes.txt
