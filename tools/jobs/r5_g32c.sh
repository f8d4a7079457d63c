#!/bin/bash
# fp32-gradient mode after a change: fast vs plain entry tests + the mode's own parity tests, then the step at B = 512 (graph) and B = 64 (eager), and the kernel table
exec < /dev/null
O=gpurun_out/r5g32c; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_round5.py -q -W ignore -k "fast_forms or trains_a_step" 2>&1 | tail -5 ) | tee $O/tests.log
( timeout 1500 python -m pytest tests/test_gpu_round4.py -q -x -W ignore -k "fp32_gradient" 2>&1 | tail -3 ) | tee $O/tests4.log
( timeout 1500 python -m pytest tests/test_gpu_prod.py -q -x -W ignore -k "backward_block_by_block and fp32" 2>&1 | tail -3 ) | tee $O/tests_net.log
( export FROST_GRAD=fp32; timeout 900 python bench.py --batch 512 --steps 10 --warmup 2 --no-cpu-baseline --no-roofline 2>$O/err32.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32 B=512 graph', d['ms_per_step'], d['value'])" ) | tee $O/modes.txt
( export FROST_GRAD=fp32; timeout 900 python bench.py --batch 64 --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-roofline 2>$O/err32.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32 B=64 eager', d['ms_per_step'], d['value'])" ) | tee -a $O/modes.txt
( cd /tmp && export TMPDIR=/tmp && cd $OLDPWD && FROST_GRAD=fp32 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o s -- python bench.py --batch 512 --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-roofline > $O/prof.log 2>&1 )
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -16 "$f" | cut -c1-120 && cp "$f" $O/g32_b512_kernel_stats.csv
find $O/prof -name "*kernel_trace.csv" -delete 2>/dev/null
