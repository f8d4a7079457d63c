#!/bin/bash
# fast fp32-gradient mode (csrc/frost_g32.hip "fast forms"): fast vs plain entry by entry, the mode's parity tests, plain vs fast at B = 64, the fast mode at B = 512 beside the bf16 step
exec < /dev/null
O=gpurun_out/r5g32; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_round5.py -q -W ignore -s 2>&1 | grep -E "g32 fast|passed|failed|Error|error|assert" | tail -40 ) > $O/tests_r5.log; tail -18 $O/tests_r5.log | cut -c1-300
( timeout 1500 python -m pytest tests/test_gpu_round4.py -q -x -W ignore -s -k "fp32_gradient" 2>&1 | grep -E "fp32-grad|passed|failed|Error|error|assert" | tail -40 ) > $O/tests_layer.log; tail -4 $O/tests_layer.log | cut -c1-300
( timeout 1500 python -m pytest tests/test_gpu_prod.py -q -x -W ignore -k "backward_block_by_block and fp32" 2>&1 | tail -6 ) > $O/tests_net.log; tail -3 $O/tests_net.log | cut -c1-300
b() { ( export "$@"; timeout 900 python bench.py --batch $B --steps $S --warmup 2 $G --no-cpu-baseline --no-roofline 2>$O/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$* B=$B $G', d['ms_per_step'], d['value'], d['config'].get('grad_dtype'))" || tail -3 $O/err.txt ); }
: > $O/modes.txt
B=64 S=5 G=--no-graph; b FROST_GRAD=bf16 >> $O/modes.txt; b FROST_GRAD=fp32 >> $O/modes.txt; b FROST_GRAD=fp32 FROST_G32_PLAIN=1 >> $O/modes.txt
B=512 S=5 G=--no-graph; b FROST_GRAD=bf16 >> $O/modes.txt; b FROST_GRAD=fp32 >> $O/modes.txt
B=512 S=10 G=; b FROST_GRAD=fp32 >> $O/modes.txt
cat $O/modes.txt
( cd /tmp && export TMPDIR=/tmp && cd $OLDPWD && FROST_GRAD=fp32 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o s -- python bench.py --batch 512 --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-roofline > $O/prof.log 2>&1 )
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f" | cut -c1-160 && cp "$f" $O/g32_b512_kernel_stats.csv
find $O/prof -name "*kernel_trace.csv" -delete 2>/dev/null
