#!/bin/bash
# persistent squeeze_conv forward (frost_sq_fwd): its tests + the block / model tests that now run through it, interleaved whole-step A/B, then the fp32-gradient mode again
exec < /dev/null
O=gpurun_out/r5sq; mkdir -p $O
( timeout 1200 python -m pytest tests/test_gpu_round5.py -q -W ignore -k "squeeze" 2>&1 | tail -12 ) > $O/tests_sq.log; tail -6 $O/tests_sq.log | cut -c1-300
( timeout 1500 python -m pytest tests/test_gpu_round4.py tests/test_gpu_block.py tests/test_gpu_model.py -q -x -W ignore -k "squeeze or block or g4" 2>&1 | tail -6 ) > $O/tests_blk.log; tail -3 $O/tests_blk.log | cut -c1-300
b() { ( export "$@"; timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>$O/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'])" || tail -3 $O/err.txt ); }
for rep in 1 2 3; do
b FROST_SQ_PERSIST=0
b FROST_SQ_PERSIST=1
done 2>&1 | tee $O/ab.txt
B=512; ( export FROST_GRAD=fp32; timeout 900 python bench.py --batch 512 --steps 10 --warmup 2 --no-cpu-baseline --no-roofline 2>$O/err32.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32 B=512', d['ms_per_step'], d['value'])" ) | tee $O/modes.txt
( timeout 600 python -m pytest tests/test_gpu_round5.py -q -W ignore -k "fast_forms" 2>&1 | tail -3 ) | tee $O/tests_g32.log
timeout 600 python tests/devtools/layer_times.py 512 2>&1 | grep -E "squeeze_conv" | head -20 | cut -c1-200 | tee $O/layer_sq.txt
