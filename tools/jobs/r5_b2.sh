#!/bin/bash
# two-chunk eight-wave blk_dw_reduce for 7 x 7 maps (k_blk_dw_reduce2): block tests, per-layer times with / without, interleaved whole-step A/B
exec < /dev/null
O=gpurun_out/r5b2; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_block.py tests/test_gpu_model.py -q -x -W ignore -k "block or g4" 2>&1 | tail -4 ) | tee $O/tests.log
for v in 0 1; do echo "FROST_BLK_B2=$v"; FROST_BLK_B2=$v timeout 600 python tests/devtools/layer_times.py 512 2>&1 | grep -E "blk_dw_reduce" | awk '{print $2, $4}' | tr '\n' ' '; echo; done | tee $O/layers.txt
b() { ( export "$@"; timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>$O/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'])" || tail -3 $O/err.txt ); }
for rep in 1 2 3; do
b FROST_BLK_B2=0
b FROST_BLK_B2=1
done 2>&1 | tee $O/ab.txt
( timeout 600 python -m pytest tests/test_gpu_dp.py -q -W ignore 2>&1 | tail -2 ) | tee $O/dp.log
( timeout 600 python -m pytest tests/test_gpu_round5.py -q -W ignore -k "fast_forms" 2>&1 | tail -2 ) | tee $O/g32.log
