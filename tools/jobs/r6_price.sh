#!/bin/bash
# round 6, first job: today's box baseline + what single families cost inside the captured step (FROST_ABL_SKIP: timing only) + per-layer table of the fp32-gradient mode
exec < /dev/null
O=gpurun_out/r6price; mkdir -p $O
b() { ( export "$@"; timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>$O/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'])" || tail -3 $O/err.txt ); }
for rep in 1 2; do
b FROST_X=base
b FROST_ABL_SKIP=frost_pw_wgrad
b FROST_ABL_SKIP=frost_weight_prep
b FROST_ABL_SKIP=frost_pw_dgrad_wide
b FROST_ABL_SKIP=frost_block_dw_bwd
b FROST_ABL_SKIP=frost_block_dw_reduce
b FROST_ABL_SKIP=frost_dw_bwd_fused,frost_dw_bwd_fused_c1
done 2>&1 | tee $O/ab.txt
FROST_GRAD=fp32 timeout 600 python tests/devtools/layer_times.py 512 2 > $O/layer_times_g32_b512.txt 2>&1
timeout 300 python bench.py --force-dp --steps 10 --warmup 5 --no-cpu-baseline --no-roofline > $O/forcedp.json 2> $O/forcedp.err
tail -c 1500 $O/forcedp.json
tail -5 $O/layer_times_g32_b512.txt
