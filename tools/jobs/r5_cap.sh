#!/bin/bash
# the statistics passes' workgroup cap (FROST_PW_STATS_CAP: fewer workgroups flush per-channel atomics, the channel-group split fills the chip instead): interleaved whole-step A/B
exec < /dev/null
O=gpurun_out/r5cap; mkdir -p $O
b() { ( export "$@"; timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>$O/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'])" || tail -3 $O/err.txt ); }
for rep in 1 2; do
b FROST_PW_STATS_CAP=0
b FROST_PW_STATS_CAP=128
b FROST_PW_STATS_CAP=256
b FROST_PW_STATS_CAP=512
done 2>&1 | tee $O/ab.txt
