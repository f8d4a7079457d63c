#!/bin/bash
# persistent squeeze forward, slot-tree version: interleaved whole-step A/B
exec < /dev/null
O=gpurun_out/r5sq2; mkdir -p $O
b() { ( export "$@"; timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>$O/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'])" || tail -3 $O/err.txt ); }
for rep in 1 2 3; do
b FROST_SQ_PERSIST=0
b FROST_SQ_PERSIST=1
done 2>&1 | tee $O/ab.txt
