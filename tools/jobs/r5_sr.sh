#!/bin/bash
# what stochastic rounding of dc costs in the step (FROST_SR=0: round to nearest; parity then degrades on per-channel sums, tests/test_gpu_prod.py)
exec < /dev/null
O=gpurun_out/r5sr; mkdir -p $O
b() { ( export "$@"; timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>$O/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'])" || tail -3 $O/err.txt ); }
for rep in 1 2; do
b FROST_SR=1
b FROST_SR=0
done 2>&1 | tee $O/ab.txt
