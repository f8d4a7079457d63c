#!/bin/bash
# bench.py (headline workload, 30 timed steps) under each of the given environment assignments (and none), twice each, interleaved: qat_env.sh K=V [K=V ...]
exec < /dev/null
for rep in 1 2; do
  for v in "" "$@"; do
    ( [ -n "$v" ] && export $v; timeout 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('qat', '$v', d['ms_per_step'], d['value'])" )
  done
done
