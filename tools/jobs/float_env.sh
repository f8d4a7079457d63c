#!/bin/bash
# bench.py --workload float under each of the given environment assignments (and none), twice each, interleaved: float_env.sh K=V [K=V ...]
exec < /dev/null
for rep in 1 2; do
  for v in "" "$@"; do
    ( [ -n "$v" ] && export $v; timeout 600 python bench.py --workload float 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('float', '$v', d['ms_per_step'], d['value'])" )
  done
done
