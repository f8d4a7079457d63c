#!/bin/bash
# Float (StatAssist warm-up) path: [tests] + bench with optional env assignments + kernel profile: float_ab.sh [K=V ...]
exec < /dev/null
mkdir -p gpurun_out/float_ab
O=gpurun_out/float_ab
timeout 900 python -m pytest tests/test_gpu_float.py -q -x > $O/tests.log 2>&1
tail -2 $O/tests.log
for v in "" "$@"; do
  for rep in 1 2; do
    ( [ -n "$v" ] && export $v; timeout 600 python bench.py --workload float 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('float', '$v', d['ms_per_step'], d['value'])" )
  done
done
bash tools/jobs/float_prof.sh cur > $O/prof.log 2>&1
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/float_prof_cur/kernel_stats.csv')):
    t = float(r['TotalDurationNs']) / 16 / 1e6
    if t > 0.25: print(r['Name'].split('(')[0][:70], r['Calls'], round(t, 3), 'avg us', round(float(r['AverageNs']) / 1e3, 1), 'max', round(float(r['MaxNs']) / 1e3))
PY
