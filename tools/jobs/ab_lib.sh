#!/bin/bash
# A/B of two library builds on ONE box: build/ab/libfrost_old.so (reference build) vs the in-tree library, interleaved; optional parity subset first.
exec < /dev/null
mkdir -p gpurun_out/ab
O=gpurun_out/ab
if [ "$1" = "tests" ]; then
  timeout 1800 python -m pytest tests/test_gpu_paths.py tests/test_gpu_block.py tests/test_gpu_prod.py -q -x > $O/tests.log 2>&1
  tail -3 $O/tests.log
fi
for rep in 1 2 3; do
  for v in new old; do
    if [ $v = old ]; then export FROST_HIP_LIB=$PWD/build/ab/libfrost_old.so; else unset FROST_HIP_LIB; fi
    timeout 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_$v.$rep.json 2> $O/bench_$v.$rep.err
    [ -s $O/bench_$v.$rep.json ] && python - $O/bench_$v.$rep.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d["ms_per_step"], d["value"])
PY
  done
done
