#!/bin/bash
exec < /dev/null
O=gpurun_out/r5mm; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_paths.py -q -x -W ignore -k "one_sweep" 2>&1 | tail -5 > $O/tests.log; tail -3 $O/tests.log | cut -c1-300
b() { ( export "$@"; timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'])" ); }
for rep in 1 2; do
b FROST_X=0
b FROST_PW_FUSE_MINMAP=0
b FROST_PW_FUSE_MINMAP=100
b FROST_PW_FUSE_MINMAP=1000
b FROST_SQ_BWD_CAT=1
b FROST_BLOCK_DWBRED=0
b FROST_PWC_RED_MAXPIX=0
done 2>&1 | tee $O/ab.txt
