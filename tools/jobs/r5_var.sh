#!/bin/bash
# interleaved whole-step A/B of library variants under build/var (FROST_HIP_LIB selects)
exec < /dev/null
O=gpurun_out/r5var; mkdir -p $O
b() { ( export "$@"; timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'])" ); }
for rep in 1 2; do
b FROST_X=0
for v in $(ls build/var/*.so); do b FROST_HIP_LIB=$PWD/$v; done
b FROST_DWB_OCC=2
b FROST_DWB_CHUNKS=1
b FROST_DWB_CHUNKS=8
done 2>&1 | tee $O/ab.txt
