#!/bin/bash
# finalize tail with batched loads (frost_common.h conv_finalize_dev): parity subset, interleaved whole-step A/B against build/var/libfrost_base.so, then the grid-barrier probe
exec < /dev/null
O=gpurun_out/r5fin; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_block.py tests/test_gpu_round3.py tests/test_gpu_paths.py -q -x -W ignore 2>&1 | tail -8 ) > $O/tests.log; tail -3 $O/tests.log | cut -c1-300
b() { ( export "$@"; timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'])" ); }
for rep in 1 2 3; do
b FROST_HIP_LIB=$PWD/build/var/libfrost_base.so
b FROST_X=new
done 2>&1 | tee $O/ab.txt
timeout 120 build/probe_gridbar > $O/gridbar.txt 2>&1; head -12 $O/gridbar.txt
