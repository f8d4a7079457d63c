#!/bin/bash
# k_blk_dw_reduce: the next chunk's coefficient row no longer drains the prefetch (select deferred to the take-over): block tests, per-layer times, interleaved whole-step A/B
exec < /dev/null
O=gpurun_out/r5rowfix; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_block.py tests/test_gpu_model.py tests/test_gpu_paths.py -q -x -W ignore -k "block or g4 or two_chunk" 2>&1 | tail -3 ) | tee $O/tests.log
for v in base new; do echo "lib=$v"; if [ $v = base ]; then export FROST_HIP_LIB=$PWD/build/var/libfrost_base.so; else unset FROST_HIP_LIB; fi; timeout 600 python tests/devtools/layer_times.py 512 2>&1 | grep -E "blk_dw_reduce" | awk '{print $2, $4}' | tr '\n' ' '; echo; done | tee $O/layers.txt
unset FROST_HIP_LIB
b() { ( export "$@"; timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>$O/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'])" || tail -3 $O/err.txt ); }
for rep in 1 2 3; do
b FROST_HIP_LIB=$PWD/build/var/libfrost_base.so
b FROST_X=new
done 2>&1 | tee $O/ab.txt
