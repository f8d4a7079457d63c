#!/bin/bash
# k_pw: the weight sums of a channel group's tiles loaded together (were four dependent L2 round trips per group in the non-resident instances): parity subset, per-layer, interleaved A/B
exec < /dev/null
O=gpurun_out/r5wsum; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_paths.py tests/test_gpu_block.py -q -x -W ignore 2>&1 | tail -3 ) | tee $O/tests.log
b() { ( export "$@"; timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>$O/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'])" || tail -3 $O/err.txt ); }
for rep in 1 2 3; do
b FROST_HIP_LIB=$PWD/build/var/libfrost_base.so
b FROST_X=new
done 2>&1 | tee $O/ab.txt
for v in base new; do echo "lib=$v"; if [ $v = base ]; then export FROST_HIP_LIB=$PWD/build/var/libfrost_base.so; else unset FROST_HIP_LIB; fi; timeout 600 python tests/devtools/layer_times.py 512 2>&1 | grep -E "layer4.1.conv1|layer5.0.conv1|layer3.3.conv1|last_layer" | grep -E "pw_" | awk '{print $2, $3, $4}' | tr '\n' ' '; echo; done | tee $O/layers.txt
