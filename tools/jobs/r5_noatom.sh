#!/bin/bash
# ceiling of a slot / tree reduction inside k_pw: the whole step with k_pw's global statistics / S1-S2 atomics switched off (timing only, wrong results)
exec < /dev/null
O=gpurun_out/r5noatom; mkdir -p $O
b() { ( export "$@"; timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>$O/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'])" || tail -3 $O/err.txt ); }
for rep in 1 2; do
b FROST_X=base
b FROST_HIP_LIB=$PWD/build/var/libfrost_noatom.so
done 2>&1 | tee $O/ab.txt
