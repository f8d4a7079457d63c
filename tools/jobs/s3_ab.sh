#!/bin/bash
# dev: interleaved A/B of build/ab/libfrost_old.so vs the in-tree library on the default bench, after an optional pytest selection:  s3_ab.sh [pytest args...]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s3
: > gpurun_out/s3/ab.txt
if [ $# -gt 0 ]; then timeout 1500 python -m pytest "$@" -q -x 2>&1 | tail -4 >> gpurun_out/s3/ab.txt; fi
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>gpurun_out/s3/bench_$tag.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['value'])"; }
for rep in 1 2 3; do
run old FROST_HIP_LIB=$PWD/build/ab/libfrost_old.so
run new A=1
done >> gpurun_out/s3/ab.txt 2>&1
