#!/bin/bash
# dev: whole-step A/B of environment switches, interleaved:  s3_env.sh "K=V [K=V ..]" ["K=V .."] ...   (first set "A=1" = the defaults)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s3
: > gpurun_out/s3/env.txt
for rep in 1 2; do
  i=0
  for set in "$@"; do
    i=$((i+1))
    env $set timeout 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>gpurun_out/s3/env_$i.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$set', d['ms_per_step'], d['value'])" >> gpurun_out/s3/env.txt
  done
done
