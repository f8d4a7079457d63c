#!/bin/bash
# Interleaved A/B of TWO TREES on one box: the working tree against build/old_tree (a `git worktree` of the commit to compare with, built locally -- library AND Python side
# differ, e.g. an ABI change):  r6_ab_tree.sh [reps]
exec < /dev/null
reps=${1:-3}
O=$PWD/gpurun_out/r6abtree; mkdir -p $O
for rep in $(seq $reps); do
  for v in new old; do
    ( [ $v = old ] && cd build/old_tree
      timeout 400 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extras 2>$O/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[ab]', '$v', d['ms_per_step'], d['value'])" || tail -3 $O/err.txt )
  done
done 2>&1 | tee $O/ab.txt
