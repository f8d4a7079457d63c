#!/bin/bash
# Interleaved whole-step A/B on ONE box:  r6_ab.sh [-t "<pytest args>"] [-n reps] [-s steps] "<spec>" ["<spec>" ...]
# spec = space-separated environment assignments for one variant ("FROST_X=1 FROST_HIP_LIB=$PWD/build/var/libfrost_foo.so"); the empty spec "" (always first) is the baseline.
exec < /dev/null
reps=2; steps=30; tests=""
while getopts "t:n:s:" o; do case $o in t) tests=$OPTARG;; n) reps=$OPTARG;; s) steps=$OPTARG;; esac; done
shift $((OPTIND - 1))
O=gpurun_out/r6ab; mkdir -p $O
if [ -n "$tests" ]; then timeout 2400 python -m pytest $tests -q -x > $O/tests.log 2>&1; tail -5 $O/tests.log; fi
for rep in $(seq $reps); do
  for v in "" "$@"; do
    ( for kv in $v; do export "$kv"; done
      timeout 400 python bench.py --steps $steps --warmup 8 --no-cpu-baseline --no-roofline --no-extras 2>$O/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[ab]', '${v:-base}', d['ms_per_step'], d['value'])" || tail -3 $O/err.txt )
  done
done 2>&1 | tee -a $O/ab.txt
