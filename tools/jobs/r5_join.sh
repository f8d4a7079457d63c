#!/bin/bash
# round 5: how does the hipGraph executor schedule the side-stream weight gradients?  Runtime knobs, one box, bench + one replay trace each
exec < /dev/null
O=gpurun_out/r5c; mkdir -p $O
run() {   # name, env...
  name=$1; shift
  ( export "$@"; timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2> $O/$name.err | tail -1 > $O/$name.json )
  python - $O/$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d["ms_per_step"], d["value"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  ( export "$@"; cd /tmp && export TMPDIR=/tmp && cd $OLDPWD && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr_$name -o t -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $O/tr_$name.log 2>&1 )
  f=$(find $O/tr_$name -name "*kernel_trace.csv" | head -1)
  if [ -n "$f" ]; then python tools/replay_nodes.py "$f" --out $O/nodes_$name.txt 2> /dev/null; rm -f "$f"; grep -m1 "k_pw_wgrad" $O/nodes_$name.txt | awk -v n=$name '{print n, "first wgrad starts at", $1, "us"}'; head -1 $O/nodes_$name.txt; fi
}
run default FROST_DUMMY=1
run join1 FROST_WG_JOIN=1
run join3 FROST_WG_JOIN=3
run join6 FROST_WG_JOIN=6
run join12 FROST_WG_JOIN=12
