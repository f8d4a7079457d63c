#!/bin/bash
# round 5: how does the hipGraph executor schedule the side-stream weight gradients?  Runtime knobs, one box, bench + one replay trace each
exec < /dev/null
O=gpurun_out/r5b; mkdir -p $O
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
run nocapture DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run batch1 DEBUG_HIP_GRAPH_BATCH_SIZE=1
run batch16 DEBUG_HIP_GRAPH_BATCH_SIZE=16
run queues2 DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run queues8 DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run noside FROST_WG_STREAM=0
