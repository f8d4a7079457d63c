#!/bin/bash
# one replayed step as a kernel timeline (rocprofv3 --kernel-trace -> tools/replay_nodes.py)
exec < /dev/null
T=${1:-t1}; O=gpurun_out/r5trace_$T; mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && cd $OLDPWD && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $O/trace.log 2>&1 )
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python tools/replay_nodes.py "$f" --out $O/replay_nodes.txt --json $O/replay_nodes.json
[ -n "$f" ] && rm -f "$f"
head -3 $O/replay_nodes.txt | cut -c1-400
grep "k_dwb" $O/replay_nodes.txt | head -20
