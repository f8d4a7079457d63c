#!/bin/bash
# round 5, first job: baseline on today's box + the node list of one replayed step + which call sites issue aten fills / copies
exec < /dev/null
O=gpurun_out/r5a; mkdir -p $O
timeout 600 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -c 400 $O/bench.json | head -c 400; echo
( cd /tmp && export TMPDIR=/tmp && cd $OLDPWD && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $O/trace.log 2>&1 )
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python tools/replay_nodes.py "$f" --out $O/replay_nodes.txt --json $O/replay_nodes.json
[ -n "$f" ] && rm -f "$f"
timeout 600 python tools/find_fills.py > $O/fills.txt 2>&1
head -45 $O/fills.txt
