#!/bin/bash
# kernel trace of replayed steps (rocprofv3 --kernel-trace): node list of one step with start / end on the timeline -> gpurun_out/r6trace/replay_nodes.txt
exec < /dev/null
O=$PWD/gpurun_out/r6trace; mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && cd $OLDPWD && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o t -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-extras > $O/run.log 2>&1 )
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python tools/replay_nodes.py "$f" --out $O/replay_nodes.txt --json $O/replay_nodes.json > $O/summary.txt 2>&1
head -40 $O/summary.txt
rm -rf $O/prof
