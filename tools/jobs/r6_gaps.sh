#!/bin/bash
# where the captured step idles: kernel trace of 8 replayed steps -> the largest gaps of each (tools/step_gaps.py)
exec < /dev/null
O=$PWD/gpurun_out/r6gaps; mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && cd $OLDPWD && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o t -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-extras > $O/run.log 2>&1 )
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python tools/step_gaps.py "$f" --top 4 > $O/gaps.txt 2>&1
tail -12 $O/gaps.txt
tail -1 $O/run.log | cut -c1-200
rm -rf $O/prof
