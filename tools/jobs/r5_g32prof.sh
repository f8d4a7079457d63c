#!/bin/bash
# kernel table of the fp32-gradient mode at B = 512 (eager step)
exec < /dev/null
O=gpurun_out/r5g32p; mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && cd $OLDPWD && FROST_GRAD=fp32 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o s -- python bench.py --batch 512 --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-roofline > $O/prof.log 2>&1 )
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -32 "$f" | cut -c1-140 && cp "$f" $O/g32_b512_kernel_stats.csv
find $O/prof -name "*kernel_trace.csv" -delete 2>/dev/null
