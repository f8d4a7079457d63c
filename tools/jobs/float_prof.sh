#!/bin/bash
# Kernel-time breakdown of the float (StatAssist warm-up) training step: [tag] [env assignments...]
exec < /dev/null
tag=${1:-default}; shift
root=$(pwd)
O=$root/gpurun_out/float_prof_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $root
for kv in "$@"; do export "$kv"; done
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python bench.py --workload float --steps 10 --warmup 3 > $O/stats.log 2>&1
f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $O/kernel_stats.csv; head -16 "$f" | cut -c1-60,150-260; fi
find $O/stats -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $O/stats.log | head -2
