#!/bin/bash
# Run ON the GPU box (via gpurun) from the repo root.  Produces under gpurun_out/prof_<tag>/:
#   stats/   rocprofv3 --kernel-trace --stats of the default bench command (hipGraph replay)
#   fetch/   PMC pass 1: FETCH_SIZE   (eager step, one kernel-trace + one counter, nothing else)
#   write/   PMC pass 2: WRITE_SIZE
# and gpurun_out/prof_<tag>/summary.json (tools/summarize_profiles.py), which is what gets copied to profiles/.
tag=${1:-r04}
B=${2:-512}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
# per-kernel numbers are taken with everything on ONE stream (the production step overlaps the pointwise weight gradients on a
# second stream, which stretches the durations of whatever runs beside them); bench.py's roofline leg measures the same way
export FROST_WG_STREAM=0
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- \
    python bench.py --batch $B --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $out/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- \
    python bench.py --batch $B --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- \
    python bench.py --batch $B --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline > $out/write.log 2>&1
# MFMA utilisation (north_star: "rocprof HBM GB/s and MFMA utilisation"): separate passes, one counter group each
for grp in "SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"; do
  d=$out/mfma_$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -o m -- \
      python bench.py --batch $B --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline > $d.log 2>&1
done
# FETCH_SIZE calibration on known byte counts (8-byte vs 16-byte per-lane loads, the depthwise staging pattern)
hipcc --offload-arch=gfx950 -O3 tools/probe_fetch.hip -o /tmp/probe_fetch 2> $out/probe_build.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/cal_fetch -o c -- /tmp/probe_fetch > $out/cal_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/cal_write -o c -- /tmp/probe_fetch > $out/cal_write.log 2>&1
python tools/summarize_profiles.py $out $B > $out/summary.log 2>&1
tail -3 $out/stats.log | cut -c1-400
tail -30 $out/summary.log
