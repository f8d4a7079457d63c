cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s3
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd $R
for wl in int8 infer; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/s3/prof_$wl -o s -- python bench.py --workload $wl --steps 20 --warmup 5 > gpurun_out/s3/prof_$wl.log 2>&1
f=$(find gpurun_out/s3/prof_$wl -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/s3/${wl}_kernel_stats.csv
find gpurun_out/s3/prof_$wl -name "*kernel_trace.csv" -delete
done
