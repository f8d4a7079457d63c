cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s3
for wl in infer int8; do for rep in 1 2; do timeout 600 python bench.py --workload $wl 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$wl', d['ms_per_step'], d['value'])"; done; done > gpurun_out/s3/infer.txt 2>&1
