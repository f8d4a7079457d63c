cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s3
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>gpurun_out/s3/bench_$tag.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['value'])"; }
for rep in 1 2; do
run new A=1
run noblk FROST_BLK_XCD=0
run nodgw FROST_DGW_XCD=0
run none FROST_BLK_XCD=0 FROST_DGW_XCD=0
done > gpurun_out/s3/bench_xcd.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_block.py tests/test_gpu_paths.py -q -x 2>&1 | tail -3 >> gpurun_out/s3/bench_xcd.txt
