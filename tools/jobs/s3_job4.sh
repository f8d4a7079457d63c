cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s3
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>gpurun_out/s3/bench_$tag.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['value'])"; }
for rep in 1 2; do
run base A=1
run res1 FROST_PW_RES_MINTILES=1
run fuse0 FROST_PW_FUSE_MINMAP=0 FROST_PW_RES_MINTILES=1
run fuse100 FROST_PW_FUSE_MINMAP=100 FROST_PW_RES_MINTILES=1
done > gpurun_out/s3/bench_env.txt 2>&1
