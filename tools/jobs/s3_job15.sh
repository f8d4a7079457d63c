cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s3
for v in 0 1; do
  for shape in "240 1440 1 1 7" "104 312 1 1 14" "120 360 1 1 14" "104 624 1 1 14" "288 1728 1 1 7" "56 336 1 1 28"; do
    echo "== slice$v $shape"; FROST_PW_WSLICE=$v python tools/bench_layer.py pw $shape 512 10 2>&1 | grep -E "pw_(fwd|bwd)"
  done
done > gpurun_out/s3/slice.txt 2>&1
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>gpurun_out/s3/bench_$tag.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['value'])"; }
for rep in 1 2; do
run slice0 FROST_PW_WSLICE=0
run slice1 A=1
done >> gpurun_out/s3/slice.txt 2>&1
for rep in 1 2; do for v in 0 1; do FROST_PW_WSLICE=$v timeout 600 python bench.py --workload int8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('int8 slice$v', d['ms_per_step'], d['value'])"; done; done >> gpurun_out/s3/slice.txt 2>&1
