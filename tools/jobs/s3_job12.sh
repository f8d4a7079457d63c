cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s3
for v in old new; do
  if [ $v = new ]; then unset FROST_HIP_LIB; else export FROST_HIP_LIB=$PWD/build/ab/libfrost_$v.so; fi
  for shape in "16 96 1 1 112" "24 144 1 1 56" "24 72 1 1 56" "96 24 1 1 56" "32 16 1 1 112" "56 168 1 1 28" "168 40 1 1 28" "144 40 1 1 28"; do
    echo "== $v $shape $(python tools/bench_layer.py pw $shape 512 6 2>&1 | grep -E "pw_bwd_fused")"
  done
done > gpurun_out/s3/fuse_xb.txt 2>&1
unset FROST_HIP_LIB
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_prod.py -q -x -k "not large224" 2>&1 | tail -3 >> gpurun_out/s3/fuse_xb.txt
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>gpurun_out/s3/bench_$tag.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['value'])"; }
for rep in 1 2; do
run old FROST_HIP_LIB=$PWD/build/ab/libfrost_old.so
run new A=1
done >> gpurun_out/s3/fuse_xb.txt 2>&1
