cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s3
for v in base; do
  for t in 96 128 192 256; do
  for shape in "240 1440 1 1 7" "1440 192 1 1 7" "104 312 1 1 14" "312 80 1 1 14"; do
    echo "== $v T$t $shape $(FROST_WG_TARGET=$t python tools/bench_layer.py pw $shape 512 10 2>&1 | grep -E "pw_wgrad")"
  done; done
done > gpurun_out/s3/wg_bf.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_prod.py -q -x -k "not large224" 2>&1 | tail -3 >> gpurun_out/s3/wg_bf.txt
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>gpurun_out/s3/bench_$tag.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['value'])"; }
for rep in 1 2; do
run pf11 FROST_HIP_LIB=$PWD/build/ab/libfrost_pf11.so
run new A=1
run new_t96 FROST_WG_TARGET=96
run new_t192 FROST_WG_TARGET=192
run new_t64 FROST_WG_TARGET=64
done >> gpurun_out/s3/wg_bf.txt 2>&1
