cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s3
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>gpurun_out/s3/bench_$tag.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['value'])"; }
run base A=1
for f in frost_pw_wgrad frost_pw_wgrad,frost_dw_wgrad frost_block_dw_bwd frost_pw_dgrad_wide frost_pw_conv_fwd_fin frost_block_dw_bwd_reduce frost_pw_conv_bwd_fused frost_dw_dgrad frost_block_dw_reduce frost_block_expand_dw_stats frost_cat_bwd frost_pw_conv_bwd; do
run skip_$(echo $f | tr ',' '+') FROST_ABL_SKIP=$f
done > gpurun_out/s3/bench_skip.txt 2>&1
run base2 A=1 >> gpurun_out/s3/bench_skip.txt 2>&1
