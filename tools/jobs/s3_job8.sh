cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s3
for v in pf11 wa1 wa2; do
  export FROST_HIP_LIB=$PWD/build/ab/libfrost_$v.so
  for t in 128 256; do
  for shape in "240 1440 1 1 7" "1440 192 1 1 7" "104 312 1 1 14"; do
    echo "== $v T$t $shape $(FROST_WG_TARGET=$t python tools/bench_layer.py pw $shape 512 10 2>&1 | grep -E "pw_wgrad")"
  done; done
done > gpurun_out/s3/wg_abl.txt 2>&1
