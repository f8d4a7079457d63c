cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s3
for v in base nogatom nofin noepi; do
  if [ $v = base ]; then unset FROST_HIP_LIB; else export FROST_HIP_LIB=$PWD/build/ab/libfrost_$v.so; fi
  for shape in "104 312 1 1 14" "240 1440 1 1 7" "80 24 1 1 14" "192 48 1 1 7" "1440 192 1 1 7" "16 96 1 1 112"; do
    echo "== $v $shape"; python tools/bench_layer.py pw $shape 512 10 2>&1 | grep -E "fwd_stats|bwd_reduce"
  done
done > gpurun_out/s3/abl.txt 2>&1
