cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s3
for v in "2048 40" "1 40" "1 64"; do
  set -- $v; export FROST_PW_RES_MINTILES=$1 FROST_PW_RES_MAXKB=$2
  for shape in "104 312 1 1 14" "120 360 1 1 14" "80 24 1 1 14" "96 24 1 1 14" "192 48 1 1 7" "192 96 1 1 7" "40 16 1 1 28"; do
    echo "== $v : $shape"; python tools/bench_layer.py pw $shape 512 10 2>&1 | grep -E "pw_"
  done
done > gpurun_out/s3/res.txt 2>&1
