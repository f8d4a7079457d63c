set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s3
for t in 64 128 256 512; do echo "== WG_TARGET $t"; FROST_WG_TARGET=$t python tools/bench_layer.py pw 240 1440 1 1 7 512 10 2>&1 | tail -8; done > gpurun_out/s3/l41c1.txt 2>&1
for t in 128 512; do echo "== WG_TARGET $t"; FROST_WG_TARGET=$t python tools/bench_layer.py pw 1440 192 1 1 7 512 10 2>&1 | tail -8; done > gpurun_out/s3/l41red.txt 2>&1
python tools/bench_layer.py pw 104 312 1 1 14 512 10 > gpurun_out/s3/l31c1.txt 2>&1
python tools/bench_layer.py pw 312 80 1 1 14 512 10 > gpurun_out/s3/l31red.txt 2>&1
