cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_wg
export FROST_HIP_LIB=$PWD/build/ab/libfrost_pf11.so
bash tools/pmc_layer.sh wg pw 240 1440 1 1 7 512 > gpurun_out/pmc_wg/summary.txt 2>&1
find gpurun_out/pmc_wg -name "*.csv" -size +5M -delete
