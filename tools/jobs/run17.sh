mkdir -p gpurun_out/pmc_l41 gpurun_out/pmc_d41
bash tools/pmc_layer.sh l41 pw 240 1440 1 1 7 512 > gpurun_out/pmc_l41.txt 2>&1
bash tools/pmc_layer.sh d41 dw 1440 1440 5 1 7 512 > gpurun_out/pmc_d41.txt 2>&1
