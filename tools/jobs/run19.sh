#!/bin/bash
# partial-resident weights A/B on the wide few-tile layers
mkdir -p gpurun_out
L='"240,1440,7" "288,1728,7" "104,624,14" "160,960,14" "192,1152,7" "60,360,14" "1728,1280,7"'
for v in 0 1; do
  echo "== FROST_PW_PRES=$v" >> gpurun_out/pres.log
  eval FROST_PW_PRES=$v timeout 300 python tests/devtools/pw_micro.py $L --n 512 >> gpurun_out/pres.log 2>&1
done
FROST_PW_PRES_KB=96 timeout 300 python tests/devtools/pw_micro.py "240,1440,7" "288,1728,7" "104,624,14" --n 512 >> gpurun_out/pres.log 2>&1
timeout 600 python -m pytest tests/test_gpu_paths.py -x -q -k "1440 or 1728 or 624" >> gpurun_out/pres.log 2>&1
tail -150 gpurun_out/pres.log
