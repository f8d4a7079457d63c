mkdir -p gpurun_out
for cfg in "small 64 4" "small 128 8" "large 224 8"; do
  for env in "" "FROST_WG_STREAM=0" "FROST_SR=0" "FROST_WG_STREAM=0 FROST_SR=0"; do
    echo "== $cfg [$env]"; env $env python tests/devtools/dbg_repro.py $cfg 2>&1 | grep rep
  done
done > gpurun_out/dbg_repro.log
