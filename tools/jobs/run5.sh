mkdir -p gpurun_out
python -m pytest tests/test_gpu_prod.py -m gpu -q -k "layer1.2.conv2 or layer3.1.reduce_conv or layer3.6.squeeze_conv or layer4.3.squeeze_conv or layer4.4.reduce_conv" 2>&1 | grep -E "^E  |step 0\]|FAILED|passed|failed|Mismatch|Max " > gpurun_out/t_fail.log
python -m pytest tests/test_gpu_round3.py -m gpu -q -k "ssdlite_head or force_dp or resume" 2>&1 | grep -E "^E  |step [01]\]|FAILED|passed|failed|Mismatch|Max |resume" >> gpurun_out/t_fail.log
