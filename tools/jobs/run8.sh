mkdir -p gpurun_out
python -m pytest tests/test_gpu_round3.py -m gpu -q 2>&1 | grep -E "^E  |step [01]\]|FAILED|passed|failed|resume|Error" > gpurun_out/t_r3.log
python -m pytest tests/test_gpu_prod.py -m gpu -q -s -k "every_large or prod_layer" 2>&1 | grep -E "^E  |step [01]\]|FAILED|passed|failed" > gpurun_out/t_all.log
