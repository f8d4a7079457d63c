mkdir -p gpurun_out
python -m pytest tests/test_gpu_round3.py -m gpu -q -s -k "hswish" 2>&1 | grep -E "^E  |hswish block|FAILED|passed|failed|Error" > gpurun_out/t_hsw.log
