mkdir -p gpurun_out
python -m pytest tests/test_gpu_convert.py tests/test_gpu_fbgemm.py -m gpu -q 2>&1 | grep -E "^E  |FAILED|passed|failed|Error" > gpurun_out/t_cv.log
