mkdir -p gpurun_out
python -m pytest tests/test_gpu_round3.py tests/test_gpu_surface.py tests/test_gpu_convert.py -m gpu -q -x 2>&1 | grep -E "^E  |FAILED|passed|failed|Error" > gpurun_out/t_surf.log
