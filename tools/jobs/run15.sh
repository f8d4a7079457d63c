mkdir -p gpurun_out
python -m pytest tests/test_gpu_round3.py -m gpu -q -k "fused_reduce" 2>&1 | grep -E "^E  |FAILED|passed|failed|Error" > gpurun_out/t_ea.log
FROST_BLOCK_EMIT_ADD=1 python -m pytest tests/test_gpu_model.py -m gpu -q -x 2>&1 | grep -E "^E  |FAILED|passed|failed|Error" >> gpurun_out/t_ea.log
