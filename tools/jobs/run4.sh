mkdir -p gpurun_out
python -m pytest tests/test_gpu_round3.py -m gpu -q -x --durations=15 2>&1 | tail -40 > gpurun_out/t_r3.log
python -m pytest tests/test_gpu_prod.py -m gpu -q --durations=10 -k "every_large" 2>&1 | tail -40 > gpurun_out/t_all.log
