mkdir -p gpurun_out
python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py tests/test_gpu_surface.py tests/test_gpu_dp.py tests/test_gpu_detect.py -m gpu -q -x 2>&1 | grep -E "^E  |FAILED|passed|failed|Error" > gpurun_out/t_blk.log
python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/bench_fuse.json 2> gpurun_out/bench_fuse.err
FROST_BLOCK_FUSE=0 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/bench_nofuse.json 2> gpurun_out/bench_nofuse.err
