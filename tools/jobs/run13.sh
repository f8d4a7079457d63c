mkdir -p gpurun_out
python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py tests/test_gpu_dp.py -m gpu -q -x 2>&1 | grep -E "^E  |FAILED|passed|failed|Error" > gpurun_out/t_blk.log
for i in 1 2; do
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > gpurun_out/b_two_fuse_$i.json
FROST_BLOCK_FUSE=0 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > gpurun_out/b_two_nofuse_$i.json
FROST_HIP_LIB=build/libfrost_single.so python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > gpurun_out/b_single_fuse_$i.json
FROST_HIP_LIB=build/libfrost_single.so FROST_BLOCK_FUSE=0 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > gpurun_out/b_single_nofuse_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/b_*.json")):
    d=json.loads(open(f).read()); print(f, d["value"], d["ms_per_step"], d["config"]["ms_per_step_median_hip_events"])
PY
