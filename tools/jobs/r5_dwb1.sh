#!/bin/bash
# one-sweep depthwise backward (frost_dwb.hip): parity tests, then interleaved whole-step A/B, then the per-layer table
exec < /dev/null
O=gpurun_out/r5dwb1; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_paths.py -q -x -W ignore -k "one_sweep or dw_" 2>&1 | tail -15 > $O/tests.log; tail -8 $O/tests.log | cut -c1-400
b() { ( export "$@"; timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'])" ); }
for rep in 1 2; do
b FROST_DW_BWD_ONE=0
b FROST_DW_BWD_ONE=1
b FROST_DW_BWD_ONE=1 FROST_DWB_CHUNKS=1
b FROST_DW_BWD_ONE=1 FROST_DWB_CHUNKS=4
done 2>&1 | tee $O/ab.txt
timeout 600 python tests/devtools/layer_times.py 512 2>&1 | grep -i "dw_\|per layer" | head -60 > $O/layer_dw.txt; grep "dw_bwd_one\|layer1.0.conv2\|layer1.2.conv2\|layer2.1.conv2" $O/layer_dw.txt | head -20
