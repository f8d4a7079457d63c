#!/bin/bash
exec < /dev/null
O=gpurun_out/r5h; mkdir -p $O
b() { ( export "$@"; timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'])" ); }
for rep in 1 2; do
b FROST_X=0
b FROST_WG_BATCH=2
b FROST_WG_BATCH=3
b FROST_WG_BATCH=6
b FROST_WSUM_WGS=512
b FROST_WSUM_WGS=32
done
( cd /tmp && export TMPDIR=/tmp && cd $OLDPWD && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $O/trace.log 2>&1 )
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python tools/replay_nodes.py "$f" --out $O/replay_nodes.txt --json $O/replay_nodes.json
[ -n "$f" ] && rm -f "$f"
head -3 $O/replay_nodes.txt | cut -c1-400
timeout 900 python -m pytest tests/test_gpu_dp.py::test_two_rank_allreduce_equals_mean_of_shard_gradients tests/test_gpu_round4.py -q -x -s -W ignore 2>&1 | grep -v "^\[W9\|Gloo\|amdgpu.ids" | tail -12 | cut -c1-700
