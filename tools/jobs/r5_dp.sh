#!/bin/bash
exec < /dev/null
O=gpurun_out/r5d; mkdir -p $O
timeout 600 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2> $O/bench.err | tail -1 > $O/bench.json
python -c "import json; d=json.loads(open('$O/bench.json').read()); print('bench', d['ms_per_step'], d['value'])"
timeout 2400 python -m pytest tests/test_gpu_dp.py -x -q -s 2>&1 | tail -25 > $O/dp_tests.log; tail -25 $O/dp_tests.log
