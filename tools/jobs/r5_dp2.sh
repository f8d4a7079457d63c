#!/bin/bash
exec < /dev/null
O=gpurun_out/r5d; mkdir -p $O
timeout 1500 python bench.py --gpus 8 --share-gpu --batch 8 --steps 3 --warmup 2 --check-allreduce --no-roofline --no-cpu-baseline > $O/r8.out 2> $O/r8.err
echo rc=$?; tail -c 1500 $O/r8.out; grep -v Warning $O/r8.err | tail -40 | cut -c1-300
