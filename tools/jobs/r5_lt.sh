#!/bin/bash
exec < /dev/null
O=gpurun_out/r5lt; mkdir -p $O
timeout 900 python tests/devtools/layer_times.py 512 > $O/layer_times_b512.txt 2>&1
grep "conv2 " $O/layer_times_b512.txt | grep -v blk | head -40
