#!/bin/bash
mkdir -p gpurun_out
FROST_PW_FUSE=0 FROST_PWC=64 FROST_PWC_RED_MAXPIX=0 timeout 900 python tests/devtools/layer_times.py 512 > gpurun_out/lt_nofuse.txt 2>&1
tail -1 gpurun_out/lt_nofuse.txt
