#!/bin/bash
# refresh profiles/r05_grad_modes.jsonl: bf16 / fp32 (fast forms) / fp32 (plain round-4 kernels) at B = 64 eager, fp32 at B = 512 captured, bf16 at B = 512 captured
exec < /dev/null
: > gpurun_out/r05_grad_modes.jsonl
timeout 900 python bench.py --batch 64 --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 >> gpurun_out/r05_grad_modes.jsonl
FROST_GRAD=fp32 timeout 900 python bench.py --batch 64 --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 >> gpurun_out/r05_grad_modes.jsonl
FROST_GRAD=fp32 FROST_G32_PLAIN=1 timeout 1500 python bench.py --batch 64 --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 >> gpurun_out/r05_grad_modes.jsonl
FROST_GRAD=fp32 timeout 900 python bench.py --batch 512 --steps 10 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 >> gpurun_out/r05_grad_modes.jsonl
timeout 900 python bench.py --batch 512 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 >> gpurun_out/r05_grad_modes.jsonl
python -c "
import json
for l in open('gpurun_out/r05_grad_modes.jsonl'):
    d=json.loads(l); print(d['config'].get('grad_dtype'), d['config'].get('per_gpu_batch'), d['config'].get('hip_graph'), d['ms_per_step'])"
