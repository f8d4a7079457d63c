#!/bin/bash
# usage (on the GPU box): tools/pmc_layer.sh <tag> <bench_layer args...>
# Two PMC passes (SQ instruction mix / SQ stall buckets) of one layer's six passes; prints per-kernel per-dispatch averages and
# the issue-rate figure  valu_per_simd_cycle = SQ_INSTS_VALU / (duration * f_clk * 1024 SIMDs).
tag=$1; shift
root=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd $root
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_$tag/p$i -o p -- python tools/bench_layer.py "$@" 1 > gpurun_out/pmc_$tag/log$i.txt 2>&1
done
python tools/pmc_summary.py gpurun_out/pmc_$tag
