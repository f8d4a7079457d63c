#!/bin/bash
# usage: tools/pmc_layer.sh <tag> <bench_layer args...>   -> prints per-kernel PMC sums
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; do
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_$tag -o p -- python tools/bench_layer.py "$@" 1 > /dev/null 2>&1
  python - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pmc_$tag/*counter_collection.csv")
rows=list(csv.DictReader(open(f[0])))
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in rows:
    agg[r["Kernel_Name"][:40]][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in agg.items():
    if "k_pw" in k or "k_dw" in k or "k_stem" in k: print(k, {a: f"{b:.3g}" for a,b in v.items()})
PY
done
