#!/bin/bash
# where the time of frost_sq_fwd goes: the kernel with nobody waiting at the barrier (FROST_SQF_DBG=1) and without fold / finalize (=3): timing only
exec < /dev/null
O=gpurun_out/r5sqdbg; mkdir -p $O
for d in 0 1; do echo "FROST_SQF_DBG=$d"; FROST_SQF_DBG=$d timeout 600 python tests/devtools/layer_times.py 512 2>&1 | grep -E "sq_fwd" | awk '{print $2, $4}' | tr '\n' ' '; echo; done | tee $O/split.txt
