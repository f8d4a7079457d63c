#!/bin/bash
# per-launch times (HIP events, eager step at B = 512) of one kernel family under several settings:  r6_lt.sh [-f family] "<env A>" "<env B>" ...
exec < /dev/null
fam=pw_fwd_stats
while getopts "f:" o; do case $o in f) fam=$OPTARG;; esac; done
shift $((OPTIND - 1))
O=gpurun_out/r6lt; mkdir -p $O; rm -f $O/lt_*.txt
i=0
for v in "$@"; do
  ( for kv in $v; do export "$kv"; done; timeout 600 python tests/devtools/layer_times.py 512 > $O/lt_$i.txt 2>$O/err_$i.txt || tail -3 $O/err_$i.txt )
  i=$((i+1))
done
FAM=$fam python - "$@" <<'PY'
import re, sys, os, glob
fam=os.environ["FAM"]
def load(f):
    d={}
    for l in open(f):
        m=re.match(r"\s*(\d+)\s+(\S+)\s+(\w+)\s+([\d.]+) us",l)
        if m and m.group(3)==fam: d[m.group(2)]=float(m.group(4))
    return d
tabs=[load(f) for f in sorted(glob.glob('gpurun_out/r6lt/lt_*.txt'), key=lambda s:int(re.findall(r"lt_(\d+)",s)[0]))]
print(f"# {fam}: " + " | ".join(sys.argv[1:]))
for k in tabs[0]:
    row=[t.get(k,float('nan')) for t in tabs]
    if max(row)-min(row)>1.5: print(f"{k:28s} "+" ".join(f"{v:8.1f}" for v in row))
print(f"{'total':28s} "+" ".join(f"{sum(t.values()):8.1f}" for t in tabs))
PY
