#!/bin/bash
# per-launch times (HIP events, eager) of the forward statistics passes under two settings:  r6_lt.sh "<env A>" "<env B>"
exec < /dev/null
O=gpurun_out/r6lt; mkdir -p $O
i=0
for v in "$@"; do
  ( for kv in $v; do export "$kv"; done; timeout 600 python tests/devtools/layer_times.py 512 > $O/lt_$i.txt 2>$O/err_$i.txt || tail -3 $O/err_$i.txt )
  i=$((i+1))
done
python - <<'PY'
import re
def load(f):
    d={}
    for l in open(f):
        m=re.match(r"\s*(\d+)\s+(\S+)\s+(\w+)\s+([\d.]+) us",l)
        if m: d[(m.group(2),m.group(3))]=float(m.group(4))
    return d
a,b=load('gpurun_out/r6lt/lt_0.txt'),load('gpurun_out/r6lt/lt_1.txt')
ta=tb=0
for k in a:
    if k in b and abs(a[k]-b[k])>2.0 and 'stats' in k[1]:
        print(f"{k[0]:28s} {k[1]:16s} {a[k]:8.1f} {b[k]:8.1f}")
    if k in b and k[1]=='pw_fwd_stats': ta+=a[k]; tb+=b[k]
print("pw_fwd_stats total", ta, tb)
PY
