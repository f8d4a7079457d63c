mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/b_fam.json
python - <<'PY'
import json
d=json.loads(open("gpurun_out/b_fam.json").read())
f=d["roofline"]["families"]
for k,v in f.items():
    if any(t in k for t in ("emit","add","cat","stats")): print(k, v["n"], v["us"])
PY
