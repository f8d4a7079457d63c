"""dev: copy the files tools/jobs/final_round.sh left under gpurun_out/ into profiles/r04_* and print the figures the docs quote."""
import json, shutil, subprocess, sys
g = "gpurun_out/"
cp = {"prof_r04/summary.json": "r04_kernels_b512.json", "bench_r04.json": "r04_bench_b512.json", "layer_times_r04_b512.txt": "r04_layer_times_b512.txt",
      "r04_kernel_stats.csv": "r04_bench_b512_kernel_stats.csv", "r04_side_workloads.jsonl": "r04_side_workloads.jsonl", "r04_other_batches.jsonl": "r04_other_batches.jsonl",
      "r04_grad_modes.jsonl": "r04_grad_modes.jsonl", "r04_infer_blocks.txt": "r04_infer_blocks.txt", "r04_float_b256_kernel_stats.csv": "r04_float_b256_kernel_stats.csv"}
for a, b in cp.items():
    shutil.copy(g + a, "profiles/" + b)
tail = open(g + "gpu_suite_r04.log").read().strip().splitlines()[-4:] + open(g + "smoke_r04.log").read().strip().splitlines()[-1:]
open("profiles/r04_gpu_suite_tail.txt", "w").write("\n".join(tail) + "\n")
d = json.loads(open(g + "bench_r04.json").read().strip().splitlines()[-1])
k = json.load(open("profiles/r04_kernels_b512.json"))
side = [json.loads(l) for l in open(g + "r04_side_workloads.jsonl")]
ob = [json.loads(l) for l in open(g + "r04_other_batches.jsonl")]
out = {"VAL": f"{d['value']:,.0f}".replace(",", " "), "VALK": f"{d['value'] / 1e3:.2f}", "MS": f"{d['ms_per_step']:.2f}", "FRAC": f"{d['roofline']['frac']:.3f}",
       "GB": f"{k['hbm_bytes_per_step'] / 1e9:.1f}", "RATIO": f"{k['traffic_ratio']:.2f}", "SUITE": tail[-2].strip("= ").split(" in ")[0] if len(tail) > 1 else "?",
       "C2": f"{side[0]['value'] / 1e3:.1f}", "INT8": f"{side[1]['value'] / 1e3:.1f}", "DET": f"{side[2]['value'] / 1e3:.2f}", "FL": f"{side[3]['value'] / 1e3:.1f}",
       "FL32": f"{side[4]['value'] / 1e3:.1f}", "OB": " / ".join(f"{o['value'] / 1e3:.1f}" for o in ob)}
print(json.dumps(out, indent=1))
if "--fill" in sys.argv:
    # first run: the docs carry @@KEY@@ placeholders; later runs: the figures of the previous run (profiles/r04_doc_numbers.json) are replaced by the new ones
    import os
    prev = json.load(open("profiles/r04_doc_numbers.json")) if os.path.exists("profiles/r04_doc_numbers.json") else {}
    for p in ("DESIGN.md", "README.md"):
        s = open(p).read()
        for key, v in out.items():
            s = s.replace("@@" + key + "@@", v)
        open(p, "w").write(s)
    changed = {k: (prev.get(k), v) for k, v in out.items() if prev.get(k) not in (None, v)}
    print("changed since the previous run (edit the docs where these are quoted):", json.dumps(changed))
    json.dump(out, open("profiles/r04_doc_numbers.json", "w"), indent=1)
