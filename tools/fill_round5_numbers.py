"""dev: copy the files tools/jobs/final_round5.sh left under gpurun_out/ into profiles/r05_* and fill the @@R5_*@@ placeholders of DESIGN.md / README.md
(a later run replaces the figures of the previous run, recorded in profiles/r05_doc_numbers.json):  python tools/fill_round5_numbers.py [--fill]"""
import csv, json, os, shutil, sys
g = "gpurun_out/"
cp = {"prof_r05/summary.json": "r05_kernels_b512.json", "bench_r05.json": "r05_bench_b512.json", "r05_kernel_stats.csv": "r05_bench_b512_kernel_stats.csv",
      "r05_side_workloads.jsonl": "r05_side_workloads.jsonl", "r05_other_batches.jsonl": "r05_other_batches.jsonl", "r05_grad_modes.jsonl": "r05_grad_modes.jsonl",
      "r05_infer_blocks.txt": "r05_infer_blocks.txt", "r05_float_b256_kernel_stats.csv": "r05_float_b256_kernel_stats.csv", "r05_g32_b512_kernel_stats.csv": "r05_g32_b512_kernel_stats.csv"}
for a, b in cp.items():
    if os.path.exists(g + a):
        shutil.copy(g + a, "profiles/" + b)
    else:
        print("missing", g + a)
if os.path.exists(g + "layer_times_r05_b512.txt"):
    open("profiles/r05_layer_times_b512.txt", "w").write("".join(l for l in open(g + "layer_times_r05_b512.txt") if "amdgpu.ids" not in l))
tail = open(g + "gpu_suite_r05.log").read().strip().splitlines()[-4:] + open(g + "smoke_r05.log").read().strip().splitlines()[-1:]
open("profiles/r05_gpu_suite_tail.txt", "w").write("\n".join(tail) + "\n")
d = json.loads(open(g + "bench_r05.json").read().strip().splitlines()[-1])
k = json.load(open("profiles/r05_kernels_b512.json"))
side = [json.loads(l) for l in open(g + "r05_side_workloads.jsonl") if l.strip()]
gm = [json.loads(l) for l in open(g + "r05_grad_modes.jsonl") if l.strip()]          # bf16 B=64, fp32 B=64, fp32 plain B=64, fp32 B=512 (captured)
fam = d["roofline"].get("families", {})
gn = d["roofline"].get("graph_nodes", {})
dwwg = 0.0
if os.path.exists("profiles/r05_g32_b512_kernel_stats.csv"):
    rows = list(csv.DictReader(open("profiles/r05_g32_b512_kernel_stats.csv")))
    steps = max(1, max(int(r["Calls"]) for r in rows if "k_g32_wq" in r["Name"]) // 69)
    dwwg = sum(float(r["TotalDurationNs"]) for r in rows if "k_g32_dw_wgrad_part" in r["Name"]) / 1e6 / steps


def ratio(name):
    v = fam.get(name, {}).get("traffic_ratio")
    return "n/a" if v is None else f"{v:.2f}"


suite = next((l for l in reversed(tail) if " passed" in l), "?").strip("= ").split(" in ")[0]
out = {"R5_IMGS": f"{d['value']:,.0f}".replace(",", " "), "R5_MS": f"{d['ms_per_step']:.2f}", "R5_FRAC": f"{d['roofline']['frac']:.3f}",
       "R5_GB": f"{k['hbm_bytes_per_step'] / 1e9:.1f}", "R5_RATIO": f"{k['traffic_ratio']:.2f}", "R5_LAUNCH": f"{k['kernel_launches_per_step']:.0f}",
       "R5_NODES": str(gn.get("nodes_total", "?")), "R5_OTHER": str(gn.get("kernels_other", 0) + gn.get("memset", 0) + gn.get("memcpy", 0) + gn.get("other_nodes", 0)),
       "R5_SUITE": suite, "R5_C2": f"{side[0]['value'] / 1e3:.1f}", "R5_INT8": f"{side[1]['value'] / 1e3:.1f}", "R5_DET": f"{side[2]['value'] / 1e3:.2f}",
       "R5_G32": f"{gm[3]['ms_per_step']:.1f}", "R5_G32X": f"{gm[3]['ms_per_step'] / d['ms_per_step']:.1f}", "R5_G32_64": f"{gm[1]['ms_per_step']:.1f}",
       "R5_G32_DWWG": f"{dwwg:.0f}", "G32NEW": f"{gm[3]['ms_per_step']:.1f}", "G32XNEW": f"{gm[3]['ms_per_step'] / d['ms_per_step']:.1f}", "G32_64NEW": f"{gm[1]['ms_per_step']:.1f}",
       "DWWGNEW": f"{dwwg:.0f}", "R5_WG_RATIO": ratio("pw_wgrad"), "R5_EXP_RATIO": ratio("blk_expand_dw"), "R5_DWS_RATIO": ratio("blk_dw_stats")}
print(json.dumps(out, indent=1))
print("side:", [(s["metric"][:40], round(s["value"])) for s in side])
print("grad modes:", [(m["config"].get("grad_dtype"), m["config"].get("per_gpu_batch"), m["ms_per_step"]) for m in gm])
if "--fill" in sys.argv:
    prev = json.load(open("profiles/r05_doc_numbers.json")) if os.path.exists("profiles/r05_doc_numbers.json") else {}
    for p in ("DESIGN.md", "README.md"):
        s = open(p).read()
        for key, v in out.items():
            s = s.replace("@@" + key + "@@", v)
        open(p, "w").write(s)
    changed = {k_: (prev.get(k_), v) for k_, v in out.items() if prev.get(k_) not in (None, v)}
    print("changed since the previous run (edit the docs where these are quoted):", json.dumps(changed))
    json.dump(out, open("profiles/r05_doc_numbers.json", "w"), indent=1)
