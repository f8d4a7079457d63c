"""dev: copy the files tools/jobs/final_round6.sh left under gpurun_out/ into profiles/r06_* and print the figures DESIGN.md / README.md quote
(profiles/r06_doc_numbers.json keeps them):  python tools/fill_round6_numbers.py"""
import json, os, shutil
g = "gpurun_out/"
cp = {"prof_r06/summary.json": "r06_kernels_b512.json", "bench_r06.json": "r06_bench_b512.json", "r06_kernel_stats.csv": "r06_bench_b512_kernel_stats.csv",
      "r06_side_workloads.jsonl": "r06_side_workloads.jsonl", "r06_other_batches.jsonl": "r06_other_batches.jsonl", "r06_grad_modes.jsonl": "r06_grad_modes.jsonl",
      "r06_infer_blocks.txt": "r06_infer_blocks.txt", "r06_float_b256_kernel_stats.csv": "r06_float_b256_kernel_stats.csv", "r06_g32_b512_kernel_stats.csv": "r06_g32_b512_kernel_stats.csv",
      "r06_replay_nodes.txt": "r06_replay_nodes.txt", "r06_replay_nodes.json": "r06_replay_nodes.json"}
for a, b in cp.items():
    if os.path.exists(g + a):
        shutil.copy(g + a, "profiles/" + b)
    else:
        print("missing", g + a)
if os.path.exists(g + "layer_times_r06_b512.txt"):
    open("profiles/r06_layer_times_b512.txt", "w").write("".join(l for l in open(g + "layer_times_r06_b512.txt") if "amdgpu.ids" not in l))
tail = open(g + "gpu_suite_r06.log").read().strip().splitlines()[-4:] + open(g + "smoke_r06.log").read().strip().splitlines()[-1:]
open("profiles/r06_gpu_suite_tail.txt", "w").write("\n".join(tail) + "\n")
d = json.loads(open(g + "bench_r06.json").read().strip().splitlines()[-1])
k = json.load(open("profiles/r06_kernels_b512.json"))
side = [json.loads(l) for l in open(g + "r06_side_workloads.jsonl") if l.strip()]
gm = [json.loads(l) for l in open(g + "r06_grad_modes.jsonl") if l.strip()]
r, c = d["roofline"], d["config"]
gn = r.get("graph_nodes") or {}
suite = next((l for l in reversed(tail) if " passed" in l), "?").strip("= ").split(" in ")[0]
out = dict(IMGS=f"{d['value']:,.0f}".replace(",", " "), MS=f"{d['ms_per_step']:.2f}", FRAC=f"{r['frac']:.3f}", GB=f"{k['hbm_bytes_per_step'] / 1e9:.1f}", RATIO=f"{k['traffic_ratio']:.2f}",
           NODES=gn.get("nodes_total"), OWN=gn.get("kernels_own"), SUITE=suite, G32=c.get("fp32_grad_ms_per_step"), DPO=c.get("dp_overhead_ms"),
           SEG=(c.get("dp_overhead") or {}).get("segmented_ms_per_step"), DOM=r["dominant_kernel"], MFMA=r["mfma"], CPU=d["cpu_baseline"]["value"],
           SIDE=[(s["metric"][:48], round(s["value"]), s.get("roofline", {}).get("frac")) for s in side],
           GRAD_MODES=[(m["config"].get("grad_dtype"), m["config"].get("per_gpu_batch"), m["ms_per_step"]) for m in gm])
print(json.dumps(out, indent=1))
fam = r.get("families", {})
for kname, v in fam.items():
    print(f"{kname:18s} n={v['n']:3d} us={v['us']:7.1f} MB={v['mb']:8.1f} GB/s={v['gbs']:6.0f} traffic={v['traffic_ratio']}")
json.dump(out, open("profiles/r06_doc_numbers.json", "w"), indent=1)
