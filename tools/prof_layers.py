"""Per-layer, per-pass HIP-event timing of one eager QAT step (which layers/kernels are furthest from their roofline)."""
import os, sys, warnings
warnings.filterwarnings("ignore")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from frostnet_amd import _lib as L, frostnet as F
from frostnet_amd import engine as EN
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
model = F.frostnet_quant_large_1_0(); F.qat_prepare(model, version=0); model.cuda().train()
x = torch.randn(B, 3, 224, 224, device="cuda").contiguous(memory_format=torch.channels_last)
t = torch.randint(0, 1000, (B,), device="cuda")
def step():
    torch.nn.functional.cross_entropy(model(x), t).backward()
step(); step(); torch.cuda.synchronize()
# re-tag: wrap call so label includes the layer name (args hold no name -> use engine hooks)
orig_launch, orig_bwd = EN.Engine._conv_launch, EN.Engine._conv_backward
cur = {"name": ""}
def launch(self, l, x_, mode, y): cur["name"] = l.name; return orig_launch(self, l, x_, mode, y)
def bwd(self, l, x_, y): cur["name"] = l.name; return orig_bwd(self, l, x_, y)
EN.Engine._conv_launch, EN.Engine._conv_backward = launch, bwd
orig_call = L.call
def call(name, *a, prof=None):
    if prof is not None: prof = (prof[0] + "|" + cur["name"], prof[1])
    return orig_call(name, *a, prof=prof)
EN.call = call
L.PROFILER = L.Profiler()
step()
s = L.PROFILER.summary()
rows = sorted(s.items(), key=lambda kv: -kv[1]["total_ms"])
tot = sum(v["total_ms"] for v in s.values())
print(f"total tagged {tot:.2f} ms")
import collections
agg = collections.defaultdict(float)
for k, v in s.items(): agg[k.split("|")[0]] += v["total_ms"]
print("by kind: " + "  ".join(f"{k}={v:.2f}" for k, v in sorted(agg.items(), key=lambda kv: -kv[1])))
bag = collections.defaultdict(float)
for k, v in s.items(): bag[k.split("|")[0]] += v["bytes_per_launch"] * v["launches"]
print("by kind GB/s: " + "  ".join(f"{k}={bag[k]/agg[k]/1e6:.0f}" for k, _ in sorted(agg.items(), key=lambda kv: -kv[1])))
print(f"total tagged bytes {sum(bag.values())/1e9:.2f} GB")
lay = collections.defaultdict(lambda: [0.0, 0.0])
for k, v in s.items():
    if "|" in k:
        n = k.split("|")[1]; lay[n][0] += v["total_ms"]; lay[n][1] += v["bytes_per_launch"] * v["launches"]
shapes = {l.name: (l.kind, l.cin_g, l.cout, l.k, l.stride) for l in model.hip_runner().E.layers}
print("per layer:")
for n, (ms, by) in sorted(lay.items(), key=lambda kv: -kv[1][0]):
    print(f"  {n:28s} {str(shapes.get(n, '')):32s} {ms:7.3f} ms {by/1e6:9.1f} MB {by/ms/1e6:8.0f} GB/s")
for k, v in rows[:int(os.environ.get("ROWS", "60"))]:
    print(f"{k:50s} {v['total_ms']:8.3f} ms  {v['bytes_per_launch']/1e6:9.1f} MB  {v['bytes_per_launch']/v['avg_ms']/1e6:8.1f} GB/s")
