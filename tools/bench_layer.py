"""Micro-benchmark of ONE conv layer's passes (HIP events, many repetitions) -- kernel tuning loop.
usage: bench_layer.py kind cin cout k stride H B [reps]"""
import os, sys, warnings
warnings.filterwarnings("ignore")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from frostnet_amd import engine as EN, _lib as L
kind, cin, cout, k, stride, H, B = sys.argv[1], *[int(v) for v in sys.argv[2:8]]
reps = int(sys.argv[8]) if len(sys.argv) > 8 else 5
dev = "cuda"
E = EN.Engine(dev); qa = EN.QArena(8, dev)
cin_g = 1 if kind == "dw" else cin
w = (torch.randn(cout, cin_g, k, k, device=dev) * (2.0 / (cout * k * k)) ** 0.5).requires_grad_(True)
gamma = (torch.rand(cout, device=dev) * 0.5 + 0.75).requires_grad_(True); beta = (torch.rand(cout, device=dev) * 0.2).requires_grad_(True)
l = EN.ConvLayer("L", kind, w, gamma, beta, torch.zeros(cout, device=dev), torch.ones(cout, device=dev),
                 torch.zeros((), dtype=torch.int64, device=dev), None, k, stride, True, qa.alloc(), qa.alloc())
E.add_layer(l)
qx = qa.alloc(); qa.set_qparams(qx, 0.02, 0)
cx = 4 if kind == "stem" else cin
x = E.new_act(B, H, H, cx, qx)
x.buf[: x.numel] = torch.randint(-128, 127, (x.numel,), dtype=torch.int8, device=dev)
L.PROFILER = L.Profiler()
for it in range(reps + 1):
    if it == 1: L.PROFILER.records.clear()
    E.begin_step()
    y = E.conv(l, x)
    y.grad = torch.randn(y.numel + 64, device=dev).to(torch.bfloat16).view(torch.int16)
    E.backward()
s = L.PROFILER.summary()
for kname, v in sorted(s.items(), key=lambda kv: -kv[1]["total_ms"]):
    if kname == "weight_prep": continue
    print(f"{kname:16s} avg {v['avg_ms']*1e3:9.1f} us   {v['bytes_per_launch']/1e6:9.1f} MB  {v['bytes_per_launch']/v['avg_ms']/1e6:8.1f} GB/s")
