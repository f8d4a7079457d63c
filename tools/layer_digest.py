"""One seeded forward + backward of a single conv layer through the engine; dumps every result to an .npz.
Used by tests/test_gpu_paths.py to hold the fast kernel paths (resident weights, DMA / LDS-staged tile I/O, channel split,
depthwise tile geometries, fused passes -- all selected by size) to the plain paths at sizes where they actually engage.
usage: layer_digest.py out.npz kind cin cout k stride H B"""
import os, sys, warnings
warnings.filterwarnings("ignore")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as ge
ge.build()
from frostnet_amd import engine as EN, _lib as L
out, kind = sys.argv[1], sys.argv[2]
cin, cout, k, stride, H, B = [int(v) for v in sys.argv[3:9]]
dev = "cuda"
g = torch.Generator(device="cpu").manual_seed(1234)
E = EN.Engine(dev); qa = EN.QArena(8, dev)
cin_g = 1 if kind == "dw" else cin
w = (torch.randn(cout, cin_g, k, k, generator=g) * (2.0 / (cout * k * k)) ** 0.5).to(dev).requires_grad_(True)
gamma = (torch.rand(cout, generator=g) * 0.5 + 0.75).to(dev).requires_grad_(True)
beta = (torch.rand(cout, generator=g) * 0.2 - 0.05).to(dev).requires_grad_(True)
rm, rv = torch.zeros(cout, device=dev), torch.ones(cout, device=dev)
relu = os.environ.get("DIGEST_RELU", "1") == "1"
l = EN.ConvLayer("L", kind, w, gamma, beta, rm, rv, torch.zeros((), dtype=torch.int64, device=dev), None, k, stride, relu, qa.alloc(), qa.alloc())
E.add_layer(l)
qx = qa.alloc(); qa.set_qparams(qx, 0.02, 3)
x = E.new_act(B, H, H, cin, qx)
x.buf[: x.numel] = torch.randint(-128, 128, (x.numel,), dtype=torch.int16, generator=g).to(torch.int8).to(dev)
x.needs_grad = True
if os.environ.get("DIGEST_CALLS"):
    L.CALL_LOG = []          # which C-ABI entries this configuration launched (written next to the results)
E.begin_step()
y = E.conv(l, x)
gy = (torch.randn(y.numel, generator=g) * 1e-3).to(dev)
y.grad = torch.cat([gy.to(torch.bfloat16).view(torch.int16), torch.zeros(64, dtype=torch.int16, device=dev)])
E.backward()
torch.cuda.synchronize()
if os.environ.get("DIGEST_CALLS"):
    open(os.environ["DIGEST_CALLS"], "w").write("\n".join(L.CALL_LOG))
np.savez(out, y=y.buf[: y.numel].cpu().numpy(), qy=l.qy.cpu().numpy(), rm=rm.cpu().numpy(), rv=rv.cpu().numpy(),
         dx=x.grad[: x.numel].cpu().numpy(), dw=w.grad.cpu().numpy(), dgamma=gamma.grad.cpu().numpy(), dbeta=beta.grad.cpu().numpy())
