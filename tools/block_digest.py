"""One seeded forward + backward of a bottleneck's conv1 -> conv2 -> reduce_conv chain (Engine.conv_pair + Engine.conv, Engine.backward); dumps every result.
Used by tests/test_gpu_block.py to hold the block-level kernels (environment switches read at import / library load, hence one subprocess per
configuration) to the layer-by-layer launches.   usage: block_digest.py out.npz cin cexp H k cout B [stride of conv2 = 1]"""
import os, sys, warnings
warnings.filterwarnings("ignore")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as ge
ge.build()
from frostnet_amd import engine as EN, _lib as L
out = sys.argv[1]
cin, cexp, H, k, cout, B = [int(v) for v in sys.argv[2:8]]
stride2 = int(sys.argv[8]) if len(sys.argv) > 8 else 1
dev = "cuda"
g = torch.Generator(device="cpu").manual_seed(4321)
E = EN.Engine(dev); qa = EN.QArena(10, dev)


def layer(name, kind, ci, co, kk, relu, stride=1):
    fan = ci if kind == "pw" else kk * kk
    w = (torch.randn(co, 1 if kind == "dw" else ci, kk, kk, generator=g) * (2.0 / fan) ** 0.5).to(dev).requires_grad_(True)
    gamma = (torch.rand(co, generator=g) * 0.5 + 0.75).to(dev).requires_grad_(True)
    beta = (torch.rand(co, generator=g) * 0.2 - 0.05).to(dev).requires_grad_(True)
    return E.add_layer(EN.ConvLayer(name, kind, w, gamma, beta, torch.zeros(co, device=dev), torch.ones(co, device=dev), torch.zeros((), dtype=torch.int64, device=dev), None,
                                    kk, stride, relu, qa.alloc(), qa.alloc()))


l1, l2, l3 = layer("conv1", "pw", cin, cexp, 1, True), layer("conv2", "dw", cexp, cexp, k, True, stride2), layer("reduce", "pw", cexp, cout, 1, False)
qx = qa.alloc(); qa.set_qparams(qx, 0.02, 3)
x = E.new_act(B, H, H, cin, qx)
x.buf[: x.numel] = torch.randint(-128, 128, (x.numel,), dtype=torch.int16, generator=g).to(torch.int8).to(dev)
x.needs_grad = True
if os.environ.get("DIGEST_CALLS"):
    L.CALL_LOG = []
E.begin_step()
if E.pair_fusable(l1, l2, x, True, True):
    y2 = E.conv_pair(l1, l2, x, l3=l3)
else:
    y2 = E.conv(l2, E.conv(l1, x))
y3 = E.conv(l3, y2)
gy = (torch.randn(y3.numel, generator=g) * 1e-3).to(dev)
if E.grad_fp32:          # FROST_GRAD=fp32: the same (bf16-representable) output gradient in the fp32-gradient parity mode -- the yardstick for the bf16 paths
    y3.grad = torch.cat([gy.to(torch.bfloat16).float(), torch.zeros(64, dtype=torch.float32, device=dev)])
else:
    y3.grad = torch.cat([gy.to(torch.bfloat16).view(torch.int16), torch.zeros(64, dtype=torch.int16, device=dev)])
E.backward()
torch.cuda.synchronize()
if os.environ.get("DIGEST_CALLS"):
    open(os.environ["DIGEST_CALLS"], "w").write("\n".join(L.CALL_LOG))
res = dict(y3=y3.buf[: y3.numel].cpu().numpy(), dx=x.grad[: x.numel].cpu().numpy())          # dx: bf16 bits, or fp32 values in the fp32-gradient mode
for i, l in enumerate((l1, l2, l3), 1):
    res.update({f"dw{i}": l.w.grad.cpu().numpy(), f"dgamma{i}": l.gamma.grad.cpu().numpy(), f"dbeta{i}": l.beta.grad.cpu().numpy(), f"qy{i}": l.qy.cpu().numpy()})
# conv1's forward statistics as the finalize saw them (the replicated tables summed) and what it derived: forward-side coefficient rows, running statistics
from frostnet_amd._lib import STATS_BYTES_PER_CH
nc, cp = STATS_BYTES_PER_CH // 24, l1.coef.numel() // L.COEF_ROWS
tab = l1.stats.view(torch.uint8)[: cp * STATS_BYTES_PER_CH].cpu().numpy().reshape(nc, cp * 24)
s1 = sum(t[: cp * 8].view(np.int64) for t in tab); s2 = sum(t[cp * 8: cp * 16].view(np.uint64) for t in tab)
mn = np.min([t[cp * 16: cp * 20].view(np.int32) for t in tab], 0); mx = np.max([t[cp * 20: cp * 24].view(np.int32) for t in tab], 0)
res.update(stats1_s1=s1, stats1_s2=s2, stats1_mn=mn, stats1_mx=mx, coef1_fwd=l1.coef.view(L.COEF_ROWS, cp)[[0, 1, 2, 3, 4, 7]].cpu().numpy(),
           rmean1=l1.rmean.cpu().numpy(), rvar1=l1.rvar.cpu().numpy())
np.savez(out, **res)
