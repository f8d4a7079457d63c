"""Localise run-to-run differences of the first QAT step: per layer (in backward order) compare the S1/S2 rows, the input gradient and dW."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from frostnet_amd import frostnet as F, engine as E, _lib as L
from test_gpu_dp import _shard

rec = []
orig = E.Engine._conv_backward
def wrap(self, l, x, y):
    gout = y.grad[: y.numel].clone()
    orig(self, l, x, y)
    S = L.COEF_ROWS
    rec.append((l.name, gout, l.coef.clone(), x.grad[: x.numel].clone() if x.grad is not None else None, l.dwq.clone()))
E.Engine._conv_backward = wrap

def once():
    rec.clear()
    torch.manual_seed(0)
    model = F.frostnet_quant_small_1_0(drop_rate=0.0)
    F.qat_prepare(model, version=0)
    model.cuda().train()
    x, tgt = _shard(0)
    x, tgt = x.cuda(), tgt.cuda()
    torch.nn.functional.cross_entropy(model(x), tgt).backward()
    torch.cuda.synchronize()
    bf = lambda t: t.view(torch.bfloat16).float().cpu()
    return [(n, bf(g), c.cpu(), (bf(gx) if gx is not None else None), d.cpu()) for n, g, c, gx, d in rec]

def rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-20))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ref = once()
nbad = 0
for r in range(reps):
    cur = once()
    lines = []
    for (n, g, c, gx, d), (_, g0, c0, gx0, d0) in zip(cur, ref):
        e = (rel(g, g0), rel(c, c0), rel(gx, gx0) if gx is not None else 0.0, 0.0)
        if max(e) > 2e-6:
            rows = [i for i in range(c.shape[0]) if rel(c[i], c0[i]) > 1e-4]
            lines.append(f"   {n}: gout {e[0]:.1e} coef {e[1]:.1e} (rows {rows}) gx {e[2]:.1e} dwq {e[3]:.1e}")
    if lines:
        nbad += 1
        print(f"rep {r}: first divergences in backward order:", flush=True)
        print("\n".join(lines[:24]), flush=True)
print(f"{nbad}/{reps} differ", flush=True)
