"""How reproducible is a short QAT training run, and does a checkpoint resume continue it?  (diagnostic behind tests/test_gpu_round3.py)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import test_gpu_round3 as T
from frostnet_amd import harness as Hn
crit = Hn.CrossEntropyLoss()
res = int(sys.argv[1]) if len(sys.argv) > 1 else 64
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
loader = T._tiny_loader(2, bs, res, 5)


def params_of(m):
    return torch.cat([p.detach().float().reshape(-1) for p in m.parameters()]).clone()


def run(nsteps, drop=None, noise=True):
    model, opt = T._new_model_and_opt(1882)
    if drop is not None:
        model.classifier[1].p = drop
    opt.is_warmup = not noise
    out = [params_of(model)]
    for s in range(nsteps):
        x, t = loader[s % 2]
        Hn.train_one_iter(model, crit, opt, x.cuda(), t.cuda())
        out.append(params_of(model))
    return out

for drop, noise in ((None, True), (0.0, True), (0.0, False)):
    a, b = run(6, drop, noise), run(6, drop, noise)
    print(f"dropout {drop} noise {noise}: per-step update mismatch between two identical runs:",
          " ".join(f"{float(((a[i+1]-a[i])-(b[i+1]-b[i])).norm()/(a[i+1]-a[i]).norm()):.2e}" for i in range(6)))
