"""Run-to-run reproducibility of ONE QAT step (fresh identical models): logits bit-equal?  gradient arena relative difference?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from frostnet_amd import frostnet as F
mode, res, bs = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
g = torch.Generator().manual_seed(5)
x = torch.randn(bs, 3, res, res, generator=g).cuda(); t = torch.randint(0, 1000, (bs,), generator=g).cuda()

def once():
    torch.manual_seed(1882)
    m = F.MODEL_REGISTRY[f"frostnet_quant_{mode}_1_0"](drop_rate=0.0)
    F.qat_prepare(m, version=0); m.cuda().train()
    out = m(x)
    torch.nn.functional.cross_entropy(out, t).backward()
    torch.cuda.synchronize()
    r = m.hip_runner()
    return out.detach().clone(), r.grad_arena.clone(), [(n, p.grad.clone()) for n, p in m.named_parameters()]

o0, g0, n0 = once()
for rep in range(3):
    o1, g1, n1 = once()
    worst = sorted(((float((a - b).norm() / (b.norm() + 1e-30)), n) for (n, a), (_, b) in zip(n1, n0)), reverse=True)[:3]
    print(f"{mode}@{res} B{bs} rep {rep}: logits equal {torch.equal(o0, o1)}  grad arena rel diff {float((g1 - g0).norm() / g0.norm()):.2e}  worst tensors {[(f'{e:.1e}', n) for e, n in worst]}", flush=True)
