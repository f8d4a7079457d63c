"""dev: run-to-run / path-to-path gradient differences of one train step (table finalize vs per-layer finalize)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import frost_oracle as O
import frostnet_amd.frostnet as F
B, R = int(sys.argv[1]), int(sys.argv[2])
def run(per_layer):
    torch.manual_seed(3)
    model = F.frostnet_quant_small_1_0(drop_rate=0.0)
    F.qat_prepare(model, version=0)
    model.cuda().train()
    runner = model.hip_runner()
    if per_layer:
        runner.E.on_layer_grads = lambda l: None
    x = torch.from_numpy(O.synth((B, 3, R, R), 5)).cuda()
    tgt = (torch.arange(B) * 7 % 1000).cuda()
    loss = torch.nn.functional.cross_entropy(model(x), tgt)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss), [(n, p.grad.detach().cpu().double()) for n, p in model.named_parameters()]
def cmp(a, b, tag):
    worst = []
    nb = {n: float(y.norm()) for n, y in b[1]}
    for (n, x), (_, y) in zip(a[1], b[1]):
        sib = n.replace("bn.weight", "weight").replace("bn.bias", "weight")
        den = max(nb[n], nb[sib], 1e-30)
        e = float((x - y).norm()) / den
        worst.append((e, n))
    worst.sort(reverse=True)
    print(tag, "loss", a[0], b[0], "worst", [(f"{e:.2e}", n) for e, n in worst[:4]], flush=True)
t1, t2, p1, p2 = run(False), run(False), run(True), run(True)
cmp(t1, t2, "table-table"); cmp(p1, p2, "perlayer-perlayer"); cmp(t1, p1, "table-perlayer")
