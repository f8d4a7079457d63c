"""dev: one QAT train step, GPU vs the CPU oracle, per-parameter gradient error (well-conditioned metric) + batch-permutation check."""
import os, sys, warnings
warnings.filterwarnings("ignore")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import __graft_entry__ as ge
ge.build()
from oracle import frost_oracle as O
from frostnet_amd import frostnet as F
torch.set_num_threads(16)
mode, B, R = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
def T(a): return torch.from_numpy(np.ascontiguousarray(a))
cfg = O.net_cfg(mode, 1.0)
spec = O.float_state_spec(cfg)
x = T(O.synth((B, 3, R, R), 11)); tgt = torch.arange(B) * 37 % 1000
P, Bf = O.make_state(spec, 5000, True); qs = O.QState(Bf)
y_ref = O.frostnet_forward(P, qs, cfg, x, True, True)
torch.nn.functional.cross_entropy(y_ref, tgt).backward()
def gpu(xs, ts):
    model = F.MODEL_REGISTRY[f"frostnet_quant_{mode}_1_0"](drop_rate=0.0)
    model.load_state_dict(O.synth_state([k for k, _ in spec], [s for _, s in spec], 5000))
    F.qat_prepare(model, version=0); model.cuda()
    y = model(xs.cuda()); torch.nn.functional.cross_entropy(y, ts.cuda()).backward(); torch.cuda.synchronize()
    return y.detach().cpu(), {n: p.grad.detach().cpu().double() for n, p in model.named_parameters()}
y, g = gpu(x, tgt)
perm = torch.randperm(B, generator=torch.Generator().manual_seed(0))
yp, gp = gpu(x[perm], tgt[perm])
ref = {O.float_to_qat_key(k) if hasattr(O, "float_to_qat_key") else k: v.grad.double() for k, v in P.items()}
names = [n for n in g if g[n].dim() == 4]
print("logits vs oracle rel", float((y - y_ref.detach()).norm() / y_ref.detach().norm()), " perm rel", float((yp - y[perm]).norm() / y.norm()))
rows = []
for n in names:
    r = ref.get(n)
    e_or = float((g[n] - r).norm() / r.norm()) if r is not None else float("nan")
    e_pm = float((g[n] - gp[n]).norm() / g[n].norm())
    rows.append((n, e_or, e_pm))
for n, a, b in rows[:6] + rows[-5:]:
    print(f"{n:40s} vs-oracle {a:.2e}   perm {b:.2e}")
print("max vs-oracle", max(r[1] for r in rows), "max perm", max(r[2] for r in rows))
