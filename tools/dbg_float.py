"""dev: per-block activation error and per-parameter gradient error of the float device path vs the fp32 CPU definition."""
import copy, os, sys, warnings
warnings.filterwarnings("ignore")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from frostnet_amd import frostnet as F
name, res, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
torch.manual_seed(5)
model = F.MODEL_REGISTRY[name](drop_rate=0.0)
g = torch.Generator().manual_seed(3)
for m in model.modules():
    if isinstance(m, torch.nn.BatchNorm2d):
        m.weight.data = torch.rand(m.num_features, generator=g) * 0.8 + 0.6
        m.bias.data = torch.rand(m.num_features, generator=g) * 0.2 - 0.1
ref = copy.deepcopy(model)
if os.environ.get("EMU", "1") == "1":      # the fp32 definition with the device path's storage roundings (layer I/O and 1x1 weights -> bf16)
    for m in ref.modules():
        if type(m).__name__ in ("ConvBNReLU", "ConvBN"):
            m.register_forward_hook(lambda mod, i, o: o.to(torch.bfloat16).float())
            m.register_forward_pre_hook(lambda mod, i: (i[0].to(torch.bfloat16).float(),))
x = torch.randn(B, 3, res, res)
tgt = (torch.arange(B) * 37) % 1000
ref.train()
caps = {}
blocks = [("conv1", ref.conv1)] + [(f"{ln}.{i}", b) for ln in ("layer1", "layer2", "layer3", "layer4", "layer5") for i, b in enumerate(getattr(ref, ln))] + [("last", ref.last_layer)]
for n, m in blocks:
    m.register_forward_hook(lambda mod, i, o, n=n: caps.__setitem__(n, o.detach()))
saved_w = {}
for m in ref.modules():
    if isinstance(m, torch.nn.Conv2d) and m.groups == 1 and m.out_channels != 1000 and os.environ.get("EMU", "1") == "1":
        saved_w[m] = m.weight.data.clone(); m.weight.data = m.weight.data.to(torch.bfloat16).float()
y_ref = ref(x)
torch.nn.functional.cross_entropy(y_ref, tgt).backward()
model.cuda().train()
r = model.hip_runner()
outs = {}
bouts = []
orig_block, orig_conv = r._block, r._conv
def blk(ent, a, training, record):
    o = orig_block(ent, a, training, record); bouts.append(o); return o
r._block = blk
def conv(l, a, training, record, out=None, ldy=None):
    o = orig_conv(l, a, training, record, out, ldy)
    if l.name in ("conv1", "last_layer"): outs[l.name] = o
    return o
r._conv = conv
y = model(x.cuda())
torch.nn.functional.cross_entropy(y, tgt.cuda()).backward()
torch.cuda.synchronize()
def rel(a, b): return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
print("stem", rel(outs["conv1"].float().cpu(), caps["conv1"]))
for i, (n, _) in enumerate(blocks[1:-1]):
    print(n, f"{rel(bouts[i].float().cpu(), caps[n]):.3e}")
print("last", rel(outs["last_layer"].float().cpu(), caps["last"]), "logits", rel(y.detach().cpu(), y_ref.detach()))
refg = {n: p.grad.double() for n, p in ref.named_parameters()}
rows = []
for n, p in model.named_parameters():
    a, b = p.grad.detach().cpu().double(), refg[n]
    den = float(b.norm())
    if n.endswith(".conv.1.weight") or n.endswith(".conv.1.bias"):
        den = max(den, float(refg[n.rsplit(".conv.1.", 1)[0] + ".conv.0.weight"].norm()))
    rows.append((float((a - b).norm()) / max(den, 1e-30), n, float(a.norm()), float(b.norm())))
rows.sort(reverse=True)
for e, n, an, bn in rows[:12]:
    print(f"{n:44s} {e:.3e}  |dev| {an:.3e} |ref| {bn:.3e}")
import numpy as np
print("median grad err", np.median([r[0] for r in rows]))
