"""Per-layer, per-pass kernel times of one FrostNet-Large QAT step (HIP events around every tagged launch, eager mode).
Usage (GPU box): python tests/devtools/layer_times.py [batch] [steps]  ->  table sorted by position in the step + totals per layer."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from frostnet_amd import _lib as L, frostnet as F, engine as E
from frostnet_amd.harness import CrossEntropyLoss
from frostnet_amd.optimizer import QSGD

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
torch.manual_seed(0)
model = F.MODEL_REGISTRY["frostnet_quant_large_1_0"]()
F.qat_prepare(model, version=0)
model.cuda().train()
opt = QSGD(model.parameters(), lr=1e-3, momentum=0.9, nesterov=True)
crit = CrossEntropyLoss()
x = torch.randn(B, 3, 224, 224, device="cuda")
t = torch.randint(0, 1000, (B,), device="cuda")

ctx = [""]
for fn in ("conv", "conv_pair", "_conv_backward", "_conv_backward_g32"):
    orig = getattr(E.Engine, fn)
    def wrap(self, l, *a, _o=orig, **k):
        ctx[0] = l.name
        try:
            return _o(self, l, *a, **k)
        finally:
            ctx[0] = ""
    setattr(E.Engine, fn, wrap)

class P(L.Profiler):
    def __init__(self):
        super().__init__()
        self.ctxs = []
_call = L.call
def step():
    opt.zero_grad(); loss = crit(model(x), t); loss.backward(); opt.step()
for _ in range(3):
    step()
torch.cuda.synchronize()
prof = L.Profiler()
# record the layer context beside each record
orig_append = prof.records.append
ctxs = []
class Rec(list):
    def append(self, r):
        ctxs.append(ctx[0]); super().append(r)
prof.records = Rec()
L.PROFILER = prof
E.L.PROFILER = prof
for _ in range(steps):
    step()
torch.cuda.synchronize()
L.PROFILER = None
n = len(prof.records) // steps
rows = collections.OrderedDict()
for i, (rec, c) in enumerate(zip(prof.records, ctxs)):
    label, nb, e0, e1 = rec
    key = (i % n, c, label)
    r = rows.setdefault(key, [0.0, nb])
    r[0] += e0.elapsed_time(e1) * 1000.0 / steps
per_layer = collections.OrderedDict()
for (pos, c, label), (us, nb) in rows.items():
    print(f"{pos:4d} {c:34s} {label:18s} {us:8.1f} us {nb / 1e6:8.1f} MB {nb / us / 1e3 if us else 0:7.0f} GB/s")
    per_layer.setdefault(c, collections.OrderedDict()).setdefault(label, [0.0])[0] += us
print("\nper layer (us): fwd_stats fwd_emit | bwd_reduce fused/dc dgrad wgrad | total")
tot = 0.0
for c, d in per_layer.items():
    s = sum(v[0] for v in d.values()); tot += s
    print(f"{c:34s} " + " ".join(f"{k.split('_', 1)[1] if '_' in k else k}={v[0]:.0f}" for k, v in d.items()) + f" | {s:.0f}")
print(f"total tagged: {tot / 1000:.2f} ms")
