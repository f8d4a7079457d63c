"""Dev: which Python call sites issue the small aten fills / copies of one eager QAT step."""
import os, sys, warnings, collections
warnings.filterwarnings("ignore")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from frostnet_amd import frostnet as F
from frostnet_amd.optimizer import QSGD
model = F.frostnet_quant_large_1_0(); F.qat_prepare(model, version=0); model.cuda().train()
opt = QSGD([{"params": [p]} for p in model.parameters()], lr=5e-3, momentum=0.9, nesterov=True, clip_by=1e-3, toss_coin=True, noise_decay=1e-2, weight_decay=1e-5)
opt.is_warmup = False
x = torch.randn(64, 3, 224, 224, device="cuda").contiguous(memory_format=torch.channels_last)
t = torch.randint(0, 1000, (64,), device="cuda")
def step():
    torch.nn.functional.cross_entropy(model(x), t).backward(); opt.step()
step(); step(); torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::fill_", "aten::copy_", "aten::zero_", "aten::zeros", "aten::zeros_like", "aten::clone", "aten::to", "aten::add_", "aten::mul_"):
        st = [s for s in ev.stack if "frostnet_amd" in s or "bench" in s or "tools" in s]
        cnt[(ev.name, st[0] if st else (ev.stack[0] if ev.stack else "?"))] += 1
for (n, s), c in cnt.most_common(40):
    print(f"{c:5d} {n:18s} {s}")
