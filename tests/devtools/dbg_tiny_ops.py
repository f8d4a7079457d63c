"""Which Python lines issue the small torch ops (copies, fills) inside one eager training step?  torch.profiler with stacks, grouped."""
import sys, collections, torch
sys.path.insert(0, ".")
import frostnet_amd.frostnet as F
from frostnet_amd import harness as H
from frostnet_amd.optimizer import QSGD
torch.manual_seed(0)
m = F.frostnet_quant_large_1_0(); F.qat_prepare(m, version=0); m.cuda().train()
x = torch.randn(16, 3, 224, 224, device="cuda"); t = torch.randint(0, 1000, (16,), device="cuda")
opt = QSGD(H.make_param_groups(m, 1e-5), lr=5e-3, momentum=0.9, nesterov=True, clip_by=1e-3, toss_coin=True, noise_decay=1e-2); opt.is_warmup = False
crit = H.CrossEntropyLoss()
def step():
    opt.zero_grad(set_to_none=True); crit(m(x), t).backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name.startswith("aten::") and e.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::clone", "aten::mul", "aten::reciprocal", "aten::add_", "aten::eq", "aten::ne", "aten::logical_not", "aten::bitwise_not", "aten::to", "aten::_to_copy"):
        st = [s for s in (e.stack or []) if "frostnet_amd" in s or "bench" in s or "torch/ao" in s or "torch/nn" in s][:3]
        cnt[(e.name, " <- ".join(st))] += 1
for (n, st), c in cnt.most_common(30):
    print(c, n, st[:400])
