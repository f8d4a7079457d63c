"""dev: throughput of one float (StatAssist warm-up) training step on the device: forward + backward + QSGD step (is_warmup)."""
import os, sys, time, warnings
warnings.filterwarnings("ignore")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from frostnet_amd import frostnet as F, harness as H, _lib as L
from frostnet_amd.optimizer import QSGD
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
model = F.frostnet_large_1_0().cuda().train()
opt = QSGD(H.make_param_groups(model, 1e-5), lr=5e-3, momentum=0.9, nesterov=True, clip_by=1e-3, toss_coin=True, noise_decay=1e-2)
crit = torch.nn.CrossEntropyLoss()
x = torch.randn(B, 3, 224, 224, device="cuda"); t = torch.randint(0, 1000, (B,), device="cuda")
for _ in range(3): H.train_one_iter(model, crit, opt, x, t)
torch.cuda.synchronize(); t0 = time.time(); n = 8
for _ in range(n): H.train_one_iter(model, crit, opt, x, t)
torch.cuda.synchronize(); dt = (time.time() - t0) / n
print(f"float train step B={B}: {dt*1e3:.1f} ms  {B/dt:.0f} img/s  peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
