import sys, warnings; warnings.filterwarnings("ignore")
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import __graft_entry__ as ge; ge.build()
from frostnet_amd import frostnet as F, ssdlite as S
from test_gpu_detect import _targets
torch.manual_seed(0)
model = S.SSDLiteFrostNet(num_classes=21, mode="large"); F.qat_prepare(model, version=0); model.cuda().train()
crit = S.MultiBoxLoss(21)
B=2
x = torch.randn(B,3,512,512,device="cuda").contiguous(memory_format=torch.channels_last)
loc, conf, pri = model(x); ll, lc = crit((loc,conf,pri), _targets(B)); (ll+lc).backward(); torch.cuda.synchronize()
for n,p in model.named_parameters():
    if n.startswith(("extras","loc.5","conf.5","loc.4","conf.4")) and n.endswith("conv.0.weight"):
        print(n, tuple(p.shape), float(p.grad.norm()), float(p.grad.abs().max()))
