"""dev: the gradient-bucket plan (frostnet_amd/parallel.py plan_buckets) of the classifier and of the SSDLite detector: closing layer, arena ranges, share of the bytes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from frostnet_amd import frostnet as F, ssdlite as S
from frostnet_amd.parallel import plan_buckets
for name in ("classifier", "detector"):
    torch.manual_seed(0)
    if name == "classifier":
        model = F.MODEL_REGISTRY["frostnet_quant_large_1_0"](drop_rate=0.0); x = torch.randn(2, 3, 224, 224, device="cuda")
    else:
        model = S.SSDLiteFrostNet(num_classes=21, mode="large", cfg=S.ssd_cfg_for(512)); x = torch.randn(2, 3, 512, 512, device="cuda")
    F.qat_prepare(model, version=0)
    model.cuda().train()
    with torch.no_grad():
        model(x)
    runner = model.hip_runner()
    tot = runner.grad_arena.numel()
    for l, ranges in plan_buckets(runner, 4):
        n = sum(b - a for a, b in ranges)
        print(name, "bucket closes at", l.name, "ranges", len(ranges), "share %.1f %%" % (100.0 * n / tot))
