"""Stress of the statistics -> finalize hand-off (last-workgroup-done tail, agent atomics + sc1 loads, no fences): two copies of the same model
run N training-mode forwards on the same inputs; the forward is bit-reproducible, so every observer / running statistic must match bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import copy, torch
from frostnet_amd import frostnet as F

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
torch.manual_seed(0)
m1 = F.MODEL_REGISTRY["frostnet_quant_large_1_0"](drop_rate=0.0)
F.qat_prepare(m1, version=0)
m2 = copy.deepcopy(m1)
m1.cuda().train(); m2.cuda().train()
g = torch.Generator(device="cuda").manual_seed(1)
bad = 0
for i in range(N):
    x = torch.randn(B, 3, 224, 224, device="cuda", generator=g)
    with torch.no_grad():
        y1 = m1(x); y2 = m2(x)
    if not torch.equal(y1, y2):
        bad += 1
        print("step", i, "logits differ", float((y1 - y2).abs().max()), flush=True)
torch.cuda.synchronize()
s1, s2 = m1.state_dict(), m2.state_dict()
nd = [k for k in s1 if not torch.equal(s1[k].float().nan_to_num(posinf=0, neginf=0), s2[k].float().nan_to_num(posinf=0, neginf=0))]
print(f"{N} forwards at B={B}: {bad} steps with differing logits, {len(nd)} state entries differ", nd[:5])
