"""Dev: per-bottleneck time of bf16 inference at B = 256 (Large @224): the layer-by-layer launches against the fused launch (csrc/frost_iblock.hip) with every candidate
tile, as measured by the "auto" policy of frostnet_amd/infer.py (3 back-to-back runs each), and the choice it makes.  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from frostnet_amd import frostnet as F, infer as I

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
model = F.MODEL_REGISTRY["frostnet_large_1_0"]().cuda().eval()
x = torch.randn(B, 3, 224, 224, device="cuda")
names = []
for li, nb in enumerate((3, 2, 7, 5, 1)):
    names += [f"layer{li + 1}.{b}" for b in range(nb)]
I._FUSED = "auto"
model.hip_infer_bf16(x)
inf = model.__dict__["_bf16_infer"]
print(f"batch {B}; microseconds per bottleneck: layer-by-layer ('plain') and fused with tile (th, tw)")
tot_p = tot_c = 0.0
for nm, ent in zip(names, inf.blocks):
    t = [v for k_, v in ent.items() if isinstance(k_, tuple) and k_[0] == "timing"][0]
    c = [v for k_, v in ent.items() if isinstance(k_, tuple) and k_[0] == "choice"][0]
    tot_p += [ms for ms, cd in t if cd == "plain"][0]
    tot_c += min(ms for ms, cd in t)
    print(f"{nm:10s} auto -> {str(c):10s}  " + "  ".join(f"{str(cd)}:{ms * 1e3:.0f}" for ms, cd in t))
print(f"sum over the bottlenecks: layer-by-layer {tot_p * 1e3:.0f} us, chosen {tot_c * 1e3:.0f} us")
