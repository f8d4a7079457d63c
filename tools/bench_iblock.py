"""Dev: per-bottleneck time of the fused bf16 inference kernel vs the layer-by-layer launches (B = 256, Large @224).  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from frostnet_amd import frostnet as F, infer as I, _lib as L

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
model = F.MODEL_REGISTRY["frostnet_large_1_0"]().cuda().eval()
x = torch.randn(B, 3, 224, 224, device="cuda")
res = {}
for fused in (True, False):
    I._FUSED = fused
    model.__dict__.pop("_bf16_infer", None)
    model.hip_infer_bf16(x)
    inf = model.__dict__["_bf16_infer"]
    # time every C-ABI call of a forward with events, grouped per block by call order
    recs = []
    orig = L.call
    def timed(name, *a, prof=None):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); orig(name, *a); e1.record(); recs.append((name, e0, e1))
    I.call = timed
    for _ in range(3):
        recs.clear(); inf(x)
    torch.cuda.synchronize()
    I.call = orig
    res[fused] = [(n, e0.elapsed_time(e1) * 1e3) for n, e0, e1 in recs]
fu, pl = res[True], res[False]
print("fused total %.0f us, plain total %.0f us" % (sum(t for _, t in fu), sum(t for _, t in pl)))
# group the plain calls per block: a block ends at frost_infer_add or at the pw after a dw
i = 0
blocks = []
cur = []
for n, t in pl:
    cur.append((n, t))
    if n == "frost_infer_dw":
        seen_dw = True
    if (n == "frost_infer_pw" and any(c[0] == "frost_infer_dw" for c in cur)):
        blocks.append(cur); cur = []
    elif n == "frost_infer_add" and blocks:
        blocks[-1].append((n, t)); cur = []
fb = [t for n, t in fu if n == "frost_infer_block"]
names = []
for li, nb in enumerate((3, 2, 7, 5, 1)):
    names += [f"layer{li + 1}.{b}" for b in range(nb)]
for k, (nm, blk) in enumerate(zip(names, [b for b in blocks if any(c[0] == "frost_infer_dw" for c in b)])):
    print(f"{nm:10s} fused {fb[k]:7.1f} us   plain {sum(t for _, t in blk):7.1f} us  ({len(blk)} launches)")

I._FUSED = "auto"
model.__dict__.pop("_bf16_infer", None)
model.hip_infer_bf16(x)
inf = model.__dict__["_bf16_infer"]
for nm, ent in zip(names, inf.blocks):
    t = [v for k_, v in ent.items() if isinstance(k_, tuple) and k_[0] == "timing"][0]
    c = [v for k_, v in ent.items() if isinstance(k_, tuple) and k_[0] == "choice"][0]
    print(f"{nm:10s} auto -> {str(c):10s}  " + "  ".join(f"{str(cd)}:{ms * 1e3:.0f}" for ms, cd in t))
