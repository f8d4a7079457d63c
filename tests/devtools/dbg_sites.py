"""dev: per-block activation mismatch (index steps) GPU vs CPU oracle, QAT train-mode forward, first step."""
import os, sys, warnings
warnings.filterwarnings("ignore")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import __graft_entry__ as ge
ge.build()
from oracle import frost_oracle as O
from frostnet_amd import frostnet as F
torch.set_num_threads(16)
mode, B, R = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
def T(a): return torch.from_numpy(np.ascontiguousarray(a))
cfg = O.net_cfg(mode, 1.0); spec = O.float_state_spec(cfg)
x = T(O.synth((B, 3, R, R), 11))
P, Bf = O.make_state(spec, 5000, True); qs = O.QState(Bf)
ref_blocks = []
orig_bf = O.block_forward
def bf(P_, qs_, prefix, x_, bc, quantized, training):
    o = orig_bf(P_, qs_, prefix, x_, bc, quantized, training); ref_blocks.append((prefix, o.detach())); return o
O.block_forward = bf
with torch.no_grad():
    y_ref = O.frostnet_forward(P, qs, cfg, x, True, True)
model = F.MODEL_REGISTRY[f"frostnet_quant_{mode}_1_0"](drop_rate=0.0)
model.load_state_dict(O.synth_state([k for k, _ in spec], [s for _, s in spec], 5000))
F.qat_prepare(model, version=0); model.cuda()
r = model.hip_runner()
dev_blocks = []
orig = r.block_forward
def gb(d, inp, training, obs):
    o = orig(d, inp, training, obs); dev_blocks.append(o); return o
r.block_forward = gb
with torch.no_grad():
    y = model(x.cuda())
torch.cuda.synchronize()
for (name, ro), do in zip(ref_blocks, dev_blocks):
    sc = float(r.qa.get(do.q)["scale"])
    d = (do.dequant().cpu() - ro).abs() / sc
    print(f"{name:12s} scale {sc:.5f}  frac>0.5 {float((d > 0.5).float().mean()):.2e}  max {float(d.max()):.2f}  rel {float((do.dequant().cpu() - ro).norm() / ro.norm()):.2e}")
print("logits rel", float((y.cpu() - y_ref).norm() / y_ref.norm()))
