"""Layer-by-layer divergence finder: oracle (CPU) vs HIP engine on the same whole-net QAT train forward."""
import sys, os, warnings
warnings.filterwarnings("ignore")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import frost_oracle as O
import __graft_entry__ as ge
ge.build()
from frostnet_amd import frostnet as F
mode, res, B = sys.argv[1] if len(sys.argv) > 1 else "small", int(sys.argv[2]) if len(sys.argv) > 2 else 64, 2
training = (sys.argv[3] != "eval") if len(sys.argv) > 3 else True
cfg = O.net_cfg(mode, 1.0)
spec = O.float_state_spec(cfg)
P, Bf = O.make_state(spec, 5000, True)
qs = O.QState(Bf); qs.trace = []
x = torch.from_numpy(O.synth((B, 3, res, res), 520))
with torch.no_grad():
    ref = O.frostnet_forward(P, qs, cfg, x, True, training)
model = F.MODEL_REGISTRY[f"frostnet_quant_{mode}_1_0"](drop_rate=0.0)
model.load_state_dict(O.synth_state([k for k, _ in spec], [s for _, s in spec], 5000))
F.qat_prepare(model, version=0); model.cuda(); model.train(training)
run = model.hip_runner(); run.E.trace = []
with torch.no_grad():
    out = run._forward_impl(x.cuda(), record=False)
torch.cuda.synchronize()
print("n sites", len(qs.trace), len(run.E.trace))
for (pn, yo), (gn, a) in zip(qs.trace, run.E.trace):
    yg = a.dequant().cpu()
    if yg.shape[1] != yo.shape[1]: yg = yg[:, :yo.shape[1]]
    sc = float(qs.sd[pn + ".scale"][0]); q = run.qa.get(a.q)
    d = (yg - yo).abs() / sc
    if float(d.max()) > 0.5 or pn.startswith("quant") or pn.startswith("conv1"): print(f"{pn[:60]:60s} {gn:28s} shape {tuple(yo.shape)} scale {sc:.6f}/{q['scale']:.6f} zp {int(qs.sd[pn + '.zero_point'][0])}/{q['zero_point']} maxd {float(d.max()):.2f} frac {float((d > 0.5).float().mean()):.5f}")
print("logits rel", float((out.cpu() - ref).norm() / ref.norm()), "scale", float(qs.sd['classifier.2.activation_post_process.scale'][0]), run.qa.get(run.cls.qy))
