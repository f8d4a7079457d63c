"""BASELINE.json config c5 on the device: SSDLite detector on the FrostNet backbone (frostnet_amd.ssdlite) in fake-quant QAT mode --
HIP engine against the CPU oracle composition (oracle.ssdlite_forward, built from the reference-pinned convbn_qat / block_forward), eval-mode
maps within one quantisation step, one training step through MultiBoxLoss with finite, sanity-bounded gradients, and c5's real size."""
import numpy as np
import pytest
import torch

from oracle import frost_oracle as O

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def relerr(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


@pytest.fixture(scope="module")
def mods():
    import __graft_entry__ as ge
    ge.build()
    from frostnet_amd import frostnet, ssdlite
    return frostnet, ssdlite


def _targets(n):
    rng = np.random.Generator(np.random.PCG64(77))
    out = []
    for i in range(n):
        k = 1 + i % 3
        c = rng.random((k, 2)) * 0.5 + 0.25
        wh = rng.random((k, 2)) * 0.3 + 0.1
        boxes = np.concatenate([c - wh / 2, c + wh / 2, rng.integers(0, 20, (k, 1)).astype(np.float64)], 1)
        if i == 0:      # one image-filling object, so that the coarse maps (4x4, 2x2: the SSDLite extras) receive positives as well
            boxes = np.concatenate([boxes, [[0.04, 0.06, 0.97, 0.93, 5.0]]], 0)
        out.append(T(boxes.astype(np.float32)))
    return out


def test_ssdlite_eval_maps_vs_oracle(mods):
    F, S = mods
    torch.set_num_threads(16)
    mode, res, B = "small", 128, 2
    cfg = O.net_cfg(mode, 1.0)
    spec, src = O.ssdlite_state_spec(cfg)
    P, Bf = O.make_state(spec, 6000, True)
    qs = O.QState(Bf)
    with torch.no_grad():                                   # two train-mode forwards populate BN statistics and observers (oracle side)
        for s_ in range(2):
            O.ssdlite_forward(P, qs, cfg, T(O.synth((B, 3, res, res), 700 + s_)), True)
    model = S.SSDLiteFrostNet(num_classes=21, mode=mode)
    assert [k for k, _ in spec] == [k for k in model.state_dict().keys()]
    F.qat_prepare(model, version=0)
    sd = {k: v.detach().clone() for k, v in P.items()}
    sd.update({k: v.detach().clone() for k, v in qs.sd.items()})
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    model.cuda().eval()
    x = T(O.synth((B, 3, res, res), 709))
    with torch.no_grad():
        ref = O.ssdlite_forward(P, qs, cfg, x, False)
        maps = model.hip_runner().forward_maps(x.cuda())
    names = [f"{h}.{i}" for i in range(6) for h in ("loc", "conf")]
    worst = 0.0
    for name, m, r in zip(names, maps, ref):
        sc = float(qs.sd[f"{name}.pw.conv.0.activation_post_process.scale"][0])
        d = (m.cpu() - r).abs() / sc
        worst = max(worst, float((d > 0.5).float().mean()))
        assert float(d.max()) <= 2.01 and float((d > 0.5).float().mean()) <= 2e-2, (name, float(d.max()), float((d > 0.5).float().mean()))
    loc, conf, pri = model(x.cuda())
    assert loc.shape == (B, 1536, 4) and conf.shape == (B, 1536, 21)
    print(f"[ssdlite small@128 eval] worst fraction of map entries off by >= 1 step: {worst:.2e}")


def test_ssdlite_train_step_and_c5_size(mods):
    F, S = mods
    from frostnet_amd.optimizer import QSGD
    torch.manual_seed(0)
    model = S.SSDLiteFrostNet(num_classes=21, mode="large")
    F.qat_prepare(model, version=0)
    model.cuda().train()
    crit = S.MultiBoxLoss(21)
    opt = QSGD([{"params": [p]} for p in model.parameters()], lr=1e-3, momentum=0.9, nesterov=True)
    B = 2
    x = torch.randn(B, 3, 512, 512, device="cuda").contiguous(memory_format=torch.channels_last)       # config c5: 512x512
    tg = _targets(B)
    losses = []
    for _ in range(2):
        loc, conf, pri = model(x)
        assert loc.shape == (B, 24528, 4) and conf.shape == (B, 24528, 21)
        ll, lc = crit((loc, conf, pri), tg)
        (ll + lc).backward()
        opt.step()
        losses.append(float(ll + lc))
    torch.cuda.synchronize()
    assert all(np.isfinite(l) for l in losses)
    gn = {n: float(p.grad.norm()) for n, p in model.named_parameters()}
    assert all(np.isfinite(v) for v in gn.values())
    assert gn["conf.0.pw.conv.0.weight"] > 0 and gn["extras.1.dw.conv.0.weight"] > 0 and gn["layer1.1.conv1.conv.0.weight"] > 0 and gn["conv1.conv.0.weight"] > 0
    # MultiBoxLoss leaves the priors of the 2x2 map without positives or mined negatives here (legitimately zero gradient there), so drive
    # every prediction with a dense synthetic loss once: each of the 269 parameter tensors must then receive a finite, non-zero gradient
    loc, conf, pri = model(x)
    g = torch.Generator(device="cuda").manual_seed(5)
    ((loc * torch.randn(loc.shape, device="cuda", generator=g)).sum() + (conf * torch.randn(conf.shape, device="cuda", generator=g)).sum()).backward()
    torch.cuda.synchronize()
    dead = [n for n, p in model.named_parameters() if not (np.isfinite(float(p.grad.norm())) and float(p.grad.norm()) > 0)]
    assert not dead, dead
    print(f"[ssdlite large@512 train] losses {losses}")
