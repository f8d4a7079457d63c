"""Ragged shapes through the whole device path: batch 1, odd resolutions (every stride-2 layer sees an odd extent: 65 -> 33 -> 17 -> 9 -> 5
-> 3), pixel counts that are not multiples of any tile, and the empty batch.  Fake-quant model in eval mode against the oracle on
the same state (bit-level: every logit within one quantisation step), train step sanity; float model head-of-network against the
fp32 stock-module definition."""
import copy

import numpy as np
import pytest
import torch

from oracle import frost_oracle as O

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def F():
    import __graft_entry__ as ge
    ge.build()
    from frostnet_amd import frostnet
    return frostnet


def _oracle_state(mode, steps=2, seed0=5000):
    cfg = O.net_cfg(mode, 1.0)
    P, B = O.make_state(O.float_state_spec(cfg), seed0, True)
    qs = O.QState(B)
    tgt = torch.tensor([3, 997])
    for step in range(steps):
        for p in P.values():
            p.grad = None
        y = O.frostnet_forward(P, qs, cfg, T(O.synth((2, 3, 64, 64), 520 + step)), True, True)
        torch.nn.functional.cross_entropy(y, tgt).backward()
    return cfg, P, qs


@pytest.mark.parametrize("batch,res", [(1, 224), (3, 65), (5, 97), (2, 127)])
def test_qat_eval_ragged_vs_oracle(F, batch, res):
    torch.set_num_threads(8)
    cfg, P, qs = _oracle_state("small")
    model = F.frostnet_quant_small_1_0(drop_rate=0.0)
    F.qat_prepare(model, version=0)
    sd = {k: v.detach().clone() for k, v in P.items()}
    sd.update({k: v.detach().clone() for k, v in qs.sd.items()})
    model.load_state_dict(sd, strict=False)
    model.cuda().eval()
    x = T(O.synth((batch, 3, res, res), 700 + res))
    with torch.no_grad():
        ref = O.frostnet_forward(P, qs, cfg, x, True, False)
        out = model(x.cuda()).cpu()
    scale = float(qs.sd["classifier.2.activation_post_process.scale"][0])
    d = (out - ref).abs() / scale
    assert out.shape == ref.shape and float(d.max()) <= 1.01 and _rel(out, ref) <= 3e-2, (float(d.max()), _rel(out, ref))


@pytest.mark.parametrize("batch,res", [(1, 224), (3, 65), (5, 97)])
def test_qat_train_step_ragged(F, batch, res):
    """One training step at a ragged shape: first-layer observers / statistics equal the oracle's (they depend on the input only), every
    gradient is finite and non-zero, running statistics moved."""
    cfg = O.net_cfg("small", 1.0)
    P, B = O.make_state(O.float_state_spec(cfg), 5000, True)
    qs = O.QState(B)
    x = T(O.synth((batch, 3, res, res), 800 + res))
    tgt = torch.arange(batch) * 7 % 1000
    y_ref = O.frostnet_forward(P, qs, cfg, x, True, True)
    model = F.frostnet_quant_small_1_0(drop_rate=0.0)
    spec = O.float_state_spec(cfg)
    model.load_state_dict(O.synth_state([k for k, _ in spec], [s for _, s in spec], 5000))
    F.qat_prepare(model, version=0)
    model.cuda()
    y = model(x.cuda())
    torch.nn.functional.cross_entropy(y, tgt.cuda()).backward()
    torch.cuda.synchronize()
    sd = model.state_dict()
    for k in ("quant.activation_post_process.scale", "conv1.conv.0.activation_post_process.scale", "conv1.conv.0.bn.running_var"):
        np.testing.assert_allclose(sd[k].float().cpu().numpy().reshape(-1), qs.sd[k].float().numpy().reshape(-1), rtol=2e-3, err_msg=k)
    assert y.shape == y_ref.shape and torch.isfinite(y).all()
    for n, p in model.named_parameters():
        assert torch.isfinite(p.grad).all(), n
    assert float(model.conv1.conv[0].weight.grad.abs().sum()) > 0


@pytest.mark.parametrize("batch,res", [(1, 224), (3, 65), (5, 97)])
def test_float_ragged(F, batch, res):
    torch.manual_seed(4)
    model = F.frostnet_small_1_0(drop_rate=0.0)
    ref = copy.deepcopy(model)
    x = torch.randn(batch, 3, res, res)
    caps = {}
    ref.conv1.register_forward_hook(lambda m, i, o: caps.__setitem__("stem", o.detach()))
    ref.layer2[0].register_forward_hook(lambda m, i, o: caps.__setitem__("l2", o.detach()))
    ref.train()
    y_ref = ref(x)
    y_ref.sum().backward()
    model.cuda().train()
    run = model.hip_runner()
    dev, orig_conv, orig_block = {}, run._conv, run._block
    def conv(l, a, training, record, out=None, ldy=None, **kw):
        o = orig_conv(l, a, training, record, out, ldy, **kw)
        if l.name == "conv1":
            dev["stem"] = o
        return o
    nblk = [0]
    def block(ent, a, training, record):
        o = orig_block(ent, a, training, record)
        nblk[0] += 1
        if nblk[0] == 4:                  # layer2.0: the second stride-2 depthwise
            dev["l2"] = o
        return o
    run._conv, run._block = conv, block
    y = model(x.cuda())
    y.sum().backward()
    torch.cuda.synchronize()
    run._conv, run._block = orig_conv, orig_block
    assert tuple(dev["stem"].float().shape) == tuple(caps["stem"].shape) and tuple(dev["l2"].float().shape) == tuple(caps["l2"].shape)
    assert _rel(dev["stem"].float().cpu(), caps["stem"]) <= 4e-3
    assert _rel(dev["l2"].float().cpu(), caps["l2"]) <= 4e-2
    assert y.shape == y_ref.shape and torch.isfinite(y).all()
    for n, p in model.named_parameters():
        assert torch.isfinite(p.grad).all(), n


def test_empty_batch_raises(F):
    """BatchNorm over an empty batch is an error in the reference (torch raises in train mode); the device path must fail with a
    Python exception, not a launch error."""
    q = F.frostnet_quant_small_1_0()
    F.qat_prepare(q, version=0)
    q.cuda()
    with pytest.raises((ValueError, RuntimeError)):
        q(torch.zeros(0, 3, 64, 64, device="cuda"))
    f = F.frostnet_small_1_0().cuda().train()
    with pytest.raises((ValueError, RuntimeError)):
        f(torch.zeros(0, 3, 64, 64, device="cuda"))
    torch.cuda.synchronize()
    out = f(torch.randn(2, 3, 64, 64, device="cuda"))          # the device is still usable afterwards
    assert torch.isfinite(out).all()


def test_full_size_properties_large_b512(F):
    """BASELINE config c3 at its full size (Large, B = 512, 224x224) through size-independent properties:
    (1) with the observers frozen (torch.quantization.disable_observer, Classification/evaluate.py:131-143) eval-mode inference is
        per-image independent: the logits of a 512-image batch equal, bit for bit, the logits of its first 8 images run alone, and a
        second run of the same batch is bit-identical (idempotence);
    (2) a training step is invariant under a permutation of the batch: logits permute (to within a few quantisation steps: fp32
        sum-of-squares order -> isolated index flips upstream), parameter-gradient norms are unchanged and the classifier's gradient agrees in direction."""
    torch.manual_seed(1882)
    model = F.frostnet_quant_large_1_0(drop_rate=0.0)
    F.qat_prepare(model, version=0)
    model.cuda().train()
    x = torch.randn(512, 3, 224, 224, device="cuda")
    tgt = torch.randint(0, 1000, (512,), device="cuda")
    for _ in range(2):                                      # observers / running statistics get real values
        torch.nn.functional.cross_entropy(model(x), tgt).backward()
    state = copy.deepcopy(model.state_dict())
    # (2) permutation invariance of a training step
    perm = torch.randperm(512, device="cuda")
    outs = []
    for xs, ts in ((x, tgt), (x[perm], tgt[perm])):
        model.load_state_dict(state)
        model.zero_grad(set_to_none=True)
        y = model(xs)
        torch.nn.functional.cross_entropy(y, ts).backward()
        outs.append((y.detach().clone(), {n: p.grad.detach().double().clone() for n, p in model.named_parameters()}))
    (y0, g0), (y1, g1) = outs
    scale = float(model.classifier[2].activation_post_process.scale)
    d = (y1 - y0[perm]).abs() / scale
    # isolated index flips upstream (fp32 sum-of-squares order) reach the 8-bit logits as a perturbation well below one step, which
    # still moves the ones sitting near a rounding boundary by exactly one step: bound the size, not the count
    assert float(d.max()) <= 3.01 and _rel(y1, y0[perm]) <= 2e-2, (float(d.max()), _rel(y1, y0[perm]), float((d > 0.5).float().mean()))
    # gradients: index flips multiply through a freshly initialised 8-bit network (SURVEY H-2: the reference moves 5.5e-2 when its
    # thread count changes; tests/devtools/dbg_sites.py shows 6e-5 flipped indices after the first block becoming 4e-3, 9e-2, 4e-1 after the
    # next three), so directions are only comparable at the tail of the backward; norms are stable everywhere
    assert float((g0["classifier.2.weight"] - g1["classifier.2.weight"]).norm() / g0["classifier.2.weight"].norm()) <= 5e-2
    for n in g0:
        if g0[n].dim() == 4:
            assert torch.isfinite(g1[n]).all() and 0.7 <= float(g1[n].norm() / g0[n].norm()) <= 1.4, (n, float(g1[n].norm() / g0[n].norm()))
    # (1) frozen observers: per-image independence and idempotence in eval mode
    model.load_state_dict(state)
    model.eval()
    model.apply(torch.quantization.disable_observer)
    assert model.hip_runner().observe is False
    with torch.no_grad():
        a = model(x)
        b = model(x)
        c = model(x[:8].contiguous())
    assert torch.equal(a, b)
    assert torch.equal(a[:8], c)
    model.apply(torch.quantization.enable_observer)
    assert model.hip_runner().observe is True


@pytest.mark.parametrize("mode,width", [("base", 1.25), ("large", 0.35), ("small", 0.75)])
def test_qat_width_multipliers_vs_oracle(F, mode, width):
    """The factory grid (frostnet.py:354-451: 3 modes x 5 width multipliers): channel counts off the 1.0 tables (multiples of 8 via
    _make_divisible) through the same kernels -- train-mode first step at B=4 @ 64, head of the network against the oracle."""
    tag = {1.25: "1_25", 0.35: "0_35", 0.75: "0_75"}[width]
    cfg = O.net_cfg(mode, width)
    spec = O.float_state_spec(cfg)
    P, B = O.make_state(spec, 5000, True)
    qs = O.QState(B)
    x = T(O.synth((4, 3, 64, 64), 900))
    y_ref = O.frostnet_forward(P, qs, cfg, x, True, True)
    model = F.MODEL_REGISTRY[f"frostnet_quant_{mode}_{tag}"](drop_rate=0.0)
    assert [k for k, _ in spec] == list(model.state_dict().keys())
    model.load_state_dict(O.synth_state([k for k, _ in spec], [s for _, s in spec], 5000))
    F.qat_prepare(model, version=0)
    model.cuda()
    y = model(x.cuda())
    torch.nn.functional.cross_entropy(y, torch.tensor([1, 2, 3, 4]).cuda()).backward()
    torch.cuda.synchronize()
    sd = model.state_dict()
    for k in ("quant.activation_post_process.scale", "conv1.conv.0.activation_post_process.scale", "conv1.conv.0.bn.running_var",
              "layer1.0.conv2.conv.0.activation_post_process.scale", "layer1.0.reduce_conv.conv.0.bn.running_mean"):
        np.testing.assert_allclose(sd[k].float().cpu().numpy().reshape(-1), qs.sd[k].float().numpy().reshape(-1), rtol=5e-3, atol=1e-5, err_msg=k)
    assert y.shape == y_ref.shape and torch.isfinite(y).all()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())


def test_checkpoint_resume_round_trip(F):
    """Classification/train.py:205-222 keeps `state_dict` + `optimizer` in the checkpoint: a model / optimizer pair restored from them on the
    device continues exactly like the original (forward bit-identical, same update up to atomic-order noise in the gradients)."""
    from frostnet_amd import harness as H
    from frostnet_amd.optimizer import QSGD

    def make():
        m = F.frostnet_quant_small_1_0(drop_rate=0.0)
        F.qat_prepare(m, version=0)
        m.cuda()
        o = QSGD(H.make_param_groups(m, 1e-5), lr=1e-2, momentum=0.9, nesterov=True, clip_by=1e-3, toss_coin=True, noise_decay=1e-2)
        return m, o
    torch.manual_seed(3)
    crit = torch.nn.CrossEntropyLoss()
    x = torch.randn(8, 3, 64, 64, device="cuda")
    t = torch.randint(0, 1000, (8,), device="cuda")
    m1, o1 = make()
    for _ in range(2):
        H.train_one_iter(m1, crit, o1, x, t)
    ck = {"state_dict": {k: v.detach().cpu().clone() for k, v in m1.state_dict().items()}, "optimizer": copy.deepcopy(o1.state_dict())}
    m2, o2 = make()
    m2.load_state_dict(ck["state_dict"])
    o2.load_state_dict(ck["optimizer"])
    assert o2.is_warmup == o1.is_warmup
    before = {n: p.detach().clone() for n, p in m1.named_parameters()}
    for (n, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert torch.equal(p1, p2), n
    l1, y1 = H.train_one_iter(m1, crit, o1, x, t)
    l2, y2 = H.train_one_iter(m2, crit, o2, x, t)
    assert torch.equal(y1, y2) and float(l1) == float(l2)                    # the forward is deterministic given the state
    # the update (momentum buffers, step counts, GradBoost statistics restored) is the same up to the run-to-run noise of the backward,
    # which at B=8 @ 64 is large relative to one step (see test_per_layer_finalize_path_matches_table_path): compare the update vectors
    for (n, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        if p1.dim() == 4:
            d1, d2 = (p1 - before[n]).double(), (p2 - before[n]).double()
            assert float((d1 - d2).norm() / d1.norm()) <= 0.5, (n, float((d1 - d2).norm() / d1.norm()))
    assert [int(o2.state[p]["step"]) for p in m2.parameters()] == [int(o1.state[p]["step"]) for p in m1.parameters()]


def test_dataparallel_wrapper_single_device(F):
    """Classification/train.py:88-92 wraps the model in nn.DataParallel; on one device that is a pass-through to the module, and the
    reference's access pattern `model.module.<attr>` keeps working."""
    torch.manual_seed(0)
    m = F.frostnet_quant_small_1_0(drop_rate=0.0)
    F.qat_prepare(m, version=0)
    m.cuda()
    dp = torch.nn.DataParallel(m, device_ids=[0])
    x = torch.randn(4, 3, 64, 64, device="cuda")
    y = dp(x)
    y.sum().backward()
    assert y.shape == (4, 1000) and torch.isfinite(y).all()
    assert dp.module is m and all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    f = F.frostnet_small_1_0(drop_rate=0.0).cuda()
    yf = torch.nn.DataParallel(f, device_ids=[0])(x)
    assert yf.shape == (4, 1000) and torch.isfinite(yf).all()


def test_features_backbone_c5_size(F):
    """BASELINE config c5's backbone at its size: frostnet_features Large @ 512x512 (frostnet_features.py:342-352 returns [x1, x2, x3, x5]),
    fake-quantised, train-mode first step, B = 2: tap shapes, the first two taps against the oracle (deeper ones are only bounded: index
    flips multiply with depth, SURVEY H-2), backward through all four taps finite."""
    from frostnet_amd import frostnet_features as FF
    torch.set_num_threads(16)
    cfg = O.net_cfg("large", 1.0)
    spec = O.float_state_spec(cfg, features=True)
    P, B = O.make_state(spec, 5000, True)
    qs = O.QState(B)
    x = T(O.synth((2, 3, 512, 512), 950))
    with torch.no_grad():
        ref = O.frostnet_forward(P, qs, cfg, x, True, True, features=True)
    net = FF.FrostNet(mode="large", width_mult=1.0, quantized=True)
    net.load_state_dict(O.synth_state([k for k, _ in spec], [s for _, s in spec], 5000), strict=False)
    F.qat_prepare(net, version=0)
    net.cuda()
    feats = net(x.cuda())
    assert [tuple(f.shape) for f in feats] == [(2, 24, 128, 128), (2, 40, 64, 64), (2, 96, 32, 32), (2, 320, 16, 16)]
    assert [tuple(f.shape) for f in feats] == [tuple(r.shape) for r in ref]
    tol = [2e-2, 5e-2, 0.5, 1.0]
    for i, (f, r) in enumerate(zip(feats, ref)):
        assert torch.isfinite(f).all() and _rel(f.detach().cpu(), r) <= tol[i], (i, _rel(f.detach().cpu(), r))
    sum((f * f).mean() for f in feats).backward()
    assert all(torch.isfinite(p.grad).all() for p in net.parameters())
