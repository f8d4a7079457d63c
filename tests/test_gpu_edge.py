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
    def conv(l, a, training, record, out=None, ldy=None):
        o = orig_conv(l, a, training, record, out, ldy)
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
