"""Per-channel mode on the device: the reference's 'fbgemm' qconfig (Classification/latency_check.py:221-226) in its
QAT flavour -- qint8 per_channel_symmetric weights with a MovingAveragePerChannelMinMaxObserver, quint8 affine activations with reduce_range
(index range 0..127) -- against fixtures generated from the reference modules (tools/gen_golden.py g12) and the oracle."""
import numpy as np
import pytest
import torch

from oracle import frost_oracle as O

pytestmark = pytest.mark.gpu
FLIP_RATE, GRAD_TOL = 5e-4, 2e-2


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def relerr(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def engine():
    import __graft_entry__ as ge
    ge.build()
    from frostnet_amd import engine
    return engine


@pytest.mark.parametrize("name", ["pw16_96", "dw5s1_144", "pw312_80_lin"])
def test_fbgemm_layer_vs_reference_golden(engine, golden, name):
    from frostnet_amd import _lib as L
    dev = "cuda"
    g = golden("g12_fbgemm_" + name)
    cin, cout, k, s, groups, H, N, xseed, gseed, relu, wseed = [int(v) for v in g["spec"]]
    keys = [str(k_) for k_ in g["init_keys"]]
    shapes = [tuple(int(x) for x in row[:n]) for row, n in zip(g["init_shapes"], g["init_ndims"])]
    sd = {k_: v.to(dev) for k_, v in O.synth_state(keys, shapes, wseed).items()}
    kind = "dw" if groups > 1 else "pw"
    E, qa = engine.Engine(dev), engine.QArena(4, dev)
    qa.t[:, L.Q_QMAX] = 127.0                                   # reduce_range activations
    E.act_qmax = 127
    w = sd["conv.0.weight"].contiguous().requires_grad_(True)
    gamma, beta = sd["conv.1.weight"].requires_grad_(True), sd["conv.1.bias"].requires_grad_(True)
    l = engine.ConvLayer("L", kind, w, gamma, beta, sd["conv.1.running_mean"], sd["conv.1.running_var"],
                         torch.zeros((), dtype=torch.int64, device=dev), None, k, s, bool(relu), qa.alloc(), qa.alloc())
    l.per_channel = True
    l.wmin = torch.full((cout,), float("inf"), device=dev)
    l.wmax = torch.full((cout,), float("-inf"), device=dev)
    E.add_layer(l)
    in_scale, in_zp = float(g["in_qp"][0]), int(g["in_qp"][1])
    qx = qa.alloc()
    qa.set_qparams(qx, in_scale, in_zp)
    for step in range(2):
        E.begin_step()
        x = E.act_from_indices(T(g["x_idx"]), qx)
        y = E.conv(l, x, training=True, observe=True)
        y.grad = engine.float_to_grad(T(O.synth((N, cout, y.h, y.w), gseed + 50 * step)).to(dev))
        yidx = y.indices().cpu()
        E.backward()
        torch.cuda.synchronize()
        assert int(yidx.max()) <= 127
        d = (yidx.to(torch.int16) - T(g[f"s{step}_yidx"]).to(torch.int16)).abs()
        assert int(d.max()) <= 1 and float((d > 0).float().mean()) <= FLIP_RATE, (name, step, int(d.max()), float((d > 0).float().mean()))
        pre = f"s{step}_sd/conv/0/"
        qy = qa.get(l.qy)
        np.testing.assert_allclose(qy["scale"], float(g[pre + "activation_post_process/scale"][0]), rtol=2e-5)
        assert qy["zero_point"] == int(g[pre + "activation_post_process/zero_point"][0])
        np.testing.assert_allclose(l.wscale[:cout].cpu().numpy(), g[pre + "weight_fake_quant/scale"], rtol=1e-6)            # per-channel scales
        np.testing.assert_allclose(l.wmin.cpu().numpy(), g[pre + "weight_fake_quant/activation_post_process/min_val"], rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(l.wmax.cpu().numpy(), g[pre + "weight_fake_quant/activation_post_process/max_val"], rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(l.rvar.cpu().numpy(), g[pre + "bn/running_var"], rtol=1e-3, atol=2e-4)
        dx = engine.grad_to_float(x.grad, x.n, x.h, x.w, x.c).cpu()
        assert relerr(dx, T(g[f"s{step}_dx"])) <= GRAD_TOL, (name, step, "dx", relerr(dx, T(g[f"s{step}_dx"])))
        pack = g[f"s{step}_dw"]
        mine = O.sample_big(l.w.grad.detach().double().cpu().numpy().reshape(-1))
        assert np.linalg.norm(mine - pack[3:]) / (np.linalg.norm(pack[3:]) + 1e-30) <= GRAD_TOL, (name, step, "dw")
        assert relerr(l.gamma.grad.cpu(), T(g[f"s{step}_dgamma"])) <= GRAD_TOL and relerr(l.beta.grad.cpu(), T(g[f"s{step}_dbeta"])) <= GRAD_TOL


def test_fbgemm_whole_net_through_the_module_surface(engine, golden):
    """`model.qconfig = get_default_qat_qconfig('fbgemm')` + prepare_qat on the module, as latency_check.py does with its zoo: per-channel weight
    buffers ([cout] scale / min_val / max_val) are aliased onto the device arrays; eval logits vs the reference golden, a training step runs."""
    from frostnet_amd import frostnet as F
    from test_oracle_golden import fbgemm_eval_case
    g = golden("g12_fbgemm_small_eval")
    cfg, P, qs, x = fbgemm_eval_case(g)
    model = F.frostnet_quant_small_1_0(drop_rate=0.0)
    F.qat_prepare(model, version=0, backend="fbgemm")
    sd = {k: v.detach().clone() for k, v in P.items()}
    sd.update({k: v.detach().clone() for k, v in qs.sd.items()})
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    model.cuda().eval()
    with torch.no_grad():
        y = model(x.cuda()).cpu()
    r = model.hip_runner()
    assert r.E.act_qmax == 127 and all(l.per_channel for l in r.E.layers)
    s_y = float(g["post_sd/classifier/2/activation_post_process/scale"][0])
    d = (y - T(g["logits"])).abs() / s_y
    print(f"[fbgemm small@64 eval] max logit delta {float(d.max()):.2f} steps, off-by-one fraction {float((d > 0.5).float().mean()):.3e}, rel {relerr(y, T(g['logits'])):.2e}")
    assert float(d.max()) <= 1.01 and relerr(y, T(g["logits"])) <= 3e-2
    k = "layer3.1.conv1.conv.0.weight_fake_quant.scale"
    assert model.state_dict()[k].shape == (model.layer3[1].conv1.conv[0].out_channels,)                       # the reference's key layout / shapes
    model.train()
    loss = model(x.cuda()).square().mean()
    loss.backward()
    assert all(torch.isfinite(p.grad).all() and float(p.grad.abs().sum()) > 0 for p in model.parameters())
    model.hip_convert()                          # a per-channel model converts to the FBGEMM engine's kernels (tests/test_gpu_convert.py holds it to the reference)
    with torch.no_grad():
        assert model(x.cuda()).shape == y.shape
