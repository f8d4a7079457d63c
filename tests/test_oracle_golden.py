"""Pin the CPU oracle (oracle/frost_oracle.py) to golden vectors produced by the REAL reference
(tools/gen_golden.py, run in the dev container against /root/reference).  CPU-only."""
import warnings

import numpy as np
import pytest
import torch

from oracle import frost_oracle as O

warnings.filterwarnings("ignore")
torch.set_num_threads(8)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def unpack_state(g, prefix):
    """fixture arrays '<prefix>a/b/c' -> {'a.b.c': tensor}"""
    out = {}
    for k in g.files:
        if k.startswith(prefix):
            out[k[len(prefix):].replace("/", ".")] = T(g[k])
    return out


def init_from_fixture(g, seed0, quantized):
    keys = [str(k) for k in g["init_keys"]]
    nd = g["init_ndims"]
    shapes = [tuple(int(x) for x in row[:n]) for row, n in zip(g["init_shapes"], nd)]
    sd = O.synth_state(keys, shapes, seed0)
    if quantized:
        sd = {O.float_to_qat_key(k): v for k, v in sd.items()}
    return O.split_state(sd)


def check_pack(grad, pack, rtol=2e-4):
    g = grad.detach().double().numpy()
    mine = np.concatenate([[g.sum(), np.abs(g).sum(), np.sqrt((g * g).sum())], O.sample_big(g)])
    scale = max(1e-12, float(np.abs(pack[3:]).max()))
    np.testing.assert_allclose(mine[3:], pack[3:], rtol=0, atol=rtol * scale)
    np.testing.assert_allclose(mine[1:3], pack[1:3], rtol=rtol)


# ------------------------------------------------------------------------------------------ G1
def test_g1_fake_quant(golden):
    g = golden("g1_fake_quant")
    x = T(g["edge_x"]).requires_grad_(True)
    y = O.fake_quant(x, 1.0, 0, 0, 255)
    y.sum().backward()
    assert np.array_equal(y.detach().numpy(), g["edge_y"])
    assert np.array_equal(x.grad.numpy(), g["edge_mask"])
    assert g["edge_y"].tolist() == [0, 2, 2, 0, 0, 254, 255, 255, 0, 0, 255]      # SURVEY G1 known answers
    assert g["edge_mask"].tolist() == [1, 1, 1, 1, 0, 1, 0, 0, 1, 0, 0]
    for name in ("act", "wgt", "act0"):
        s, zp, qmin, qmax, seed = g[name + "_qp"]
        x = T(O.synth((4099,), int(seed)) * (3.0 if name != "wgt" else 0.5)).requires_grad_(True)
        gr = T(O.synth((4099,), int(seed) + 100))
        y = O.fake_quant(x, s, int(zp), int(qmin), int(qmax))
        y.backward(gr)
        assert np.array_equal(y.detach().numpy(), g[name + "_y"])
        assert np.array_equal(x.grad.numpy(), g[name + "_dx"])


# ------------------------------------------------------------------------------------------ G2
@pytest.mark.parametrize("ver,rule", [(0, "127.5"), (1, "127")])
def test_g2_observer(golden, ver, rule):
    g = golden("g2_observer")
    for name, kind in (("act", O.ACT), ("wgt", O.WGT)):
        scale = float(g[f"v{ver}_{name}_inscale"])
        qs = O.QState(rule=rule)
        for step in range(3):
            x = T(O.synth((3, 8, 5, 5), 200 + step) * scale * (1 + 0.3 * step) + (0.4 if name == "act" else 0.0))
            y = qs.fq_site("m", x, kind)
            row = g[f"v{ver}_{name}_traj"][step]
            mine = [float(qs.sd["m.activation_post_process.min_val"]), float(qs.sd["m.activation_post_process.max_val"]),
                    float(qs.sd["m.scale"][0]), float(qs.sd["m.zero_point"][0])]
            if ver == 0:
                assert np.array_equal(np.float32(mine), row), (name, step, mine, row)
                assert np.array_equal(y.numpy(), g[f"v{ver}_{name}_y{step}"])
            else:  # fused observer computes the scale in a different precision: last-ulp tolerance (SURVEY H-3)
                np.testing.assert_allclose(np.float32(mine), row, rtol=3e-7)


# ------------------------------------------------------------------------------------------ G3
G3 = ["stem", "pw16_96", "dw3s2_96", "dw5s2_144", "dw5s1_624", "pw624_96_lin", "pw288_1728", "pw1728_320_lin"]


@pytest.mark.parametrize("name", G3)
def test_g3_layer(golden, name):
    g = golden("g3_" + name)
    cin, cout, k, s, groups, H, N, xseed, gseed, relu, wseed = [int(v) for v in g["spec"]]
    in_scale, in_zp = float(g["in_qp"][0]), int(g["in_qp"][1])
    P, B = init_from_fixture(g, wseed, True)
    P = {"L." + k_: v for k_, v in P.items()}
    qs = O.QState({"L." + k_: v for k_, v in B.items()})
    x = ((T(g["x_idx"].astype(np.float32)) - in_zp) * in_scale).requires_grad_(True)
    for step in range(2):
        x.grad = None
        for p in P.values():
            p.grad = None
        y = O.convbn_qat(P, qs, "L", x, s, (k - 1) // 2, groups, bool(relu), True)
        gr = T(O.synth(tuple(y.shape), gseed + 50 * step))
        y.backward(gr)
        sd = qs.sd
        a = "L.conv.0.activation_post_process"
        idx = O.fq_index(y.detach(), sd[a + ".scale"][0], sd[a + ".zero_point"][0])
        assert np.array_equal(idx.numpy().astype(np.uint8), g[f"s{step}_yidx"])
        exp = unpack_state(g, f"s{step}_sd/")
        for key, v in exp.items():
            mine = sd.get("L." + key)
            if mine is None:
                assert key.endswith("enabled") or key.endswith("eps"), key
                continue
            np.testing.assert_allclose(mine.reshape(-1).double().numpy(), v.reshape(-1).double().numpy(), rtol=1e-6,
                                       atol=1e-7, err_msg=key)
        np.testing.assert_allclose(x.grad.numpy(), g[f"s{step}_dx"], rtol=1e-4, atol=1e-6)
        check_pack(P["L.conv.0.weight"].grad, g[f"s{step}_dw"])
        np.testing.assert_allclose(P["L.conv.0.bn.weight"].grad.numpy(), g[f"s{step}_dgamma"], rtol=2e-3, atol=2e-4)
        np.testing.assert_allclose(P["L.conv.0.bn.bias"].grad.numpy(), g[f"s{step}_dbeta"], rtol=1e-4, atol=1e-5)


def classifier_case(g):
    """Inputs of the classifier fixture (SURVEY 8c G3 'classifier 1280->1000'): shared with the GPU test."""
    N, H, xseed, gseed, wseed = [int(v) for v in g["spec"]]
    in_scale, in_zp = float(g["in_qp"][0]), int(g["in_qp"][1])
    w = O.synth_state(["classifier.2.weight", "classifier.2.bias"], [(1000, 1280, 1, 1), (1000,)], wseed)
    x = (T(g["x_idx"].astype(np.float32)) - in_zp) * in_scale
    return N, H, gseed, in_scale, in_zp, w, x


def test_g3_classifier(golden):
    """nnqat.Conv2d head (frostnet.py:295-299) teacher-forced: logits indices bit-exact, gradients 1e-4, observer state 1e-6."""
    g = golden("g3_classifier")
    N, H, gseed, in_scale, in_zp, w, x = classifier_case(g)
    P = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    qs = O.QState()
    x = x.requires_grad_(True)
    for step in range(2):
        x.grad = None
        for p in P.values():
            p.grad = None
        y = O.classifier_forward(P, qs, x, True)
        y.backward(T(O.synth(tuple(y.shape), gseed + 50 * step)))
        a = "classifier.2.activation_post_process"
        idx = O.fq_index(y.detach(), qs.sd[a + ".scale"][0], qs.sd[a + ".zero_point"][0]).reshape(N, 1000)
        assert np.array_equal(idx.numpy().astype(np.uint8), g[f"s{step}_yidx"])
        np.testing.assert_allclose(y.detach().reshape(N, 1000).numpy(), g[f"s{step}_y"], rtol=1e-6, atol=1e-7)
        for key, v in unpack_state(g, f"s{step}_sd/").items():
            mine = qs.sd.get(key)
            if mine is None:
                assert key.endswith("enabled") or key.endswith("eps"), key
                continue
            np.testing.assert_allclose(mine.reshape(-1).double().numpy(), v.reshape(-1).double().numpy(), rtol=1e-6, atol=1e-7, err_msg=key)
        np.testing.assert_allclose(x.grad[:, :, 0, 0].numpy(), g[f"s{step}_dx00"], rtol=1e-4, atol=1e-7)
        check_pack(P["classifier.2.weight"].grad, g[f"s{step}_dw"])
        np.testing.assert_allclose(P["classifier.2.bias"].grad.numpy(), g[f"s{step}_db"], rtol=1e-4, atol=1e-6)


# ------------------------------------------------------------------------------------------ G4
G4 = ["dw_e1", "mb", "cas_res", "cas_nores", "cas_s2"]


@pytest.mark.parametrize("name", G4)
@pytest.mark.parametrize("tag", ["q", "f"])
def test_g4_block(golden, name, tag):
    g = golden(f"g4_{name}_{tag}")
    cin, cout, k, s, e, r, H, N, xseed, gseed, wseed = [int(v) for v in g["spec"]]
    quantized = tag == "q"
    bc = O.block_cfg(cin, cout, k, e, r, s)
    P, B = init_from_fixture(g, wseed, quantized)
    P = {"B." + k_: v for k_, v in P.items()}
    qs = O.QState({"B." + k_: v for k_, v in B.items()})
    if quantized:
        x = ((T(g["x_idx"].astype(np.float32)) - float(g["in_qp"][1])) * float(g["in_qp"][0])).requires_grad_(True)
    else:
        x = T(O.synth((N, cin, H, H), xseed)).requires_grad_(True)
    for step in range(2):
        x.grad = None
        for p in P.values():
            p.grad = None
        y = O.block_forward(P, qs, "B", x, bc, quantized, True)
        y.backward(T(O.synth(tuple(y.shape), gseed + 50 * step)))
        np.testing.assert_allclose(y.detach().numpy(), g[f"s{step}_y"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(x.grad.numpy(), g[f"s{step}_dx"], rtol=1e-4, atol=1e-6)
        for key in g.files:
            if key.startswith(f"s{step}_grad/"):
                pn = "B." + key[len(f"s{step}_grad/"):].replace("/", ".")
                check_pack(P[pn].grad, g[key], rtol=5e-4)
        for key, v in unpack_state(g, f"s{step}_sd/").items():
            mine = qs.sd.get("B." + key)
            if mine is None:
                continue
            np.testing.assert_allclose(mine.reshape(-1).double().numpy(), v.reshape(-1).double().numpy(), rtol=1e-6,
                                       atol=1e-7, err_msg=key)


G4T = ["l31", "l33", "l36", "l41", "l43", "l50"]


@pytest.mark.parametrize("name", G4T)
def test_g4_block_true_shapes(golden, name):
    """The bottlenecks of FrostNet-Large's 14x14 / 7x7 stages at their TRUE shapes (the shapes the device's block kernels take): the
    oracle's indices are the reference's except on ties of the reference's own fp32 sums (<= 1 step on <= 1e-4 of the elements)."""
    g = golden(f"g4t_{name}_q")
    cin, cout, k, s, e, r, H, N, xseed, gseed, wseed = [int(v) for v in g["spec"]]
    bc = O.block_cfg(cin, cout, k, e, r, s)
    P, B = init_from_fixture(g, wseed, True)
    P = {"B." + k_: v for k_, v in P.items()}
    qs = O.QState({"B." + k_: v for k_, v in B.items()})
    x = ((T(g["x_idx"].astype(np.float32)) - float(g["in_qp"][1])) * float(g["in_qp"][0])).requires_grad_(True)
    for step in range(2):
        x.grad = None
        for p in P.values():
            p.grad = None
        y = O.block_forward(P, qs, "B", x, bc, True, True)
        y.backward(T(O.synth(tuple(y.shape), gseed + 50 * step)))
        sc, zp = float(g[f"s{step}_yqp"][0]), float(g[f"s{step}_yqp"][1])
        idx = torch.round(y.detach() / sc + zp).to(torch.int16)
        d = (idx - T(g[f"s{step}_yidx"]).to(torch.int16)).abs()
        assert int(d.max()) <= 1 and float((d > 0).float().mean()) <= 1e-4, (name, step, int(d.max()), float((d > 0).float().mean()))
        np.testing.assert_allclose(x.grad.numpy(), g[f"s{step}_dx"], rtol=1e-3, atol=1e-5 * float(np.abs(g[f"s{step}_dx"]).max()))
        for key in g.files:
            if key.startswith(f"s{step}_grad/"):
                pn = "B." + key[len(f"s{step}_grad/"):].replace("/", ".")
                check_pack(P[pn].grad, g[key], rtol=1e-3)
        for key, v in unpack_state(g, f"s{step}_sd/").items():
            mine = qs.sd.get("B." + key)
            if mine is None:
                continue
            np.testing.assert_allclose(mine.reshape(-1).double().numpy(), v.reshape(-1).double().numpy(), rtol=1e-6,
                                       atol=1e-7, err_msg=key)


# ------------------------------------------------------------------------------------------ G5
def _net(mode, quantized, seed0=5000):
    cfg = O.net_cfg(mode, 1.0)
    P, B = O.make_state(O.float_state_spec(cfg), seed0, quantized)
    return cfg, P, O.QState(B)


@pytest.mark.parametrize("mode", ["small", "large"])
def test_g5_fp32_eval(golden, mode):
    g = golden("g5_fp32_eval")
    B, res, seed, wseed = [int(v) for v in g[f"fp32_eval_{mode}_spec"]]
    cfg, P, qs = _net(mode, False, wseed)
    with torch.no_grad():
        y = O.frostnet_forward(P, qs, cfg, T(O.synth((B, 3, res, res), seed)), False, False)
    np.testing.assert_allclose(y.numpy(), g[f"fp32_eval_{mode}_logits"], rtol=1e-5, atol=1e-5)


def _check_grads(P, names, norms, sums, rtol):
    mine_n = np.array([float(P[n].grad.double().norm()) for n in names])
    np.testing.assert_allclose(mine_n, norms, rtol=rtol, atol=1e-7)


def test_g5_fp32_train(golden):
    g = golden("g5_fp32_train")
    B, res, seed, wseed = [int(v) for v in g["spec"]]
    cfg, P, qs = _net("large", False, wseed)
    y = O.frostnet_forward(P, qs, cfg, T(O.synth((B, 3, res, res), seed)), False, True)
    loss = torch.nn.functional.cross_entropy(y, T(g["target"]))
    loss.backward()
    np.testing.assert_allclose(y.detach().numpy(), g["logits"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(float(loss), float(g["loss"]), rtol=1e-6)
    names = [str(n) for n in g["param_names"]]
    assert names == list(P.keys())
    _check_grads(P, names, g["grad_norms"], g["grad_sums"], 1e-4)
    np.testing.assert_allclose(P["conv1.conv.0.weight"].grad.numpy(), g["grad_stem"], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(qs.sd["last_layer.conv.1.running_var"].numpy(), g["rv_last"], rtol=1e-5)


def test_g5_qat_large(golden):
    g = golden("g5_qat_large")
    B, res, seed, wseed = [int(v) for v in g["spec"]]
    cfg, P, qs = _net("large", True, wseed)
    tgt = T(g["target"])
    names = [str(n) for n in g["s0_param_names"]]
    assert names == list(P.keys())
    for step in range(2):
        for p in P.values():
            p.grad = None
        y = O.frostnet_forward(P, qs, cfg, T(O.synth((B, 3, res, res), seed + step)), True, True)
        loss = torch.nn.functional.cross_entropy(y, tgt)
        loss.backward()
        np.testing.assert_allclose(y.detach().numpy(), g[f"s{step}_logits"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(float(loss), float(g[f"s{step}_loss"]), rtol=1e-6)
        _check_grads(P, names, g[f"s{step}_grad_norms"], g[f"s{step}_grad_sums"], 1e-3)
        qk = [str(k) for k in g[f"s{step}_qkeys"]]
        default = {"scale": 1.0, "zero_point": 0.0, "min_val": float("inf"), "max_val": float("-inf")}
        mine = np.array([float(qs.sd[k].reshape(-1)[0]) if k in qs.sd else default[k.rsplit(".", 1)[1]] for k in qk])
        exp = g[f"s{step}_qvals"]
        fin = np.isfinite(exp)
        assert np.array_equal(mine[~fin], exp[~fin])      # never-executed FloatFunctionals stay (inf,-inf)
        np.testing.assert_allclose(mine[fin], exp[fin], rtol=1e-6, atol=1e-7)
    with torch.no_grad():
        ye = O.frostnet_forward(P, qs, cfg, T(O.synth((B, 3, res, res), 529)), True, False)
    np.testing.assert_allclose(ye.numpy(), g["eval_logits"], rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------------------------------ G6
HP = {
    "QSGD": ("QSGD", dict(lr=5e-3, momentum=0.9, weight_decay=1e-5, nesterov=True, clip_by=1e-3, toss_coin=True,
                          noise_decay=1e-2)),
    "QSGD_plain": ("QSGD", dict(lr=1e-2, momentum=0.9, weight_decay=0.0, nesterov=False, clip_by=0.0, toss_coin=False,
                                noise_decay=5e-2)),
    "QRMS": ("QRMS", dict(lr=1e-3, alpha=0.9, momentum=0.9, eps=1e-8, weight_decay=1e-5, clip_by=1e-3, toss_coin=True,
                          noise_decay=1e-2)),
    "QAdam": ("QAdam", dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4, amsgrad=False, clip_by=1e-3,
                            toss_coin=True, noise_decay=1e-2)),
    "QAdam_ams": ("QAdam", dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=True, clip_by=1e-3,
                                toss_coin=True, noise_decay=1e-2)),
    "QAdamW": ("QAdamW", dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False, clip_by=1e-3,
                              toss_coin=True, noise_decay=1e-2)),
}


@pytest.mark.parametrize("name", list(HP))
def test_g6_optimizer(golden, name):
    g = golden("g6_optimizers")
    n = int(g["n"])
    kind, hp = HP[name]
    p = T(O.synth((n,), 600)) * 0.1
    state = {}
    for step in range(6):
        grad = T(O.synth((n,), 610 + step)) * 0.01
        O.gradboost_step(kind, p, grad, state, hp, boost=step >= 3, noise=T(g[f"{name}_noise"][step]),
                         coin=T(g[f"{name}_coin"][step]))
        np.testing.assert_allclose(p.numpy(), g[f"{name}_p"][step], rtol=1e-6, atol=1e-9, err_msg=f"step {step}")
    np.testing.assert_allclose(grad.numpy(), g[f"{name}_gfinal"], rtol=1e-6, atol=1e-10)
    for k in g.files:
        if k.startswith(name + "_state_"):
            key = k[len(name) + 7:]
            v = state[key]
            v = v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            np.testing.assert_allclose(v, g[k], rtol=1e-6, atol=1e-12, err_msg=key)
    assert float(state["exp_min"].abs().max()) == 0.0        # exp_min stays 0 forever


# ------------------------------------------------------------------------------------------ G7
def test_g7_scalars(golden):
    g = golden("g7_scalars")
    for (ep, it), lr in zip(g["lr_points"], g["lr_values"]):
        assert O.cosine_lr(5e-3, 0.0, 5, 400, int(ep), int(it), 10) == pytest.approx(float(lr), rel=1e-12, abs=1e-18)
    for mode in ("large", "base", "small"):
        for wm, tag in ((1.0, "1_0"), (0.5, "0_5"), (1.25, "1_25")):
            spec = O.float_state_spec(O.net_cfg(mode, wm))
            assert [k for k, _ in spec] == [str(k) for k in g[f"{mode}_{tag}_float_keys"]]
            pspec = [(k, s) for k, s in spec if k.endswith(".weight") or k.endswith(".bias")]
            assert [k for k, _ in pspec] == [str(k) for k in g[f"{mode}_{tag}_param_names"]]
            assert [int(np.prod(s)) for _, s in pspec] == g[f"{mode}_{tag}_param_numel"].tolist()
    spec = O.float_state_spec(O.net_cfg("large", 1.0))
    groups = {0.0: 0, 1.0: 0, 0.01: 0}
    for k, s in spec:
        if k.endswith(".weight") or k.endswith(".bias"):
            groups[O.param_group_rule(s, 1.0)] += int(np.prod(s))
    assert [groups[0.0], groups[1.0], groups[0.01]] == g["large_group_counts"].tolist() == [234792, 5523232, 49032]


# ------------------------------------------------------------------------------------------ G8
def test_g8_features(golden):
    g = golden("g8_features")
    B, res, seed, wseed = [int(v) for v in g["spec"]]
    cfg = O.net_cfg("large", 1.0)
    spec = O.float_state_spec(cfg, features=True)
    assert [k for k, _ in spec] == [str(k) for k in g["keys"]]
    P, Bf = O.make_state(spec, wseed, False)
    with torch.no_grad():
        fs = O.frostnet_forward(P, O.QState(Bf), cfg, T(O.synth((B, 3, res, res), seed)), False, False, features=True)
    for i, f in enumerate(fs):
        assert list(f.shape) == g[f"f{i}_shape"].tolist()
        np.testing.assert_allclose(float(f.double().abs().sum()), float(g[f"f{i}_abssum"]), rtol=1e-5)
        np.testing.assert_allclose(f[0, :8, :4, :4].numpy(), g[f"f{i}_crop"], rtol=1e-4, atol=1e-5)


# ------------------------------------------------------------------------------------------ G9 converted int8 inference
def convert_case(g, mode):
    """State of the reference's QAT model right before torch.quantization.convert (weights by seed, everything else from the fixture)."""
    B, R, tseed, xseed, wseed = [int(v) for v in g["spec"]]
    cfg = O.net_cfg(mode, 1.0)
    P, _ = O.make_state(O.float_state_spec(cfg), wseed, True)
    qs = O.QState(unpack_state(g, "pre_sd/"))
    return cfg, P, qs, T(O.synth((B, 3, R, R), xseed))


@pytest.mark.parametrize("mode", ["small", "large"])
def test_g9_converted_inference(golden, mode):
    """oracle.converted_forward (integer restatement of convert + QNNPACK kernels) == the reference's converted model, index for index:
    every block output (CRC of the uint8 indices + stored indices), qparams, and the dequantised logits -- bit-exact."""
    import zlib
    g = golden(f"g9_convert_{mode}")
    cfg, P, qs, x = convert_case(g, mode)
    trace = []
    with torch.no_grad():
        y = O.converted_forward(P, qs, cfg, x, True, trace)
    for name, q, s, z in trace:
        key = "blk/" + name.replace(".", "/")
        assert [s, float(z)] == g[key + "/qp"].tolist(), name
        idx = q.to(torch.uint8).numpy()
        assert np.uint32(zlib.crc32(np.ascontiguousarray(idx).tobytes())) == g[key + "/crc"], name
        ref = g[key + "/idx"]
        assert np.array_equal(idx if idx.size <= 40000 else idx[:, :8, :6, :6], ref), name
    assert np.array_equal(y.numpy(), g["logits"])


def convert_case_fbgemm(g, mode):
    B, R, tseed, xseed, wseed = [int(v) for v in g["spec"]]
    cfg = O.net_cfg(mode, 1.0)
    P, _ = O.make_state(O.float_state_spec(cfg), wseed, True)
    qs = O.QState(unpack_state(g, "pre_sd/"), qconfig="fbgemm")
    return cfg, P, qs, T(O.synth((B, 3, R, R), xseed))


@pytest.mark.parametrize("mode", ["small", "large"])
def test_g13_converted_inference_fbgemm(golden, mode):
    """oracle.converted_forward(engine='fbgemm') == the reference's model prepared with the 'fbgemm' qconfig and converted on the FBGEMM engine
    (Classification/latency_check.py:221-226): per-channel weights, float-bias requantisation, float add -- every block output and the logits bit-exact."""
    import zlib
    g = golden(f"g13_convert_fbgemm_{mode}")
    cfg, P, qs, x = convert_case_fbgemm(g, mode)
    trace = []
    with torch.no_grad():
        y = O.converted_forward(P, qs, cfg, x, True, trace, engine="fbgemm")
    for name, q, s, z in trace:
        key = "blk/" + name.replace(".", "/")
        assert [s, float(z)] == g[key + "/qp"].tolist(), name
        idx = q.to(torch.uint8).numpy()
        assert np.uint32(zlib.crc32(np.ascontiguousarray(idx).tobytes())) == g[key + "/crc"], name
        ref = g[key + "/idx"]
        assert np.array_equal(idx if idx.size <= 40000 else idx[:, :8, :6, :6], ref), name
    assert np.array_equal(y.numpy(), g["logits"])


# ------------------------------------------------------------------------------------------ G10 quantizable hard-swish
def test_g10_hswish(golden):
    g = golden("g10_hswish")
    in_scale, in_zp = float(g["in_qp"][0]), int(g["in_qp"][1])
    N, C, H, W, xseed, gseed = [int(v) for v in g["spec"]]
    qs = O.QState()
    for step in range(3):
        x = ((T(g[f"s{step}_xidx"].astype(np.float32)) - in_zp) * in_scale).requires_grad_(True)
        y = O.hswish_qat(qs, "act", x)
        y.backward(T(O.synth((N, C, H, W), gseed + step)))
        np.testing.assert_array_equal(y.detach().numpy(), g[f"s{step}_y"])          # bit-exact
        np.testing.assert_array_equal(x.grad.numpy(), g[f"s{step}_dx"])
        a = "act.quant_mul1.activation_post_process"
        mine = np.float32([qs.sd[a + ".scale"][0], qs.sd[a + ".zero_point"][0], qs.sd[a + ".activation_post_process.min_val"],
                           qs.sd[a + ".activation_post_process.max_val"]])
        assert np.array_equal(mine, g[f"s{step}_qp"]), (step, mine, g[f"s{step}_qp"])


# ------------------------------------------------------------------------------------------ G12 per-channel / reduce_range (fbgemm qconfig)
G12 = ["pw16_96", "dw5s1_144", "pw312_80_lin"]


def test_g14_hswish_converted(golden):
    """The converted model's hard-swish as a table of the quint8 input index (oracle.converted_hswish_table) against the REFERENCE's `_Hswish` converted by stock
    torch and run on both CPU engines (tools/gen_golden.py g14): every table entry and the output qparams, both branches of add_scalar."""
    g = golden("g14_hswish_converted")
    branches = set()
    for ci in range(len(g["cases"])):
        for eng in ("qnnpack", "fbgemm"):
            sx, zx, sm, zm, so, zo = g[f"c{ci}_{eng}_qp"]
            tab, s_out, z_out = O.converted_hswish_table(sx, zx, sm, zm)
            assert np.array_equal(tab, g[f"c{ci}_{eng}_table"]), (ci, eng, np.nonzero(tab != g[f"c{ci}_{eng}_table"])[0][:8])
            assert s_out == so and z_out == zo, (ci, eng)
            branches.add(bool(zx - int(np.rint(3.0 / sx)) < 0))
    assert branches == {True, False}


@pytest.mark.parametrize("name", G12)
def test_g12_fbgemm_layer(golden, name):
    """The reference's 'fbgemm' qconfig (Classification/latency_check.py:221-226), QAT flavour: per-channel symmetric weights
    (MovingAveragePerChannelMinMaxObserver) + 7-bit affine activations; teacher-forced layer, 2 steps, forward + backward."""
    g = golden("g12_fbgemm_" + name)
    cin, cout, k, s, groups, H, N, xseed, gseed, relu, wseed = [int(v) for v in g["spec"]]
    in_scale, in_zp = float(g["in_qp"][0]), int(g["in_qp"][1])
    P, B = init_from_fixture(g, wseed, True)
    P = {"L." + k_: v for k_, v in P.items()}
    qs = O.QState({"L." + k_: v for k_, v in B.items()}, qconfig="fbgemm")
    x = ((T(g["x_idx"].astype(np.float32)) - in_zp) * in_scale).requires_grad_(True)
    for step in range(2):
        x.grad = None
        for p in P.values():
            p.grad = None
        y = O.convbn_qat(P, qs, "L", x, s, (k - 1) // 2, groups, bool(relu), True)
        y.backward(T(O.synth(tuple(y.shape), gseed + 50 * step)))
        sd = qs.sd
        a = "L.conv.0.activation_post_process"
        idx = O.fq_index(y.detach(), sd[a + ".scale"][0], sd[a + ".zero_point"][0])
        assert int(idx.max()) <= 127
        assert np.array_equal(idx.numpy().astype(np.uint8), g[f"s{step}_yidx"])
        for key, v in unpack_state(g, f"s{step}_sd/").items():
            mine = sd.get("L." + key)
            if mine is None:
                assert key.endswith("enabled") or key.endswith("eps"), key
                continue
            np.testing.assert_allclose(mine.reshape(-1).double().numpy(), v.reshape(-1).double().numpy(), rtol=1e-6, atol=1e-7, err_msg=key)
        np.testing.assert_allclose(x.grad.numpy(), g[f"s{step}_dx"], rtol=1e-4, atol=1e-6)
        check_pack(P["L.conv.0.weight"].grad, g[f"s{step}_dw"])
        np.testing.assert_allclose(P["L.conv.0.bn.weight"].grad.numpy(), g[f"s{step}_dgamma"], rtol=2e-3, atol=2e-4)
        np.testing.assert_allclose(P["L.conv.0.bn.bias"].grad.numpy(), g[f"s{step}_dbeta"], rtol=1e-4, atol=1e-5)


def fbgemm_eval_case(g):
    B, R, tseed, xseed, wseed = [int(v) for v in g["spec"]]
    cfg = O.net_cfg("small", 1.0)
    P, _ = O.make_state(O.float_state_spec(cfg), wseed, True)
    qs = O.QState(unpack_state(g, "pre_sd/"), qconfig="fbgemm")
    return cfg, P, qs, T(O.synth((B, 3, R, R), xseed))


def test_g12_fbgemm_small_eval(golden):
    g = golden("g12_fbgemm_small_eval")
    cfg, P, qs, x = fbgemm_eval_case(g)
    with torch.no_grad():
        y = O.frostnet_forward(P, qs, cfg, x, True, False)
    np.testing.assert_allclose(y.numpy(), g["logits"], rtol=1e-5, atol=1e-6)
