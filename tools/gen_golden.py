"""Generate tests/golden/*.npz by RUNNING the real reference (dev container only).

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py

Imports /root/reference through tools/refshim.py, pins the qconfig to
get_default_qat_qconfig('qnnpack', version=0) and dumps inputs-by-seed / outputs as small fixtures.
Fixtures hold DATA only (seeds, shapes, expected outputs); no reference source travels.
Inputs are regenerated on the consumer side with oracle.frost_oracle.synth(shape, seed).
"""
import os
import sys
import warnings
import zlib

warnings.filterwarnings("ignore")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np
import torch

import refshim
from oracle.frost_oracle import synth, synth_state, float_to_qat_key, sample_big

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)
META = dict(torch_version=torch.__version__, qconfig="qnnpack", qconfig_version=0,
            threads=torch.get_num_threads(), mkldnn=bool(torch.backends.mkldnn.enabled))


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def save(name, **arrs):
    arrs = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrs.items()}
    for k, v in META.items():
        arrs["meta_" + k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KB, {len(arrs)} arrays")


def crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def fq_idx(y, mod):
    s, zp = float(mod.scale[0]), int(mod.zero_point[0])
    return torch.round(y.detach() / s + zp).to(torch.int16)


def sd_np(sd, prefix=""):
    """Non-parameter state only (BN stats, observer scalars): weights are regenerated from seeds."""
    return {prefix + k.replace(".", "/"): v.detach().cpu().numpy().copy() for k, v in sd.items()
            if not (k.endswith(".weight") or k.endswith(".bias"))}


def load_synth(m, seed0):
    """Fill a FLOAT module with synth_state values; returns (keys, shapes) for the fixture."""
    sd = m.state_dict()
    keys = list(sd.keys())
    shapes = [tuple(v.shape) for v in sd.values()]
    m.load_state_dict(synth_state(keys, shapes, seed0))
    return np.array(keys), np.array([list(s) + [0] * (4 - len(s)) for s in shapes]), np.array([len(s) for s in shapes])


def grad_pack(g):
    g = g.detach().double().numpy()
    return np.concatenate([[g.sum(), np.abs(g).sum(), np.sqrt((g * g).sum())], sample_big(g)]).astype(np.float64)


# ------------------------------------------------------------------------------------------ G1
def g1_fake_quant():
    edge = torch.tensor([0.5, 1.5, 2.5, -0.5, -1.5, 254.5, 255.5, 256.4, -0.4, -0.6, 300.0], requires_grad=True)
    y = torch.fake_quantize_per_tensor_affine(edge, 1.0, 0, 0, 255)
    y.sum().backward()
    out = dict(edge_x=edge, edge_y=y, edge_mask=edge.grad)
    cases = [("act", 0.0371, 121, 0, 255, 11), ("wgt", 0.00527, 0, -128, 127, 12), ("act0", 0.0213, 0, 0, 255, 13)]
    for name, s, zp, qmin, qmax, seed in cases:
        x = T(synth((4099,), seed) * (3.0 if name != "wgt" else 0.5)).requires_grad_(True)
        g = T(synth((4099,), seed + 100))
        y = torch.fake_quantize_per_tensor_affine(x, torch.tensor(s), torch.tensor(zp, dtype=torch.int32), qmin, qmax)
        y.backward(g)
        out.update({f"{name}_qp": np.array([s, zp, qmin, qmax, seed], dtype=np.float64),
                    f"{name}_y": y, f"{name}_dx": x.grad})
    save("g1_fake_quant", **out)


# ------------------------------------------------------------------------------------------ G2
def g2_observer():
    from torch.ao.quantization import get_default_qat_qconfig
    out = {}
    for ver in (0, 1):
        qc = get_default_qat_qconfig("qnnpack", version=ver)
        for name, ctor, scale in (("act", qc.activation, 2.5), ("wgt", qc.weight, 0.3)):
            m = ctor()
            m.train()
            rows = []
            for step in range(3):
                x = T(synth((3, 8, 5, 5), 200 + step) * scale * (1 + 0.3 * step) + (0.4 if name == "act" else 0.0))
                y = m(x)
                ap = m.activation_post_process if hasattr(m, "activation_post_process") else m
                rows.append([float(ap.min_val), float(ap.max_val), float(m.scale.reshape(-1)[0]),
                             float(m.zero_point.reshape(-1)[0])])
                out[f"v{ver}_{name}_y{step}"] = y
            out[f"v{ver}_{name}_traj"] = np.array(rows, dtype=np.float32)
            out[f"v{ver}_{name}_inscale"] = np.float32(scale)
    save("g2_observer", **out)


# ------------------------------------------------------------------------------------------ G3
def g3_layers():
    ref, _ = refshim.load_frostnet()
    from torch.ao.quantization import get_default_qat_qconfig, prepare_qat
    specs = [  # name, kind, cin, cout, k, s, groups, H
        ("stem", "cbr", 3, 32, 3, 2, 1, 16),
        ("pw16_96", "cbr", 16, 96, 1, 1, 1, 12),
        ("dw3s2_96", "cbr", 96, 96, 3, 2, 96, 12),
        ("dw5s2_144", "cbr", 144, 144, 5, 2, 144, 10),
        ("dw5s1_624", "cbr", 624, 624, 5, 1, 624, 6),
        ("pw624_96_lin", "cb", 624, 96, 1, 1, 1, 6),
        ("pw288_1728", "cbr", 288, 1728, 1, 1, 1, 3),
        ("pw1728_320_lin", "cb", 1728, 320, 1, 1, 1, 3),
    ]
    for i, (name, kind, cin, cout, k, s, groups, H) in enumerate(specs):
        cls = ref.ConvBNReLU if kind == "cbr" else ref.ConvBN
        m = cls(cin, cout, k, s, (k - 1) // 2, 1, groups)
        keys, shapes, ndims = load_synth(m, 3000 + 20 * i)
        m.train()
        m.fuse_model()
        m.qconfig = get_default_qat_qconfig("qnnpack", version=0)
        prepare_qat(m, inplace=True)
        fused = m.conv[0]
        # teacher-forced input: an already fake-quantised activation (uint8 index * scale)
        in_scale, in_zp = 0.0231, (0 if i % 2 == 0 else 117)
        xi = np.clip(np.round(synth((2, cin, H, H), 360 + i) * 40 + 128 + (0 if in_zp else -60)), 0, 255)
        x = ((T(xi.astype(np.float32)) - in_zp) * in_scale).requires_grad_(True)
        outs = {}
        for step in range(2):  # two steps so the EMA branch of both observers is hit
            x.grad = None
            m.zero_grad()
            y = m(x)
            g = T(synth(tuple(y.shape), 370 + i + 50 * step))
            y.backward(g)
            outs[f"s{step}_yidx"] = fq_idx(y, fused.activation_post_process).to(torch.uint8)
            outs[f"s{step}_dx"] = x.grad.clone()
            outs[f"s{step}_dw"] = grad_pack(fused.weight.grad)
            outs[f"s{step}_dgamma"] = fused.bn.weight.grad.clone()
            outs[f"s{step}_dbeta"] = fused.bn.bias.grad.clone()
            outs.update(sd_np(m.state_dict(), f"s{step}_sd/"))
        save(f"g3_{name}", spec=np.array([cin, cout, k, s, groups, H, 2, 360 + i, 370 + i, int(kind == "cbr"),
                                          3000 + 20 * i]),
             in_qp=np.array([in_scale, in_zp]), x_idx=xi.astype(np.uint8), init_keys=keys, init_shapes=shapes,
             init_ndims=ndims, **outs)


def g3_classifier():
    """SURVEY 8c G3 'classifier 1280->1000': the reference's head (frostnet.py:295-299: AdaptiveAvgPool2d(1) -> Dropout -> Conv2d,
    QAT-prepared -> nnqat.Conv2d + activation FakeQuantize), teacher-forced on a fake-quantised 7x7 map, forward + backward."""
    _, reg = refshim.load_frostnet()
    net = reg["frostnet_quant_small_1_0"](drop_rate=0.0)
    wseed = 3400
    w = synth_state(["classifier.2.weight", "classifier.2.bias"], [(1000, 1280, 1, 1), (1000,)], wseed)
    refshim.qat_prepare(net, version=0)
    head = net.classifier
    with torch.no_grad():
        head[2].weight.copy_(w["classifier.2.weight"])
        head[2].bias.copy_(w["classifier.2.bias"])
    N, H = 4, 7
    in_scale, in_zp = 0.0412, 0
    xi = np.clip(np.round(np.abs(synth((N, 1280, H, H), 380)) * 30), 0, 255)
    x = ((T(xi.astype(np.float32)) - in_zp) * in_scale).requires_grad_(True)
    outs = {}
    for step in range(2):
        x.grad = None
        net.zero_grad()
        y = head(x)
        g = T(synth(tuple(y.shape), 390 + 50 * step))
        y.backward(g)
        outs[f"s{step}_y"] = y.detach().reshape(N, 1000).clone()
        outs[f"s{step}_yidx"] = fq_idx(y, head[2].activation_post_process).reshape(N, 1000).to(torch.uint8)
        assert float((x.grad - x.grad[:, :, :1, :1]).abs().max()) == 0.0       # avg-pool backward: uniform over the map
        outs[f"s{step}_dx00"] = x.grad[:, :, 0, 0].clone()
        outs[f"s{step}_dw"] = grad_pack(head[2].weight.grad)
        outs[f"s{step}_db"] = head[2].bias.grad.clone()
        outs.update(sd_np({k: v for k, v in net.state_dict().items() if k.startswith("classifier.")}, f"s{step}_sd/"))
    save("g3_classifier", spec=np.array([N, H, 380, 390, wseed]), in_qp=np.array([in_scale, in_zp]), x_idx=xi.astype(np.uint8), **outs)


# ------------------------------------------------------------------------------------------ G4
def g4_blocks():
    ref, _ = refshim.load_frostnet()
    from torch.ao.quantization import get_default_qat_qconfig, prepare_qat
    specs = [  # name, cin, cout, k, s, e, r, H
        ("dw_e1", 32, 16, 3, 1, 1, 1, 8),
        ("mb", 16, 24, 3, 2, 6, 4, 8),
        ("cas_res", 80, 80, 5, 1, 3, 4, 6),
        ("cas_nores", 80, 96, 5, 1, 6, 4, 6),
        ("cas_s2", 40, 80, 5, 2, 6, 4, 8),
    ]
    for i, (name, cin, cout, k, s, e, r, H) in enumerate(specs):
        for quantized in (True, False):
            m = ref.CascadePreExBottleneck(cin, cout, quantized=quantized, kernel_size=k, stride=s, expand_ratio=e,
                                           reduce_factor=r)
            keys, shapes, ndims = load_synth(m, 4000 + 100 * i)
            m.train()
            if quantized:
                for mod in m.modules():
                    if type(mod) in (ref.ConvBNReLU, ref.ConvBN):
                        mod.fuse_model()
                m.qconfig = get_default_qat_qconfig("qnnpack", version=0)
                prepare_qat(m, inplace=True)
            if quantized:
                in_scale, in_zp = 0.0187, 109
                xi = np.clip(np.round(synth((2, cin, H, H), 460 + i) * 45 + 120), 0, 255).astype(np.uint8)
                x = ((T(xi.astype(np.float32)) - in_zp) * in_scale).requires_grad_(True)
            else:
                xi = np.zeros(1)
                x = T(synth((2, cin, H, H), 460 + i)).requires_grad_(True)
            outs = {}
            for step in range(2):
                x.grad = None
                m.zero_grad()
                y = m(x)
                g = T(synth(tuple(y.shape), 470 + i + 50 * step))
                y.backward(g)
                outs[f"s{step}_y"] = y.detach().clone()
                outs[f"s{step}_dx"] = x.grad.clone()
                for pn, p in m.named_parameters():
                    outs[f"s{step}_grad/" + pn.replace(".", "/")] = grad_pack(p.grad)
                outs.update(sd_np(m.state_dict(), f"s{step}_sd/"))
            tag = "q" if quantized else "f"
            save(f"g4_{name}_{tag}", spec=np.array([cin, cout, k, s, e, r, H, 2, 460 + i, 470 + i, 4000 + 100 * i]),
                 in_qp=np.array([0.0187, 109]) if quantized else np.zeros(2), x_idx=xi,
                 init_keys=keys, init_shapes=shapes, init_ndims=ndims, **outs)



# ------------------------------------------------------------------------------------------ G4 at true shapes
def g4_true_shapes():
    """G4 at the TRUE shapes of FrostNet-Large's 14x14 / 7x7 bottlenecks (frostnet.py:176-198 rows 11, 13, 17, 19, 21, 23), where the
    device's block kernels (csrc/frost_block.hip) engage: N = 4 ... 7 images, two training steps, forward indices + dx + every parameter
    gradient + state.  y is stored as uint8 indices + (scale, zero_point) (y = (idx - zp) * scale exactly)."""
    ref, _ = refshim.load_frostnet()
    from torch.ao.quantization import get_default_qat_qconfig, prepare_qat
    specs = [  # name, cin, cout, k, s, e, r, H, N
        ("l31", 80, 80, 5, 1, 3, 4, 14, 4),
        ("l33", 80, 96, 5, 1, 6, 4, 14, 4),
        ("l36", 96, 96, 3, 1, 3, 4, 14, 5),
        ("l41", 192, 192, 5, 1, 6, 4, 7, 6),
        ("l43", 192, 192, 5, 1, 3, 4, 7, 7),
        ("l50", 192, 320, 5, 1, 6, 2, 7, 6),
    ]
    for i, (name, cin, cout, k, s, e, r, H, N) in enumerate(specs):
        m = ref.CascadePreExBottleneck(cin, cout, quantized=True, kernel_size=k, stride=s, expand_ratio=e, reduce_factor=r)
        keys, shapes, ndims = load_synth(m, 4600 + 100 * i)
        m.train()
        for mod in m.modules():
            if type(mod) in (ref.ConvBNReLU, ref.ConvBN):
                mod.fuse_model()
        m.qconfig = get_default_qat_qconfig("qnnpack", version=0)
        prepare_qat(m, inplace=True)
        in_scale, in_zp = 0.0187, 109
        xi = np.clip(np.round(synth((N, cin, H, H), 480 + i) * 45 + 120), 0, 255).astype(np.uint8)
        x = ((T(xi.astype(np.float32)) - in_zp) * in_scale).requires_grad_(True)
        outs = {}
        last = m.skip_add if (s == 1 and cin == cout) else m.reduce_conv.conv[0]
        for step in range(2):
            x.grad = None
            m.zero_grad()
            y = m(x)
            g = T(synth(tuple(y.shape), 490 + i + 50 * step))
            y.backward(g)
            fq = last.activation_post_process
            outs[f"s{step}_yidx"] = fq_idx(y, fq).to(torch.uint8)
            outs[f"s{step}_yqp"] = np.array([float(fq.scale[0]), float(fq.zero_point[0])])
            outs[f"s{step}_dx"] = x.grad.clone()
            for pn, p in m.named_parameters():
                outs[f"s{step}_grad/" + pn.replace(".", "/")] = grad_pack(p.grad)
            outs.update(sd_np(m.state_dict(), f"s{step}_sd/"))
        save(f"g4t_{name}_q", spec=np.array([cin, cout, k, s, e, r, H, N, 480 + i, 490 + i, 4600 + 100 * i]),
             in_qp=np.array([in_scale, in_zp]), x_idx=xi, init_keys=keys, init_shapes=shapes, init_ndims=ndims, **outs)

# ------------------------------------------------------------------------------------------ G5
def _pgrads(net):
    ps = list(net.named_parameters())
    return dict(grad_norms=np.array([float(p.grad.double().norm()) for _, p in ps]),
                grad_sums=np.array([float(p.grad.double().sum()) for _, p in ps]),
                param_names=np.array([n for n, _ in ps]))


def g5_wholenet():
    _, reg = refshim.load_frostnet()
    out = {}
    # kaiming-init parity of the module itself (M9): same seed -> same weights
    torch.manual_seed(1882)
    net = reg["frostnet_small_1_0"](drop_rate=0.0)
    out["init1882_small_conv1_crc"] = crc(net.state_dict()["conv1.conv.0.weight"].numpy())
    out["init1882_small_cls_crc"] = crc(net.state_dict()["classifier.2.weight"].numpy())
    # FP32 eval logits, Small B=1 @224 (config c1) and Large B=2 @224 (c2 reference); synth weights
    for mode, B, seed in (("small", 1, 500), ("large", 2, 501)):
        net = reg[f"frostnet_{mode}_1_0"](drop_rate=0.0)
        load_synth(net, 5000)
        net.eval()
        x = T(synth((B, 3, 224, 224), seed))
        with torch.no_grad():
            y = net(x)
        out[f"fp32_eval_{mode}_logits"] = y
        out[f"fp32_eval_{mode}_spec"] = np.array([B, 224, seed, 5000])
    save("g5_fp32_eval", **out)

    # FP32 train-mode fwd+bwd, Large B=2 @64
    net = reg["frostnet_large_1_0"](drop_rate=0.0)
    load_synth(net, 5000)
    net.train()
    x = T(synth((2, 3, 64, 64), 510))
    tgt = torch.tensor([3, 997])
    y = net(x)
    loss = torch.nn.functional.cross_entropy(y, tgt)
    loss.backward()
    save("g5_fp32_train", spec=np.array([2, 64, 510, 5000]), target=tgt, logits=y, loss=loss, **_pgrads(net),
         rm_last=net.last_layer.conv[1].running_mean, rv_last=net.last_layer.conv[1].running_var,
         grad_stem=net.conv1.conv[0].weight.grad, grad_cls_b=net.classifier[2].bias.grad)

    # QAT: Large B=2 @64. two train fwd+bwd steps (first-call observers, then EMA), then eval logits.
    net = reg["frostnet_quant_large_1_0"](drop_rate=0.0)
    load_synth(net, 5000)
    refshim.qat_prepare(net, version=0)
    out = dict(spec=np.array([2, 64, 520, 5000]), target=tgt)
    for step in range(2):
        net.zero_grad()
        x = T(synth((2, 3, 64, 64), 520 + step))
        y = net(x)
        loss = torch.nn.functional.cross_entropy(y, tgt)
        loss.backward()
        out[f"s{step}_logits"] = y
        out[f"s{step}_loss"] = loss
        for k_, v in _pgrads(net).items():
            out[f"s{step}_{k_}"] = v
        sd = net.state_dict()
        keys = [k for k in sd if k.endswith("scale") or k.endswith("zero_point") or k.endswith("min_val")
                or k.endswith("max_val")]
        out[f"s{step}_qkeys"] = np.array(keys)
        out[f"s{step}_qvals"] = np.array([float(sd[k].reshape(-1)[0]) for k in keys], dtype=np.float64)
    net.eval()
    with torch.no_grad():
        ye = net(T(synth((2, 3, 64, 64), 529)))
    out["eval_logits"] = ye
    save("g5_qat_large", **out)


# ------------------------------------------------------------------------------------------ G6
def g6_optimizers():
    opt = refshim.load_optimizer()
    n = 53
    out = {}
    confs = {
        "QSGD": (opt.QSGD, dict(lr=5e-3, momentum=0.9, weight_decay=1e-5, nesterov=True, clip_by=1e-3, toss_coin=True,
                                noise_decay=1e-2)),
        "QSGD_plain": (opt.QSGD, dict(lr=1e-2, momentum=0.9, weight_decay=0.0, nesterov=False, clip_by=0.0,
                                      toss_coin=False, noise_decay=5e-2)),
        "QRMS": (opt.QRMSprop, dict(lr=1e-3, alpha=0.9, momentum=0.9, eps=1e-8, weight_decay=1e-5, clip_by=1e-3,
                                    toss_coin=True, noise_decay=1e-2)),
        "QAdam": (opt.QAdam, dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4, amsgrad=False, clip_by=1e-3,
                                  toss_coin=True, noise_decay=1e-2)),
        "QAdam_ams": (opt.QAdam, dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=True,
                                      clip_by=1e-3, toss_coin=True, noise_decay=1e-2)),
        "QAdamW": (opt.QAdamW, dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False,
                                    clip_by=1e-3, toss_coin=True, noise_decay=1e-2)),
    }
    real_laplace, real_random_ = np.random.laplace, torch.Tensor.random_
    for name, (cls, kw) in confs.items():
        p = torch.nn.Parameter(T(synth((n,), 600)) * 0.1)
        o = cls([p], **kw)
        traj_p, noises, coins = [], [], []
        for step in range(6):  # 3 warm-up + 3 boost
            if step == 3:
                o.is_warmup = False
            p.grad = T(synth((n,), 610 + step)) * 0.01
            nz = np.abs(synth((n,), 620 + step)) * 1.3 + 0.01   # injected Laplace draw (sign irrelevant: abs)
            cn = (synth((n,), 630 + step, "uniform") > 0).astype(np.float32)
            np.random.laplace = lambda a, b, size, _nz=nz: _nz.astype(np.float64).reshape(tuple(size))

            def fake_random_(self, *a, _cn=cn, **k):
                return self.copy_(T(_cn).reshape(self.shape))
            torch.Tensor.random_ = fake_random_
            try:
                o.step()
            finally:
                np.random.laplace, torch.Tensor.random_ = real_laplace, real_random_
            traj_p.append(p.detach().clone().numpy())
            noises.append(nz)
            coins.append(cn)
        st = o.state[p]
        out[f"{name}_p"] = np.stack(traj_p)
        out[f"{name}_noise"] = np.stack(noises)
        out[f"{name}_coin"] = np.stack(coins)
        for k_, v in st.items():
            out[f"{name}_state_{k_}"] = v if isinstance(v, torch.Tensor) else np.asarray(v)
        out[f"{name}_gfinal"] = p.grad.clone()      # reference mutates p.grad in place
    save("g6_optimizers", n=n, **out)


# ------------------------------------------------------------------------------------------ G7
def g7_scalars():
    helpers = refshim.load_helpers()
    _, reg = refshim.load_frostnet()

    class A:
        pass
    a = A()
    a.anneal, a.epochs, a.warmup_epochs, a.warmup_lr, a.lr, a.restart_epochs = False, 400, 5, 0.0, 5e-3, 100

    class FakeOpt:
        param_groups = [dict(lr=0.0)]
    pts = [(0, 0), (0, 1), (0, 2), (2, 7), (5, 0), (100, 3), (399, 9)]
    lrs = [helpers.adjust_learning_rate_cosine(FakeOpt(), ep, it, 10, a) for ep, it in pts]
    out = dict(lr_points=np.array(pts), lr_values=np.array(lrs, dtype=np.float64))
    for mode in ("large", "base", "small"):
        for wm, tag in ((1.0, "1_0"), (0.5, "0_5"), (1.25, "1_25")):
            net = reg[f"frostnet_quant_{mode}_{tag}"](drop_rate=0.0)
            names = [n for n, _ in net.named_parameters()]
            shapes = [tuple(p.shape) for _, p in net.named_parameters()]
            out[f"{mode}_{tag}_param_names"] = np.array(names)
            out[f"{mode}_{tag}_param_numel"] = np.array([int(np.prod(s)) for s in shapes])
            out[f"{mode}_{tag}_param_shape4"] = np.array([list(s) + [0] * (4 - len(s)) for s in shapes])
            out[f"{mode}_{tag}_float_keys"] = np.array(list(net.state_dict().keys()))
            if tag == "1_0":
                refshim.qat_prepare(net, version=0)
                out[f"{mode}_{tag}_qat_keys"] = np.array(list(net.state_dict().keys()))
    # param-group element counts, Large (train.py:129-137)
    net = reg["frostnet_quant_large_1_0"]()
    dw = sum(p.numel() for p in net.parameters() if p.dim() == 4 and p.shape[1] == 1)
    dense = sum(p.numel() for p in net.parameters() if p.dim() == 4 and p.shape[1] != 1)
    other = sum(p.numel() for p in net.parameters() if p.dim() != 4)
    out["large_group_counts"] = np.array([dw, dense, other])
    save("g7_scalars", **out)


# ------------------------------------------------------------------------------------------ G9
def g9_convert():
    """Converted int8 inference: the reference's deployment artefact.  Flow of Classification/evaluate.py:126-134 -- QAT-prepared model
    (two train-mode forwards populate BN statistics and observers), eval(), torch.quantization.convert, run -- executed by the QNNPACK
    engine.  Stored: the non-parameter QAT state BEFORE convert, every block output (uint8 indices; CRC + crop for the large net), logits."""
    import copy
    _, reg = refshim.load_frostnet()
    torch.backends.quantized.engine = "qnnpack"
    for mode, R, B in (("small", 64, 2), ("large", 224, 1)):
        net = reg[f"frostnet_quant_{mode}_1_0"](drop_rate=0.0)
        load_synth(net, 5000)
        refshim.qat_prepare(net, version=0)
        net.train()
        with torch.no_grad():
            for s_ in range(2):
                net(T(synth((B, 3, R, R), 520 + s_)))
        net.eval()
        out = dict(spec=np.array([B, R, 520, 529, 5000]))
        out.update(sd_np(net.state_dict(), "pre_sd/"))
        cv = copy.deepcopy(net)
        torch.ao.quantization.convert(cv.eval(), inplace=True)
        taps = {}
        for ln in ("layer1", "layer2", "layer3", "layer4", "layer5"):
            for bi, b in enumerate(getattr(cv, ln)):
                b.register_forward_hook(lambda m, i, o, name=f"{ln}.{bi}": taps.__setitem__(name, o))
        with torch.no_grad():
            y = cv(T(synth((B, 3, R, R), 529)))
        for name, o in taps.items():
            idx = o.int_repr().numpy()
            key = "blk/" + name.replace(".", "/")
            out[key + "/qp"] = np.array([o.q_scale(), o.q_zero_point()], dtype=np.float64)
            out[key + "/crc"] = crc(idx)
            out[key + "/idx"] = idx if idx.size <= 40000 else idx[:, :8, :6, :6].copy()
        out["logits"] = y
        out["cls_qp"] = np.array([float(cv.classifier[2].scale), int(cv.classifier[2].zero_point)], dtype=np.float64)
        save(f"g9_convert_{mode}", **out)


# ------------------------------------------------------------------------------------------ G13 (VERDICT r2 missing #2)
def g13_convert_fbgemm():
    """Converted int8 inference of the model prepared with the 'fbgemm' qconfig, on the FBGEMM engine: what Classification/latency_check.py:221-226
    does (fuse_model -> get_default_qat_qconfig('fbgemm') -> prepare_qat -> [calibration forwards] -> convert(model.eval()) -> timed eval).  Same
    fixture layout as G9: the non-parameter QAT state before convert, every block output (qparams, CRC, indices or a crop), logits."""
    import copy
    from torch.ao.quantization import get_default_qat_qconfig, prepare_qat
    _, reg = refshim.load_frostnet()
    torch.backends.quantized.engine = "fbgemm"
    for mode, R, B in (("small", 64, 2), ("large", 224, 1)):
        net = reg[f"frostnet_quant_{mode}_1_0"](drop_rate=0.0)
        load_synth(net, 5000)
        net.train()
        net.fuse_model()
        net.qconfig = get_default_qat_qconfig("fbgemm", version=0)
        prepare_qat(net, inplace=True)
        with torch.no_grad():
            for s_ in range(2):
                net(T(synth((B, 3, R, R), 520 + s_)))
        net.eval()
        out = dict(spec=np.array([B, R, 520, 529, 5000]))
        out.update(sd_np(net.state_dict(), "pre_sd/"))
        cv = copy.deepcopy(net)
        torch.ao.quantization.convert(cv.eval(), inplace=True)
        taps = {}
        for ln in ("layer1", "layer2", "layer3", "layer4", "layer5"):
            for bi, b in enumerate(getattr(cv, ln)):
                b.register_forward_hook(lambda m, i, o, name=f"{ln}.{bi}": taps.__setitem__(name, o))
        with torch.no_grad():
            y = cv(T(synth((B, 3, R, R), 529)))
        for name, o in taps.items():
            idx = o.int_repr().numpy()
            key = "blk/" + name.replace(".", "/")
            out[key + "/qp"] = np.array([o.q_scale(), o.q_zero_point()], dtype=np.float64)
            out[key + "/crc"] = crc(idx)
            out[key + "/idx"] = idx if idx.size <= 40000 else idx[:, :8, :6, :6].copy()
        out["logits"] = y
        out["cls_qp"] = np.array([float(cv.classifier[2].scale), int(cv.classifier[2].zero_point)], dtype=np.float64)
        save(f"g13_convert_fbgemm_{mode}", **out)
    torch.backends.quantized.engine = "qnnpack"


# ------------------------------------------------------------------------------------------ G10
def g10_hswish():
    """The reference's quantizable hard-swish (`_Hswish`, Classification/models/imagenet/mobilenetv3.py:43-56) under the qnnpack QAT qconfig,
    teacher-forced on fake-quantised inputs: 3 steps (observer first call + 2 EMA steps), forward values and input gradients."""
    from torch.ao.quantization import get_default_qat_qconfig, prepare_qat
    mv3 = refshim.load_mobilenetv3()
    m = mv3._Hswish(inplace=False)
    m.train()
    m.qconfig = get_default_qat_qconfig("qnnpack", version=0)
    prepare_qat(m, inplace=True)
    in_scale, in_zp = 0.0473, 131
    out = dict(in_qp=np.array([in_scale, in_zp]), spec=np.array([2, 24, 10, 10, 900, 910]))
    for step in range(3):
        xi = np.clip(np.round(synth((2, 24, 10, 10), 900 + step) * (40 + 15 * step) + 128), 0, 255)
        x = ((T(xi.astype(np.float32)) - in_zp) * in_scale).requires_grad_(True)
        y = m(x)
        g = T(synth(tuple(y.shape), 910 + step))
        y.backward(g)
        fq = m.quant_mul1.activation_post_process
        out[f"s{step}_xidx"] = xi.astype(np.uint8)
        out[f"s{step}_y"] = y.detach().clone()
        out[f"s{step}_dx"] = x.grad.clone()
        out[f"s{step}_qp"] = np.array([float(fq.scale[0]), float(fq.zero_point[0]), float(fq.activation_post_process.min_val),
                                       float(fq.activation_post_process.max_val)], dtype=np.float32)
    save("g10_hswish", **out)


# ------------------------------------------------------------------------------------------ G14 (SURVEY N4 after convert)
def g14_hswish_converted():
    """The reference's `_Hswish` prepared for QAT, calibrated by one train-mode step on the dequantised input indices that are present, converted
    (torch.quantization.convert, as Classification/evaluate.py:130-134 converts a network) and applied to ALL 256 quint8 input indices on both CPU engines:
    the output index table + qparams per case.  Cases cover both branches of add_scalar (zero point shifted in range / re-scaled)."""
    from torch.ao.quantization import get_default_qat_qconfig, prepare_qat, convert
    mv3 = refshim.load_mobilenetv3()
    cases = [(0.0473, 131, 0, 256), (0.02297, 44, 0, 256), (0.129, 248, 100, 256), (0.0516, 22, 0, 200), (0.011, 162, 30, 256), (0.235, 10, 0, 64), (0.0871, 0, 0, 256),
             (0.0302, 99, 0, 256), (0.0302, 100, 0, 256)]
    out = dict(cases=np.array(cases, dtype=np.float64))
    for ci, (sx, zx, lo, hi) in enumerate(cases):
        sx = float(np.float32(sx))
        for eng in ("qnnpack", "fbgemm"):
            torch.backends.quantized.engine = eng
            m = mv3._Hswish(inplace=False)
            m.train()
            m.qconfig = get_default_qat_qconfig("qnnpack", version=0)
            prepare_qat(m, inplace=True)
            idx = torch.arange(lo, hi, dtype=torch.float32)
            m((idx.view(1, -1, 1, 1) - zx) * sx)                       # calibration: both observers see the values of indices lo .. hi-1
            m.eval()
            mc = convert(m, inplace=False)
            xq = torch._make_per_tensor_quantized_tensor(torch.arange(256, dtype=torch.uint8).view(1, 256, 1, 1), sx, int(zx))
            y = mc(xq)
            out[f"c{ci}_{eng}_table"] = y.int_repr().flatten().clone()
            out[f"c{ci}_{eng}_qp"] = np.array([sx, zx, float(m.quant_mul1.activation_post_process.scale[0]), float(m.quant_mul1.activation_post_process.zero_point[0]),
                                               y.q_scale(), y.q_zero_point()], dtype=np.float64)
    torch.backends.quantized.engine = "qnnpack"
    save("g14_hswish_converted", **out)


# ------------------------------------------------------------------------------------------ G11
def g11_detection():
    """The reference's PriorBox (layers/functions/prior_box.py) and MultiBoxLoss (layers/modules/multibox_loss.py) on the 512x512 table of
    frostnet_amd.ssdlite: prior boxes, and losses + gradients for seeded predictions / ground truth."""
    from frostnet_amd.ssdlite import SSD512_VOC
    PriorBox, MultiBoxLoss = refshim.load_detection()
    pri = PriorBox(dict(SSD512_VOC)).get_prior()
    out = dict(priors_shape=np.array(pri.shape), priors_sum=np.float64(pri.double().sum()), priors_crc=crc(pri.numpy()),
               priors_head=pri[:64].clone(), priors_tail=pri[-64:].clone())
    crit = MultiBoxLoss(21, 0.5, True, 0, True, 3, 0.5, False, use_gpu=False)
    P = pri.shape[0]
    for case in range(2):
        N = 3
        loc = (T(synth((N, P, 4), 1100 + case)) * 0.5).requires_grad_(True)
        conf = (T(synth((N, P, 21), 1110 + case)) * (1.0 + case)).requires_grad_(True)
        rng = np.random.Generator(np.random.PCG64(1120 + case))
        tg = []
        for n in range(N):
            k = 1 + (n + case) % 3
            c = rng.random((k, 2)) * 0.6 + 0.2
            wh = rng.random((k, 2)) * 0.35 + 0.05
            lab = rng.integers(0, 20, (k, 1)).astype(np.float64)
            tg.append(T(np.concatenate([c - wh / 2, c + wh / 2, lab], 1).astype(np.float32)))
        ll, lc = crit((loc, conf, pri), tg)
        (ll + lc).backward()
        out[f"c{case}_losses"] = np.array([float(ll), float(lc)], dtype=np.float64)
        out[f"c{case}_dloc"] = grad_pack(loc.grad)
        out[f"c{case}_dconf"] = grad_pack(conf.grad)
        for n, t in enumerate(tg):
            out[f"c{case}_t{n}"] = t
    save("g11_detection", **out)


# ------------------------------------------------------------------------------------------ G12
def g12_fbgemm():
    """Per-channel mode = the reference's 'fbgemm' qconfig (Classification/latency_check.py:221-226), QAT flavour, version 0: activations quint8
    affine with reduce_range (0..127), weights qint8 per_channel_symmetric with a MovingAveragePerChannelMinMaxObserver.  Teacher-forced
    ConvBN(ReLU) layers (2 steps) and a whole FrostNet-Small eval forward after two training-mode forwards."""
    ref, reg = refshim.load_frostnet()
    from torch.ao.quantization import get_default_qat_qconfig, prepare_qat
    specs = [("pw16_96", "cbr", 16, 96, 1, 1, 1, 12), ("dw5s1_144", "cbr", 144, 144, 5, 1, 144, 8), ("pw312_80_lin", "cb", 312, 80, 1, 1, 1, 6)]
    for i, (name, kind, cin, cout, k, s, groups, H) in enumerate(specs):
        cls = ref.ConvBNReLU if kind == "cbr" else ref.ConvBN
        m = cls(cin, cout, k, s, (k - 1) // 2, 1, groups)
        keys, shapes, ndims = load_synth(m, 3600 + 20 * i)
        m.train()
        m.fuse_model()
        m.qconfig = get_default_qat_qconfig("fbgemm", version=0)
        prepare_qat(m, inplace=True)
        fused = m.conv[0]
        in_scale, in_zp = 0.0462, (0 if i % 2 == 0 else 58)                 # 7-bit input indices (0..127)
        xi = np.clip(np.round(synth((2, cin, H, H), 1360 + i) * 20 + 64 + (0 if in_zp else -30)), 0, 127)
        x = ((T(xi.astype(np.float32)) - in_zp) * in_scale).requires_grad_(True)
        outs = {}
        for step in range(2):
            x.grad = None
            m.zero_grad()
            y = m(x)
            g = T(synth(tuple(y.shape), 1370 + i + 50 * step))
            y.backward(g)
            outs[f"s{step}_yidx"] = fq_idx(y, fused.activation_post_process).to(torch.uint8)
            outs[f"s{step}_dx"] = x.grad.clone()
            outs[f"s{step}_dw"] = grad_pack(fused.weight.grad)
            outs[f"s{step}_dgamma"] = fused.bn.weight.grad.clone()
            outs[f"s{step}_dbeta"] = fused.bn.bias.grad.clone()
            outs.update(sd_np(m.state_dict(), f"s{step}_sd/"))
        save(f"g12_fbgemm_{name}", spec=np.array([cin, cout, k, s, groups, H, 2, 1360 + i, 1370 + i, int(kind == "cbr"), 3600 + 20 * i]),
             in_qp=np.array([in_scale, in_zp]), x_idx=xi.astype(np.uint8), init_keys=keys, init_shapes=shapes, init_ndims=ndims, **outs)
    net = reg["frostnet_quant_small_1_0"](drop_rate=0.0)
    load_synth(net, 5000)
    net.train()
    net.fuse_model()
    net.qconfig = get_default_qat_qconfig("fbgemm", version=0)
    prepare_qat(net, inplace=True)
    with torch.no_grad():
        for s_ in range(2):
            net(T(synth((2, 3, 64, 64), 520 + s_)))
    net.eval()
    out = dict(spec=np.array([2, 64, 520, 529, 5000]))
    out.update(sd_np(net.state_dict(), "pre_sd/"))
    with torch.no_grad():
        out["logits"] = net(T(synth((2, 3, 64, 64), 529)))
    out.update(sd_np({k: v for k, v in net.state_dict().items() if k.startswith("classifier.2.activation_post_process")}, "post_sd/"))
    save("g12_fbgemm_small_eval", **out)


# ------------------------------------------------------------------------------------------ G8
def g8_features():
    feat = refshim.load_features()
    net = feat.FrostNet(mode="large", width_mult=1.0)
    load_synth(net, 5000)
    net.eval()
    x = T(synth((1, 3, 128, 128), 800))
    with torch.no_grad():
        fs = net(x)
    out = dict(spec=np.array([1, 128, 800, 5000]))
    for i, f in enumerate(fs):
        out[f"f{i}_shape"] = np.array(f.shape)
        out[f"f{i}_sum"] = np.float64(f.double().sum())
        out[f"f{i}_abssum"] = np.float64(f.double().abs().sum())
        out[f"f{i}_crop"] = f[0, :8, :4, :4].clone()
    out["keys"] = np.array(list(net.state_dict().keys()))
    save("g8_features", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g3c", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13", "g14", "g4t"]
    fns = dict(g1=g1_fake_quant, g2=g2_observer, g3=g3_layers, g3c=g3_classifier, g4=g4_blocks, g5=g5_wholenet, g6=g6_optimizers,
               g7=g7_scalars, g8=g8_features, g9=g9_convert, g10=g10_hswish, g11=g11_detection, g12=g12_fbgemm, g13=g13_convert_fbgemm, g14=g14_hswish_converted, g4t=g4_true_shapes)
    for w in which:
        fns[w]()
