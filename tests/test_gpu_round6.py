"""Round-6 additions on the GPU (every case calls through the C ABI of libfrost_hip.so)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    import __graft_entry__ as ge
    ge.build()
    from frostnet_amd import _lib
    assert torch.cuda.is_available()
    return _lib


def _qat_model(name="frostnet_quant_large_1_0", per_channel=False):
    from frostnet_amd import frostnet as F
    torch.manual_seed(11)
    m = F.MODEL_REGISTRY[name]()
    if per_channel:
        F.qat_prepare(m, backend="fbgemm")
    else:
        F.qat_prepare(m, version=0)
    return m.cuda().train()


@pytest.mark.parametrize("name,per_channel", [("frostnet_quant_large_1_0", False), ("frostnet_quant_small_0_5", False), ("frostnet_quant_base_1_0", True)])
def test_step_prologue_three_launches_equal_the_seven(L, name, per_channel):
    """frost_step_prologue (flat workgroup map: sigma snapshot + weight range | observer / scales | packs + weight sums; csrc/frost_elem.hip) against
    frost_save_sigma + frost_weight_prep + frost_stats_init_table on the same state: every pack, weight sum, scale row, weight FakeQuantize record, sigma row and the
    statistics arena bit for bit -- over THREE consecutive steps (the observers are moving averages: torch MovingAverageMinMaxObserver, Q2 of SURVEY 8(a)), for the whole
    table and for a part of it (the runner's two-part preparation)."""
    from frostnet_amd.engine import ptr, stream
    try:
        model = _qat_model(name, per_channel)
    except TypeError:
        pytest.skip("qat_prepare has no backend switch")
    run = model.hip_runner()
    E = run.E
    E._ensure_tables()
    n = len(E.layers)
    # a weight perturbation between steps, so that the ranges move
    state = lambda: dict(q=run.qa.t.clone(), mm=[l.minmax2.clone() for l in E.layers], ws=[l.wscale.clone() for l in E.layers],
                         wmin=[l.wmin.clone() if l.per_channel else None for l in E.layers], wmax=[l.wmax.clone() if l.per_channel else None for l in E.layers])

    def restore(st):
        run.qa.t.copy_(st["q"])
        for l, a, b, c, d in zip(E.layers, st["mm"], st["ws"], st["wmin"], st["wmax"]):
            l.minmax2.copy_(a); l.wscale.copy_(b)
            if c is not None:
                l.wmin.copy_(c); l.wmax.copy_(d)

    def outputs():
        return dict(q=run.qa.t.clone(), packs=[l.wq_pack.clone() for l in E.layers], wt=[l.wt_pack.clone() if l.wt_pack is not None else None for l in E.layers],
                    wsum=[l.wsum.clone() for l in E.layers], ws=[l.wscale.clone() for l in E.layers], sig=[l.sigma.clone() for l in E.layers], stats=E._stats.clone(),
                    mm=[l.minmax2.clone() for l in E.layers])

    def scrub():
        for l in E.layers:
            l.wq_pack.fill_(77); l.wsum.fill_(-5); l.sigma.fill_(-1.0)
            if l.wt_pack is not None:
                l.wt_pack.fill_(0x3f80)
        E._stats.fill_(0x5a)

    def seven(lo, hi):
        m = hi - lo
        tab = C.c_void_p(E._table.data_ptr() + lo * C.sizeof(L.FrostWDesc))
        L.call("frost_save_sigma", tab, C.c_void_p(E._sigma_ptrs.data_ptr() + 8 * lo), m, stream())
        L.call("frost_weight_prep", tab, m, max(l.w.numel() for l in E.layers[lo:hi]), E.rule127, 1, stream())
        L.call("frost_stats_init_table", ptr(E._stats), C.c_void_p(E._cpads.data_ptr() + 4 * lo), C.c_void_p(E._offs.data_ptr() + 8 * lo), m, stream())

    def three(lo, hi):
        rows = []
        for i, l in enumerate(E.layers[lo:hi]):
            nsl = max(1, min(64, -(-l.w.numel() // 2048)))
            rows += [[i, sl, nsl, 0] for sl in range(nsl)]
        wg = torch.tensor(rows, dtype=torch.int32, device="cuda")
        L.call("frost_step_prologue", C.c_void_p(E._table.data_ptr() + lo * C.sizeof(L.FrostWDesc)), hi - lo, ptr(wg), wg.shape[0],
               C.c_void_p(E._sigma_ptrs.data_ptr() + 8 * lo), ptr(E._stats), C.c_void_p(E._cpads.data_ptr() + 4 * lo), C.c_void_p(E._offs.data_ptr() + 8 * lo),
               E.rule127, 1, stream())

    g = torch.Generator(device="cuda").manual_seed(3)
    for lo, hi in ((0, n), (n // 3, n - 2)):
        for step in range(3):
            with torch.no_grad():
                for l in E.layers:
                    l.w.mul_(1.0 + 0.05 * step).add_(torch.randn(l.w.shape, device="cuda", generator=g) * 0.01)
            st = state()
            scrub(); seven(lo, hi); a = outputs()
            restore(st)
            scrub(); three(lo, hi); b = outputs()
            torch.cuda.synchronize()
            assert torch.equal(a["q"].view(torch.int32), b["q"].view(torch.int32)), (lo, hi, step, "records")
            assert torch.equal(a["stats"], b["stats"])
            for i in range(n):
                for key in ("packs", "wsum", "ws", "sig", "mm"):
                    assert torch.equal(a[key][i].view(torch.int32) if a[key][i].dtype == torch.float32 else a[key][i],
                                       b[key][i].view(torch.int32) if b[key][i].dtype == torch.float32 else b[key][i]), (E.layers[i].name, key, lo, hi, step)
                if a["wt"][i] is not None:
                    assert torch.equal(a["wt"][i], b["wt"][i]), (E.layers[i].name, "wt_pack")
    # and the engine's own prologue takes the three-launch entry
    log = []
    L.CALL_LOG = log
    try:
        E.begin_step()
    finally:
        L.CALL_LOG = None
    assert "frost_step_prologue" in log and "frost_weight_prep" not in log


# ------------------------------------------------------------------------------------------------ conv1's reduce pass inside the image-resident depthwise backward
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64).ravel(), np.asarray(b, dtype=np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def _bf16(a):
    return (np.asarray(a).astype(np.uint16).astype(np.uint32) << 16).view(np.float32)


def _block(tmp, tag, case, env):
    out = os.path.join(tmp, f"{tag}.npz")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "block_digest.py"), out] + [str(v) for v in case], check=True, env=dict(os.environ, **env), cwd=ROOT,
                   stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    return np.load(out)


# (cin of conv1, expanded width, map, k, cout of reduce_conv, images): the stride-1 CAS bottlenecks of FrostNet-Large's 14 x 14 / 7 x 7 stages (frostnet.py:176-198: 104 -> 312
# k5, 104 -> 624 k5, 120 -> 360 k5 / k3 @14; 240 -> 1440, 240 -> 720, 288 -> 1728 k5 @7) and ragged relatives: a partial last 64-channel chunk, a K that is no multiple of 64 or
# 16, one image, k = 3 at 7 x 7
C1_CASES = [(104, 312, 14, 5, 80, 5), (104, 624, 14, 5, 96, 3), (120, 360, 14, 3, 96, 4), (240, 1440, 7, 5, 192, 6), (240, 720, 7, 5, 192, 9), (288, 1728, 7, 5, 320, 4),
            (56, 168, 14, 3, 40, 1), (72, 200, 7, 3, 48, 7), (320, 328, 7, 5, 64, 3), (40, 72, 14, 5, 24, 2)]


@pytest.mark.parametrize("case", C1_CASES, ids=lambda c: "_".join(str(v) for v in c))
def test_conv1_reduce_pass_inside_the_image_resident_depthwise_backward(case, tmp_path):
    """frost_block_dw_bwd_c1 (csrc/frost_block.hip, k_blk_dw_bwd<.., KS1 > 0>): the depthwise backward of a 14 x 14 / 7 x 7 bottleneck carries the reduce pass of the
    conv1 in front of it (CascadePreExBottleneck, /root/reference/frostnet.py:134-137; the reference's backward is autograd through nniqat.ConvBnReLU2d).  conv2 / reduce_conv
    gradients and dx of the depthwise layer do not move at all (same kernel, same draws); conv1's S1 / S2 now see the dx values BEFORE their bf16 rounding, so conv1's
    gradients move at the bf16-noise level and must not be further from the fp32-gradient mode (FROST_GRAD=fp32) than the separate reduce pass is."""
    calls = os.path.join(str(tmp_path), "calls.txt")
    on = _block(str(tmp_path), "on", case, {"FROST_SR": "0", "FROST_BLOCK_C1": "1", "FROST_BLK_C1": "3", "DIGEST_CALLS": calls})          # (7 x 7 maps too: off by default, +0.14 ms in the step)
    log = open(calls).read().split("\n")
    assert "frost_block_dw_bwd_c1" in log and "frost_block_dw_bwd" not in log, log
    red_on = log.count("frost_pw_conv_bwd") + log.count("frost_pwc_conv_bwd")
    calls2 = os.path.join(str(tmp_path), "calls2.txt")
    off = _block(str(tmp_path), "off", case, {"FROST_SR": "0", "FROST_BLOCK_C1": "0", "DIGEST_CALLS": calls2})
    log2 = open(calls2).read().split("\n")
    assert "frost_block_dw_bwd" in log2 and "frost_block_dw_bwd_c1" not in log2
    assert red_on == log2.count("frost_pw_conv_bwd") + log2.count("frost_pwc_conv_bwd") - 1, (log, log2)          # exactly conv1's reduce pass is gone
    ref = _block(str(tmp_path), "ref", case, {"FROST_GRAD": "fp32"})
    assert on["y3"].tobytes() == off["y3"].tobytes() == ref["y3"].tobytes()
    for k in ("dw2", "dgamma2", "dbeta2", "dw3", "dgamma3", "dbeta3"):
        assert _relerr(on[k], off[k]) <= 1e-4, k
    assert _relerr(_bf16(on["dx"]), _bf16(off["dx"])) <= 5e-3
    e_on = {k: _relerr(on[k], ref[k]) for k in ("dw1", "dgamma1", "dbeta1")}
    e_off = {k: _relerr(off[k], ref[k]) for k in ("dw1", "dgamma1", "dbeta1")}
    print("conv1 vs fp32-gradient mode: fold", e_on, "separate", e_off)
    assert e_on["dbeta1"] <= 1.25 * e_off["dbeta1"] + 1e-4 and e_on["dgamma1"] <= 1.25 * e_off["dgamma1"] + 1e-4 and e_on["dw1"] <= 1.25 * e_off["dw1"] + 1e-4, (e_on, e_off)
    assert _relerr(_bf16(on["dx"]), ref["dx"]) <= 1.25 * _relerr(_bf16(off["dx"]), ref["dx"]) + 1e-4


# ------------------------------------------------------------------------------------------------ mixed gradient precision
def test_mixed_gradient_mode_sits_between_bf16_and_fp32():
    """model.grad_precision = "mixed" (FROST_GRAD=mixed): the fp32-gradient kernels (csrc/frost_g32.hip) for every layer whose output map is at most 14 x 14,
    the production bf16 kernels above, one fp32 -> bf16 conversion where they meet.  Reference: loss.backward() is fp32 autograd throughout
    (Classification/utils/helper_functions.py:139-143).  Against the fp32-gradient mode of the same step (the yardstick that is itself within 1e-3 of the reference golden,
    tests/test_gpu_round4.py): every parameter gradient of layer3 ... layer5 / last_layer / classifier within 5e-3 in the mixed mode (bf16 mode: up to 5e-2 on conv1's
    dgamma), and the shallow layers no worse than the bf16 mode's own error x 1.5."""
    from frostnet_amd import frostnet as F
    from frostnet_amd.harness import CrossEntropyLoss
    torch.manual_seed(3)
    x = torch.randn(12, 3, 224, 224, device="cuda")
    t = torch.randint(0, 1000, (12,), device="cuda")
    grads = {}
    for mode in ("fp32", "mixed", "bf16"):
        torch.manual_seed(21)
        m = F.MODEL_REGISTRY["frostnet_quant_large_1_0"]()
        F.qat_prepare(m, version=0)
        m.cuda().train()
        m.classifier[1].p = 0.0
        m.grad_precision = mode
        crit = CrossEntropyLoss()
        loss = crit(m(x), t)
        loss.backward()
        torch.cuda.synchronize()
        grads[mode] = {n: p.grad.detach().double().cpu() for n, p in m.named_parameters() if p.grad is not None}
    def rel(a, b):
        return float((a - b).norm() / (b.norm() + 1e-30))
    # (the BatchNorm bias of a reduce_conv has a gradient that cancels to ~1e-7 of its terms in exact arithmetic -- the next block re-normalises: a relative error means
    # nothing there; parameters whose fp32 gradient norm is below 1e-5 are left out of the relative comparison and held to an absolute bound instead)
    tiny = [n for n in grads["fp32"] if float(grads["fp32"][n].norm()) < 1e-5]
    for n in tiny:
        assert float((grads["mixed"][n] - grads["fp32"][n]).norm()) <= 2e-2, n
    deep = [n for n in grads["fp32"] if n not in tiny and n.split(".")[0] in ("layer3", "layer4", "layer5", "last_layer", "classifier") and not n.startswith("layer3.0.s") and not n.startswith("layer3.0.conv1")]
    shallow = [n for n in grads["fp32"] if n not in deep and n not in tiny]
    worst_deep = max(rel(grads["mixed"][n], grads["fp32"][n]) for n in deep)
    worst_deep_bf16 = max(rel(grads["bf16"][n], grads["fp32"][n]) for n in deep)
    worst_sh = max(rel(grads["mixed"][n], grads["fp32"][n]) for n in shallow)
    worst_sh_bf16 = max(rel(grads["bf16"][n], grads["fp32"][n]) for n in shallow)
    print(f"mixed vs fp32: deep {worst_deep:.2e} (bf16 mode {worst_deep_bf16:.2e}), shallow {worst_sh:.2e} (bf16 mode {worst_sh_bf16:.2e})")
    assert worst_deep <= 5e-3, worst_deep
    assert worst_sh <= 1.5 * worst_sh_bf16 + 1e-3, (worst_sh, worst_sh_bf16)


# ------------------------------------------------------------------------------------------------------------------------------------------------------
# hard-swish through the float (StatAssist warm-up) kernels and bf16 inference: activation code 2 of the `relu` argument (csrc/frost_float.hip f_act /
# f_dhswish, csrc/frost_common.h hswish_f).  The definition is the stock-module network with act='hswish' (frostnet.py ConvBNHswish = the reference's
# _ConvBNHswish, Classification/models/imagenet/mobilenetv3.py:43-88) evaluated by torch on the CPU.
def _randomize_bn(model, seed):
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data = torch.rand(m.num_features, generator=g) * 0.8 + 0.6
            m.bias.data = torch.rand(m.num_features, generator=g) * 0.2 - 0.1
            m.running_mean.data = torch.randn(m.num_features, generator=g) * 0.1
            m.running_var.data = torch.rand(m.num_features, generator=g) * 0.5 + 0.5


def _rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _grad_errors(dev_model, ref_model):
    ref = {n: p.grad.double() for n, p in ref_model.named_parameters()}
    out = {}
    for n, p in dev_model.named_parameters():
        a, b = p.grad.detach().cpu().double(), ref[n]
        den = float(b.norm())
        if n.endswith(".conv.1.weight") or n.endswith(".conv.1.bias"):          # BN gamma / beta: measured against the conv's gradient norm when that is larger
            den = max(den, float(ref[n.rsplit(".conv.1.", 1)[0] + ".conv.0.weight"].norm()))
        out[n] = float((a - b).norm()) / max(den, 1e-30)
    return out


def _kink_budget(ref, run_ref):
    """hard-swish's derivative jumps by 1/2 where relu6's input crosses 0 or 6; an fp32 evaluation can land on the other side of a kink than the fp64 one
    only within round-off of it.  Candidates: relu6 inputs within 3e-6 * rms of 0 or 6; budget = 3 * sqrt(candidates / activations) norm-wise."""
    cnt = [0, 0]

    def hook(m, i):                                   # a PRE-hook: the reference's ReLU6 is in-place, afterwards its input is already clamped
        t = i[0].detach()
        eps = 3e-6 * t.pow(2).mean().sqrt()
        cnt[0] += int(((t.abs() < eps) | ((t - 6.0).abs() < eps)).sum()); cnt[1] += t.numel()
    hs = [m.register_forward_pre_hook(hook) for m in ref.modules() if isinstance(m, torch.nn.ReLU6)]
    out = run_ref()
    for h in hs:
        h.remove()
    return out, cnt[0], 3.0 * (cnt[0] / max(cnt[1], 1)) ** 0.5


HS_BLOCKS = [(32, 16, 3, 1, 1, 1), (16, 24, 3, 2, 6, 4), (80, 96, 5, 1, 6, 4), (40, 80, 5, 2, 6, 4)]


@pytest.mark.parametrize("cfg", HS_BLOCKS)
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_hswish_float_block(L, cfg, prec):
    """A hard-swish Frost bottleneck, teacher-forced, train mode, N=8 at 16x16.  fp32 mode against the fp64 stock modules: y 5e-6, running statistics 1e-5, dx and
    every parameter gradient 2e-5 + the kink budget (the bounds of the ReLU blocks, tests/test_gpu_float.py test_fp32_mode_block).  bf16 mode against the fp32 stock
    modules: y 1e-2, dx and gradients 5e-2 (hard-swish is continuous: no mask flips, so tighter than the ReLU blocks' 1e-1), running statistics 5e-3."""
    import copy
    import numpy as np
    from frostnet_amd import frostnet as F
    from frostnet_amd.float_train import FloatRunner
    cin, cout, k, s, e, r = cfg
    torch.manual_seed(11)
    m = F.CascadePreExBottleneck(cin, cout, quantized=False, kernel_size=k, stride=s, expand_ratio=e, reduce_factor=r, act="hswish")
    assert any(isinstance(x, F.ConvBNHswish) for x in m.modules())
    _randomize_bn(m, 5)
    ref = copy.deepcopy(m).double().train() if prec == "fp32" else copy.deepcopy(m).train()
    x = torch.randn(8, cin, 16, 16)
    xr = x.to(torch.float64 if prec == "fp32" else torch.float32).requires_grad_(True)
    yr, cands, budget = _kink_budget(ref, lambda: ref(xr))
    gy = torch.randn(yr.shape)
    yr.backward(gy.to(yr.dtype))
    m.cuda().train()
    run = FloatRunner.for_block(m, precision=prec)
    assert all(l.relu == (0 if l.name.endswith("reduce_conv") else 2) for l in run.layers)
    y, dx = run.block_step(x.cuda(), gy.cuda())
    torch.cuda.synchronize()
    ey, edx = _rel(y.cpu(), yr.detach()), _rel(dx.cpu(), xr.grad)
    errs = _grad_errors(m, ref)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:2]
    print(f"[hswish block {cfg} {prec}] y {ey:.1e} dx {edx:.1e} worst grad {worst[0][1]:.1e} ({worst[0][0]}); {cands} kink candidates")
    sd, sr = m.state_dict(), ref.state_dict()
    if prec == "fp32":
        assert ey <= 5e-6, ey
        assert edx <= 2e-5 + budget and worst[0][1] <= 2e-5 + budget, (edx, worst, budget)
        rt, at = 1e-5, 1e-6
    else:
        assert ey <= 1e-2, ey
        assert edx <= 5e-2 and worst[0][1] <= 5e-2, (edx, worst)
        rt, at = 5e-3, 2e-3
    for key in sr:
        if key.endswith("running_mean") or key.endswith("running_var"):
            np.testing.assert_allclose(sd[key].cpu().numpy(), sr[key].float().numpy(), rtol=rt, atol=at, err_msg=key)


def test_hswish_float_train_step_end_to_end(L):
    """One train step of the whole hard-swish network (frostnet_small_1_0, act='hswish', B=8 at 64x64) in fp32 mode against the fp64 evaluation of the stock
    modules -- the bounds of the ReLU network's gate (tests/test_gpu_float.py test_fp32_mode_train_step_end_to_end): logits 3e-5, loss 1e-6, all gradients
    1e-4 + 10x the kink budget, running statistics 2e-4, eval-mode logits 3e-5; then the bf16 mode of the same step: logits 0.3 (a sanity bound, as for ReLU: tests/test_gpu_float.py)."""
    import copy
    import numpy as np
    from frostnet_amd import frostnet as F
    torch.manual_seed(5)
    model = F.MODEL_REGISTRY["frostnet_small_1_0"](drop_rate=0.0, act="hswish")
    _randomize_bn(model, 3)
    ref = copy.deepcopy(model).double().train()
    x = torch.randn(8, 3, 64, 64)
    tgt = (torch.arange(8) * 37) % 1000
    y_ref, cands, budget = _kink_budget(ref, lambda: ref(x.double()))
    loss_ref = torch.nn.functional.cross_entropy(y_ref, tgt)
    loss_ref.backward()
    bf = copy.deepcopy(model)
    model.float_precision = "fp32"
    model.cuda().train()
    run = model.hip_runner()
    assert type(run).__name__ == "FloatRunner" and run.precision == "fp32" and run.stem.relu == 2 and run.last.relu == 2
    y = model(x.cuda())
    loss = torch.nn.functional.cross_entropy(y, tgt.cuda())
    loss.backward()
    torch.cuda.synchronize()
    cat = lambda mod: torch.cat([p.grad.detach().double().cpu().reshape(-1) for p in mod.parameters()])
    ey, eg = _rel(y.detach().cpu(), y_ref.detach()), _rel(cat(model), cat(ref))
    print(f"[hswish e2e fp32] logits {ey:.1e} gradient (all parameters) {eg:.1e}; {cands} kink candidates, budget {budget:.1e}")
    assert ey <= 3e-5, ey
    assert abs(float(loss) - float(loss_ref)) <= 1e-6 * abs(float(loss_ref))
    assert eg <= 1e-4 + 10 * budget, (eg, budget)
    sd, sr = model.state_dict(), ref.state_dict()
    for k in sr:
        if k.endswith("running_mean") or k.endswith("running_var"):
            np.testing.assert_allclose(sd[k].cpu().numpy(), sr[k].float().numpy(), rtol=2e-4, atol=1e-5, err_msg=k)
    model.eval(); ref.eval()
    with torch.no_grad():
        e, e_ref = model(x.cuda()).cpu(), ref(x.double())
    assert _rel(e, e_ref) <= 3e-5, _rel(e, e_ref)
    # bf16 mode, same step from the same initial state
    bf.cuda().train()
    yb = bf(x.cuda())
    torch.nn.functional.cross_entropy(yb, tgt.cuda()).backward()
    torch.cuda.synchronize()
    eb, egb = _rel(yb.detach().cpu(), y_ref.detach()), _rel(cat(bf), cat(ref))
    print(f"[hswish e2e bf16] logits {eb:.1e} gradient {egb:.1e}")
    assert bf.hip_runner().precision == "bf16" and eb <= 0.3 and egb <= 0.5, (eb, egb)      # sanity bounds, as for the ReLU network (train-mode amplification)


@pytest.mark.parametrize("npix,cin,cout", [(256 * 196, 104, 624), (256 * 49, 288, 1728), (256 * 49 + 37, 240, 1440), (6272, 16, 96)])
def test_hswish_bf16_inference_pointwise_layer_vs_fp64(L, npix, cin, cout):
    """frost_infer_pw with activation code 2 on both of its kernels (k_pw's bf16 mode for rows <= 256 B, the stand-alone GEMM above): operands as the kernel sees them,
    reference in fp64 -- what remains is the fp32 accumulation order and one bf16 rounding (the bound of the ReLU cases, tests/test_gpu_round3.py)."""
    from frostnet_amd import infer
    dev = "cuda"
    g = torch.Generator().manual_seed(100 + cin + cout)
    conv, bn = torch.nn.Conv2d(cin, cout, 1, bias=False), torch.nn.BatchNorm2d(cout)
    conv.weight.data = torch.randn(cout, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5
    bn.weight.data = torch.rand(cout, generator=g) * 0.8 + 0.6
    bn.bias.data = torch.rand(cout, generator=g) * 0.4 - 0.2
    bn.running_mean.data = torch.randn(cout, generator=g) * 0.1
    bn.running_var.data = torch.rand(cout, generator=g) * 0.5 + 0.5
    seq = torch.nn.Sequential(conv, bn).to(dev).eval()
    l = infer._ILayer(seq, 2, dev)
    arr = (L.FrostIDesc * 1)()
    arr[0] = l.desc()
    table = L.struct_to_tensor(arr, torch.device(dev))
    L.call("frost_infer_weight_prep", L.ptr(table), 1, L.stream())
    x = (torch.randn(npix, cin, generator=g) * 2.5).to(torch.bfloat16)            # wide enough that z crosses both kinks (-3 and +3)
    xb = torch.zeros(npix * cin + 64, dtype=torch.int16, device=dev)
    xb[: npix * cin] = x.view(torch.int16).reshape(-1).to(dev)
    y = torch.empty(npix * cout + 64, dtype=torch.int16, device=dev)
    L.call("frost_infer_pw", L.ptr(xb), L.ptr(l.pack), L.ptr(l.biasf), npix, cin, cout, 2, L.ptr(y), L.stream())
    torch.cuda.synchronize()
    out = y[: npix * cout].view(torch.bfloat16).float().view(npix, cout).cpu()
    sf = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach().cpu()
    wf = (conv.weight.detach().cpu().view(cout, cin) * sf[:, None]).to(torch.bfloat16).double()
    z = x.double() @ wf.t() + (bn.bias.detach().cpu() - bn.running_mean.detach().cpu() * sf).double()
    ref = z * (z + 3.0).clamp(0.0, 6.0) / 6.0
    assert float((z < -3).float().mean()) > 1e-3 and float((z > 3).float().mean()) > 1e-3
    err = (out.double() - ref).abs()
    tol = 2.0 ** -8 * ref.abs() + 4e-3                      # one bf16 rounding + fp32 accumulation slack (amplified by up to 1.5 = the largest slope below z = 3)
    assert float((err > tol).float().mean()) <= 1e-4, float(err.max())
    assert _rel(out, ref) <= 3e-3, _rel(out, ref)


def test_hswish_bf16_inference_network(L):
    """bf16 inference of the hard-swish network (eval mode, BatchNorm folded; stem, depthwise and 1x1 kernels with activation code 2, layer by layer) against
    the fp32 definition on the CPU: logits within 1.5e-2 norm-wise and the same argmax (the ReLU network's bound is 8e-3 at 70 layers), idempotent, and independent
    of the batch an image travels in."""
    from frostnet_amd import frostnet as F
    torch.manual_seed(7)
    model = F.MODEL_REGISTRY["frostnet_large_1_0"](act="hswish")
    _randomize_bn(model, 11)
    model.eval()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(16, 3, 224, 224, generator=g)
    with torch.no_grad():
        ref = model(x[:8])
    model.cuda()
    out = model.hip_infer_bf16(x.cuda())
    assert torch.equal(out, model.hip_infer_bf16(x.cuda()))
    small = model.hip_infer_bf16(x[:8].cuda().contiguous())
    assert _rel(out[:8], small) <= 5e-3
    rel = _rel(out[:8].cpu(), ref)
    print(f"[hswish bf16 inference] vs fp32 definition {rel:.2e}")
    assert rel <= 1.5e-2, rel
    assert int((out[:8].cpu().argmax(1) == ref.argmax(1)).sum()) == 8


def test_hswish_converted_table_vs_golden_and_oracle(L):
    """frost_hswish_converted (csrc/frost_convert.hip: the converted model's add_scalar -> relu6 -> mul -> mul_scalar as a table of the quint8 input index) on all 256
    indices: bit-equal to the REFERENCE's converted `_Hswish` on both CPU engines (tests/golden/g14, tools/gen_golden.py) and to the oracle's restatement over 300
    random (input, quant_mul1) records, including re-scaled add_scalar outputs; the output record is quant_mul1's at scale double(s) / 6."""
    import os
    import numpy as np
    from frostnet_amd import engine
    from oracle import frost_oracle as O
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g14_hswish_converted.npz"))
    dev = "cuda"
    E, qa = engine.Engine(dev), engine.QArena(4, dev)
    qx, qs, qo = qa.alloc(), qa.alloc(), qa.alloc()
    idx = torch.arange(256, dtype=torch.uint8).view(1, 256, 1, 1)

    def device_table(sx, zx, sm, zm):
        qa.set_qparams(qx, float(sx), int(zx)); qa.set_qparams(qs, float(sm), int(zm))
        x = E.act_from_indices(idx, qx)
        y = E.hswish_converted(x, qs, qo)
        torch.cuda.synchronize()
        out = qa.get(qo)
        return y.indices().cpu().flatten().numpy(), out["scale"], out["zero_point"]
    for ci in range(len(g["cases"])):
        for eng in ("qnnpack", "fbgemm"):
            sx, zx, sm, zm, so, zo = g[f"c{ci}_{eng}_qp"]
            tab, s_out, z_out = device_table(sx, zx, sm, zm)
            assert np.array_equal(tab, g[f"c{ci}_{eng}_table"]), (ci, eng)
            assert np.float32(s_out) == np.float32(so) and int(z_out) == int(zo), (ci, eng, s_out, so)
    rng = np.random.default_rng(7)
    rescaled = 0
    for _ in range(300):
        sx, zx = float(np.float32(rng.uniform(0.008, 0.25))), int(rng.integers(0, 256))
        sm, zm = float(np.float32(rng.uniform(0.004, 0.3))), int(rng.integers(0, 90))
        ref, s_ref, z_ref = O.converted_hswish_table(sx, zx, sm, zm)
        tab, s_out, z_out = device_table(sx, zx, sm, zm)
        assert np.array_equal(tab, ref), (sx, zx, sm, zm, np.nonzero(tab != ref)[0][:8])
        assert np.float32(s_out) == np.float32(s_ref) and int(z_out) == z_ref
        rescaled += zx - int(np.rint(3.0 / sx)) < 0
    assert 20 <= rescaled <= 280, rescaled


@pytest.mark.parametrize("backend", ["qnnpack", "fbgemm"])
def test_hswish_network_convert_and_export_vs_stock_torch_cpu(L, backend):
    """hip_convert() of a QAT hard-swish FrostNet (Classification/evaluate.py:130-143 on the act='hswish' network): the device's converted logits against the SAME
    module tree converted by stock torch and run by the CPU engine, loading the state_dict exported from the device (strict) -- bit for bit, per-tensor QNNPACK
    and per-channel FBGEMM flows (FBGEMM: stock torch untouched; QNNPACK: see the note on torch's relu6 below)."""
    import warnings
    from frostnet_amd import frostnet as F
    torch.manual_seed(9)

    def make():
        m = F.FrostNet(nclass=1000, mode="small", quantized=True, drop_rate=0.0, act="hswish")
        F.qat_prepare(m, version=0, **({"backend": "fbgemm"} if backend == "fbgemm" else {}))
        return m
    model = make().cuda().train()
    g = torch.Generator().manual_seed(21)
    with torch.no_grad():
        for _ in range(3):                                      # calibration: BatchNorm statistics and every observer move
            model(torch.randn(4, 3, 96, 96, generator=g).cuda())
    x = torch.randn(2, 3, 96, 96, generator=g)
    model.hip_convert()
    with torch.no_grad():
        y_dev = model(x.cuda()).cpu()
        assert torch.equal(y_dev, model(x.cuda()).cpu())
    exported = model.hip_export_converted()
    assert "conv1.act.quant_mul1.scale" in exported and "layer3.0.conv2.act.quant_mul1.zero_point" in exported
    torch.backends.quantized.engine = backend
    mc = make().cpu().eval()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mc = torch.quantization.convert(mc, inplace=False)
    missing, unexpected = mc.load_state_dict(exported, strict=True)
    assert not missing and not unexpected
    if backend == "qnnpack":
        # the installed torch (2.10) has a defect here: on the QNNPACK engine quantized::relu6 / hardtanh of a CHANNELS-LAST tensor (what every QNNPACK conv returns)
        # writes the clamped NHWC memory into an NCHW-strided result, i.e. it scrambles positions (quantized relu, add_scalar, mul and the FBGEMM engine do not; on a
        # contiguous tensor QNNPACK's relu6 is the plain clamp, which golden g14 pins from the reference's own module).  The yardstick therefore hands its ReLU6
        # modules contiguous tensors; everything else is stock.
        import types
        for mod in mc.modules():
            if type(mod).__name__ == "ReLU6":
                mod.forward = types.MethodType(lambda self, t: torch.ops.quantized.relu6(t.contiguous(), False), mod)
    with torch.no_grad():
        y_cpu = mc(x)
    torch.backends.quantized.engine = "qnnpack"
    assert torch.isfinite(y_dev).all() and float(y_dev.std()) > 0
    assert torch.equal(y_cpu, y_dev), float((y_cpu - y_dev).abs().max())


# ------------------------------------------------------------------------------------------------------------------------------------------------------
# the forward statistics pass of the wide pointwise layers in the chunked layout (csrc/frost_pwc.hip MODE 3; frost_pw_conv_fwd_fin routes to it)
@pytest.mark.parametrize("case", [(240, 1440, 7, 5, 192, 21), (288, 1728, 7, 5, 320, 9), (104, 624, 14, 5, 96, 10), (120, 360, 14, 3, 96, 7), (104, 312, 14, 5, 80, 17),
                                  (144, 864, 14, 5, 192, 6, 2)], ids=lambda c: "_".join(str(v) for v in c))
def test_chunked_forward_statistics_equal_the_tile_kernel(L, case, tmp_path):
    """conv1's batch statistics (sum, sum of squares, min, max per channel: the replicated tables added up) from k_pwc<3> against k_pw's statistics pass
    (FROST_PWC_STATS=0) on the same seeded bottleneck: bit for bit (same per-lane grouping of the fp32 square sums, integer sums across workgroups), and therefore
    every forward coefficient row, output record, running statistic and the block's output; ragged pixel counts (17 x 196 = 52 tiles + 4 pixels) included."""
    import os
    import subprocess
    import sys
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def digest(tag, env):
        out = os.path.join(str(tmp_path), f"{tag}.npz")
        log = os.path.join(str(tmp_path), f"{tag}.calls")
        subprocess.run([sys.executable, os.path.join(root, "tools", "block_digest.py"), out] + [str(v) for v in case], check=True, cwd=root,
                       env=dict(os.environ, DIGEST_CALLS=log, **env), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        return np.load(out)
    new, old = digest("pwc", {"FROST_PWC_STATS": "1"}), digest("pw", {"FROST_PWC_STATS": "0"})      # (1 = every layer the chunked kernels take; the default takes Cout >= 1024)
    for key in ("stats1_s1", "stats1_s2", "stats1_mn", "stats1_mx", "coef1_fwd", "rmean1", "rvar1", "qy1", "qy2", "qy3", "y3"):
        assert new[key].tobytes() == old[key].tobytes(), key
    assert int(np.abs(new["stats1_s1"]).max()) > 0 and int(new["stats1_mn"].min()) < int(new["stats1_mx"].max())
