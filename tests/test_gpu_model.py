"""GPU parity of the Frost bottleneck and of the whole network through the nn.Module surface (frostnet_amd.FrostNet
-> FrostRunner -> libfrost_hip.so), against the reference goldens and the CPU oracle.

Gates follow SURVEY.md H-2 / BASELINE.md section 4: teacher-forced blocks, eval-mode end-to-end and one-step training
statistics; QAT-train end-to-end logits are only sanity-bounded (the reference itself moves 5.5e-2 rel when its thread
count changes)."""
import os

import numpy as np
import pytest
import torch

from oracle import frost_oracle as O

pytestmark = pytest.mark.gpu
TAIL_NORM_TOL = 5e-2      # classifier gradient norms of the first step vs the reference golden (the logits feeding them already differ by isolated index flips)
GRAD_TOL = 5e-2      # bf16 gradient storage + bf16 MFMA operands through 4-6 chained layers


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def relerr(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def fa():
    import __graft_entry__ as ge
    ge.build()
    import frostnet_amd
    from frostnet_amd import engine, frostnet, runner
    return dict(engine=engine, frostnet=frostnet, runner=runner)


def load_float_state(module, keys, shapes, seed0):
    sd = O.synth_state(keys, shapes, seed0)
    module.load_state_dict(sd)


G4 = ["dw_e1", "mb", "cas_res", "cas_nores", "cas_s2"]


@pytest.mark.parametrize("name", G4)
def test_g4_block(fa, golden, name):
    from torch.ao.quantization import get_default_qat_qconfig, prepare_qat
    engine, F, R = fa["engine"], fa["frostnet"], fa["runner"]
    g = golden(f"g4_{name}_q")
    cin, cout, k, s, e, r, H, N, xseed, gseed, wseed = [int(v) for v in g["spec"]]
    m = F.CascadePreExBottleneck(cin, cout, quantized=True, kernel_size=k, stride=s, expand_ratio=e, reduce_factor=r)
    keys = [str(k_) for k_ in g["init_keys"]]
    shapes = [tuple(int(x) for x in row[:n]) for row, n in zip(g["init_shapes"], g["init_ndims"])]
    assert keys == list(m.state_dict().keys())            # state_dict layout parity with the reference
    load_float_state(m, keys, shapes, wseed)
    m.train()
    for mod in m.modules():
        if type(mod) in (F.ConvBNReLU, F.ConvBN):
            mod.fuse_model()
    m.qconfig = get_default_qat_qconfig("qnnpack", version=0)
    prepare_qat(m, inplace=True)
    m.cuda()
    run = R.FrostRunner.for_block(m)
    qx = run.qa.alloc()
    in_scale, in_zp = float(g["in_qp"][0]), int(g["in_qp"][1])
    run.qa.set_qparams(qx, in_scale, in_zp)
    xi = T(g["x_idx"])
    xf = (xi.float() - in_zp) * in_scale
    qx[4], qx[5] = float(xf.min()), float(xf.max())
    for step in range(2):
        run.E.begin_step()
        x = run.E.act_from_indices(xi, qx)
        y = run.block_forward(run.block, x, True, True)
        yf = y.dequant().cpu()
        ysc = float(run.qa.get(y.q)["scale"])
        gr = T(O.synth(tuple(g[f"s{step}_y"].shape), gseed + 50 * step)).cuda()
        y.grad = engine.float_to_grad(gr)
        run.bind_grads()
        run.E.backward()
        torch.cuda.synchronize()
        d = (yf - T(g[f"s{step}_y"])).abs() / ysc
        assert float(d.max()) <= 2.01 and float((d > 0.5).float().mean()) <= 1e-2, (name, step, float(d.max()), float((d > 0.5).float().mean()))
        dx = engine.grad_to_float(x.grad, x.n, x.h, x.w, x.c).cpu()
        assert relerr(dx, T(g[f"s{step}_dx"])) <= GRAD_TOL, (name, step, "dx", relerr(dx, T(g[f"s{step}_dx"])))
        for pn, p in m.named_parameters():
            pack = g[f"s{step}_grad/" + pn.replace(".", "/")]
            mine = O.sample_big(p.grad.detach().double().cpu().numpy().reshape(-1))
            err = np.linalg.norm(mine - pack[3:]) / (np.linalg.norm(pack[3:]) + 1e-30)
            assert err <= GRAD_TOL, (name, step, pn, err)
        # observer / BN state written back into the module's own buffers (state_dict parity)
        sd = m.state_dict()
        for key in g.files:
            if key.startswith(f"s{step}_sd/") and (key.endswith("scale") or key.endswith("running_var")):
                mk = key[len(f"s{step}_sd/"):].replace("/", ".")
                np.testing.assert_allclose(sd[mk].detach().float().cpu().numpy().reshape(-1), g[key].reshape(-1), rtol=2e-3, atol=1e-5, err_msg=mk)


G4T = ["l31", "l33", "l36", "l41", "l43", "l50"]
G4T_GRAD = 2e-2       # every gradient of the block against an fp64 evaluation of the reference's formulas (bf16 gradient storage through 4-6 chained layers)


@pytest.mark.parametrize("name", G4T)
def test_g4_block_true_shapes(fa, golden, name):
    """The 14x14 / 7x7 bottlenecks of FrostNet-Large at their TRUE shapes (frostnet.py:176-198), i.e. with the block kernels of csrc/frost_block.hip
    ENGAGED (asserted: conv1 emit + conv2 statistics, conv2 emit + reduce_conv GEMM, image-resident depthwise backward) -- against the REFERENCE
    golden (tests/golden/g4t_*.npz: forward indices, dx, every parameter gradient, observer / BatchNorm state, two steps) and, for the gradients,
    against an fp64 evaluation of the same formulas by the oracle (the yardstick of tests/test_gpu_prod.py: the reference's own fp32 sums are
    1e-3 ... 1e-2 from the exact value at these pixel counts)."""
    from torch.ao.quantization import get_default_qat_qconfig, prepare_qat
    from frostnet_amd import _lib as L
    engine, F, R = fa["engine"], fa["frostnet"], fa["runner"]
    g = golden(f"g4t_{name}_q")
    cin, cout, k, s, e, r, H, N, xseed, gseed, wseed = [int(v) for v in g["spec"]]
    m = F.CascadePreExBottleneck(cin, cout, quantized=True, kernel_size=k, stride=s, expand_ratio=e, reduce_factor=r)
    keys = [str(k_) for k_ in g["init_keys"]]
    shapes = [tuple(int(x) for x in row[:n]) for row, n in zip(g["init_shapes"], g["init_ndims"])]
    assert keys == list(m.state_dict().keys())
    load_float_state(m, keys, shapes, wseed)
    # fp64 yardstick: the oracle on the same state in double precision
    sd64 = {O.float_to_qat_key(k_): (v.clone().double() if v.is_floating_point() else v.clone()) for k_, v in O.synth_state(keys, shapes, wseed).items()}
    P64, B64 = O.split_state(sd64)
    P64 = {"B." + k_: v for k_, v in P64.items()}
    qs64 = O.QState({"B." + k_: v for k_, v in B64.items()})
    bc = O.block_cfg(cin, cout, k, e, r, s)
    m.train()
    for mod in m.modules():
        if type(mod) in (F.ConvBNReLU, F.ConvBN):
            mod.fuse_model()
    m.qconfig = get_default_qat_qconfig("qnnpack", version=0)
    prepare_qat(m, inplace=True)
    m.cuda()
    run = R.FrostRunner.for_block(m)
    qx = run.qa.alloc()
    in_scale, in_zp = float(g["in_qp"][0]), int(g["in_qp"][1])
    run.qa.set_qparams(qx, in_scale, in_zp)
    xi = T(g["x_idx"])
    xf = (xi.float() - in_zp) * in_scale
    qx[4], qx[5] = float(xf.min()), float(xf.max())
    x64 = ((xi.double() - in_zp) * in_scale).requires_grad_(True)
    for step in range(2):
        gr = T(O.synth((N, cout, H, H), gseed + 50 * step))
        x64.grad = None
        for p in P64.values():
            p.grad = None
        y64 = O.block_forward(P64, qs64, "B", x64, bc, True, True)
        y64.backward(gr.bfloat16().double())                      # the device receives bf16 gradients
        L.CALL_LOG = []
        try:
            run.E.begin_step()
            x = run.E.act_from_indices(xi, qx)
            y = run.block_forward(run.block, x, True, True)
            yidx = y.indices().cpu()
            y.grad = engine.float_to_grad(gr.cuda())
            run.bind_grads()
            run.E.backward()
            torch.cuda.synchronize()
            log = list(L.CALL_LOG)
        finally:
            L.CALL_LOG = None
        # the kernels the bench times at these stages are the ones under test
        if not os.environ.get("FROST_G4T_ANY_PATH"):          # (dev knob: the same comparison with the block kernels switched off by FROST_BLOCK_*=0)
            assert ("frost_block_expand_dw_stats" in log) or ("frost_block_dw_stats" in log), log
            # (round 6: at 14 x 14 the image-resident depthwise backward carries conv1's reduce pass -- frost_block_dw_bwd_c1 -- by default)
            assert "frost_block_dw_reduce" in log and ("frost_block_dw_bwd" in log or "frost_block_dw_bwd_c1" in log) and "frost_block_dw_bwd_reduce" in log, log
            if str(name).startswith("l3"):          # the 14 x 14 bottlenecks
                assert "frost_block_dw_bwd_c1" in log, log
            assert "frost_dw_conv_fwd" not in log and "frost_dw_dgrad" not in log, log
            if engine._SQ_BWD_CAT:            # (A/B switch FROST_SQ_BWD_CAT=1: quant_cat's backward + the squeeze_conv's reduce pass as one launch)
                assert "frost_sq_bwd_cat" in log and "frost_cat_bwd" not in log, log
            if cin == cout and s == 1:        # residual block: skip_add's backward rides in the reduce_conv's element-wise passes (no stand-alone launch)
                assert log.count("frost_pw_ew_add_bwd") == 2 and "frost_add_bwd" not in log, log
        d = (yidx.to(torch.int16) - T(g[f"s{step}_yidx"]).to(torch.int16)).abs()
        flips = float((d > 0).float().mean())
        assert int(d.max()) <= 2 and flips <= 2e-3, (name, step, int(d.max()), flips)
        dx = engine.grad_to_float(x.grad, x.n, x.h, x.w, x.c).cpu()
        e_ref, e_64 = relerr(dx, T(g[f"s{step}_dx"])), relerr(dx, x64.grad)
        print(f"[g4t {name} step {step}] index flips {flips:.2e} (max {int(d.max())}); dx vs reference {e_ref:.2e}, vs fp64 {e_64:.2e}")
        # Yardsticks: the REFERENCE golden (fp32 autograd) and the oracle in fp64.  Neither alone is the truth for a chained block: the fp64 forward lands on
        # different indices than the reference's fp32 forward at ties (then ITS gradients differ through the masks: l33, 3-5e-2 apart), and the reference's fp32
        # sums carry their own cancellation noise.  A gradient passes if it is within G4T_GRAD of EITHER.
        bad = []
        if min(e_64, e_ref) > G4T_GRAD:
            bad.append(("dx", e_ref, e_64))
        for pn, p in m.named_parameters():
            pack = g[f"s{step}_grad/" + pn.replace(".", "/")]
            mine = p.grad.detach().double().cpu()
            e_ref = np.linalg.norm(O.sample_big(mine.numpy().reshape(-1)) - pack[3:]) / (np.linalg.norm(pack[3:]) + 1e-30)
            p64 = P64["B." + pn]
            e_64 = relerr(mine, p64.grad)
            tol = G4T_GRAD
            if pn.endswith("bn.bias") and "reduce_conv" not in pn:
                # dbeta of a layer followed by another BatchNorm is mathematically ~0 (all rounding noise): bounded against the scale of dgamma instead
                gam = P64["B." + pn.replace("bn.bias", "bn.weight")].grad
                e_64 = float((mine - p64.grad).norm() / (max(float(p64.grad.norm()), float(gam.norm())) + 1e-30))
            if pn.startswith("conv1.") and ".bn." in pn:
                # conv1 feeds a DEPTHWISE conv + train-mode BatchNorm: the loss is invariant to a per-channel scale / shift of conv1's output (up to border
                # and clamping effects), so its dgamma / dbeta are small residuals of sums that cancel -- the bf16 storage of the incoming gradient
                # (one part in 2^9 of the un-cancelled sum) reads as 2-3e-2 of the residual; same number on the layer-by-layer kernels (FROST_BLOCK_*=0)
                tol = 5e-2
            print(f"    {pn:40s} vs reference {e_ref:.2e}, vs fp64 {e_64:.2e}")
            if min(e_64, e_ref) > tol:
                bad.append((pn, e_ref, e_64))
        assert not bad, (name, step, bad)
        sd = m.state_dict()
        for key in g.files:
            if key.startswith(f"s{step}_sd/") and (key.endswith("scale") or key.endswith("running_var") or key.endswith("running_mean") or key.endswith("min_val") or key.endswith("max_val")):
                mk = key[len(f"s{step}_sd/"):].replace("/", ".")
                np.testing.assert_allclose(sd[mk].detach().float().cpu().numpy().reshape(-1), g[key].reshape(-1), rtol=2e-3, atol=2e-4, err_msg=mk)


def _oracle_state_after_train(mode, res, steps, seed0=5000):
    cfg = O.net_cfg(mode, 1.0)
    P, B = O.make_state(O.float_state_spec(cfg), seed0, True)
    qs = O.QState(B)
    tgt = torch.tensor([3, 997])
    for step in range(steps):
        for p in P.values():
            p.grad = None
        y = O.frostnet_forward(P, qs, cfg, T(O.synth((2, 3, res, res), 520 + step)), True, True)
        torch.nn.functional.cross_entropy(y, tgt).backward()
    return cfg, P, qs


@pytest.mark.parametrize("mode,res", [("small", 64), ("large", 64), ("large", 224)])
def test_eval_end_to_end_vs_oracle(fa, mode, res):
    """Same trained-for-2-steps state on both sides -> eval-mode logits (BN frozen, observers live as in the reference).
    Stated tolerance: every logit within ONE quantisation step of the oracle's (the logits are themselves 8-bit
    fake-quantised, step ~0.02, so a single step on ~5% of them already reads as ~1e-2 norm-wise) and rel-err <= 3e-2."""
    F = fa["frostnet"]
    torch.set_num_threads(8)
    cfg, P, qs = _oracle_state_after_train(mode, 64, 2)
    model = F.MODEL_REGISTRY[f"frostnet_quant_{mode}_1_0"](drop_rate=0.0)
    F.qat_prepare(model, version=0)
    sd = {k: v.detach().clone() for k, v in P.items()}
    sd.update({k: v.detach().clone() for k, v in qs.sd.items()})
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(("skip_add" in k or "quant_cat" in k or k.endswith("enabled") or k.endswith("eps")) for k in missing), missing
    model.cuda().eval()
    x = T(O.synth((2, 3, res, res), 529))
    with torch.no_grad():
        ref = O.frostnet_forward(P, qs, cfg, x, True, False)
        out = model(x.cuda()).cpu()
    scale = float(qs.sd["classifier.2.activation_post_process.scale"][0])
    d = (out - ref).abs() / scale
    print(f"[{mode}@{res}] eval logits: max index delta {float(d.max()):.2f}, frac differing {float((d > 0.5).float().mean()):.4f}, rel {relerr(out, ref):.2e}")
    assert relerr(out, ref) <= 3e-2 and float(d.max()) <= 1.01


def test_train_step_large(fa, golden):
    """One QAT training step through the public surface: loss.backward() + QSGD.step(); checked against the golden
    (reference) step loosely at the logits (chaotic, SURVEY H-2) and tightly at the first-layer observers."""
    F = fa["frostnet"]
    from frostnet_amd.optimizer import QSGD
    g = golden("g5_qat_large")
    B, res, seed, wseed = [int(v) for v in g["spec"]]
    model = F.frostnet_quant_large_1_0(drop_rate=0.0)
    spec = O.float_state_spec(O.net_cfg("large", 1.0))
    assert [k for k, _ in spec] == list(model.state_dict().keys())
    model.load_state_dict(O.synth_state([k for k, _ in spec], [s for _, s in spec], wseed))
    F.qat_prepare(model, version=0)
    model.cuda()
    names = [n for n, _ in model.named_parameters()]
    assert names == [str(n) for n in g["s0_param_names"]]
    opt = QSGD([{"params": [p], "weight_decay": O.param_group_rule(tuple(p.shape), 1e-5)} for p in model.parameters()],
               lr=5e-3, momentum=0.9, nesterov=True, clip_by=1e-3, toss_coin=True, noise_decay=1e-2)
    tgt = T(g["target"]).cuda()
    losses = []
    for step in range(2):
        opt.zero_grad()
        y = model(T(O.synth((B, 3, res, res), seed + step)).cuda())
        loss = torch.nn.functional.cross_entropy(y, tgt)
        loss.backward()
        gn = np.array([float(p.grad.double().norm()) for p in model.parameters()])
        ref_gn = g[f"s{step}_grad_norms"]
        ratio = gn / (ref_gn + 1e-12)
        print(f"step {step}: loss {float(loss):.4f} (ref {float(g[f's{step}_loss']):.4f}) logits rel {relerr(y.detach().cpu(), T(g[f's{step}_logits'])):.3f} "
              f"grad-norm ratio median {np.median(ratio):.3f} p5 {np.percentile(ratio, 5):.3f} p95 {np.percentile(ratio, 95):.3f}")
        assert np.isfinite(float(loss))
        assert abs(float(loss) - float(g[f"s{step}_loss"])) <= 1.5      # chaotic at B=2@64 (BN over 8 samples), SURVEY H-2
        assert 0.33 <= np.median(ratio) <= 3.0
        # the tail of the backward (classifier, last_layer: the first gradients produced, before the chaos of the layers below) against the REFERENCE's
        # gradient norms: a 2x scaling bug anywhere in the head cannot hide in the loose band above
        tail = {n: float(r_) for n, r_ in zip(names, ratio) if n.startswith("classifier.") or n.startswith("last_layer.")}
        print(f"    tail gradient-norm ratios vs the reference: { {k: round(v, 3) for k, v in tail.items()} }")
        if step == 0:
            assert all(abs(v - 1.0) <= TAIL_NORM_TOL for k, v in tail.items() if k.startswith("classifier.")), tail
        if step == 0:
            sd = model.state_dict()
            qk = [str(k) for k in g["s0_qkeys"]]
            for k, v in zip(qk, g["s0_qvals"]):
                if k.startswith("quant.") or k.startswith("conv1."):
                    np.testing.assert_allclose(float(sd[k].reshape(-1)[0]), v, rtol=1e-4, atol=1e-6, err_msg=k)
            opt.is_warmup = False
        opt.step()
        losses.append(float(loss))


def test_features_backbone_qat_gpu_vs_oracle(fa):
    """Config-5 backbone: quantised feature maps [x1,x2,x3,x5] from the HIP engine vs the oracle (train-mode forward, Small@64:
    bit-identical sites upstream => feature maps equal up to isolated 1-step flips), plus a backward smoke through all four taps."""
    from frostnet_amd import frostnet_features as FF
    from frostnet_amd import frostnet as F
    torch.set_num_threads(8)
    mode, res = "small", 64
    cfg = O.net_cfg(mode, 1.0)
    spec = O.float_state_spec(cfg, features=True)
    P, B = O.make_state(spec, 5000, True)
    qs = O.QState(B)
    x = T(O.synth((2, 3, res, res), 520))
    with torch.no_grad():
        ref = O.frostnet_forward(P, qs, cfg, x, True, True, features=True)
    net = FF.FrostNet(mode=mode, width_mult=1.0, quantized=True)
    net.load_state_dict(O.synth_state([k for k, _ in spec], [s for _, s in spec], 5000), strict=False)
    F.qat_prepare(net, version=0)
    net.cuda()
    feats = net(x.cuda())
    assert [tuple(f.shape) for f in feats] == [tuple(r.shape) for r in ref]
    for i, (f, r) in enumerate(zip(feats, ref)):
        assert relerr(f.detach().cpu(), r) <= 2e-2, (i, relerr(f.detach().cpu(), r))
    loss = sum((f * f).mean() for f in feats)
    loss.backward()
    gn = [float(p.grad.norm()) for p in net.parameters()]
    assert all(np.isfinite(gn)) and gn[0] > 0


def test_per_layer_finalize_path_matches_table_path(fa):
    """The data-parallel path finalises weight gradients per layer (so buckets can start their all-reduce during the backward),
    the single-GPU path in one table launch (with the pointwise weight gradients on the side stream): same gradients.

    Two separate steps are compared, so the bound is the run-to-run noise of one step (fp32 atomic order in the S1/S2
    reductions -> a bf16 rounding flip of dc), measured at <= 1.2e-2 of a layer's gradient norm over 36 runs
    (tests/devtools/dbg_flake.py).  BatchNorm gamma/beta gradients are measured against their convolution's gradient norm: at
    initialisation (gamma=1, beta=0, the next layer normalises again) they are exactly zero in exact arithmetic, so their own
    norm is cancellation noise and a relative error against it is meaningless."""
    F = fa["frostnet"]
    grads = []
    for per_layer in (False, True):
        torch.manual_seed(3)
        model = F.frostnet_quant_small_1_0(drop_rate=0.0)
        F.qat_prepare(model, version=0)
        model.cuda().train()
        runner = model.hip_runner()
        seen = []
        if per_layer:
            runner.E.on_layer_grads = lambda l: seen.append(l.name)
        x = T(O.synth((4, 3, 64, 64), 5)).cuda()
        tgt = torch.tensor([1, 2, 3, 4]).cuda()
        torch.nn.functional.cross_entropy(model(x), tgt).backward()
        grads.append({n: p.grad.detach().cpu().double() for n, p in model.named_parameters()})
        if per_layer:
            assert len(seen) == len(runner.E.layers)
    a, b = grads
    for n in a:
        sib = n.replace("bn.weight", "weight").replace("bn.bias", "weight")
        den = max(float(b[n].norm()), float(b[sib].norm()), 1e-30)
        assert float((a[n] - b[n]).norm()) / den <= 5e-2, n


def test_qat_training_reduces_loss_on_one_batch(fa):
    """helper_functions.py:139-143 repeated on ONE batch: forward, CE, backward, GradBoost-SGD step (StatAssist statistics only, i.e.
    is_warmup=True: no injected noise, so the run is deterministic up to atomic order).  If the gradients or the update were wrong in
    sign or scale the loss would not fall."""
    F = fa["frostnet"]
    from frostnet_amd import harness as H
    from frostnet_amd.optimizer import QSGD
    torch.manual_seed(7)
    model = F.frostnet_quant_small_1_0(drop_rate=0.0)
    F.qat_prepare(model, version=0)
    model.cuda()
    opt = QSGD(H.make_param_groups(model, 1e-5), lr=2e-2, momentum=0.9, nesterov=True, clip_by=1e-3, toss_coin=True, noise_decay=1e-2)
    assert opt.is_warmup
    crit = torch.nn.CrossEntropyLoss()
    x = torch.randn(32, 3, 96, 96, device="cuda")
    t = torch.randint(0, 1000, (32,), device="cuda")
    losses = []
    for _ in range(12):
        loss, _ = H.train_one_iter(model, crit, opt, x, t)
        losses.append(float(loss))
    assert all(np.isfinite(losses)), losses
    assert min(losses[-3:]) < 0.8 * losses[0], losses
