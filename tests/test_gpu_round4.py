"""Round-4 additions on the GPU: sub-module flag switches, float-runner dropout stream in checkpoints, frozen-BatchNorm training
(frostnet_features.py:354-359 `_freeze_stages`), the fp32-gradient parity mode, the fused-block inference kernels."""
import numpy as np
import pytest
import torch

from oracle import frost_oracle as O

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def relerr(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def F():
    import __graft_entry__ as ge
    ge.build()
    from frostnet_amd import frostnet
    assert torch.cuda.is_available()
    return frostnet


def test_submodule_flag_switches_reach_the_runner(F):
    """`model.layer3.apply(disable_fake_quant)` (a sub-module, not the root) must be refused like the root-level call, and in eval a
    site re-enabled through a sub-module `.apply` must observe again although the host summary said "all observers off"."""
    import torch.ao.quantization as aoq
    torch.manual_seed(3)
    model = F.MODEL_REGISTRY["frostnet_quant_small_1_0"](drop_rate=0.0)
    F.qat_prepare(model, version=0)
    model.cuda().train()
    x = torch.randn(4, 3, 64, 64, device="cuda")
    with torch.no_grad():
        model(x)
    model.layer3.apply(aoq.disable_fake_quant)
    with pytest.raises(NotImplementedError, match="fake_quant_enabled"):
        with torch.no_grad():
            model(x)
    model.layer3.apply(aoq.enable_fake_quant)
    with torch.no_grad():
        model(x)
    model.eval()
    model.apply(aoq.disable_observer)
    site = model.layer2[0].conv1.conv[0].activation_post_process
    with torch.no_grad():
        model(x)
    before = site.activation_post_process.max_val.clone()
    with torch.no_grad():
        model(3.0 * x)
    assert torch.equal(before, site.activation_post_process.max_val)          # every observer is off: nothing moves
    model.layer2[0].conv1.apply(aoq.enable_observer)                          # ONE site back on, through a sub-module
    with torch.no_grad():
        model(3.0 * x)
    assert not torch.equal(before, site.activation_post_process.max_val), "the re-enabled observer did not run"
    # a flag written straight into the buffer (no apply at all): honoured per site on the device; the host summary ("everything is frozen: skip the statistics
    # passes") learns of it through `flags_dirty`
    model.layer2[0].conv1.apply(aoq.disable_observer)
    other = model.layer1[1].conv1.conv[0].activation_post_process
    b2 = other.activation_post_process.max_val.clone()
    other.observer_enabled[0] = 1
    model.hip_runner().flags_dirty = True
    with torch.no_grad():
        model(5.0 * x)
    assert not torch.equal(b2, other.activation_post_process.max_val)


def test_float_runner_dropout_stream_is_checkpointed(F, tmp_path):
    """the float (StatAssist warm-up) runner draws dropout from the shared device Philox stream; harness.save_checkpoint / load_checkpoint
    must carry it, and restoring must write the draw counter in place."""
    from frostnet_amd import harness
    from frostnet_amd.optimizer import QSGD
    torch.manual_seed(5)
    model = F.MODEL_REGISTRY["frostnet_quant_small_1_0"](drop_rate=0.2).cuda().train()
    opt = QSGD([{"params": [p]} for p in model.parameters()], lr=1e-3, momentum=0.9)
    x = torch.randn(4, 3, 64, 64, device="cuda")
    for _ in range(2):
        model(x).sum().backward()
        opt.zero_grad()
    r = model.hip_runner()
    assert type(r).__name__ == "FloatRunner" and r.rng_state()["draws"] == 2
    path = str(tmp_path / "ck.pth.tar")
    harness.save_checkpoint(harness.checkpoint_state(model, opt, epoch=0), path)
    ctr = r._drop_ctr
    model(x).sum().backward()
    assert r.rng_state()["draws"] == 3
    ck = harness.load_checkpoint(model, opt, path)
    assert "hip_rng" in ck and r.rng_state()["draws"] == 2
    assert model.hip_runner()._drop_ctr.data_ptr() == ctr.data_ptr()          # restored in place


# ------------------------------------------------------------------------------------------ frozen BatchNorm (VERDICT r3 missing #3)
FROZEN_LAYERS = [("pw104_312_14", 104, 312, 1, 1, 1, 14, 16, 1, 117),      # chunked (k_pwc) dc pass
                 ("pw312_80_14_lin", 312, 80, 1, 1, 1, 14, 16, 0, 0),      # kept-output element-wise passes
                 ("pw24_72_56", 24, 72, 1, 1, 1, 56, 64, 1, 117),          # fused dc + dgrad + wgrad kernel
                 ("dw5s1_312_14", 312, 312, 5, 1, 312, 14, 16, 1, 0),      # image-resident block depthwise backward
                 ("dw3s2_96_112", 96, 96, 3, 2, 96, 112, 2, 1, 0)]         # tiled depthwise kernels


@pytest.mark.parametrize("case", FROZEN_LAYERS, ids=[c[0] for c in FROZEN_LAYERS])
def test_frozen_batchnorm_layer_vs_oracle(F, case):
    """A ConvBN(ReLU) whose BatchNorm is in eval mode inside a TRAINING forward/backward (`_freeze_stages`, frostnet_features.py:354-359): the reference
    normalises with the running statistics (nothing updated) and autograd differentiates that graph.  Oracle = convbn_qat(training=False) under
    autograd (fp32 for indices / state, fp64 as the gradient yardstick); device = Engine.conv with bn.eval()."""
    from frostnet_amd import engine
    name, cin, cout, k, s, groups, H, N, relu, in_zp = case
    dev, seed = "cuda", 7700 + 13 * FROZEN_LAYERS.index(case)
    torch.set_num_threads(16)
    spec = O._convbn_spec("L", cin, cout, k, groups)
    sd = O.synth_state([k_ for k_, _ in spec], [s_ for _, s_ in spec], seed)
    in_scale = 0.0231
    xi = np.clip(np.round(O.synth((N, cin, H, H), seed + 1) * 40 + 128 + (0 if in_zp else -60)), 0, 255).astype(np.uint8)
    P, B = O.split_state({O.float_to_qat_key(k_): v.clone() for k_, v in sd.items()})
    qs = O.QState(B)
    P64, B64 = O.split_state({O.float_to_qat_key(k_): (v.clone().double() if v.is_floating_point() else v.clone()) for k_, v in sd.items()})
    qs64 = O.QState(B64)
    xo = ((T(xi.astype(np.float32)) - in_zp) * in_scale).requires_grad_(True)
    xo64 = ((T(xi.astype(np.float64)) - in_zp) * in_scale).requires_grad_(True)
    kind = "dw" if groups > 1 else "pw"
    E, qa = engine.Engine(dev), engine.QArena(4, dev)
    w = sd["L.conv.0.weight"].to(dev).contiguous().requires_grad_(True)
    gamma, beta = sd["L.conv.1.weight"].to(dev).requires_grad_(True), sd["L.conv.1.bias"].to(dev).requires_grad_(True)
    l = engine.ConvLayer("L", kind, w, gamma, beta, sd["L.conv.1.running_mean"].to(dev), sd["L.conv.1.running_var"].to(dev),
                         torch.zeros((), dtype=torch.int64, device=dev), None, k, s, bool(relu), qa.alloc(), qa.alloc())
    l.bn_mod = torch.nn.BatchNorm2d(cout).eval()            # the flag Engine.conv reads
    E.add_layer(l)
    qx = qa.alloc()
    qa.set_qparams(qx, in_scale, in_zp)
    rm0, rv0 = l.rmean.clone(), l.rvar.clone()
    for step in range(2):
        gr = T(O.synth((N, cout, H // s, H // s), seed + 2 + 50 * step))
        outs = []
        for Pq, q_, x_, g_ in ((P, qs, xo, gr), (P64, qs64, xo64, gr.bfloat16().double())):
            x_.grad = None
            for p in Pq.values():
                p.grad = None
            yo = O.convbn_qat(Pq, q_, "L", x_, s, (k - 1) // 2, groups, bool(relu), False)      # training=False: running statistics, under autograd
            yo.backward(g_)
            outs.append(yo.detach())
        a = "L.conv.0.activation_post_process"
        idx_o = O.fq_index(outs[0], qs.sd[a + ".scale"][0], qs.sd[a + ".zero_point"][0])
        E.begin_step()
        x = E.act_from_indices(T(xi), qx)
        y = E.conv(l, x, training=True, observe=True)
        assert l.frozen
        y.grad = engine.float_to_grad(gr.to(dev))
        yidx = y.indices().cpu()
        E.backward()
        torch.cuda.synchronize()
        d = (yidx.to(torch.int16) - idx_o.to(torch.int16)).abs()
        assert int(d.max()) <= 1 and float((d > 0).float().mean()) <= 1e-4, (name, step, int(d.max()), float((d > 0).float().mean()))
        assert torch.equal(l.rmean, rm0) and torch.equal(l.rvar, rv0) and int(l.nbt) == 0            # frozen: nothing moved
        np.testing.assert_allclose(qa.get(l.qy)["scale"], float(qs.sd[a + ".scale"][0]), rtol=2e-5)
        dx = engine.grad_to_float(x.grad, x.n, x.h, x.w, x.c).cpu()
        errs = dict(dx=relerr(dx, xo64.grad), dW=relerr(l.w.grad.cpu(), P64["L.conv.0.weight"].grad),
                    dgamma=relerr(l.gamma.grad.cpu(), P64["L.conv.0.bn.weight"].grad), dbeta=relerr(l.beta.grad.cpu(), P64["L.conv.0.bn.bias"].grad))
        print(f"[frozen {name} step {step}] " + " ".join(f"{k_} {v:.2e}" for k_, v in errs.items()))
        assert all(v <= 2e-2 for v in errs.values()), (name, step, errs)


def test_freeze_stages_training_step_on_the_features_backbone(F):
    """`_freeze_stages()` + `.train()`-mode forward/backward on the QAT feature backbone (mmdet's norm_eval fine-tuning): runs, leaves every running
    statistic untouched, produces finite parameter gradients, and the observers still move."""
    from frostnet_amd import frostnet_features as FF
    torch.manual_seed(9)
    m = FF.FrostNet(mode="small", width_mult=1.0, quantized=True)
    m._init_weights()
    F.qat_prepare(m, version=0)
    m.cuda().train()
    x = torch.randn(2, 3, 96, 96, device="cuda")
    sum(f.sum() for f in m(x)).backward()                 # one ordinary step first: running statistics and observers initialised
    m._freeze_stages()
    before = {k: v.clone() for k, v in m.state_dict().items() if "running_" in k or k.endswith("num_batches_tracked")}
    obs = m.layer3[0].conv1.conv[0].activation_post_process.activation_post_process.max_val.clone()
    for p in m.parameters():
        p.grad = None
    sum(f.float().pow(2).mean() for f in m(2.0 * x)).backward()
    after = m.state_dict()
    assert all(torch.equal(v, after[k]) for k, v in before.items())
    assert not torch.equal(obs, m.layer3[0].conv1.conv[0].activation_post_process.activation_post_process.max_val)
    gs = [p.grad for p in m.parameters()]
    assert all(g is not None and bool(torch.isfinite(g).all()) for g in gs) and sum(float(g.abs().sum()) for g in gs) > 0


# ------------------------------------------------------------------------------------------ fp32-gradient parity mode
G32_LAYERS = [("stem_224", 3, 32, 3, 2, 1, 224, 2, 1, 117), ("pw16_96_112", 16, 96, 1, 1, 1, 112, 2, 1, 0), ("pw96_24_56_lin", 96, 24, 1, 1, 1, 56, 4, 0, 0),
              ("dw3s2_96_112", 96, 96, 3, 2, 96, 112, 2, 1, 0), ("dw3s1_72_56", 72, 72, 3, 1, 72, 56, 4, 1, 0), ("dw5s2_144_56", 144, 144, 5, 2, 144, 56, 4, 1, 0),
              ("dw5s1_624_14", 624, 624, 5, 1, 624, 14, 8, 1, 0), ("pw104_624_14", 104, 624, 1, 1, 1, 14, 16, 1, 117), ("pw624_96_14_lin", 624, 96, 1, 1, 1, 14, 16, 0, 0),
              ("pw240_1440_7", 240, 1440, 1, 1, 1, 7, 32, 1, 117), ("dw5s1_1440_7", 1440, 1440, 5, 1, 1440, 7, 32, 1, 0), ("pw1728_320_7_lin", 1728, 320, 1, 1, 1, 7, 32, 0, 0),
              ("pw24_144_56", 24, 144, 1, 1, 1, 56, 4, 1, 117)]
G32_TOL = 1e-3


@pytest.mark.parametrize("case", G32_LAYERS, ids=[c[0] for c in G32_LAYERS])
def test_fp32_gradient_mode_layer_vs_oracle(F, case):
    """The QAT backward with fp32 gradient storage (Engine.grad_fp32; csrc/frost_g32.hip) on the 13 production layer shapes of tests/test_gpu_prod.py
    (smaller batches: plain kernels): dx, dW, dgamma, dbeta within 1e-3 of an fp64 evaluation of the reference's formulas -- the north-star tolerance -- where
    the production bf16 backward measures 2-9e-3 on the same quantities."""
    from frostnet_amd import engine
    name, cin, cout, k, s, groups, H, N, relu, in_zp = case
    dev, seed = "cuda", 8800 + 17 * G32_LAYERS.index(case)
    torch.set_num_threads(16)
    spec = O._convbn_spec("L", cin, cout, k, groups)
    sd = O.synth_state([k_ for k_, _ in spec], [s_ for _, s_ in spec], seed)
    in_scale = 0.0231
    xi = np.clip(np.round(O.synth((N, cin, H, H), seed + 1) * 40 + 128 + (0 if in_zp else -60)), 0, 255).astype(np.uint8)
    # two yardsticks: the reference's formulas evaluated in fp32 (= the reference itself: it decides borderline cases -- the weight with the largest magnitude sits
    # exactly on the clipping boundary 127.5 of its fake-quantiser, and whether its gradient is masked depends on the last bit of x * (1 / scale)) and in fp64
    # (free of the reference's own summation noise).  A gradient passes if it is within tolerance of either.
    ev = []
    for dt in (torch.float32, torch.float64):
        Pq, Bq = O.split_state({O.float_to_qat_key(k_): (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k_, v in sd.items()})
        ev.append((Pq, O.QState(Bq), ((T(xi).to(dt) - in_zp) * in_scale).requires_grad_(True), dt))
    kind = "stem" if (groups == 1 and k == 3) else ("dw" if groups > 1 else "pw")
    E, qa = engine.Engine(dev), engine.QArena(4, dev)
    E.grad_fp32 = True
    w = sd["L.conv.0.weight"].to(dev).contiguous().requires_grad_(True)
    gamma, beta = sd["L.conv.1.weight"].to(dev).requires_grad_(True), sd["L.conv.1.bias"].to(dev).requires_grad_(True)
    l = engine.ConvLayer("L", kind, w, gamma, beta, sd["L.conv.1.running_mean"].to(dev), sd["L.conv.1.running_var"].to(dev),
                         torch.zeros((), dtype=torch.int64, device=dev), None, k, s, bool(relu), qa.alloc(), qa.alloc())
    E.add_layer(l)
    qx = qa.alloc()
    qa.set_qparams(qx, in_scale, in_zp)
    xi_t = T(xi)
    if kind == "stem":
        xi_t = torch.cat([xi_t, torch.full_like(xi_t[:, :1], in_zp)], 1)
    pad = (k - 1) // 2
    Ho = (H + 2 * pad - k) // s + 1
    for step in range(2):
        gr = T(O.synth((N, cout, Ho, Ho), seed + 2 + 50 * step))
        idxs = []
        for Pq, qsq, xq, dt in ev:
            xq.grad = None
            for p in Pq.values():
                p.grad = None
            yo = O.convbn_qat(Pq, qsq, "L", xq, s, pad, groups, bool(relu), True)
            yo.backward(gr.to(dt))
            idxs.append(O.fq_index(yo.detach(), qsq.sd["L.conv.0.activation_post_process.scale"][0], qsq.sd["L.conv.0.activation_post_process.zero_point"][0]))
        E.begin_step()
        x = E.act_from_indices(xi_t, qx)
        y = E.conv(l, x, training=True, observe=True)
        yidx = y.indices().cpu()
        y.grad = engine.float_to_grad(gr.to(dev), fp32=True)
        E.backward()
        torch.cuda.synchronize()
        flips = min(float((yidx.to(torch.int16) != ix.to(torch.int16)).float().mean()) for ix in idxs)
        errs = {}
        for nm, mine, key in (("dW", l.w.grad, "L.conv.0.weight"), ("dgamma", l.gamma.grad, "L.conv.0.bn.weight"), ("dbeta", l.beta.grad, "L.conv.0.bn.bias")):
            errs[nm] = min(relerr(mine.cpu(), Pq[key].grad) for Pq, _, _, _ in ev)
        if kind != "stem":
            dxd = engine.grad_to_float(x.grad, x.n, x.h, x.w, x.c).cpu()
            errs["dx"] = min(relerr(dxd, xq.grad) for _, _, xq, _ in ev)
        print(f"[fp32-grad {name} step {step}] forward flips {flips:.1e}; " + " ".join(f"{k_} {v:.2e}" for k_, v in errs.items()))
        # a forward index that differs from the oracle's (a tie of the reference's own rounding) moves a mask: budget 3 x sqrt(flip fraction) on top
        tol = G32_TOL + 3.0 * flips ** 0.5
        assert all(v <= tol for v in errs.values()), (name, step, errs, tol)


def test_fp32_gradient_mode_block_vs_reference_golden(F, golden):
    """One whole bottleneck (g4t l31: 80 -> 80, k5, 14 x 14, CAS + residual, true shape) through the module surface with `grad_precision = 'fp32'`: dx and
    every parameter gradient against the REFERENCE golden and the fp64 oracle -- the same comparison test_gpu_model.py::test_g4_block_true_shapes makes for the
    bf16 production backward at 2e-2 / 5e-2, here at 2e-3 (cat / add / every conv kind chained; forward index ties of the reference are the floor)."""
    from torch.ao.quantization import get_default_qat_qconfig, prepare_qat
    from frostnet_amd import engine, runner as R
    g = golden("g4t_l31_q")
    cin, cout, k, s, e, r, H, N, xseed, gseed, wseed = [int(v) for v in g["spec"]]
    m = F.CascadePreExBottleneck(cin, cout, quantized=True, kernel_size=k, stride=s, expand_ratio=e, reduce_factor=r)
    keys = [str(k_) for k_ in g["init_keys"]]
    shapes = [tuple(int(x) for x in row[:n]) for row, n in zip(g["init_shapes"], g["init_ndims"])]
    m.load_state_dict(O.synth_state(keys, shapes, wseed))
    m.train()
    for mod in m.modules():
        if type(mod) in (F.ConvBNReLU, F.ConvBN):
            mod.fuse_model()
    m.qconfig = get_default_qat_qconfig("qnnpack", version=0)
    prepare_qat(m, inplace=True)
    m.cuda()
    run = R.FrostRunner.for_block(m)
    run.E.grad_fp32 = True
    qx = run.qa.alloc()
    in_scale, in_zp = float(g["in_qp"][0]), int(g["in_qp"][1])
    run.qa.set_qparams(qx, in_scale, in_zp)
    xi = T(g["x_idx"])
    xf = (xi.float() - in_zp) * in_scale
    qx[4], qx[5] = float(xf.min()), float(xf.max())
    for step in range(2):
        gr = T(O.synth((N, cout, H, H), gseed + 50 * step))
        run.E.begin_step()
        x = run.E.act_from_indices(xi, qx)
        y = run.block_forward(run.block, x, True, True)
        flips = float((y.indices().cpu().to(torch.int16) != T(g[f"s{step}_yidx"]).to(torch.int16)).float().mean())
        y.grad = engine.float_to_grad(gr.cuda(), fp32=True)
        run.bind_grads()
        run.E.backward()
        torch.cuda.synchronize()
        dx = engine.grad_to_float(x.grad, x.n, x.h, x.w, x.c).cpu()
        worst = {"dx": relerr(dx, T(g[f"s{step}_dx"]))}
        for pn, p in m.named_parameters():
            pack = g[f"s{step}_grad/" + pn.replace(".", "/")]
            mine = O.sample_big(p.grad.detach().double().cpu().numpy().reshape(-1))
            worst[pn] = float(np.linalg.norm(mine - pack[3:]) / (np.linalg.norm(pack[3:]) + 1e-30))
        top = sorted(worst.items(), key=lambda kv: -kv[1])[:4]
        print(f"[fp32-grad block l31 step {step}] output index flips vs the reference {flips:.1e}; worst gradients vs the reference: " + ", ".join(f"{k_} {v:.2e}" for k_, v in top))
        assert all(v <= 2e-3 + 3.0 * flips ** 0.5 for v in worst.values()), top


# ------------------------------------------------------------------------------------------ squeeze emit + cat in one launch
@pytest.mark.parametrize("cin,r,H,n", [(40, 16, 28, 5), (80, 24, 14, 7), (96, 24, 14, 33), (192, 48, 7, 9), (192, 96, 7, 3), (120, 32, 9, 2)])
def test_squeeze_emit_cat_fused_is_bit_identical(F, cin, r, H, n):
    """frost_sq_emit_cat (squeeze_conv emit + quant_cat.cat requantisation from one staged tile, frostnet.py:127-129) against the two launches it replaces
    (frost_pw_conv_fwd mode 1 + frost_cat_requant): squeezed activation, cat output and the cat's record bit-identical over two steps (the second from moved
    observers), ragged last tiles included."""
    from frostnet_amd import engine, _lib as L
    dev = "cuda"

    def run(fused):
        old = engine._BLOCK_SQCAT
        engine._BLOCK_SQCAT = fused
        try:
            g = torch.Generator(device="cpu").manual_seed(31)
            E, qa = engine.Engine(dev), engine.QArena(8, dev)
            w = (torch.randn(r, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5).to(dev).requires_grad_(True)
            gamma = (torch.rand(r, generator=g) * 0.5 + 0.75).to(dev).requires_grad_(True)
            beta = (torch.rand(r, generator=g) * 0.2 - 0.05).to(dev).requires_grad_(True)
            l = E.add_layer(engine.ConvLayer("sq", "pw", w, gamma, beta, torch.zeros(r, device=dev), torch.ones(r, device=dev), torch.zeros((), dtype=torch.int64, device=dev),
                                             None, 1, 1, True, qa.alloc(), qa.alloc()))
            qx, qcat = qa.alloc(), qa.alloc()
            qa.set_qparams(qx, 0.021, 117)
            qx[4], qx[5] = -2.4, 2.9
            outs = []
            for step in range(2):
                x = E.new_act(n, H, H, cin, qx)
                x.buf[: x.numel] = torch.randint(-128, 128, (x.numel,), dtype=torch.int16, generator=g).to(torch.int8).to(dev)
                E.begin_step()
                L.CALL_LOG = []
                sq = E.conv(l, x, True, True, cat=(x.q, qcat))
                y = E.cat(sq, x, qcat, True)
                torch.cuda.synchronize()
                log, L.CALL_LOG = list(L.CALL_LOG), None
                assert (("frost_sq_emit_cat" in log) or ("frost_sq_fwd" in log)) == fused and ("frost_cat_requant" in log) == (not fused), log
                outs.append((sq.buf[: sq.numel].clone(), y.buf[: y.numel].clone(), qcat.clone(), l.qy.clone()))
            return outs
        finally:
            engine._BLOCK_SQCAT = old
            L.CALL_LOG = None
    a, b = run(True), run(False)
    for sa, sb in zip(a, b):
        for ta, tb in zip(sa, sb):
            assert torch.equal(ta, tb)


@pytest.mark.parametrize("shape,cl", [((5, 3, 97, 131), False), ((3, 3, 64, 64), True), ((2, 3, 300, 520), True), ((1, 3, 31, 17), False)])
def test_converted_stem_in_one_launch_is_bit_identical(shape, cl):
    """frost_stem_converted (QuantStub + quantized conv1 from the fp32 image) against frost_quantize_input + frost_stem_im2col + the int8 GEMM emit: every block
    output and the logits of the converted model must be identical -- odd sizes, NCHW and channels_last, a map wider than one 128-column tile, a map smaller
    than one tile, a batch that is not a multiple of the classifier's 16-image MFMA tile.  (The reference fixtures pin both to torch: tests/test_gpu_convert.py.)"""
    import __graft_entry__ as ge
    ge.build()
    from frostnet_amd import frostnet as F, engine as E, _lib as L
    torch.manual_seed(11)
    model = F.MODEL_REGISTRY["frostnet_quant_small_1_0"](drop_rate=0.0)
    F.qat_prepare(model, version=0)
    model.cuda().train()
    for _ in range(2):                                     # observers and running statistics see data before convert()
        model(torch.randn(4, 3, 64, 64, device="cuda") * 1.5)
    model.hip_convert()
    r = model.hip_runner()
    x = torch.randn(*shape, device="cuda") * 1.5
    if cl:
        x = x.contiguous(memory_format=torch.channels_last)
    outs = {}
    for fused in (True, False):
        E._STEM_FUSED_CVT = fused
        L.CALL_LOG = []
        try:
            taps = []
            with torch.no_grad():
                y = r._forward_converted(x, taps)
            torch.cuda.synchronize()
            assert ("frost_stem_converted" in L.CALL_LOG) == fused and ("frost_stem_im2col" in L.CALL_LOG) == (not fused)
            outs[fused] = (y.clone(), [t.indices().clone() for t in taps], r.E.last_logit_idx.clone())
        finally:
            L.CALL_LOG = None
            E._STEM_FUSED_CVT = True
    assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][2], outs[False][2])
    for a, b in zip(outs[True][1], outs[False][1]):
        assert torch.equal(a, b)
