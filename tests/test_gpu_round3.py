"""Round-3 parity / coverage additions (VERDICT r2 "next round" item 3 and "missing" items 4-5):

  (a) config c2 at its REAL size -- bf16 inference of FrostNet-Large at B = 256 through properties (per-image independence against a B = 8 run of
      the same images, idempotence, the fp32 definition on a sample) plus a per-layer check of every kernel the c2 model selects by pixel count
      (k_pw's bf16 mode, the stand-alone GEMM with 256- and with 128-pixel tiles) against an fp64 GEMM + bias + ReLU of the same bf16 operands;
  (c) the SSDLite extras / prediction layers of config c5, teacher-forced WITH gradients at their real 8x8 / 4x4 / 2x2 maps against oracle.convbn_qat;
  (d) `bench.py --force-dp`: the RCCL ("nccl") backend + segmented hipGraph + all-reduce-between-segments path executes on this box every round;
  the epoch-level loop (harness.train / val) and checkpoint resume: a run resumed from `save_checkpoint` continues bit-identically (parameters,
  GradBoost noise stream, dropout stream).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import frost_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def engine():
    import __graft_entry__ as ge
    ge.build()
    from frostnet_amd import engine
    assert torch.cuda.is_available()
    return engine


def _randomize_bn(model, seed):
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data = torch.rand(m.num_features, generator=g) * 0.8 + 0.6
            m.bias.data = torch.rand(m.num_features, generator=g) * 0.2 - 0.1
            m.running_mean.data = torch.randn(m.num_features, generator=g) * 0.1
            m.running_var.data = torch.rand(m.num_features, generator=g) * 0.5 + 0.5


# ------------------------------------------------------------------------------------------------ (a) c2 at B = 256
def test_c2_bf16_inference_at_batch_256(engine):
    from frostnet_amd import frostnet as F
    torch.manual_seed(7)
    model = F.MODEL_REGISTRY["frostnet_large_1_0"]()
    _randomize_bn(model, 11)
    model.eval()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(256, 3, 224, 224, generator=g)
    with torch.no_grad():
        ref8 = model(x[:8])                              # the fp32 definition (stock torch modules, CPU) on a sample of the batch
    model.cuda()
    xd = x.cuda()
    out = model.hip_infer_bf16(xd)
    again = model.hip_infer_bf16(xd)
    assert torch.equal(out, again)                       # idempotent: no state, no atomics-order dependence in the inference path
    small = model.hip_infer_bf16(xd[:8].contiguous())
    # per-image independence: image i's logits do not depend on the batch it travels in.  The kernel INSTANCE does (pixel-count thresholds pick
    # the tile shapes), so the comparison is at bf16 rounding level, not bitwise
    rel_ind = float((out[:8] - small).norm() / small.norm())
    assert rel_ind <= 5e-3, rel_ind                      # (measured 0.0 -- the same instances are chosen at B = 8 and B = 256 today -- the bound leaves room for a bf16 rounding apart)
    assert int((out[:8].argmax(1) == small.argmax(1)).sum()) == 8
    rel = float((out[:8].cpu() - ref8).norm() / ref8.norm())
    # bf16 activations through 70 layers against the fp32 definition: measured 2.5e-3 (round 6, gpurun_out/r6_side.log); bound = 3 x that (was a flat 3e-2: VERDICT r5 #7)
    assert rel <= 8e-3, rel
    assert int((out[:8].cpu().argmax(1) == ref8.argmax(1)).sum()) == 8
    # a permutation of the batch permutes the logits (no cross-image coupling anywhere in the eval graph)
    perm = torch.randperm(256, generator=g)
    outp = model.hip_infer_bf16(xd[perm.cuda()].contiguous())
    assert float((outp - out[perm.cuda()]).norm() / out.norm()) <= 1e-6
    print(f"[c2 B=256] independence {rel_ind:.2e}, vs fp32 definition {rel:.2e}")


#             npix            cin   cout  relu   which kernel frost_infer_pw selects
INFER_PW = [(256 * 196,       104,  624,  1),   # rows <= 256 B: k_pw's bf16 mode (DMA-staged tiles)
            (256 * 784,       168,  40,   0),   # rows > 256 B: the stand-alone GEMM, 256-pixel tiles (784 x 1 workgroups)
            (256 * 49,        1440, 192,  0),   # ... 128-pixel tiles (a 256-pixel launch would leave CUs idle), two channel chunks
            (256 * 49,        288,  1728, 1),   # ... 256-pixel tiles, 14 channel chunks
            (256 * 49 + 37,   240,  1440, 1),   # ragged pixel count
            (256 * 3136 // 8, 16,   96,   1)]   # narrow high-resolution layer (32-byte rows)


@pytest.mark.parametrize("npix,cin,cout,relu", INFER_PW)
def test_bf16_inference_pointwise_layer_vs_fp64_gemm(engine, npix, cin, cout, relu):
    """One inference layer y = act(x . W'^T + b') (frostnet.py:14-60 with BatchNorm folded), operands exactly as the kernel sees them (bf16 x, bf16
    folded weights), reference in fp64: the only differences left are the fp32 accumulation order and ONE bf16 rounding of the output."""
    from frostnet_amd import _lib as L, infer
    dev = "cuda"
    g = torch.Generator().manual_seed(100 + cin + cout)
    conv = torch.nn.Conv2d(cin, cout, 1, bias=False)
    bn = torch.nn.BatchNorm2d(cout)
    conv.weight.data = torch.randn(cout, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5
    bn.weight.data = torch.rand(cout, generator=g) * 0.8 + 0.6
    bn.bias.data = torch.rand(cout, generator=g) * 0.4 - 0.2
    bn.running_mean.data = torch.randn(cout, generator=g) * 0.1
    bn.running_var.data = torch.rand(cout, generator=g) * 0.5 + 0.5
    seq = torch.nn.Sequential(conv, bn).to(dev).eval()
    l = infer._ILayer(seq, bool(relu), dev)
    arr = (L.FrostIDesc * 1)()
    arr[0] = l.desc()
    table = L.struct_to_tensor(arr, torch.device(dev))
    L.call("frost_infer_weight_prep", L.ptr(table), 1, L.stream())
    x = (torch.randn(npix, cin, generator=g) * 1.5).to(torch.bfloat16)
    xb = torch.zeros(npix * cin + 64, dtype=torch.int16, device=dev)
    xb[: npix * cin] = x.view(torch.int16).reshape(-1).to(dev)
    y = torch.empty(npix * cout + 64, dtype=torch.int16, device=dev)
    L.call("frost_infer_pw", L.ptr(xb), L.ptr(l.pack), L.ptr(l.biasf), npix, cin, cout, int(relu), L.ptr(y), L.stream())
    torch.cuda.synchronize()
    out = y[: npix * cout].view(torch.bfloat16).float().view(npix, cout).cpu()
    # reference: the folded weights rounded to bf16 as the pack holds them, fp64 accumulation
    sf = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach().cpu()
    wf = (conv.weight.detach().cpu().view(cout, cin) * sf[:, None]).to(torch.bfloat16).double()
    bf = (bn.bias.detach().cpu() - bn.running_mean.detach().cpu() * sf).double()
    ref = x.double() @ wf.t() + bf
    if relu:
        ref = ref.clamp_min(0)
    err = (out.double() - ref).abs()
    tol = 2.0 ** -8 * ref.abs() + 2e-3                    # one bf16 rounding (half an ulp = 2^-9 relative) + fp32 accumulation slack
    frac = float((err > tol).float().mean())
    assert frac <= 1e-4, (frac, float(err.max()))
    rel = float((out.double() - ref).norm() / ref.norm())
    assert rel <= 3e-3, rel


# ------------------------------------------------------------------------------------------------ (c) SSDLite extras / heads, teacher-forced
#          name               cin  cout  k  s  groups  H  relu    (config c5 @512: sources 64^2 / 32^2 / 16^2, extras at 8^2 / 4^2 / 2^2; per-GPU batch 32)
SSD_LAYERS = [("extras.0.pw2",  256, 512, 1, 1, 1,    8, 1),
              ("extras.1.pw1",  512, 128, 1, 1, 1,    8, 1),
              ("extras.1.dw",   128, 128, 3, 2, 128,  8, 1),
              ("extras.1.pw2",  128, 256, 1, 1, 1,    4, 1),
              ("extras.2.dw",   128, 128, 3, 2, 128,  4, 1),
              ("extras.2.pw2",  128, 256, 1, 1, 1,    2, 1),
              ("loc.3.dw",      512, 512, 3, 1, 512,  8, 1),
              ("loc.3.pw",      512, 24,  1, 1, 1,    8, 0),
              ("conf.4.dw",     256, 256, 3, 1, 256,  4, 1),
              ("conf.4.pw",     256, 88,  1, 1, 1,    4, 0),
              ("loc.5.dw",      256, 256, 3, 1, 256,  2, 1),
              ("conf.5.pw",     256, 88,  1, 1, 1,    2, 0)]


@pytest.mark.parametrize("case", SSD_LAYERS, ids=[c[0] for c in SSD_LAYERS])
def test_ssdlite_head_layers_teacher_forced_with_gradients(engine, case):
    """Object_Detection/ssd_qmv2.py:285-303 conventions (ConvBN head layers) as built in frostnet_amd/ssdlite.py: forward indices, observer / BN state
    and all four gradients of each extras / head layer shape at its true map size, against oracle.convbn_qat (fp32) and its fp64 evaluation."""
    from test_gpu_prod import run_layer_case
    name, cin, cout, k, s, groups, H, relu = case
    i = SSD_LAYERS.index(case)
    run_layer_case(engine, "ssd_" + name, cin, cout, k, s, groups, H, 32, relu, 9900 + 11 * i, 0 if i % 2 == 0 else 117, steps=2)


# ------------------------------------------------------------------------------------------------ (d) RCCL backend + segmented step on this box
def test_bench_force_dp_runs_the_rccl_backend():
    """`bench.py --force-dp`: a 1-rank process group on the `nccl` (= RCCL) backend, the backward as a chain of hipGraph segments with the bucket
    all-reduces issued between them on RCCL's stream -- the code path of the 8-GPU run (Classification/train.py:88-92's DataParallel replaced by one
    process per GPU).  Checks that it completes, prints ONE JSON line, really captured the segments and reports the exchange's timing."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29655", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dp", "--steps", "3", "--warmup", "2", "--batch", "64", "--no-cpu-baseline",
                          "--no-roofline"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert "backend=nccl" in out.stderr
    assert rec["n_gpus"] == 1 and rec["steps"] == 3 and rec["value"] > 0
    ga = rec["config"]["grad_allreduce"]
    assert rec["config"]["hip_graph"] is True and ga["fallback"] is None, ga
    assert ga["mode"].startswith("4 buckets") and len(ga["buckets"]) == 4 and sum(b["bytes"] for b in ga["buckets"]) == 4 * 5807056
    assert all(b["allreduce_us"] > 0 for b in ga["buckets"]) and ga["step_ms_without_collectives"] > 0


# ------------------------------------------------------------------------------------------------ epoch loops + checkpoint resume
def _tiny_loader(n_batches, batch, res, seed):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn(batch, 3, res, res, generator=g), torch.randint(0, 1000, (batch,), generator=g)) for _ in range(n_batches)]


def _new_model_and_opt(seed):
    from frostnet_amd import frostnet as F
    from frostnet_amd import harness as Hn
    from frostnet_amd.optimizer import QSGD
    torch.manual_seed(seed)
    model = F.MODEL_REGISTRY["frostnet_quant_small_1_0"]()          # drop_rate 0.2: the dropout stream is part of what must resume
    F.qat_prepare(model, version=0)
    model.cuda().train()
    opt = QSGD(Hn.make_param_groups(model, 1e-5), lr=5e-3, momentum=0.9, nesterov=True, clip_by=1e-3, toss_coin=True, noise_decay=1e-2, weight_decay=1e-5)
    opt.is_warmup = False
    return model, opt


def test_epoch_loops_and_checkpoint_resume(engine, tmp_path):
    """helper_functions.py:99-163 / 306-350 / 400-407 counterparts.  `train` returns the iteration means; `val` leaves the observers live (the
    reference only calls model.eval()); `save_checkpoint` / `load_checkpoint` restore EVERYTHING a continued run depends on: parameters and
    buffers (BN statistics, observer ranges, qparams) bit for bit, the optimizer state (momentum, GradBoost statistics, step counters), and the two
    device random streams -- checked by drawing from them: the next dropout mask and the next GradBoost step (same injected gradients) of the
    resumed pair equal those of the pair that kept running, bit for bit.  (Whole epochs cannot be compared: two identical runs of this network
    drift apart within a few steps -- the backward's fp32 atomics perturb gradients along directions the loss is blind to, tests/devtools/dbg_repro.py.)"""
    from frostnet_amd import harness as Hn
    from frostnet_amd.runner import dropout_mask
    crit = Hn.CrossEntropyLoss()
    loader = _tiny_loader(2, 4, 64, 5)

    class A:                     # the reference's args attributes read by adjust_learning_rate_cosine
        lrsch, anneal, epochs, warmup_epochs, warmup_lr, lr, dataset_len = "cos_lr", False, 6, 1, 1e-4, 5e-3, 2
    model, opt = _new_model_and_opt(1882)
    stats = [Hn.train(loader, model, crit, opt, ep, 6, A) for ep in range(2)]
    assert all(np.isfinite(v) for st in stats for v in st) and all(0.0 <= st[1] <= 100.0 and st[1] <= st[2] <= 100.0 for st in stats)
    assert abs(opt.param_groups[0]["lr"] - Hn.cosine_lr(5e-3, 1e-4, 1, 6, 1, 1, 2)) < 1e-12        # the last iteration's per-iteration cosine LR
    # val(): eval-mode BatchNorm, observers still moving
    q = model.hip_runner().qa.t
    before = q[:, 0].clone()
    lv, a1, a5 = Hn.val(loader, model, crit)
    assert np.isfinite(lv) and not torch.equal(before, q[:, 0]) and not model.training
    model.train()

    path = str(tmp_path / "checkpoint.pth.tar")
    Hn.save_checkpoint(Hn.checkpoint_state(model, opt, 1, lossTr=stats[-1][0], lr=opt.param_groups[0]["lr"]), path)
    model3, opt3 = _new_model_and_opt(4242)               # different seed: everything that matters must come from the checkpoint
    ck = Hn.load_checkpoint(model3, opt3, path)
    assert ck["epoch"] == 2 and ck["hip_rng"]["draws"] == 4 and ck["lossTr"] == stats[-1][0]
    sd, sd3 = model.state_dict(), model3.state_dict()
    assert list(sd) == list(sd3)
    for k in sd:
        assert torch.equal(sd[k].cpu(), sd3[k].cpu()), k
    o, o3 = opt.state_dict(), opt3.state_dict()
    for i in o["state"]:
        for k, v in o["state"][i].items():
            w = o3["state"][i][k]
            assert (torch.equal(v.cpu(), w.cpu()) if torch.is_tensor(v) else v == w), (i, k)
    assert opt3.is_warmup is False or opt3.is_warmup == opt.is_warmup
    opt3.is_warmup = opt.is_warmup
    # the dropout stream continues: draw 5 of both runners is the same mask (and differs from a rewound stream's)
    r, r3 = model.hip_runner(), model3.hip_runner()
    m, m3 = dropout_mask(r, 4 * 1280, 0.8), dropout_mask(r3, 4 * 1280, 0.8)
    torch.cuda.synchronize()
    assert torch.equal(m, m3) and r.rng_state() == r3.rng_state() and r.rng_state()["draws"] == 5
    r3.set_rng_state(dict(r3.rng_state(), draws=0))
    assert not torch.equal(dropout_mask(r3, 4 * 1280, 0.8), m)
    # the GradBoost stream continues: one step on identical gradients gives identical parameters
    g = torch.Generator(device="cuda").manual_seed(9)
    for p, p3 in zip(model.parameters(), model3.parameters()):
        p.grad = torch.randn(p.shape, device="cuda", generator=g) * 1e-3
        p3.grad = p.grad.clone()
    opt.step()
    opt3.step()
    torch.cuda.synchronize()
    for (n, p), p3 in zip(model.named_parameters(), model3.parameters()):
        assert torch.equal(p, p3), n


# ------------------------------------------------------------------------------------------------ quantizable hard-swish reachable from the module surface
def test_hswish_bottleneck_device_vs_stock_modules(engine):
    """`act="hswish"` puts the reference's quantizable `_Hswish` (Classification/models/imagenet/mobilenetv3.py:43-56, restated as
    frostnet_amd.frostnet.Hswish with the same FloatFunctional attribute names) behind every activated layer of a Frost bottleneck.  CPU = the
    stock torch QAT modules prepare_qat builds from it (what the reference's zoo executes); device = ConvBn2d emit + Engine.hswish.  Two training
    steps, teacher-forced input: outputs within one quantisation step, input and parameter gradients at the bf16-storage level."""
    import copy
    from torch.ao.quantization import get_default_qat_qconfig, prepare_qat
    from frostnet_amd import frostnet as F, runner as R
    torch.manual_seed(3)
    torch.set_num_threads(16)
    m = F.CascadePreExBottleneck(80, 80, quantized=True, kernel_size=5, stride=1, expand_ratio=3, reduce_factor=4, act="hswish")
    assert isinstance(m.conv1, F.ConvBNHswish) and isinstance(m.reduce_conv, F.ConvBN)
    _randomize_bn(m, 21)
    m.train()
    for mod in m.modules():
        if type(mod) in (F.ConvBNReLU, F.ConvBN, F.ConvBNHswish):
            mod.fuse_model()
    m.qconfig = get_default_qat_qconfig("qnnpack", version=0)
    prepare_qat(m, inplace=True)
    keys = list(m.state_dict().keys())
    assert "conv1.act.quant_mul1.activation_post_process.scale" in keys and "conv1.act.relu6.activation_post_process.scale" in keys      # the reference's key names
    ref = copy.deepcopy(m)
    m.cuda()
    run = R.FrostRunner.for_block(m)
    qx = run.qa.alloc()
    in_scale, in_zp, N, H = 0.0417, 117, 8, 14
    run.qa.set_qparams(qx, in_scale, in_zp)
    xi = torch.from_numpy(np.clip(np.round(O.synth((N, 80, H, H), 31) * 35 + 120), 0, 255).astype(np.uint8))
    xf = (xi.float() - in_zp) * in_scale
    qx[4], qx[5] = float(xf.min()), float(xf.max())

    def relerr(a, b):
        return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
    for step in range(2):
        gr = torch.from_numpy(O.synth((N, 80, H, H), 60 + step))
        xr = xf.clone().requires_grad_(True)
        ref.zero_grad()
        yr = ref(xr)
        yr.backward(gr.bfloat16().float())
        run.E.begin_step()
        x = run.E.act_from_indices(xi, qx)
        y = run.block_forward(run.block, x, True, True)
        yd = y.dequant().cpu()
        y.grad = engine.float_to_grad(gr.cuda())
        run.bind_grads()
        run.E.backward()
        torch.cuda.synchronize()
        ysc = float(ref.skip_add.activation_post_process.scale[0])
        d = (yd - yr.detach()).abs() / ysc
        e_dx = relerr(engine.grad_to_float(x.grad, x.n, x.h, x.w, x.c).cpu(), xr.grad)
        worst = max((relerr(p.grad.cpu(), dict(ref.named_parameters())[n].grad), n) for n, p in m.named_parameters())
        print(f"[hswish block step {step}] y: max {float(d.max()):.2f} steps, off by > 0.5 step {float((d > 0.5).float().mean()):.2e}; dx {e_dx:.2e}; worst parameter gradient {worst[0]:.2e} ({worst[1]})")
        assert float(d.max()) <= 2.01 and float((d > 0.5).float().mean()) <= 2e-2
        assert e_dx <= 5e-2 and worst[0] <= 5e-2, (e_dx, worst)
        sd_d, sd_r = m.state_dict(), ref.state_dict()
        for k in ("conv1.act.relu6.activation_post_process.scale", "conv1.act.quant_mul1.activation_post_process.scale", "conv2.act.quant_mul1.activation_post_process.scale",
                  "squeeze_conv.act.quant_mul1.activation_post_process.activation_post_process.max_val"):
            np.testing.assert_allclose(sd_d[k].float().cpu().numpy().reshape(-1), sd_r[k].float().numpy().reshape(-1), rtol=2e-3, atol=1e-6, err_msg=k)


def test_hswish_network_builds_and_trains_in_every_mode(engine):
    """FrostNet(act='hswish'): a QAT training step on the device (finite loss, non-zero gradients everywhere), eval forward, convert(); the float model trains on
    the float kernels and runs bf16 inference (rounds 3-5 refused those three; the numerics are held by tests/test_gpu_round6.py test_hswish_*)."""
    from frostnet_amd import frostnet as F
    torch.manual_seed(5)
    model = F.FrostNet(nclass=1000, mode="small", quantized=True, drop_rate=0.0, act="hswish")
    F.qat_prepare(model, version=0)
    model.cuda().train()
    x = torch.randn(4, 3, 64, 64, device="cuda")
    tgt = torch.tensor([1, 2, 3, 4], device="cuda")
    loss = torch.nn.functional.cross_entropy(model(x), tgt)
    loss.backward()
    torch.cuda.synchronize()
    assert np.isfinite(float(loss))
    dead = [n for n, p in model.named_parameters() if p.grad is None or not np.isfinite(float(p.grad.norm())) or float(p.grad.norm()) == 0.0]
    assert not dead, dead
    model.eval()
    with torch.no_grad():
        assert model(x).shape == (4, 1000)
        model.hip_convert()
        assert torch.isfinite(model(x)).all()
    fm = F.FrostNet(mode="small", act="hswish", drop_rate=0.0).cuda().train()
    lf = torch.nn.functional.cross_entropy(fm(x), tgt)
    lf.backward()
    torch.cuda.synchronize()
    assert type(fm.hip_runner()).__name__ == "FloatRunner" and np.isfinite(float(lf))
    dead = [n for n, p in fm.named_parameters() if p.grad is None or not np.isfinite(float(p.grad.norm())) or float(p.grad.norm()) == 0.0]
    assert not dead, dead
    fm.eval()
    assert torch.isfinite(fm.hip_infer_bf16(x)).all()


def test_convert_is_idempotent_and_refuses_a_silent_revert(engine):
    """the converted state lives on the device executor.  A second hip_convert() must not move the weight observers again, and a rebuilt
    executor (parameters moved) must not silently fall back to the fake-quant eval graph -- a different model (tests/test_gpu_convert.py)."""
    import copy
    from frostnet_amd import frostnet as F
    torch.manual_seed(11)
    model = F.MODEL_REGISTRY["frostnet_quant_small_1_0"](drop_rate=0.0)
    F.qat_prepare(model, version=0)
    model.cuda().train()
    x = torch.randn(4, 3, 64, 64, device="cuda")
    with torch.no_grad():
        model(x)
    model.hip_convert()
    with torch.no_grad():
        a = model(x).clone()
    wmax = model.conv1.conv[0].weight_fake_quant.activation_post_process.max_val.clone()
    model.hip_convert()                                   # no-op
    with torch.no_grad():
        b = model(x)
    assert torch.equal(a, b) and torch.equal(wmax, model.conv1.conv[0].weight_fake_quant.activation_post_process.max_val)
    snap = copy.deepcopy(model)                           # a copy carries the module tree and the "converted" mark, not the frozen int8 weights (they live on the
    for _ in range(2):                                    # executor): it refuses to run -- on every call -- instead of serving the fake-quant eval graph
        with pytest.raises(RuntimeError, match="hip_convert"):
            with torch.no_grad():
                snap.eval()(x)
    model.conv1.conv[0].weight.data = model.conv1.conv[0].weight.data.clone()        # a parameter moved: the executor must be rebuilt ...
    for _ in range(3):                                    # ... and every later call still refuses (the guard used to fire only once)
        with pytest.raises(RuntimeError, match="hip_convert"):
            with torch.no_grad():
                model(x)
    assert "_hip_runner" not in model.__dict__


def test_fused_reduce_emit_and_add_range_pass_equals_the_two_launches(engine):
    """frost_pw_ew_emit_add (block-boundary fusion) against frost_pw_ew(mode 2) + frost_add_minmax_observe on the same kept conv output:
    the int8 output and every field of the add's qrecord (EMA'd min / max, scale, zero point, fake-quantised range) must be bit-identical, for both the
    first observation (copy) and the moving average, and the arrival ticket must come back armed."""
    from frostnet_amd import _lib as L
    dev = "cuda"
    g = torch.Generator().manual_seed(21)
    for npix, cout in ((512 * 49, 192), (512 * 196, 80), (777, 96)):
        cint = torch.randint(-60000, 60000, (npix * cout + 64,), generator=g, dtype=torch.int32).to(dev)
        cpad = (cout + 15) // 16 * 16
        coef = torch.zeros(L.COEF_ROWS, cpad, device=dev)
        coef[L.COEF_A, :cout] = (torch.rand(cout, generator=g) * 4e-5 + 2e-5).to(dev)
        coef[L.COEF_B, :cout] = (torch.rand(cout, generator=g) * 0.4 - 0.2).to(dev)
        qa_, qy_ = engine.QArena(6, dev), None
        qy, qa, qs1, qs2 = qa_.alloc(), qa_.alloc(), qa_.alloc(), qa_.alloc()
        engine.QArena.set_qparams(qy, 0.0213, 121)
        engine.QArena.set_qparams(qa, 0.0377, 109)
        a = torch.randint(-128, 127, (npix * cout + 64,), generator=g, dtype=torch.int8).to(dev)
        y1, y2 = torch.zeros(npix * cout + 64, dtype=torch.int8, device=dev), torch.zeros(npix * cout + 64, dtype=torch.int8, device=dev)
        nst = L.load_library().frost_add_state_floats()       # {2 unused, ticket, per-workgroup range slots}: zeroed once
        st1, st2 = torch.zeros(nst, device=dev), torch.zeros(nst, device=dev)
        for rep in range(2):                                  # rep 0: first observation; rep 1: exponential moving average
            L.call("frost_pw_ew", L.ptr(cint), npix, cout, L.ptr(coef), L.ptr(qy), 0, 2, None, L.ptr(y1), L.stream())
            L.call("frost_add_minmax_observe", L.ptr(a), L.ptr(qa), L.ptr(y1), L.ptr(qy), npix * cout, L.ptr(st1), L.ptr(qs1), 1, L.stream())
            L.call("frost_pw_ew_emit_add", L.ptr(cint), npix, cout, L.ptr(coef), L.ptr(qy), 0, L.ptr(a), L.ptr(qa), L.ptr(y2), L.ptr(st2), L.ptr(qs2), 1, L.stream())
            torch.cuda.synchronize()
            assert torch.equal(y1[: npix * cout], y2[: npix * cout])
            assert torch.equal(qs1.view(torch.int32)[:8], qs2.view(torch.int32)[:8]), (qs1, qs2)
            assert int(st1.view(torch.int32)[2: 2 + L.TICKET_WORDS].abs().sum()) == 0 and int(st2.view(torch.int32)[2: 2 + L.TICKET_WORDS].abs().sum()) == 0      # tickets re-armed
            cint = (cint // 2).contiguous()                   # different data for the second observation


def test_statistics_finalize_handoff_stress():
    """The statistics -> finalize hand-off of every conv layer (last-workgroup-done ticket, agent-scope atomics read back by agent-scope loads, no fences on
    gfx950: csrc/frost_common.h) under a stress run: two copies of FrostNet-Large run 60 training-mode forwards on the same inputs; the forward is
    bit-reproducible, so one stale statistic anywhere shows up as a differing logit or state entry (keep this as a test, not a devtool)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "devtools", "stress_finalize.py"), "60", "16"], capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "0 steps with differing logits, 0 state entries differ" in out.stdout, out.stdout[-1000:]


def test_flag_summary_is_host_side_and_follows_apply():
    """The runner's `observe` summary is read from the device only after train() / eval() / model.apply(...) (no per-forward device -> host read),
    a disabled fake-quantizer is refused in training as well as in eval, and per-site flags keep working without any host involvement."""
    import torch
    from frostnet_amd import frostnet as F
    torch.manual_seed(0)
    m = F.MODEL_REGISTRY["frostnet_quant_small_0_5"]()
    F.qat_prepare(m, version=0)
    m.cuda().train()
    x = torch.randn(4, 3, 64, 64, device="cuda")
    m(x)
    r = m.hip_runner()
    assert r.flags_dirty is False
    reads = []
    orig = r.read_flags
    r.read_flags = lambda: (reads.append(1), orig())[1]
    m(x); m(x)
    assert not reads, "a training forward read the flags although nothing changed"
    m.eval()
    with torch.no_grad():
        m(x); m(x)
    assert len(reads) == 1, reads                      # once after the mode switch
    m.apply(torch.quantization.disable_observer)
    with torch.no_grad():
        m(x)
    assert len(reads) == 2 and r._obs_cached is False
    m.apply(torch.quantization.enable_observer)
    m.apply(torch.quantization.disable_fake_quant)
    m.train()
    with pytest.raises(NotImplementedError, match="fake_quant_enabled"):
        m(x)
    m.apply(torch.quantization.enable_fake_quant)
    m(x)


def test_input_quantisation_channels_last_fast_path_equals_generic():
    """QuantStub on a channels_last RGB batch takes the vectorised kernel (four pixels per thread); an NCHW-contiguous copy of the same values takes the generic
    strided kernel: identical bytes, including the zero-point pad channel."""
    from frostnet_amd import engine as EN
    torch.manual_seed(3)
    E, qa = EN.Engine("cuda"), EN.QArena(4, "cuda")
    q = qa.alloc()
    qa.set_qparams(q, 0.0187, 117)
    x = torch.randn(6, 3, 32, 20, device="cuda") * 2.0
    a = E.quantize_input(x.contiguous(), q, observe=False)
    b = E.quantize_input(x.contiguous(memory_format=torch.channels_last), q, observe=False)
    torch.cuda.synchronize()
    assert a.c == 4 and torch.equal(a.buf[: a.numel], b.buf[: b.numel])
