"""GPU tests of the host-side contract around the kernels: the Philox GradBoost path (the one real training uses), per-module
observer switches, optimizer checkpoint reload, model copies, device checks, the one-forward-one-backward guard."""
import copy
import math

import numpy as np
import pytest
import torch

from oracle import frost_oracle as O

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.fixture(scope="module")
def fa():
    import __graft_entry__ as ge
    ge.build()
    from frostnet_amd import frostnet, optimizer
    assert torch.cuda.is_available()
    return dict(F=frostnet, opt=optimizer)


# ------------------------------------------------------------------------------------------ Philox GradBoost path
def _philox4x32_10(c, k):
    """numpy Philox4x32-10 (Salmon et al., SC'11), c: [n,4] uint32 counters, k: (k0, k1)."""
    c = c.astype(np.uint64).copy()
    k0, k1 = np.uint64(k[0]), np.uint64(k[1])
    M0, M1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[:, 0], M1 * c[:, 2]
        n0 = ((p1 >> np.uint64(32)) ^ c[:, 1] ^ k0) & MASK
        n1 = p1 & MASK
        n2 = ((p0 >> np.uint64(32)) ^ c[:, 3] ^ k1) & MASK
        n3 = p0 & MASK
        c = np.stack([n0, n1, n2, n3], 1)
        k0 = (k0 + np.uint64(0x9E3779B9)) & MASK
        k1 = (k1 + np.uint64(0xBB67AE85)) & MASK
    return c.astype(np.uint32)


def _noise_probe(fo, sizes, seed, steps=1, toss=True):
    """QSGD configured so that the injected noise can be read back exactly: g = 1 everywhere, first step => exp_max = 1, exp_min = 0,
    noise_decay = 0, no clipping, no weight decay, lr = 0  =>  p.grad after the step = 1 + |Laplace| * coin."""
    ps = [torch.nn.Parameter(torch.zeros(n, device="cuda")) for n in sizes]
    opt = fo.QSGD([{"params": [p]} for p in ps], lr=0.0, momentum=0.0, weight_decay=0.0, clip_by=0.0, toss_coin=toss, noise_decay=0.0)
    opt._seed = seed
    opt.is_warmup = False
    out = []
    for _ in range(steps):
        for p in ps:
            p.grad = torch.ones_like(p)
        opt.step()
        torch.cuda.synchronize()
        out.append([(p.grad.detach().cpu().double().numpy() - 1.0) for p in ps])
    return opt, ps, out


def test_philox_noise_known_answer_and_moments(fa):
    """optimizer.py:170-189 draws |Laplace(0,1)| (= Exp(1)) and a fair coin per element; here they come from Philox4x32-10 on
    device (frost_optim.hip:35-41): counter = (global element index, step), key = seed; lap = -log((r0>>8 + 0.5) / 2^24), coin = r1>>31."""
    fo = fa["opt"]
    sizes = [200_003, 50_000, 77]
    seed = 1882
    opt, ps, out = _noise_probe(fo, sizes, seed, steps=2)
    # step 1: exp_max = |g| = 1 exactly.  Known answer against a host Philox for every element of the stream
    n_tot = sum(sizes)
    idx = np.arange(n_tot, dtype=np.uint64)
    ctr = np.stack([idx & 0xFFFFFFFF, idx >> 32, np.full(n_tot, 1, np.uint64), np.zeros(n_tot, np.uint64)], 1)
    r = _philox4x32_10(ctr, (seed & 0xFFFFFFFF, seed >> 32))
    u = ((r[:, 0] >> 8).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
    lap = -np.log(u.astype(np.float64))
    coin = (r[:, 1] >> 31).astype(np.float64)
    got = np.concatenate(out[0])
    np.testing.assert_allclose(got, lap * coin, rtol=2e-6, atol=1e-7)
    # the coin state is written (optimizer.py:183 keeps it in state['coin_toss']): after the second step it holds step 2's coins
    ctr[:, 2] = 2
    coin2 = (_philox4x32_10(ctr, (seed & 0xFFFFFFFF, seed >> 32))[:, 1] >> 31).astype(np.float32)
    coins = np.concatenate([opt.state[p]["coin_toss"].cpu().numpy() for p in ps])
    assert np.array_equal(coins, coin2) and not np.array_equal(coin2, coin.astype(np.float32))
    # moments of the distribution the reference samples: |Laplace(0,1)| has mean 1, variance 1; the coin is fair
    heads = coin == 1
    assert abs(coin.mean() - 0.5) < 5e-3
    assert abs(got[heads].mean() - 1.0) < 1.5e-2 and abs(got[heads].var() - 1.0) < 5e-2
    assert float(got[~heads].max()) == 0.0
    # tail: P(x > 3) = e^-3
    assert abs((got[heads] > 3.0).mean() - math.exp(-3.0)) < 3e-3


def test_philox_streams_do_not_overlap(fa):
    fo = fa["opt"]
    # tensors share ONE counter space (prefix offsets): equal-sized tensors in one step get different draws
    _, _, out = _noise_probe(fo, [4096, 4096], 7, steps=3, toss=False)
    a0, b0 = out[0]
    assert not np.array_equal(a0, b0) and np.corrcoef(a0, b0)[0, 1] < 0.1
    # steps advance the stream: step 2 / step 3 of the same tensor are fresh (their scale differs -- exp_max inflates -- so compare ranks)
    r1, r2 = np.argsort(np.argsort(out[1][0])), np.argsort(np.argsort(out[2][0]))
    assert abs(np.corrcoef(r1, r2)[0, 1]) < 0.1
    # same seed, same step -> same draws (replayable); another seed -> different
    _, _, again = _noise_probe(fo, [4096, 4096], 7, steps=1, toss=False)
    _, _, other = _noise_probe(fo, [4096, 4096], 8, steps=1, toss=False)
    assert np.array_equal(again[0][0], a0) and not np.array_equal(other[0][0], a0)


def test_optimizer_state_dict_reload_continues(fa):
    """the cached device table must not survive load_state_dict; the loaded moments / step counts are used and the
    Philox stream continues from the step count instead of restarting."""
    fo = fa["opt"]

    def make():
        torch.manual_seed(0)
        ps = [torch.nn.Parameter(torch.randn(n, device="cuda")) for n in (1000, 333)]
        opt = fo.QSGD([{"params": [p], "weight_decay": wd} for p, wd in zip(ps, (1e-4, 0.0))], lr=1e-2, momentum=0.9, nesterov=True)
        opt.is_warmup = False
        return ps, opt

    def grads(ps, step):
        for i, p in enumerate(ps):
            p.grad = (T(O.synth((p.numel(),), 900 + 10 * step + i)) * 0.01).cuda()

    ps_a, opt_a = make()
    for s in range(4):
        grads(ps_a, s)
        opt_a.step()
    ps_b, opt_b = make()
    for s in range(2):
        grads(ps_b, s)
        opt_b.step()
    sd = copy.deepcopy(opt_b.state_dict())
    saved = [p.detach().clone() for p in ps_b]
    ps_c, opt_c = make()
    grads(ps_c, 0)
    opt_c.step()                                   # builds a plan that load_state_dict must invalidate
    with torch.no_grad():
        for p, v in zip(ps_c, saved):
            p.copy_(v)
    opt_c.load_state_dict(sd)
    opt_c.is_warmup = False
    for g in opt_c.param_groups:
        g["lr"] = 1e-2
    for s in range(2, 4):
        grads(ps_c, s)
        opt_c.step()
    torch.cuda.synchronize()
    for pa, pc in zip(ps_a, ps_c):
        assert torch.equal(pa, pc)                 # resumed run == uninterrupted run, bit for bit (same noise stream)
    assert opt_c.state[ps_c[0]]["step"] == 4
    with pytest.raises(NotImplementedError):
        bad = fo.QSGD([{"params": [ps_a[0]], "momentum": 0.9}, {"params": [ps_a[1]], "momentum": 0.5}], lr=1e-2)
        bad.step()


# ------------------------------------------------------------------------------------------ module surface
def _small(fa, **kw):
    F = fa["F"]
    torch.manual_seed(3)
    m = F.frostnet_quant_small_1_0(drop_rate=0.0, **kw)
    F.qat_prepare(m, version=0)
    return m.cuda()


def test_observer_switch_is_per_module_and_device_resident(fa):
    """torch.quantization.disable_observer applied to ONE sub-module freezes that module's observers only (no
    network-wide 'last call wins', no instance patching)."""
    m = _small(fa).train()
    x = torch.randn(4, 3, 64, 64, device="cuda")
    with torch.no_grad():
        m(x)
    blk = m.layer3[1]
    blk.apply(torch.quantization.disable_observer)
    assert "disable_observer" not in vars(blk.conv1.conv[0].activation_post_process)          # nothing patched onto instances
    before = {k: v.clone() for k, v in m.state_dict().items() if k.endswith("min_val") or k.endswith("max_val") or k.endswith("scale")}
    with torch.no_grad():
        m(x * 3.0)
    after = m.state_dict()
    frozen = [k for k in before if k.startswith("layer3.1.")]
    moving = [k for k in before if k.startswith("layer3.2.") and "activation_post_process" in k and "weight_fake_quant" not in k]
    assert frozen and moving
    for k in frozen:
        assert torch.equal(before[k], after[k]), k
    assert any(not torch.equal(before[k], after[k]) for k in moving)
    assert m.hip_runner().observe is True
    m.apply(torch.quantization.disable_observer)
    assert m.hip_runner().observe is False
    blk.apply(torch.quantization.enable_observer)
    assert m.hip_runner().observe is True
    m.apply(torch.quantization.disable_fake_quant)
    with pytest.raises(NotImplementedError):
        m.eval()(x)


def test_deepcopy_is_independent(fa, tmp_path):
    """EMA / best-model snapshots (timm ModelEma in the published recipe): a deep copy owns its observers and runner."""
    m = _small(fa).train()
    x = torch.randn(4, 3, 64, 64, device="cuda")
    with torch.no_grad():
        m(x)
    c = copy.deepcopy(m)
    assert "_hip_runner" not in vars(c)
    c.apply(torch.quantization.disable_observer)
    assert m.hip_runner().observe is True                      # the original keeps observing
    k = "conv1.conv.0.activation_post_process.activation_post_process.max_val"
    v_m, v_c = m.state_dict()[k].clone(), c.state_dict()[k].clone()
    with torch.no_grad():
        m(x * 2.0)
        c(x * 2.0)
    assert not torch.equal(m.state_dict()[k], v_m) and torch.equal(c.state_dict()[k], v_c)
    assert c.hip_runner() is not m.hip_runner() and c.hip_runner().model is c
    # state_dict round trip through a file into a fresh model (torch.save(model) itself is impossible for ANY prepare_qat'ed module:
    # the qconfig holds local closures -- stock torch limitation, not ours)
    torch.save({"state_dict": m.state_dict(), "epoch": 1}, tmp_path / "m.pt")          # Classification/train.py:210-218
    m2 = _small(fa)
    m2.load_state_dict(torch.load(tmp_path / "m.pt")["state_dict"])
    with torch.no_grad():
        m.eval(); m2.eval()
        assert torch.equal(m(x), m2(x))


def test_guards(fa):
    m = _small(fa).train()
    x = torch.randn(2, 3, 64, 64, device="cuda")
    # one recorded forward per model: the first graph's backward must not silently use the second forward's tape
    l1 = m(x).sum()
    l2 = m(x).sum()
    with pytest.raises(RuntimeError, match="tape"):
        l1.backward()
    l2.backward()
    with pytest.raises(NotImplementedError):
        m(x.clone().requires_grad_(True))
    # frozen BatchNorm inside a training forward (`_freeze_stages`) is honoured per layer on the fake-quant path -- its running statistics do not move
    # (parity with the oracle: tests/test_gpu_round4.py) -- and refused, not silently normalised with batch statistics, on the float path
    bn = m.layer2[0].conv1.conv[0].bn
    bn.eval()
    rm, nbt = bn.running_mean.clone(), int(bn.num_batches_tracked)
    other = m.layer2[0].conv2.conv[0].bn.running_mean.clone()
    m(x).sum().backward()
    assert torch.equal(rm, bn.running_mean) and int(bn.num_batches_tracked) == nbt and not torch.equal(other, m.layer2[0].conv2.conv[0].bn.running_mean)
    m.train()
    fm = fa["F"].MODEL_REGISTRY["frostnet_small_1_0"]().cuda().train()
    fm.layer2[0].conv1.conv[1].eval()
    with pytest.raises(NotImplementedError, match="BatchNorm"):
        fm(x)
    # a DataParallel replica (shares the original's runner and pointers) is refused
    rep = m._replicate_for_data_parallel()
    with pytest.raises(RuntimeError, match="one process per"):
        rep(x)
    if torch.cuda.device_count() > 1:
        with pytest.raises(RuntimeError, match="is on"):
            m(x.to("cuda:1"))


def test_checkpoint_ingest_ema_and_module_prefix(fa, tmp_path):
    """frostnet_features.py:10-35 (F2): timm checkpoint with `state_dict_ema` and `module.` prefixes round-trips into the HIP model."""
    from frostnet_amd import frostnet_features as FF
    torch.manual_seed(5)
    src = FF.FrostNet(mode="small", width_mult=1.0)
    src._init_weights()
    ema = {("module." + k): (v + 0.25 if v.dtype.is_floating_point else v) for k, v in src.state_dict().items()}
    ema["module.classifier.2.weight"] = torch.zeros(3)                      # a head the backbone does not have (non-strict load)
    path = tmp_path / "ckpt.pth.tar"
    torch.save({"epoch": 3, "state_dict": {"module." + k: v for k, v in src.state_dict().items()}, "state_dict_ema": ema}, path)
    dst = FF.FrostNet(mode="small", width_mult=1.0, quantized=True)
    rec = dst.init_weights(str(path))
    assert rec.unexpected_keys == ["classifier.2.weight"] and not rec.missing_keys
    k = "layer3.0.conv1.conv.1.running_var"
    assert torch.equal(dst.state_dict()[k], src.state_dict()[k] + 0.25)      # EMA weights preferred
    assert torch.equal(FF.load_state_dict(str(path), use_ema=False)["conv1.conv.0.weight"], src.state_dict()["conv1.conv.0.weight"])
    with pytest.raises(FileNotFoundError):
        FF.load_state_dict(str(tmp_path / "nope.pth"))
    fa["F"].qat_prepare(dst, version=0)
    dst.cuda().eval()
    with torch.no_grad():
        feats = dst(torch.randn(2, 3, 64, 64, device="cuda"))
    assert [f.shape[1] for f in feats] == [24, 40, 96, 320] and all(torch.isfinite(f).all() for f in feats)


def test_flops_counter_style_hooks_survive(fa):
    """the reference wraps the model with add_flops_counting_methods (Classification/train.py:80-84: methods bound onto the module,
    forward hooks and `__flops__` attributes on every conv) BEFORE moving it to the GPU and preparing QAT.  Same treatment here: the CPU pass
    counts through the stock modules, the HIP path afterwards is unaffected by the leftovers."""
    F = fa["F"]
    torch.manual_seed(1)
    m = F.frostnet_quant_small_1_0(drop_rate=0.0)
    macs = {"n": 0}

    def hook(mod, inp, out):
        mod.__flops__ += out.numel() // out.shape[0] * (mod.in_channels // mod.groups) * mod.kernel_size[0] * mod.kernel_size[1]
        macs["n"] += 1
    for mod in m.modules():
        if isinstance(mod, torch.nn.Conv2d):
            mod.__flops__ = 0
            mod.__flops_handle__ = mod.register_forward_hook(hook)
    m.compute_average_flops_cost = (lambda self: sum(mod.__flops__ for mod in self.modules() if isinstance(mod, torch.nn.Conv2d))).__get__(m)
    m.eval()
    with torch.no_grad():
        m(torch.zeros(1, 3, 224, 224))
    assert macs["n"] == 54 and m.compute_average_flops_cost() == 315_344_912          # FrostNet-Small @224, 54 convs
    F.qat_prepare(m, version=0)
    m.cuda().train()
    x = torch.randn(2, 3, 64, 64, device="cuda")
    loss = m(x).sum()
    loss.backward()
    assert torch.isfinite(loss) and all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


def test_cross_entropy_and_dropout_kernels(fa):
    """K13: nn.CrossEntropyLoss forward + backward and the Dropout mask as hand-written kernels (frostnet_amd.harness.CrossEntropyLoss,
    frost_dropout_mask) against torch's fp32 definitions."""
    from frostnet_amd import harness as H, _lib as L
    torch.manual_seed(0)
    x = (torch.randn(37, 1000, device="cuda") * 3).requires_grad_(True)
    t = torch.randint(0, 1000, (37,), device="cuda")
    loss = H.CrossEntropyLoss()(x, t)
    (loss * 2.5).backward()
    xr = x.detach().clone().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(xr, t)
    (ref * 2.5).backward()
    torch.testing.assert_close(loss, ref, rtol=2e-6, atol=1e-6)
    torch.testing.assert_close(x.grad, xr.grad, rtol=1e-5, atol=1e-8)
    # dropout mask: values in {0, 1/keep}, keep rate, fresh draw per call, reproducible for the same (seed, draw)
    ctr = torch.zeros(2, dtype=torch.int64, device="cuda")           # {draw counter, arrival ticket}
    n, keep = 512 * 1280, 0.8
    a, b = torch.empty(n, device="cuda"), torch.empty(n, device="cuda")
    L.call("frost_dropout_mask", L.ptr(ctr), 1882, n, keep, L.ptr(a), L.stream())
    L.call("frost_dropout_mask", L.ptr(ctr), 1882, n, keep, L.ptr(b), L.stream())
    assert ctr.tolist() == [2, 0]
    assert set(torch.unique(a).tolist()) == {0.0, 1.25}
    assert abs(float((a > 0).float().mean()) - keep) < 3e-3 and abs(float((b > 0).float().mean()) - keep) < 3e-3
    assert not torch.equal(a, b) and abs(float(((a > 0) == (b > 0)).float().mean()) - (keep * keep + (1 - keep) ** 2)) < 5e-3
    ctr.zero_()
    c = torch.empty(n, device="cuda")
    L.call("frost_dropout_mask", L.ptr(ctr), 1882, n, keep, L.ptr(c), L.stream())
    assert torch.equal(a, c)


# ------------------------------------------------------------------------------------------ gradient accumulation (torch semantics)
@pytest.mark.parametrize("quant", [True, False])
def test_backward_accumulates_until_zero_grad(fa, quant):
    """`loss.backward()` twice without `zero_grad()` leaves the SUM of the two gradients in p.grad (torch semantics: the kernels write the flat
    arena, the runner carries uncleared gradients over); `zero_grad()` -- either flavour -- starts over.  Compared against a twin model that
    clears in between.  Tolerance: the stochastic bf16 rounding of dc (QAT); the float model runs in its fp32 mode, where the sum is exact to rounding."""
    F = fa["F"]

    def make():
        torch.manual_seed(3)
        m = (F.frostnet_quant_small_1_0 if quant else F.frostnet_small_1_0)(drop_rate=0.0)
        if quant:
            F.qat_prepare(m, version=0)
        else:
            m.float_precision = "fp32"           # the bf16 float path's run-to-run spread (atomics order x bf16 storage; tests/devtools/dbg_accumulate.py)
        return m.cuda().train()                  # would hide the mechanism under test; in fp32 the sum is exact to 1e-6
    x1, x2 = torch.randn(4, 3, 64, 64, device="cuda"), torch.randn(4, 3, 64, 64, device="cuda")
    t = torch.tensor([1, 2, 3, 4], device="cuda")
    ce = torch.nn.functional.cross_entropy
    flat = lambda m: torch.cat([p.grad.detach().reshape(-1) for p in m.parameters()]).clone()
    a = make()                                   # twin: separate gradients
    ce(a(x1), t).backward(); g1 = flat(a)
    a.zero_grad()                                # set_to_none=True: p.grad is None
    assert all(p.grad is None for p in a.parameters())
    ce(a(x2), t).backward(); g2 = flat(a)
    b = make()                                   # accumulating model: same two batches, nothing cleared in between
    ce(b(x1), t).backward()
    ce(b(x2), t).backward()
    gacc = flat(b)
    rel = float((gacc - (g1 + g2)).norm() / (g1 + g2).norm())
    print(f"[accumulate quant={quant}] |g_acc - (g1 + g2)| / |g1 + g2| = {rel:.2e}")
    assert rel <= (3e-2 if quant else 1e-4), rel
    assert float((gacc - g2).norm() / g2.norm()) > 0.3                    # and it is not just the last gradient
    c = copy.deepcopy(b)                         # same state from here on: clearing in place vs clearing to None give the same next gradient
    b.zero_grad(set_to_none=False)
    c.zero_grad()
    ce(b(x2), t).backward(); ce(c(x2), t).backward()
    assert float((flat(b) - flat(c)).norm() / flat(c).norm()) <= (3e-2 if quant else 1e-4)
