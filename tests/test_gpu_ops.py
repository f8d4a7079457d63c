"""GPU parity tests, per operator, through the C ABI (libfrost_hip.so) against (a) the golden vectors produced by
the real reference and (b) the CPU oracle on the same seeded inputs.

Tolerances (stated once, used everywhere):
  * integer indices of fake-quantised activations: |delta| <= 1 and flip rate <= FLIP_RATE.  The HIP path computes the
    conv exactly in integers and the BN affine as one fma, the reference as an fp32 conv + division + batch_norm, so a
    value that lands within ~1e-6 of a rounding boundary may fall on the other side (SURVEY H-2: the reference
    disagrees with itself at this level when only its thread count changes).
  * observer / qparam / running-stat scalars: rel 2e-5.
  * gradients: activations' gradients are stored in bf16 and the dgrad/wgrad GEMMs take bf16 operands with fp32
    accumulation -> norm-wise relative error <= GRAD_TOL.
"""
import numpy as np
import pytest
import torch

from oracle import frost_oracle as O

pytestmark = pytest.mark.gpu
FLIP_RATE = 5e-4       # measured 0 .. 3.4e-4 (K = 624 / 1728 layers; the reference's fp32 conv sum rounds where the integer sum does not)
GRAD_TOL = 2e-2


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.fixture(scope="module")
def eng_mod():
    import __graft_entry__ as ge
    ge.build()
    from frostnet_amd import engine
    assert torch.cuda.is_available()
    return engine


def relerr(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def idx_compare(mine, ref):
    d = (mine.to(torch.int16) - ref.to(torch.int16)).abs()
    return int(d.max()), float((d > 0).float().mean())


# ------------------------------------------------------------------------------------------ G1 / G2
def test_fake_quant_kat(eng_mod, golden):
    from frostnet_amd._lib import call, ptr, stream
    g = golden("g1_fake_quant")
    dev = "cuda"
    qa = eng_mod.QArena(4, dev)
    x = T(g["edge_x"]).to(dev)
    rec = qa.alloc()
    qa.set_qparams(rec, 1.0, 0)
    y = torch.empty_like(x)
    m = torch.empty(x.numel(), dtype=torch.uint8, device=dev)
    call("frost_fake_quant_f32", ptr(x), x.numel(), ptr(rec), 0, 255, ptr(y), ptr(m), stream())
    assert y.cpu().tolist() == [0, 2, 2, 0, 0, 254, 255, 255, 0, 0, 255]
    assert m.cpu().tolist() == [1, 1, 1, 1, 0, 1, 0, 0, 1, 0, 0]
    for name in ("act", "wgt", "act0"):
        s, zp, qmin, qmax, seed = g[name + "_qp"]
        x = T(O.synth((4099,), int(seed)) * (3.0 if name != "wgt" else 0.5)).to(dev)
        gr = T(O.synth((4099,), int(seed) + 100)).to(dev)
        rec = qa.alloc()
        qa.set_qparams(rec, s, int(zp))
        y, m, dx = torch.empty_like(x), torch.empty(x.numel(), dtype=torch.uint8, device=dev), torch.empty_like(x)
        call("frost_fake_quant_f32", ptr(x), x.numel(), ptr(rec), int(qmin), int(qmax), ptr(y), ptr(m), stream())
        call("frost_fake_quant_bwd_f32", ptr(gr), ptr(m), x.numel(), ptr(dx), stream())
        assert np.array_equal(y.cpu().numpy(), g[name + "_y"])          # bit-exact
        assert np.array_equal(dx.cpu().numpy(), g[name + "_dx"])


@pytest.mark.parametrize("ver", [0, 1])
def test_observer_traj(eng_mod, golden, ver):
    from frostnet_amd._lib import call, ptr, stream
    g = golden("g2_observer")
    dev = "cuda"
    qa = eng_mod.QArena(2, dev)
    for name, sym in (("act", 0), ("wgt", 1)):
        scale = float(g[f"v{ver}_{name}_inscale"])
        rec = qa.alloc()
        mm = torch.empty(2, device=dev)
        for step in range(3):
            x = T(O.synth((3, 8, 5, 5), 200 + step) * scale * (1 + 0.3 * step) + (0.4 if name == "act" else 0.0)).to(dev)
            call("frost_fill_minmax", ptr(mm), 1, stream())
            call("frost_minmax_f32", ptr(x), x.numel(), ptr(mm), stream())
            call("frost_observer_update", ptr(rec), ptr(mm), sym, ver, 1, stream())
            y = torch.empty_like(x)
            call("frost_fake_quant_f32", ptr(x), x.numel(), ptr(rec), -128 if sym else 0, 127 if sym else 255, ptr(y), None, stream())
            q = qa.get(rec)
            row = g[f"v{ver}_{name}_traj"][step]
            mine = np.float32([q["min_val"], q["max_val"], q["scale"], q["zero_point"]])
            if ver == 0:
                assert np.array_equal(mine, row), (name, step, mine, row)
                assert np.array_equal(y.cpu().numpy(), g[f"v{ver}_{name}_y{step}"])
            else:
                np.testing.assert_allclose(mine, row, rtol=3e-7)


# ------------------------------------------------------------------------------------------ G3 teacher-forced layers
G3 = ["stem", "pw16_96", "dw3s2_96", "dw5s2_144", "dw5s1_624", "pw624_96_lin", "pw288_1728", "pw1728_320_lin"]


def make_layer(engine, g, dev, name="L"):
    cin, cout, k, s, groups, H, N, xseed, gseed, relu, wseed = [int(v) for v in g["spec"]]
    keys = [str(k_) for k_ in g["init_keys"]]
    shapes = [tuple(int(x) for x in row[:n]) for row, n in zip(g["init_shapes"], g["init_ndims"])]
    sd = {k_: v.to(dev) for k_, v in O.synth_state(keys, shapes, wseed).items()}
    kind = "stem" if (groups == 1 and k == 3) else ("dw" if groups > 1 else "pw")
    E = engine.Engine(dev)
    qa = engine.QArena(4, dev)
    w = sd["conv.0.weight"].contiguous().requires_grad_(True)
    gamma, beta = sd["conv.1.weight"].requires_grad_(True), sd["conv.1.bias"].requires_grad_(True)
    l = engine.ConvLayer(name, kind, w, gamma, beta, sd["conv.1.running_mean"], sd["conv.1.running_var"],
                         torch.zeros((), dtype=torch.int64, device=dev), None, k, s, bool(relu), qa.alloc(), qa.alloc())
    E.add_layer(l)
    return E, qa, l


@pytest.mark.parametrize("name", G3)
def test_g3_layer(eng_mod, golden, name):
    engine = eng_mod
    dev = "cuda"
    g = golden("g3_" + name)
    cin, cout, k, s, groups, H, N, xseed, gseed, relu, wseed = [int(v) for v in g["spec"]]
    E, qa, l = make_layer(engine, g, dev)
    in_scale, in_zp = float(g["in_qp"][0]), int(g["in_qp"][1])
    qx = qa.alloc()
    qa.set_qparams(qx, in_scale, in_zp)
    xi = T(g["x_idx"])
    if l.kind == "stem":  # 3 channels -> pad to 4 with the zero point
        xi = torch.cat([xi, torch.full_like(xi[:, :1], in_zp)], 1)
    for step in range(2):
        E.begin_step()
        x = E.act_from_indices(xi, qx)
        y = E.conv(l, x, training=True, observe=True)
        gr = T(O.synth((N, cout, y.h, y.w), gseed + 50 * step)).to(dev)
        y.grad = engine.float_to_grad(gr)
        yidx = y.indices().cpu()
        E.backward()
        torch.cuda.synchronize()
        mx, rate = idx_compare(yidx, T(g[f"s{step}_yidx"]))
        assert mx <= 1 and rate <= FLIP_RATE, (name, step, mx, rate)
        qy, qw = qa.get(l.qy), qa.get(l.qw)
        pre = f"s{step}_sd/conv/0/"
        np.testing.assert_allclose(qy["scale"], float(g[pre + "activation_post_process/scale"][0]), rtol=2e-5)
        assert qy["zero_point"] == int(g[pre + "activation_post_process/zero_point"][0])
        np.testing.assert_allclose(qy["min_val"], float(g[pre + "activation_post_process/activation_post_process/min_val"]), rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(qy["max_val"], float(g[pre + "activation_post_process/activation_post_process/max_val"]), rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(qw["scale"], float(g[pre + "weight_fake_quant/scale"][0]), rtol=1e-6)
        np.testing.assert_allclose(l.rmean.cpu().numpy(), g[pre + "bn/running_mean"], rtol=1e-3, atol=2e-4)  # ref fp32 conv rounding at K=1728
        np.testing.assert_allclose(l.rvar.cpu().numpy(), g[pre + "bn/running_var"], rtol=1e-3, atol=2e-4)
        assert int(l.nbt) == step + 1
        # gradients
        if l.kind != "stem":
            dx = engine.grad_to_float(x.grad, x.n, x.h, x.w, x.c).cpu()
            assert relerr(dx, T(g[f"s{step}_dx"])) <= GRAD_TOL, (name, step, relerr(dx, T(g[f"s{step}_dx"])))
        pack = g[f"s{step}_dw"]
        dw = l.w.grad.detach().double().cpu().numpy().reshape(-1)
        mine = O.sample_big(dw)
        e = np.linalg.norm(mine - pack[3:]) / (np.linalg.norm(pack[3:]) + 1e-30)
        assert e <= GRAD_TOL, (name, step, "dw", e)
        assert relerr(l.gamma.grad.cpu(), T(g[f"s{step}_dgamma"])) <= GRAD_TOL
        assert relerr(l.beta.grad.cpu(), T(g[f"s{step}_dbeta"])) <= GRAD_TOL


# ------------------------------------------------------------------------------------------ cat / add vs oracle
def test_cat_add(eng_mod):
    engine = eng_mod
    dev = "cuda"
    E = engine.Engine(dev)
    qa = engine.QArena(8, dev)
    N, H = 2, 9
    ia = np.clip(np.round(O.synth((N, 24, H, H), 70) * 50 + 100), 0, 255).astype(np.uint8)
    ib = np.clip(np.round(O.synth((N, 80, H, H), 71) * 40 + 128), 0, 255).astype(np.uint8)
    ic = np.clip(np.round(O.synth((N, 80, H, H), 72) * 60 + 90), 0, 255).astype(np.uint8)
    qs = O.QState()
    recs = {}
    for nm, s, zp in (("a", 0.021, 0), ("b", 0.013, 121), ("c", 0.017, 97)):
        recs[nm] = qa.alloc()
        qa.set_qparams(recs[nm], s, zp)
    fa = (T(ia.astype(np.float32)) - 0) * 0.021
    fb = (T(ib.astype(np.float32)) - 121) * 0.013
    fc = (T(ic.astype(np.float32)) - 97) * 0.017
    # the cat observer reads the producers' fake-quantised min/max from their qrecords: fill them like a producer would
    for nm, f in (("a", fa), ("b", fb), ("c", fc)):
        recs[nm][4] = float(f.min()); recs[nm][5] = float(f.max())
    A, B, Cc = E.act_from_indices(T(ia), recs["a"]), E.act_from_indices(T(ib), recs["b"]), E.act_from_indices(T(ic), recs["c"])
    qcat, qadd = qa.alloc(), qa.alloc()
    for step in range(2):
        ycat = E.cat(A, B, qcat)
        yadd = E.add(B, Cc, qadd)
        ocat = qs.fq_site("cat", torch.cat([fa, fb], 1), O.ACT)
        oadd = qs.fq_site("add", fb + fc, O.ACT)
        for y, o, key, q in ((ycat, ocat, "cat", qcat), (yadd, oadd, "add", qadd)):
            sc, zp = qs.sd[key + ".scale"][0], qs.sd[key + ".zero_point"][0]
            got = qa.get(q)
            assert got["zero_point"] == int(zp) and abs(got["scale"] - float(sc)) <= 1e-7 * float(sc) + 1e-12
            assert torch.equal(y.indices().cpu().to(torch.int64), O.fq_index(o, sc, zp))      # bit-exact
    # backward masks
    E.tape = []
    qcat2, qadd2 = qa.alloc(), qa.alloc()
    qa.set_qparams(qcat2, 0.012, 60)      # narrow range so some values clamp
    qa.set_qparams(qadd2, 0.015, 40)
    ycat = E.cat(A, B, qcat2, observe=False)
    yadd = E.add(B, Cc, qadd2, observe=False)
    g1, g2 = T(O.synth((N, 104, H, H), 75)).to(dev), T(O.synth((N, 80, H, H), 76)).to(dev)
    ycat.grad, yadd.grad = engine.float_to_grad(g1), engine.float_to_grad(g2)
    E.backward()
    xa, xb, xc = fa.clone().requires_grad_(True), fb.clone().requires_grad_(True), fc.clone().requires_grad_(True)
    o1 = O.fake_quant(torch.cat([xa, xb], 1), 0.012, 60, 0, 255)
    o2 = O.fake_quant(xb + xc, 0.015, 40, 0, 255)
    (o1 * g1.cpu().bfloat16().float()).sum().backward()
    (o2 * g2.cpu().bfloat16().float()).sum().backward()
    ga = engine.grad_to_float(A.grad, N, H, H, 24).cpu()
    gb = engine.grad_to_float(B.grad, N, H, H, 80).cpu()
    gc = engine.grad_to_float(Cc.grad, N, H, H, 80).cpu()
    assert torch.equal(ga, xa.grad)
    assert relerr(gb, xb.grad) <= 4e-3          # sum of two bf16 contributions, re-rounded to bf16
    assert torch.equal(gc, xc.grad)


# ------------------------------------------------------------------------------------------ GradBoost kernel
OPT = {
    "QSGD": (0, dict(lr=5e-3, momentum=0.9, weight_decay=1e-5, nesterov=1, clip_by=1e-3, toss_coin=1, noise_decay=1e-2)),
    "QSGD_plain": (0, dict(lr=1e-2, momentum=0.9, weight_decay=0.0, nesterov=0, clip_by=0.0, toss_coin=0, noise_decay=5e-2)),
    "QRMS": (1, dict(lr=1e-3, alpha=0.9, momentum=0.9, eps=1e-8, weight_decay=1e-5, clip_by=1e-3, toss_coin=1, noise_decay=1e-2)),
    "QAdam": (2, dict(lr=1e-3, eps=1e-8, weight_decay=1e-4, amsgrad=0, clip_by=1e-3, toss_coin=1, noise_decay=1e-2)),
    "QAdam_ams": (2, dict(lr=1e-3, eps=1e-8, weight_decay=0.0, amsgrad=1, clip_by=1e-3, toss_coin=1, noise_decay=1e-2)),
    "QAdamW": (3, dict(lr=1e-3, eps=1e-8, weight_decay=1e-2, amsgrad=0, clip_by=1e-3, toss_coin=1, noise_decay=1e-2)),
}


@pytest.mark.parametrize("name", list(OPT))
def test_gradboost_kernel(eng_mod, golden, name):
    from frostnet_amd import optimizer as fo
    g = golden("g6_optimizers")
    n = int(g["n"])
    dev = "cuda"
    kind, hp = OPT[name]
    p = torch.nn.Parameter((T(O.synth((n,), 600)) * 0.1).to(dev))
    cls = [fo.QSGD, fo.QRMSprop, fo.QAdam, fo.QAdamW][kind]
    kw = dict(lr=hp["lr"], weight_decay=hp["weight_decay"], clip_by=hp["clip_by"], toss_coin=bool(hp["toss_coin"]),
              noise_decay=hp["noise_decay"])
    if kind == 0:
        kw.update(momentum=hp["momentum"], nesterov=bool(hp["nesterov"]))
    elif kind == 1:
        kw.update(alpha=hp["alpha"], momentum=hp["momentum"], eps=hp["eps"])
    else:
        kw.update(betas=(0.9, 0.999), eps=hp["eps"], amsgrad=bool(hp["amsgrad"]))
    opt = cls([p], **kw)
    for step in range(6):
        if step == 3:
            opt.is_warmup = False
        p.grad = (T(O.synth((n,), 610 + step)) * 0.01).to(dev)
        opt.inject(T(g[f"{name}_noise"][step]).to(dev), T(g[f"{name}_coin"][step]).to(dev))
        opt.step()
        np.testing.assert_allclose(p.detach().cpu().numpy(), g[f"{name}_p"][step], rtol=1e-5, atol=1e-9, err_msg=f"step {step}")   # fp32 op-order (fma vs mul+add) noise
    np.testing.assert_allclose(p.grad.cpu().numpy(), g[f"{name}_gfinal"], rtol=1e-5, atol=1e-10)
    st = opt.state[p]
    for k in g.files:
        if k.startswith(name + "_state_"):
            key = k[len(name) + 7:]
            v = st[key]
            v = v.cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            np.testing.assert_allclose(v, g[k], rtol=1e-4, atol=1e-12, err_msg=key)


@pytest.mark.parametrize("npix,cin,cout,acc", [(25088, 288, 1728, 0), (1000, 104, 312, 1), (513, 56, 336, 0), (4096, 320, 1280, 1), (300, 24, 72, 0), (2048, 1440, 192, 0)])
def test_dgrad_wide_kernel_vs_torch(eng_mod, npix, cin, cout, acc):
    """frost_pw_dgrad_wide (k_dgrad_wide, frost_wgrad.hip) against a torch fp32 GEMM of the same bf16 operands: dx = s_w * dc . Wq (+ dx), rounded to
    bf16 once.  Shapes: the 7x7 expand layer at full size, ragged pixel counts (not multiples of 256 / 16), Cout not a multiple of 32, Cin not a
    multiple of 16, several input-channel chunks, accumulate on and off.  Tolerance: one bf16 ulp (the fp32 summation order differs)."""
    from frostnet_amd import _lib as L
    g = torch.Generator(device="cuda").manual_seed(npix + cin)
    dc = (torch.randn(npix, cout, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
    wq = torch.randint(-128, 128, (cout, cin), device="cuda", generator=g).float()          # integers: exact in bf16
    sw = 0.0123
    qw = torch.zeros(L.Q_STRIDE, device="cuda"); qw[L.Q_SCALE] = sw
    cit, kb = (cin + 15) // 16, (cout + 31) // 32
    wp = torch.zeros(cit * 16, kb * 32, device="cuda"); wp[:cin, :cout] = wq.t()
    # wt_pack[cit][kb][lane][8] = W[co = kb*32 + 8*(lane>>4) + e][ci = cit*16 + (lane&15)]
    pack = wp.view(cit, 16, kb, 4, 8).permute(0, 2, 3, 1, 4).contiguous().to(torch.bfloat16)       # [cit][kb][g][i][e] -> lane = g*16 + i
    prev = (torch.randn(npix, cin, device="cuda", generator=g)).to(torch.bfloat16)
    dx = prev.clone() if acc else torch.full((npix, cin), float("nan"), device="cuda").to(torch.bfloat16)
    assert L.load_library().frost_pw_dgrad_wide_ok(npix, cin, cout) == 1
    L.call("frost_pw_dgrad_wide", L.ptr(dc.view(torch.int16)), L.ptr(pack.view(torch.int16)), L.ptr(qw), npix, cin, cout, L.ptr(dx.view(torch.int16)), acc, L.stream())
    torch.cuda.synchronize()
    ref = (dc.float() @ wq) * sw + (prev.float() if acc else 0.0)
    got = dx.float()
    assert bool(torch.isfinite(got).all())
    err = (got - ref).abs() / (ref.abs() + 1e-3)
    assert float(err.max()) <= 2.0 ** -7, float(err.max())
    assert relerr(got, ref) <= 3e-3
