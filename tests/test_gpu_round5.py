"""Round-5 additions on the GPU: the FAST forms of the fp32-gradient mode (csrc/frost_g32.hip, section "fast forms": int8-MFMA conv output, fp32-MFMA data / weight
gradients, two-stage deterministic sums) against the plain one-thread-per-output kernels of round 4, entry by entry through the C ABI, on production layer shapes
(reference: loss.backward() is fp32 autograd, Classification/utils/helper_functions.py:139-143; the mode-level parity against the oracle / the reference golden is
tests/test_gpu_round4.py::test_fp32_gradient_mode_*)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    import __graft_entry__ as ge
    ge.build()
    from frostnet_amd import _lib
    assert torch.cuda.is_available()
    yield _lib
    _lib.load_library().frost_g32_set_plain(0)


def relerr(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


# (name, kind, cin, cout, k, stride, H, N): kind 0 pointwise, 1 depthwise, 2 stem on the im2col'd input.  Widths / maps of FrostNet-Large (frostnet.py:176-198) at small
# batches, plus ragged cases: a K that is no multiple of 64 / 16, a pixel count that is no multiple of the 64-pixel wave tile, cout > 1024 (two quad blocks).
CASES = [("pw16_96_28", 0, 16, 96, 1, 1, 28, 3), ("pw104_624_14", 0, 104, 624, 1, 1, 14, 3), ("pw624_104_14", 0, 624, 104, 1, 1, 14, 2), ("pw288_1728_7", 0, 288, 1728, 1, 1, 7, 5),
         ("pw1728_320_7", 0, 1728, 320, 1, 1, 7, 3), ("pw40_24_9", 0, 40, 24, 1, 1, 9, 1), ("pw24_8_5", 0, 24, 8, 1, 1, 5, 1),
         ("dw96_k3s2_28", 1, 96, 96, 3, 2, 28, 2), ("dw144_k5s2_14", 1, 144, 144, 5, 2, 14, 3), ("dw624_k3s1_14", 1, 624, 624, 3, 1, 14, 2), ("dw1440_k5s1_7", 1, 1440, 1440, 5, 1, 7, 3),
         ("dw32_k3s1_9", 1, 32, 32, 3, 1, 9, 1), ("stem_3_32_20", 2, 3, 32, 3, 1, 20, 2)]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_fp32_gradient_fast_forms_equal_the_plain_kernels(L, case):
    """Every frost_g32_* entry, fast form vs plain form on the same inputs: the integer conv output and dc bit-equal, S1 / S2 to fp32 rounding of the same fp64 sums,
    the data gradients (fp32 MFMA accumulation vs fp64 sums) norm-wise <= 2e-6, the weight gradients <= 1e-5 -- two to three orders inside the mode's 1e-3 bound."""
    from frostnet_amd.engine import ptr, stream
    name, kind, cin, cout, k, stride, H, N = case
    lib = L.load_library()
    dev = "cuda"
    g = torch.Generator(device="cpu").manual_seed(500 + CASES.index(case))
    pad = (k - 1) // 2
    if kind == 1:
        h = w = H
        ho = (H + 2 * pad - k) // stride + 1
        xc, cin_g, per = cin, 1, k * k
        nin, npo = N * H * H, N * ho * ho
        geo = (kind, N, H, H, xc, cin_g, cout, k, stride)
    else:
        ho = H
        xc = 40 if kind == 2 else cin
        cin_g = cin
        per = cin * 9 if kind == 2 else cin
        nin = npo = N * H * H
        geo = (kind, N, H, H, xc, cin_g, cout, 1 if kind == 0 else 3, 1)
    x = torch.randint(-128, 128, (nin * xc + 64,), generator=g, dtype=torch.int8).to(dev)
    qw = torch.randint(-128, 128, (cout * per + 64,), generator=g, dtype=torch.int8).to(dev)
    zp = 117
    qx = torch.zeros(16, dtype=torch.float32)
    qx[2] = 0.0231
    qx[3] = torch.tensor([zp], dtype=torch.int32).view(torch.float32)[0]
    qx[6] = 1.0 / 0.0231
    qy = torch.zeros(16, dtype=torch.float32)
    qy[2] = 0.05
    qy[3] = torch.tensor([3], dtype=torch.int32).view(torch.float32)[0]
    qy[6] = 20.0
    qwr = torch.zeros(16, dtype=torch.float32)
    qwr[2] = 0.0123
    qx, qy, qwr = qx.to(dev), qy.to(dev), qwr.to(dev)
    cpad = (cout + 15) // 16 * 16
    coef0 = torch.zeros(8 * cpad, dtype=torch.float32)           # rows A, B, M, R, K1, ... (include/frost_hip.h FROST_COEF_*)
    kk = cin_g * (k * k if kind == 1 else (9 if kind == 2 else 1))
    coef0.view(8, cpad)[0, :cout] = (torch.rand(cout, generator=g) + 0.5) * 0.05 / (40.0 * kk ** 0.5)      # A: acc -> a few output levels
    coef0.view(8, cpad)[1, :cout] = torch.randn(cout, generator=g) * 0.05
    coef0.view(8, cpad)[2, :cout] = torch.randn(cout, generator=g) * 10.0
    coef0.view(8, cpad)[3, :cout] = (torch.rand(cout, generator=g) + 0.5) / (40.0 * kk ** 0.5)
    coef0.view(8, cpad)[4, :cout] = torch.rand(cout, generator=g) + 0.5
    gout = (torch.randn(npo * cout + 64, generator=g) * 1e-3).to(dev)
    scr = torch.empty(int(lib.frost_g32_scratch_bytes()), dtype=torch.uint8, device=dev)
    s = stream()
    res = {}
    for plain in (1, 0):
        lib.frost_g32_set_plain(plain)
        coef = coef0.clone().to(dev)
        acc = torch.zeros(npo * cout + 64, dtype=torch.int32, device=dev)
        dc = torch.zeros(npo * cout + 64, dtype=torch.float32, device=dev)
        L.call("frost_g32_conv_acc", ptr(x), ptr(qx), ptr(qw), *geo, ptr(acc), s)
        L.call("frost_g32_reduce", ptr(acc), npo, cout, ptr(coef), ptr(qy), 1, ptr(gout), ptr(scr), s)
        s12 = coef.view(8, cpad).clone()
        L.call("frost_g32_dc", ptr(acc), npo, cout, ptr(coef), ptr(qy), 1, ptr(gout), ptr(dc), s)
        out = {"acc": acc[: npo * cout].clone(), "coef": s12, "dc": dc[: npo * cout].clone()}
        if not plain and kind != 1:          # the reduce / dc passes that recompute the integer conv output instead of reading `acc` (frost_g32_reduce_x / _dc_x)
            assert lib.frost_g32_x_ok(kind, npo, xc, cin_g, cout)
            coefx = coef0.clone().to(dev)
            dcx = torch.zeros(npo * cout + 64, dtype=torch.float32, device=dev)
            gx_ = (kind, N, H, H, xc, cin_g, cout)
            L.call("frost_g32_reduce_x", ptr(x), ptr(qx), ptr(qw), *gx_, ptr(coefx), ptr(qy), 1, ptr(gout), ptr(scr), s)
            out["coefx"] = coefx.view(8, cpad).clone()
            L.call("frost_g32_dc_x", ptr(x), ptr(qx), ptr(qw), *gx_, ptr(coefx), ptr(qy), 1, ptr(gout), ptr(dcx), s)
            out["dcx"] = dcx[: npo * cout].clone()
        if kind != 2:
            for accum in (0, 1):
                gx = torch.full((nin * xc + 64,), 0.25, dtype=torch.float32, device=dev)
                L.call("frost_g32_dgrad", ptr(dc), ptr(qw), ptr(qwr), None, *geo, ptr(gx), accum, s)
                out[f"gx{accum}"] = gx[: nin * xc].clone()
        dwq = torch.zeros(cout * per, dtype=torch.float32, device=dev)
        L.call("frost_g32_wgrad", ptr(dc), ptr(x), ptr(qx), *geo, ptr(dwq), ptr(scr), s)
        out["dwq"] = dwq
        torch.cuda.synchronize()
        res[plain] = out
    lib.frost_g32_set_plain(0)
    a, b = res[0], res[1]
    assert torch.equal(a["acc"], b["acc"]), (name, "integer conv output")
    assert float(b["acc"].float().abs().max()) > 0
    srow = [r for r in range(8) if not torch.equal(b["coef"][r], coef0.view(8, cpad).to(dev)[r])]          # the rows the reduce pass wrote: S1, S2
    assert len(srow) == 2, srow
    for r in srow:
        assert relerr(a["coef"][r], b["coef"][r]) <= 1e-6, (name, "S row", r, relerr(a["coef"][r], b["coef"][r]))
    for r in range(8):
        if r not in srow:
            assert torch.equal(a["coef"][r], b["coef"][r])
    assert relerr(a["dc"], b["dc"]) <= 1e-6, (name, "dc", relerr(a["dc"], b["dc"]))          # (the S rows differ in their last bit)
    if "coefx" in a:          # recomputing forms: S rows from fp32 lane sums of ~200 values (fp64 across lanes / waves), dc from them
        for r in srow:
            assert relerr(a["coefx"][r], b["coef"][r]) <= 5e-6, (name, "S row (recomputing form)", r, relerr(a["coefx"][r], b["coef"][r]))
        assert relerr(a["dcx"], b["dc"]) <= 5e-6, (name, "dc (recomputing form)", relerr(a["dcx"], b["dc"]))
    errs = {key: relerr(a[key], b[key]) for key in a if (key.startswith("gx") or key == "dwq") and key in b}
    print(f"[g32 fast vs plain {name}] " + " ".join(f"{k_} {v:.1e}" for k_, v in errs.items()))
    assert all(v <= (1e-5 if k_ == "dwq" else 2e-6) for k_, v in errs.items()), (name, errs)          # (the weight gradient: fp32 sums over runs of <= 1024 pixels, fp64 across runs)
    assert float(b["dwq"].abs().max()) > 0 and float(b["dc"].abs().max()) > 0


def test_fp32_gradient_mode_trains_a_step_at_the_bf16_modes_loss(L):
    """The mode end to end through the module surface: one QAT step of FrostNet-Small with `grad_precision = "fp32"` next to the production bf16 step from the same state --
    same forward (bit-equal logits), parameter gradients of the classifier / last_layer (the tail of the backward, before chaos amplifies) within the bf16 mode's own bound."""
    from frostnet_amd import frostnet as F
    grads = {}
    for mode in ("bf16", "fp32"):
        torch.manual_seed(11)
        model = F.MODEL_REGISTRY["frostnet_quant_small_1_0"](drop_rate=0.0)
        F.qat_prepare(model, version=0)
        model.cuda().train()
        model.grad_precision = mode
        x = torch.randn(8, 3, 96, 96, generator=torch.Generator().manual_seed(5)).cuda()
        y = torch.arange(8, device="cuda") % 1000
        out = model(x)
        loss = torch.nn.functional.cross_entropy(out, y)
        loss.backward()
        torch.cuda.synchronize()
        grads[mode] = ({n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}, out.detach().clone())
    assert torch.equal(grads["bf16"][1], grads["fp32"][1])
    ga, gb = grads["bf16"][0], grads["fp32"][0]
    assert set(ga) == set(gb) and len(ga) > 100
    for n in ga:
        if n.startswith("classifier.") or n.startswith("last_layer."):
            assert relerr(ga[n], gb[n]) <= 3e-2, (n, relerr(ga[n], gb[n]))
    assert all(torch.isfinite(v).all() for v in gb.values())


# ------------------------------------------------------------------------------------------ squeeze_conv forward as one persistent launch
@pytest.mark.parametrize("cin,r,H,n", [(80, 24, 14, 7), (96, 24, 14, 33), (96, 24, 14, 512), (192, 48, 7, 9), (192, 48, 7, 512), (192, 96, 7, 130), (120, 32, 9, 2), (40, 16, 28, 5)])
def test_squeeze_forward_persistent_launch_equals_the_layer_launches(L, cin, r, H, n):
    """frost_sq_fwd (statistics -> device-wide barrier with the finalize inside -> emit + cat from the kept accumulators; frostnet.py:127-129) against the launches it
    replaces (frost_pw_conv_fwd_fin + frost_sq_emit_cat): squeezed activation, cat output, both FakeQuantize records, coefficient rows, running statistics and
    num_batches_tracked over three steps (moving observers), ragged last tiles, the production grids of the 14 x 14 / 7 x 7 stages at B = 512 (784 / 196 workgroups)
    included; the barrier's give-up flag must stay clear and its generation word must advance once per launch."""
    from frostnet_amd import engine
    dev = "cuda"

    def run(persist):
        old = engine._SQ_PERSIST
        engine._SQ_PERSIST = persist
        try:
            g = torch.Generator(device="cpu").manual_seed(77)
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
            for step in range(3):
                x = E.new_act(n, H, H, cin, qx)
                x.buf[: x.numel] = torch.randint(-128, 128, (x.numel,), dtype=torch.int16, generator=g).to(torch.int8).to(dev)
                E.begin_step()
                L.CALL_LOG = []
                sq = E.conv(l, x, True, True, cat=(x.q, qcat))
                y = E.cat(sq, x, qcat, True)
                torch.cuda.synchronize()
                log, L.CALL_LOG = list(L.CALL_LOG), None
                assert ("frost_sq_fwd" in log) == persist and ("frost_sq_emit_cat" in log) == (not persist), log
                outs.append((sq.buf[: sq.numel].clone(), y.buf[: y.numel].clone(), qcat.clone(), l.qy.clone(), l.coef.clone(), l.rmean.clone(), l.rvar.clone(), l.nbt.clone()))
            ctl = l.fin_counter.cpu()
            if persist:
                assert int(ctl[37]) == 0, "a workgroup gave up waiting at the device-wide barrier"
                assert int(ctl[36]) == 3 and int(ctl[:36].abs().sum()) == 0, ctl          # one generation per launch; every ticket counter re-armed
            return outs
        finally:
            engine._SQ_PERSIST = old
            L.CALL_LOG = None
    if not L.load_library().frost_sq_fwd_ok(n * H * H, cin, r):
        pytest.skip("grid does not fit the device at once: the layer launches are the path")
    a, b = run(True), run(False)
    for step, (sa, sb) in enumerate(zip(a, b)):
        # the statistics are exact integers in the persistent kernel (k_pw forms the sum of squares in fp32 within a tile): coefficient rows / running variance agree to
        # fp32 rounding of values ~1e-7 apart, activations and records wherever that moves no rounding tie (a flip budget of 1e-5 of the elements, one level)
        for i in (0, 1):
            d = (sa[i].to(torch.int16) - sb[i].to(torch.int16)).abs()
            assert int(d.max()) <= 1 and float((d != 0).float().mean()) <= 1e-5, (step, i, int(d.max()), float((d != 0).float().mean()))
        for i in (2, 3, 4, 5, 6):
            assert torch.allclose(sa[i], sb[i], rtol=2e-6, atol=1e-9), (step, i, float((sa[i] - sb[i]).abs().max()))
        assert torch.equal(sa[7], sb[7])


def test_squeeze_forward_persistent_refuses_a_grid_that_cannot_be_resident(L):
    """frost_sq_fwd_ok is the co-residency guard of the device-wide barrier: a 28 x 28 map at B = 512 (3136 tiles) is beyond any occupancy of 256 CUs and must be refused
    (the engine then takes frost_pw_conv_fwd_fin + frost_sq_emit_cat), the 14 x 14 / 7 x 7 grids of FrostNet-Large at B = 512 accepted."""
    lib = L.load_library()
    assert lib.frost_sq_fwd_ok(512 * 28 * 28, 40, 16) == 0
    assert lib.frost_sq_fwd_ok(512 * 14 * 14, 96, 24) == 1 and lib.frost_sq_fwd_ok(512 * 14 * 14, 80, 24) == 1
    assert lib.frost_sq_fwd_ok(512 * 7 * 7, 192, 48) == 1 and lib.frost_sq_fwd_ok(512 * 7 * 7, 192, 96) == 1
    assert lib.frost_sq_fwd_ok(1000, 196, 48) == 0          # cin beyond the staged tile
