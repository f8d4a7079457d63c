"""Config c2: bf16 inference of the float model on the HIP kernels vs the fp32 definition of the same module (the stock-module
CPU path, which tests/test_oracle_golden.py pins to the reference on G5).  Tolerance: bf16 storage of weights and of every layer
output (8 mantissa bits, ~70 roundings deep) -> norm-wise relative error of the logits <= 3e-2, arg-max agreement on >= 7/8 images."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _randomize_bn(model, seed):
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data = torch.rand(m.num_features, generator=g) * 0.8 + 0.6
            m.bias.data = torch.rand(m.num_features, generator=g) * 0.2 - 0.1
            m.running_mean.data = torch.randn(m.num_features, generator=g) * 0.1
            m.running_var.data = torch.rand(m.num_features, generator=g) * 0.5 + 0.5


@pytest.mark.parametrize("name,res,batch", [("frostnet_small_1_0", 64, 4), ("frostnet_large_1_0", 224, 8), ("frostnet_base_0_75", 96, 3)])
def test_bf16_inference_vs_fp32_definition(name, res, batch):
    import __graft_entry__ as ge
    ge.build()
    from frostnet_amd import frostnet as F
    torch.manual_seed(7)
    model = F.MODEL_REGISTRY[name]()
    _randomize_bn(model, 11)
    model.eval()
    x = torch.randn(batch, 3, res, res)
    with torch.no_grad():
        ref = model(x)                                  # CPU: stock torch modules = the reference's definition
    model.cuda()
    out = model.hip_infer_bf16(x.cuda()).cpu()
    out_cl = model.hip_infer_bf16(x.cuda().contiguous(memory_format=torch.channels_last)).cpu()
    assert torch.equal(out, out_cl)                     # layout of the input must not matter
    rel = float((out - ref).norm() / ref.norm())
    agree = int((out.argmax(1) == ref.argmax(1)).sum())
    assert rel <= 3e-2, rel
    assert agree >= batch - max(1, batch // 8), (agree, batch)
    with pytest.raises(RuntimeError):
        model.train(); model.hip_infer_bf16(x.cuda())


@pytest.mark.parametrize("name,res,batch", [("frostnet_large_1_0", 224, 6), ("frostnet_large_1_0", 160, 3), ("frostnet_small_1_0", 97, 5), ("frostnet_base_1_25", 128, 2)])
def test_fused_block_inference_equals_layer_by_layer(name, res, batch):
    """frost_infer_block (csrc/frost_iblock.hip: squeeze -> cat -> conv1 -> depthwise -> reduce_conv -> + x in ONE launch per bottleneck, expanded tensors in LDS)
    against the layer-by-layer kernels it replaces: same rounding points (every layer output rounded to bf16 once), so the logits agree to the summation order of
    the GEMMs -- held to 1e-2 norm-wise (one bf16 step is 4e-3) -- and both meet the fp32 definition's tolerance.  224 px exercises whole-image, half-image and
    8 x 16 / 8 x 8 / 7 x 7 tiles with stride 1 and 2; 160 / 97 / 128 px ragged edge tiles and odd maps; base_1_25 other channel counts (partial 64-channel chunks)."""
    import __graft_entry__ as ge
    ge.build()
    from frostnet_amd import frostnet as F, infer as I, _lib as L
    torch.manual_seed(17)
    model = F.MODEL_REGISTRY[name]()
    _randomize_bn(model, 23)
    model.eval()
    x = torch.randn(batch, 3, res, res)
    with torch.no_grad():
        ref = model(x)
    model.cuda()
    xg = x.cuda()
    old = I._FUSED
    try:
        I._FUSED = True                                   # every bottleneck the kernel takes (the default, "auto", fuses only where it measured faster)
        L.CALL_LOG = []
        fused = model.hip_infer_bf16(xg).cpu()
        log = list(L.CALL_LOG)
        L.CALL_LOG = None
        I._FUSED = False
        model.__dict__.pop("_bf16_infer", None)
        plain = model.hip_infer_bf16(xg).cpu()
        I._FUSED = "auto"                                 # the default: per bottleneck whichever of {layer-by-layer, fused with one of the candidate tiles} measured fastest
        model.__dict__.pop("_bf16_infer", None)
        auto = model.hip_infer_bf16(xg).cpu()
        auto2 = model.hip_infer_bf16(xg).cpu()            # second call: the cached choices
        picks = [v for ent in model.__dict__["_bf16_infer"].blocks for k_, v in ent.items() if isinstance(k_, tuple) and k_[0] == "choice"]
    finally:
        I._FUSED = old
        L.CALL_LOG = None
    nblocks = sum(len(getattr(model, f"layer{i}")) for i in range(1, 6))
    nfused = log.count("frost_infer_block")
    assert nfused == nblocks if name.endswith("1_0") else nfused >= nblocks // 2, (nfused, nblocks)      # (other widths: the widest blocks exceed the kernel's K budget)
    r_fp, r_pl, r_ref = float((fused - plain).norm() / plain.norm()), float((plain - ref).norm() / ref.norm()), float((fused - ref).norm() / ref.norm())
    print(f"[{name}@{res}] fused vs layer-by-layer {r_fp:.2e}; vs the fp32 definition: fused {r_ref:.2e}, layer-by-layer {r_pl:.2e}")
    assert r_fp <= 1e-2 and r_ref <= 3e-2, (r_fp, r_ref, r_pl)
    assert torch.equal(auto, plain) and torch.equal(auto2, plain) and len(picks) == nblocks, picks       # whatever was picked, the result is the same
    print(f"    measured choices: {picks}")


@pytest.mark.parametrize("name,res,batch", [("frostnet_large_1_0", 224, 3), ("frostnet_large_1_0", 97, 2), ("frostnet_small_1_0", 131, 2), ("frostnet_base_1_25", 64, 5)])
def test_wave_block_kernel_is_bit_identical_to_layer_by_layer(name, res, batch):
    """frost_infer_block_w (csrc/frost_iblockw.hip: one WAVE per 4 x 4 / 4 x 8 output tile, no workgroup barriers, conv1's operand straight from HBM) on every
    bottleneck without a squeeze conv it takes, block by block on the layer-by-layer path's own inputs: the outputs must be the same bits (same rounding points,
    same summation order: one K step per 32-channel chunk, taps in (ky, kx) order) -- maps of 112 ... 8 pixels, stride 1 / 2, k 3 / 5, ragged edge tiles, residual."""
    import __graft_entry__ as ge
    ge.build()
    from frostnet_amd import frostnet as F, infer as I, _lib as L
    torch.manual_seed(29)
    model = F.MODEL_REGISTRY[name]()
    _randomize_bn(model, 31)
    model.eval().cuda()
    x = torch.randn(batch, 3, res, res, device="cuda")
    model.hip_infer_bf16(x)                               # builds the runner and its weight packs
    run = model.__dict__["_bf16_infer"]
    lib = L.load_library()
    n, _, h, w = x.shape
    ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    a = torch.empty(n * ho * wo * run.stem.cout, dtype=torch.int16, device="cuda")
    L.call("frost_infer_stem", L.ptr(x), n, h, w, x.stride(0), x.stride(1), x.stride(2), x.stride(3), L.ptr(run.stem.pack), L.ptr(run.stem.biasf), run.stem.cout, 1, L.ptr(a), L.stream())
    c, h, w = run.stem.cout, ho, wo
    taken = 0
    for ent in run.blocks:
        ref, c2, h2, w2 = run._block_plain(ent, a, c, n, h, w)
        l2, l3 = ent["conv2"], ent["reduce"]
        if ent["squeeze"] is None:
            for tw in (4, 8):
                if lib.frost_infer_block_w_ok(c, 0, l2.cout, l3.cout, l2.k, l2.stride, 1 if ent["conv1"] is not None else 0, 4, tw):
                    out, c3, h3, w3 = run._block_wave(ent, a, c, n, h, w, ("w", 4, tw))
                    torch.cuda.synchronize()
                    m = n * h2 * w2 * c2
                    assert (c3, h3, w3) == (c2, h2, w2) and torch.equal(out[:m], ref[:m]), (name, res, c, l2.cout, l3.cout, l2.k, l2.stride, tw, int((out[:m] != ref[:m]).sum()))
                    taken += 1
        a, c, h, w = ref, c2, h2, w2
    assert taken >= (6 if name.endswith("large_1_0") else 2), taken


def test_fused_block_kernel_with_16_waves_equals_8_waves_and_layer_by_layer():
    """The 16-wave instances of frost_infer_block (whole 7 x 7 / 8 x 8 maps: one workgroup per image, so only more waves shorten its serial phases) against the
    8-wave instances and the layer launches, block by block on FrostNet-Large @224: the wave count only deals the same tiles to more waves -- same bits."""
    import __graft_entry__ as ge
    ge.build()
    from frostnet_amd import frostnet as F, _lib as L
    torch.manual_seed(41)
    model = F.MODEL_REGISTRY["frostnet_large_1_0"]()
    _randomize_bn(model, 43)
    model.eval().cuda()
    x = torch.randn(3, 3, 224, 224, device="cuda")
    model.hip_infer_bf16(x)
    run = model.__dict__["_bf16_infer"]
    n, _, h, w = x.shape
    ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    a = torch.empty(n * ho * wo * run.stem.cout, dtype=torch.int16, device="cuda")
    L.call("frost_infer_stem", L.ptr(x), n, h, w, x.stride(0), x.stride(1), x.stride(2), x.stride(3), L.ptr(run.stem.pack), L.ptr(run.stem.biasf), run.stem.cout, 1, L.ptr(a), L.stream())
    c, h, w = run.stem.cout, ho, wo
    taken = 0
    for ent in run.blocks:
        ref, c2, h2, w2 = run._block_plain(ent, a, c, n, h, w)
        if h2 * w2 <= 64:
            m = n * h2 * w2 * c2
            try:
                o16 = run._block_fused(ent, a, c, n, h, w, (h2, w2, 16, 64))[0]
            except RuntimeError:          # a geometry the kernel (or this wave count) does not take
                o16 = None
            if o16 is not None:
                o8 = run._block_fused(ent, a, c, n, h, w, (h2, w2, 8, 64))[0]
                torch.cuda.synchronize()
                assert torch.equal(o16[:m], o8[:m]) and torch.equal(o16[:m], ref[:m]), (c, c2, h2, w2)
                taken += 1
        a, c, h, w = ref, c2, h2, w2
    assert taken >= 4, taken


@pytest.mark.parametrize("name,n,h,w", [("frostnet_large_1_0", 3, 224, 224), ("frostnet_small_1_0", 2, 97, 131), ("frostnet_base_0_75", 1, 600, 520), ("frostnet_large_1_0", 2, 31, 17)])
def test_direct_stem_is_bit_identical_to_im2col_gemm(name, n, h, w):
    """frost_infer_stem (conv1 straight from the fp32 image, tile staged in LDS) against frost_infer_stem_im2col + frost_infer_pw on the same packs: the same
    bf16 operands enter the same two MFMAs per tile in the same order, so every output element must match bit for bit -- NCHW and channels_last inputs, odd sizes,
    maps wider than one 128-column tile, maps smaller than one tile."""
    import __graft_entry__ as ge
    ge.build()
    from frostnet_amd import frostnet as F, infer as I, _lib as L
    from frostnet_amd._lib import call, ptr, stream
    torch.manual_seed(3)
    model = F.MODEL_REGISTRY[name]()
    _randomize_bn(model, 5)
    model.eval().cuda()
    inf = I.Bf16Inference(model)
    st = inf.stem
    assert L.load_library().frost_infer_stem_ok(st.cout)
    inf._prepare_weights()
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    npix = n * ho * wo
    for x in (torch.randn(n, 3, h, w, device="cuda"), torch.randn(n, 3, h, w, device="cuda").contiguous(memory_format=torch.channels_last) * 3.0):
        col = torch.empty(npix * 64 + 64, dtype=torch.int16, device="cuda")
        call("frost_infer_stem_im2col", ptr(x), n, h, w, *x.stride(), ptr(col), stream())
        want = inf._pw(st, col, npix, 64)[: npix * st.cout]
        got = torch.full((npix * st.cout,), 0x7fc0, dtype=torch.int16, device="cuda")
        call("frost_infer_stem", ptr(x), n, h, w, *x.stride(), ptr(st.pack), ptr(st.biasf), st.cout, 1, ptr(got), stream())
        torch.cuda.synchronize()
        assert torch.equal(got, want), int((got != want).sum())
        assert float(got.view(torch.bfloat16).float().abs().max()) > 0


def test_weight_prep_is_cached_and_follows_updates():
    """The folded packs are rebuilt when a parameter changes in place (version counter), when a tensor is re-assigned (data pointer), on refresh() -- and not otherwise."""
    import __graft_entry__ as ge
    ge.build()
    from frostnet_amd import frostnet as F, _lib as L
    torch.manual_seed(5)
    model = F.MODEL_REGISTRY["frostnet_small_1_0"]()
    _randomize_bn(model, 9)
    model.eval().cuda()
    x = torch.randn(2, 3, 64, 64, device="cuda")
    y0 = model.hip_infer_bf16(x).clone()
    L.CALL_LOG = []
    try:
        y1 = model.hip_infer_bf16(x)
        assert "frost_infer_weight_prep" not in L.CALL_LOG and torch.equal(y0, y1)
        with torch.no_grad():
            model.classifier[2].bias.add_(1.0)                         # not a folded tensor: no re-prep, but the head reads it live
            model.conv1.conv[1].weight.mul_(1.5)                            # in place
        del L.CALL_LOG[:]
        y2 = model.hip_infer_bf16(x)
        assert "frost_infer_weight_prep" in L.CALL_LOG and not torch.equal(y2, y1 + 1.0)
        model.last_layer.conv[0].weight.data = model.last_layer.conv[0].weight.data * 0.5      # re-assigned
        del L.CALL_LOG[:]
        y3 = model.hip_infer_bf16(x)
        assert "frost_infer_weight_prep" in L.CALL_LOG and not torch.equal(y3, y2)
        ref = F.MODEL_REGISTRY["frostnet_small_1_0"]()
        ref.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
        ref.eval()
        with torch.no_grad():
            want = ref(x.cpu())
        assert float((y3.cpu() - want).norm() / want.norm()) <= 3e-2
        del L.CALL_LOG[:]
        model.__dict__["_bf16_infer"].refresh()
        assert torch.equal(model.hip_infer_bf16(x), y3) and "frost_infer_weight_prep" in L.CALL_LOG
    finally:
        L.CALL_LOG = None
