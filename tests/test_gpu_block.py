"""Block-level fused forward kernels (csrc/frost_block.hip) against the layer-by-layer launches they replace.

conv1 -> conv2 -> reduce_conv of a CascadePreExBottleneck (/root/reference/frostnet.py:134-138) at the 14x14 / 7x7 stages runs as
  conv1 statistics (k_pw) -> frost_block_expand_dw_stats -> frost_block_dw_reduce -> reduce emit (frost_pw_ew)
instead of six launches.  Same integer accumulators, same quantisation expressions, exact integer statistics: EVERY result must be bit-identical --
conv1 / conv2 / reduce outputs, the kept integer conv output, BN coefficient rows, running statistics and the observers' records.  The layer-by-layer
kernels themselves are held to the oracle and the reference goldens in test_gpu_ops / test_gpu_prod / test_gpu_model.  The block kernels engage only in
training mode with live observers on 14x14 / 7x7 maps, i.e. NOT in the 64-97 px training tests nor in the 224-px eval tests; their direct comparisons with
the reference / the oracle are tests/test_gpu_model.py::test_g4_block_true_shapes (reference goldens g4t_* at the true shapes, forward + backward, engagement
asserted) and tests/test_gpu_prod.py::test_large224_train_forward_block_by_block (Large @224 training forward against the oracle)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

# (cin, cexp, H, k, cout, batch): FrostNet-Large's 7x7 / 14x14 bottlenecks, a partial last 64-channel chunk (1440 = 22.5 x 64, 360 = 5.6 x 64),
# 8-mod-16 input rows (104, 120), batches that are not multiples of the images-per-workgroup split, and the widest reduce (1728 -> 320)
CASES = [(240, 1440, 7, 5, 192, 5), (192, 1152, 7, 3, 192, 64), (288, 1728, 7, 5, 320, 9), (104, 624, 14, 5, 96, 3), (120, 360, 14, 3, 96, 33),
         (160, 960, 14, 5, 96, 6), (104, 312, 14, 5, 80, 17)]


def _build(cin, cexp, k, cout, seed):
    from frostnet_amd import engine
    dev = "cuda"
    g = torch.Generator(device="cpu").manual_seed(seed)
    E, qa = engine.Engine(dev), engine.QArena(10, dev)

    def layer(name, kind, ci, co, kk, relu):
        fan = ci if kind == "pw" else kk * kk
        w = (torch.randn(co, 1 if kind == "dw" else ci, kk, kk, generator=g) * (2.0 / fan) ** 0.5).to(dev).requires_grad_(True)
        gamma = (torch.rand(co, generator=g) * 0.5 + 0.75).to(dev).requires_grad_(True)
        beta = (torch.rand(co, generator=g) * 0.2 - 0.05).to(dev).requires_grad_(True)
        return E.add_layer(engine.ConvLayer(name, kind, w, gamma, beta, torch.zeros(co, device=dev), torch.ones(co, device=dev),
                                            torch.zeros((), dtype=torch.int64, device=dev), None, kk, 1, relu, qa.alloc(), qa.alloc()))
    l1, l2, l3 = layer("conv1", "pw", cin, cexp, 1, True), layer("conv2", "dw", cexp, cexp, k, True), layer("reduce", "pw", cexp, cout, 1, False)
    qx = qa.alloc()
    qa.set_qparams(qx, 0.02, 3)
    return E, l1, l2, l3, qx


def _run(case, fused, steps=2):
    cin, cexp, H, k, cout, n = case
    E, l1, l2, l3, qx = _build(cin, cexp, k, cout, 11)
    g = torch.Generator(device="cpu").manual_seed(5)
    out = []
    for _ in range(steps):            # the second step starts from moved running statistics / observer records
        x = E.new_act(n, H, H, cin, qx)
        x.buf[: x.numel] = torch.randint(-128, 128, (x.numel,), dtype=torch.int16, generator=g).to(torch.int8).to("cuda")
        E.begin_step()
        if fused:
            assert E.pair_fusable(l1, l2, x, True, True)
            y2 = E.conv_pair(l1, l2, x, l3=l3)
            assert y2.kept_next is not None, "reduce_conv was expected on the fused path"
        else:
            y2 = E.conv(l2, E.conv(l1, x))
        y3 = E.conv(l3, y2)
        torch.cuda.synchronize()
        y1 = E.tape[0][3]
        out.append(dict(y1=y1.buf[: y1.numel].clone(), y2=y2.buf[: y2.numel].clone(), y3=y3.buf[: y3.numel].clone(), cint=y3.cint[: y3.numel].clone(),
                        **{f"{nm}{i}": getattr(l, nm).clone() for i, l in enumerate((l1, l2, l3), 1) for nm in ("qy", "coef", "rmean", "rvar", "nbt")}))
    return out


@pytest.mark.parametrize("case", CASES, ids=lambda c: "_".join(str(v) for v in c))
def test_block_kernels_bit_identical_to_layer_launches(case):
    ref, got = _run(case, False), _run(case, True)
    for step, (a, b) in enumerate(zip(ref, got)):
        for key in a:
            assert torch.equal(a[key].reshape(-1).view(torch.uint8), b[key].reshape(-1).view(torch.uint8)), (step, key)


def test_block_kernels_refuse_unsupported_shapes():
    from frostnet_amd import _lib as L
    lib = L.load_library()
    assert lib.frost_block_supported(14, 14, 5, 1, 104, 624) == 1 and lib.frost_block_supported(7, 7, 3, 1, 192, 1152) == 1
    assert lib.frost_block_supported(28, 28, 3, 1, 56, 168) == 0          # 28x28: layer-by-layer
    assert lib.frost_block_supported(14, 14, 5, 2, 104, 624) == 0          # stride 2: layer-by-layer
    assert lib.frost_block_supported(7, 7, 5, 1, 328, 1968) == 0           # input rows beyond the resident tile
    assert lib.frost_block_dw_reduce_supported(7, 7, 5, 1, 1728, 320) == 1 and lib.frost_block_dw_reduce_supported(14, 14, 5, 1, 624, 160) == 0
    with pytest.raises(RuntimeError, match="unsupported shape"):
        L.call("frost_block_expand_dw_stats", None, None, None, None, None, None, None, 1, 28, 28, 56, 168, None, None, 3, None, None, None)


# ---- the backward kernels of the same blocks (frost_block_dw_bwd_reduce[_dgrad], frost_block_dw_bwd) against the layer-by-layer backward -------------------------
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAYERWISE = dict(FROST_BLOCK_PAIR="0", FROST_BLOCK_DWRED="0", FROST_BLOCK_DWBWD="0", FROST_BLOCK_DWBRED="0")
BWD_CASES = [(240, 1440, 7, 5, 192, 21), (192, 1152, 7, 3, 192, 40), (288, 1728, 7, 5, 320, 9), (104, 624, 14, 5, 96, 10), (120, 360, 14, 3, 96, 7), (104, 312, 14, 5, 80, 17)]


def _digest(tmp, tag, case, env):
    out = os.path.join(tmp, f"{tag}.npz")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "block_digest.py"), out] + [str(v) for v in case], check=True, env=dict(os.environ, **env), cwd=ROOT,
                   stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    return np.load(out)


def _bf(a):
    return (a.astype(np.uint16).astype(np.uint32) << 16).view(np.float32).astype(np.float64)


def _rel(a, b):
    a, b = a.astype(np.float64).ravel(), b.astype(np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


@pytest.mark.parametrize("case", BWD_CASES, ids=lambda c: "_".join(str(v) for v in c))
def test_block_backward_matches_layer_backward(case, tmp_path):
    """All block kernels on (default) vs all off.  The forward is bit-identical; dc is rounded stochastically with a generator seeded from the workgroup / thread
    index, so two tilings of the same layer draw different noise and everything downstream of a dc agrees to the bf16 rounding level (the bound every tiling
    variant is held to in test_gpu_paths) -- through three layers here."""
    blk, lay = _digest(str(tmp_path), "blk", case, {}), _digest(str(tmp_path), "lay", case, LAYERWISE)
    assert blk["y3"].tobytes() == lay["y3"].tobytes()
    for i in (1, 2, 3):
        assert blk[f"qy{i}"].tobytes() == lay[f"qy{i}"].tobytes()
    assert _rel(_bf(blk["dx"]), _bf(lay["dx"])) <= 3e-2
    # the reduce layer sees identical inputs on both paths; conv2 / conv1 sit one / two stochastic roundings downstream, and at these small pixel counts
    # (a few thousand) their d-gamma is a cancelling sum dominated by that noise: the tight statement is the next test (no stochastic rounding)
    assert _rel(blk["dw3"], lay["dw3"]) <= 1e-4 and _rel(blk["dgamma3"], lay["dgamma3"]) <= 1e-4 and _rel(blk["dbeta3"], lay["dbeta3"]) <= 1e-5
    for i in (1, 2):
        assert _rel(blk[f"dw{i}"], lay[f"dw{i}"]) <= 5e-2, i
        assert _rel(blk[f"dgamma{i}"], lay[f"dgamma{i}"]) <= 0.3, i


@pytest.mark.parametrize("case", BWD_CASES[:4], ids=lambda c: "_".join(str(v) for v in c))
def test_block_backward_without_stochastic_rounding(case, tmp_path):
    """FROST_SR=0 (round-to-nearest dc: no random draws): what is left between the two paths is the order of float atomics in the reduce passes and weight
    gradients -- the data gradient may move on the few elements fed by a dc that sits on a bf16 rounding boundary, nothing else."""
    # (FROST_BLOCK_C1=0: with conv1's reduce pass folded into the depthwise backward -- round 6, the default -- conv1's S1 / S2 are sums of the fp32 dx values instead of the
    # bf16-rounded ones, which moves every dc of conv1 by a rounding-level amount: that form is held to the fp32-gradient mode in tests/test_gpu_round6.py)
    blk = _digest(str(tmp_path), "blk", case, {"FROST_SR": "0", "FROST_BLOCK_C1": "0"})
    lay = _digest(str(tmp_path), "lay", case, dict(LAYERWISE, FROST_SR="0"))
    a, b = _bf(blk["dx"]), _bf(lay["dx"])
    assert float((a != b).mean()) <= 4e-2 and _rel(a, b) <= 3e-3          # (run-to-run order of the float atomics in the reduce passes moves a few dc roundings)
    for i in (1, 2, 3):
        assert _rel(blk[f"dw{i}"], lay[f"dw{i}"]) <= 2e-3 and _rel(blk[f"dgamma{i}"], lay[f"dgamma{i}"]) <= 3e-2, i       # d-gamma: a cancelling sum (see above)


# ---- frost_block_fwd / frost_block_bwd: the chain as ONE C-ABI call each -----------------------------------------------------------------
def _block_desc(E, l1, l2, l3, x, y1, y2, y3, cint):
    import ctypes as C
    from frostnet_amd import _lib as L

    def fin(l):
        return L.FrostFinDesc(l.qw.data_ptr(), l.gamma.data_ptr(), l.beta.data_ptr(), l.rmean.data_ptr(), l.rvar.data_ptr(), l.nbt.data_ptr(), l.coef.data_ptr(),
                              l.qy.data_ptr(), l.fin_counter.data_ptr(), 1, int(l.relu), 1, 0, l.wscale.data_ptr(), None, None)

    def layer(l):
        return L.FrostBlockLayer(l.wq_pack.data_ptr(), l.wsum.data_ptr(), l.stats.data_ptr(), fin(l), l.wt_pack.data_ptr() if l.wt_pack is not None else None,
                                 l.dwq.data_ptr(), l.cout, l.k)
    return L.FrostBlockDesc(x.n, x.h, x.w, x.c, x.buf.data_ptr(), x.q.data_ptr(), layer(l1), layer(l2), layer(l3), y1.data_ptr(), y2.data_ptr(), cint.data_ptr(), y3.data_ptr())


@pytest.mark.parametrize("case", [(240, 1440, 7, 5, 192, 13), (104, 624, 14, 5, 96, 6), (192, 1152, 7, 3, 192, 32)], ids=lambda c: "_".join(str(v) for v in c))
def test_frost_block_fwd_bwd_entries_match_engine_sequence(case):
    """frost_block_fwd / frost_block_bwd called through ctypes on caller-owned buffers against Engine.conv_pair + Engine.conv + Engine.backward on an identical
    copy of the layers: the forward bit for bit; the backward runs the same kernels on the same grids (same stochastic-rounding draws), so the data gradient and
    the raw weight-gradient sums agree up to the order of float atomics."""
    import ctypes as C
    from frostnet_amd import _lib as L, engine as EN
    cin, cexp, H, k, cout, n = case
    res = {}
    for mode in ("engine", "entry"):
        E, l1, l2, l3, qx = _build(cin, cexp, k, cout, 23)
        g = torch.Generator(device="cpu").manual_seed(9)
        x = E.new_act(n, H, H, cin, qx)
        x.buf[: x.numel] = torch.randint(-128, 128, (x.numel,), dtype=torch.int16, generator=g).to(torch.int8).to("cuda")
        gy = (torch.randn(n * H * H * cout, generator=g) * 1e-3).to("cuda").to(torch.bfloat16).view(torch.int16)
        gy = torch.cat([gy, torch.zeros(64, dtype=torch.int16, device="cuda")])
        E.begin_step()
        if mode == "engine":
            x.needs_grad = True
            y3 = E.conv(l3, E.conv_pair(l1, l2, x, l3=l3))
            y1, y2 = E.tape[0][3], E.tape[1][3]
            fwd = dict(y1=y1.buf[: y1.numel].clone(), y2=y2.buf[: y2.numel].clone(), y3=y3.buf[: y3.numel].clone(), cint=y3.cint[: y3.numel].clone())
            y3.grad = gy
            E._prepare_dwq(); E._pending = []; E._side = None; E._keep = []       # Engine.backward's set-up without its weight-gradient finalize (the raw sums are compared)
            for op in reversed(E.tape):
                E._conv_backward(op[1], op[2], op[3])
            torch.cuda.synchronize()
            bwd = dict(dx=x.grad[: x.numel].clone(), **{f"dwq{i}": l.dwq.clone() for i, l in enumerate((l1, l2, l3), 1)})
        else:
            npix = n * H * H
            E._prepare_dwq()                                  # allocates and zeroes the raw weight-gradient sums
            y1 = torch.empty(npix * cexp + 64, dtype=torch.int8, device="cuda"); y2 = torch.empty_like(y1)
            y3 = torch.empty(npix * cout + 64, dtype=torch.int8, device="cuda"); cint = torch.empty(npix * cout + 64, dtype=torch.int32, device="cuda")
            d = _block_desc(E, l1, l2, l3, x, y1, y2, y3, cint)
            L.call("frost_block_fwd", C.byref(d), L.stream())
            fwd = dict(y1=y1[: npix * cexp].clone(), y2=y2[: npix * cexp].clone(), y3=y3[: npix * cout].clone(), cint=cint[: npix * cout].clone())
            bf = lambda m: torch.empty(m + 64, dtype=torch.int16, device="cuda")
            dc3, g2, g1, dc1, dx = bf(npix * cout), bf(npix * cexp), bf(npix * cexp), bf(npix * cexp), bf(npix * cin)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            b = L.FrostBlockBwd(gy.data_ptr(), dc3.data_ptr(), g2.data_ptr(), g1.data_ptr(), dc1.data_ptr(), dx.data_ptr(), C.c_void_p(side.cuda_stream))
            L.call("frost_block_bwd", C.byref(d), C.byref(b), L.stream())
            torch.cuda.synchronize()
            bwd = dict(dx=dx[: npix * cin].clone(), **{f"dwq{i}": l.dwq.clone() for i, l in enumerate((l1, l2, l3), 1)})
        res[mode] = (fwd, bwd)
    (fa, ba), (fb, bb) = res["engine"], res["entry"]
    for key in fa:
        assert torch.equal(fa[key].reshape(-1).view(torch.uint8), fb[key].reshape(-1).view(torch.uint8)), key
    dxa, dxb = ba["dx"].view(torch.bfloat16).double(), bb["dx"].view(torch.bfloat16).double()
    mism, rel = float((dxa != dxb).double().mean()), float((dxa - dxb).norm() / dxa.norm())
    rels = [float((ba[f"dwq{i}"].double() - bb[f"dwq{i}"].double()).norm() / (ba[f"dwq{i}"].double().norm() + 1e-30)) for i in (1, 2, 3)]
    print(f"[entries vs engine {case}] dx differs on {mism:.2e} of the elements, rel {rel:.2e}; dWq rel {rels[0]:.2e} {rels[1]:.2e} {rels[2]:.2e}")
    assert mism <= 4e-2 and rel <= 3e-3, (mism, rel)
    for i in (1, 2, 3):
        assert rels[i - 1] <= 2e-3, (i, rels)
