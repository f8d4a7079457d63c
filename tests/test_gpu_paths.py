"""Full-size path equivalence on the GPU: the kernels choose resident weights, DMA / LDS-staged tile I/O, channel-group
splitting, depthwise tile geometries, XCD-aware block maps, shape-specialised instances and fused passes (incl. the fused pointwise backward) by layer size, so the small teacher-forced golden layers do not reach
them.  Here one seeded layer runs at a size where the fast paths engage, once with the defaults and once with every such
path switched off (environment switches read by libfrost_hip.so at load, hence one subprocess per configuration), and the
two results must agree: quantised outputs identical up to rare rounding-boundary flips (the fp32 part of the variance sum
depends on the tile-to-lane assignment), statistics to 1e-6, gradients to the bf16 rounding noise of the dc tensor."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLAIN = dict(FROST_PW_RESMASK="0", FROST_PW_IO="0", FROST_PW_GL="0", FROST_PW_CSPLIT="0", FROST_DW_GEO="0", FROST_DW_FUSE="0", FROST_PW_FUSE="0", FROST_DW_XCD="0", FROST_WG_XCD="0",
             FROST_PW_SPEC="0", FROST_DW_SPEC="0", FROST_DGRAD_WIDE="0", FROST_INFER_WIDE="0", FROST_PW_KEEP="0", FROST_BLOCK_DWBWD="0", FROST_BLOCK_DWBRED="0", FROST_PWC="0", FROST_PWC_EMIT="0", FROST_DW_BWD_ONE="0", FROST_DW_STREAM="0")
FAST = dict()
CASES = [("pw", 16, 96, 1, 1, 112, 64), ("pw", 32, 16, 1, 1, 112, 64), ("pw", 72, 24, 1, 1, 56, 128), ("pw", 144, 40, 1, 1, 28, 512), ("dw", 72, 72, 3, 1, 56, 64), ("pw", 56, 168, 1, 1, 28, 128), ("pw", 40, 16, 1, 1, 28, 64), ("pw", 56, 336, 1, 1, 28, 512), ("pw", 96, 24, 1, 1, 56, 128), ("pw", 24, 144, 1, 1, 56, 128), ("pw", 240, 1440, 1, 1, 7, 512),
         ("pw", 1728, 320, 1, 1, 7, 256), ("dw", 96, 96, 3, 2, 112, 32), ("dw", 32, 32, 3, 1, 112, 32), ("dw", 1440, 1440, 5, 1, 7, 256),
         ("dw", 144, 144, 5, 2, 56, 64),
         # 14x14 / 7x7 depthwise layers: the fused backward (dc + weight gradient + data gradient from an LDS plane, csrc/frost_block.hip) vs the three separate kernels;
         # partial last 64-channel chunks (360, 1440), batches that do not divide by the images-per-workgroup split
         ("dw", 624, 624, 5, 1, 14, 19), ("dw", 360, 360, 3, 1, 14, 33), ("dw", 1152, 1152, 3, 1, 7, 45), ("dw", 312, 312, 5, 1, 14, 64),
         # wide-K reduce layers on the kept-conv-output path (rows of 8 mod 16 bytes, a ragged pixel count, three channel chunks) vs all four passes through k_pw
         # wide expand layers on the chunked reduce / dc kernel (csrc/frost_pwc.hip) vs k_pw: a partial last chunk (1440, 360), 8-mod-16 input rows (104), a ragged last 64-pixel tile
         ("pw", 104, 624, 1, 1, 14, 19), ("pw", 120, 360, 1, 1, 14, 33), ("pw", 320, 1280, 1, 1, 7, 45), ("pw", 288, 1728, 1, 1, 7, 64),
         ("pw", 312, 80, 1, 1, 14, 67), ("pw", 360, 96, 1, 1, 14, 33), ("pw", 1440, 192, 1, 1, 7, 70), ("pw", 624, 96, 1, 1, 14, 128)]


def run(tmp, tag, case, env_extra):
    out = os.path.join(tmp, f"{tag}.npz")
    env = dict(os.environ, **env_extra)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "layer_digest.py"), out] + [str(v) for v in case], check=True, env=env,
                   cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    return np.load(out)


def bf16_to_f32(a):
    return (a.astype(np.uint16).astype(np.uint32) << 16).view(np.float32)


def relerr(a, b):
    a, b = a.astype(np.float64).ravel(), b.astype(np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


@pytest.mark.parametrize("case", CASES, ids=lambda c: "_".join(str(v) for v in c))
def test_fast_paths_match_plain_paths(case, tmp_path):
    fast = run(str(tmp_path), "fast", case, FAST)
    plain = run(str(tmp_path), "plain", case, PLAIN)
    d = np.abs(fast["y"].astype(np.int16) - plain["y"].astype(np.int16))
    assert d.max() <= 1 and float((d > 0).mean()) <= 1e-4, ("y", int(d.max()), float((d > 0).mean()))
    np.testing.assert_allclose(fast["qy"][:3], plain["qy"][:3], rtol=1e-6)          # observer min, max, scale
    assert fast["qy"][3].tobytes() == plain["qy"][3].tobytes()                         # zero point (integer bits)
    np.testing.assert_allclose(fast["rm"], plain["rm"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(fast["rv"], plain["rv"], rtol=1e-6, atol=1e-7)
    # dc is rounded to bf16 STOCHASTICALLY (unbiased, frost_common.h) with a generator seeded from the workgroup / thread index: two tilings of
    # the same layer draw different rounding noise, so everything downstream of dc agrees to the bf16 rounding level (2^-9 / sqrt(3) per
    # element), not to atomic-order noise
    assert relerr(bf16_to_f32(fast["dx"]), bf16_to_f32(plain["dx"])) <= 8e-3
    for k in ("dw", "dgamma"):
        assert relerr(fast[k], plain[k]) <= 1e-2, k
    assert relerr(fast["dbeta"], plain["dbeta"]) <= 1e-3


# The matrix-core depthwise formulation (csrc/frost_dwm.hip: Toeplitz bands on v_mfma_i32_16x16x64_i8) against the lane-=-channel stencil kernels:
# 64- and 32-channel blocks, partial channel blocks, 3x3 and 5x5, maps that are / are not multiples of the 16 x 16 tile.  Same layer, same inputs;
# the backward kernels are shared, so everything but the forward statistics' summation order is identical.
DWM_CASES = [("dw", 72, 72, 3, 1, 56, 8), ("dw", 624, 624, 5, 1, 14, 16), ("dw", 32, 32, 3, 1, 112, 4), ("dw", 168, 168, 3, 1, 28, 8), ("dw", 360, 360, 5, 1, 14, 6),
             ("dw", 1440, 1440, 5, 1, 7, 16)]


@pytest.mark.parametrize("cb", ["32", "64"])
@pytest.mark.parametrize("case", DWM_CASES, ids=lambda c: "_".join(str(v) for v in c))
def test_dw_matrix_core_path_matches_stencil_path(case, cb, tmp_path):
    mfma = run(str(tmp_path), "mfma", case, {"FROST_DW_MFMA": "1", "FROST_DWM_CB": cb})
    sten = run(str(tmp_path), "sten", case, {"FROST_DW_MFMA": "0"})
    d = np.abs(mfma["y"].astype(np.int16) - sten["y"].astype(np.int16))
    assert d.max() <= 1 and float((d > 0).mean()) <= 1e-4, ("y", int(d.max()), float((d > 0).mean()))
    np.testing.assert_allclose(mfma["qy"][:3], sten["qy"][:3], rtol=1e-6)
    assert mfma["qy"][3].tobytes() == sten["qy"][3].tobytes()
    np.testing.assert_allclose(mfma["rm"], sten["rm"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(mfma["rv"], sten["rv"], rtol=1e-6, atol=1e-7)


# Fused depthwise backward with round-to-nearest dc (FROST_SR=0: no random draws): dc is then a deterministic function of the inputs, the data gradient is
# summed in k_dw3_dgrad's order, so dx must be BIT-IDENTICAL to the three-kernel path; the weight gradient differs by the order of its fp32 atomics only.
@pytest.mark.parametrize("case", [("dw", 1440, 1440, 5, 1, 7, 21), ("dw", 624, 624, 5, 1, 14, 10), ("dw", 360, 360, 3, 1, 14, 7), ("dw", 1152, 1152, 3, 1, 7, 16)],
                         ids=lambda c: "_".join(str(v) for v in c))
def test_fused_depthwise_backward_exact_without_stochastic_rounding(case, tmp_path):
    fused = run(str(tmp_path), "fused", case, {"FROST_SR": "0", "FROST_BLOCK_DWBWD": "2"})
    sep = run(str(tmp_path), "sep", case, {"FROST_SR": "0", "FROST_BLOCK_DWBWD": "0", "FROST_BLOCK_DWBRED": "0"})
    assert fused["y"].tobytes() == sep["y"].tobytes()
    # S1 / S2 of the reduce pass are float atomics (order differs from run to run at the 1e-7 level), so a dc exactly on a bf16 rounding boundary may fall
    # either way and moves the up to k*k data-gradient elements it feeds: everything else is identical
    a, b = bf16_to_f32(fused["dx"]).astype(np.float64), bf16_to_f32(sep["dx"]).astype(np.float64)
    differ = a != b
    assert float(differ.mean()) <= 2e-3, float(differ.mean())
    assert relerr(a, b) <= 2e-4
    assert relerr(fused["dw"], sep["dw"]) <= 2e-5 and relerr(fused["dgamma"], sep["dgamma"]) <= 2e-5 and relerr(fused["dbeta"], sep["dbeta"]) <= 2e-5


# The one-sweep depthwise backward of the tiled (high-resolution) k = 3 stride-1 layers (csrc/frost_dwb.hip: dc stays in registers, weight gradient and data gradient in the
# same strip-streaming pass) against frost_dw_conv_bwd_dc_wgrad + frost_dw_dgrad, with round-to-nearest dc (FROST_SR=0): same dc, the data gradient summed in k_dw3_dgrad's
# order -> dx BIT-IDENTICAL up to dc elements on a bf16 rounding boundary (S1 / S2 are float atomics); the weight gradient differs by the order of its fp32 sums only.
# 32- and 64-channel blocks, partial channel blocks (72, 40, 168), maps that are not a multiple of the strip width (30, 28), several row chunks, fewer images than XCDs.
# Stride 2 (k = 3, 5): 112 -> 56, 56 -> 28, 28 -> 14, 14 -> 7 and ragged relatives (a last strip that is half outside the map, channel blocks of 8 .. 64 live lanes).
@pytest.mark.parametrize("case", [("dw", 32, 32, 3, 1, 112, 16), ("dw", 72, 72, 3, 1, 56, 12), ("dw", 240, 240, 3, 1, 28, 9), ("dw", 40, 40, 3, 1, 30, 5), ("dw", 168, 168, 3, 1, 28, 3),
                                  ("dw", 96, 96, 3, 2, 112, 10), ("dw", 144, 144, 5, 2, 56, 12), ("dw", 336, 336, 5, 2, 28, 9), ("dw", 672, 672, 5, 2, 14, 11),
                                  ("dw", 40, 40, 3, 2, 30, 5), ("dw", 72, 72, 5, 2, 20, 3), ("dw", 168, 168, 3, 2, 28, 8)],
                         ids=lambda c: "_".join(str(v) for v in c))
@pytest.mark.parametrize("chunks", ["0", "3"])
def test_one_sweep_depthwise_backward_exact_without_stochastic_rounding(case, chunks, tmp_path):
    one = run(str(tmp_path), "one", case, {"FROST_SR": "0", "FROST_DW_BWD_ONE": "1", "FROST_DWB_CHUNKS": chunks, "FROST_DWB_MINW": "8", "DIGEST_CALLS": os.path.join(str(tmp_path), "calls.txt")})
    sep = run(str(tmp_path), "sep", case, {"FROST_SR": "0", "FROST_DW_BWD_ONE": "0"})
    assert "frost_dw_bwd_fused" in open(os.path.join(str(tmp_path), "calls.txt")).read()
    assert one["y"].tobytes() == sep["y"].tobytes()
    a, b = bf16_to_f32(one["dx"]).astype(np.float64), bf16_to_f32(sep["dx"]).astype(np.float64)
    differ = a != b
    assert float(differ.mean()) <= 2e-3, float(differ.mean())
    assert relerr(a, b) <= 2e-4
    # (the weight-gradient sums are fp32 in both paths but grouped differently -- per lane across a whole strip here, per tile there: 4e-5 on the smallest case)
    assert relerr(one["dw"], sep["dw"]) <= 1e-4 and relerr(one["dgamma"], sep["dgamma"]) <= 1e-4 and relerr(one["dbeta"], sep["dbeta"]) <= 2e-5


def test_one_sweep_depthwise_backward_with_stochastic_rounding_matches_at_the_bf16_level(tmp_path):
    case = ("dw", 72, 72, 3, 1, 56, 16)
    one = run(str(tmp_path), "one", case, {"FROST_DW_BWD_ONE": "1"})
    sep = run(str(tmp_path), "sep", case, {"FROST_DW_BWD_ONE": "0"})
    assert relerr(bf16_to_f32(one["dx"]), bf16_to_f32(sep["dx"])) <= 8e-3
    assert relerr(one["dw"], sep["dw"]) <= 1e-2 and relerr(one["dgamma"], sep["dgamma"]) <= 1e-2 and relerr(one["dbeta"], sep["dbeta"]) <= 1e-3


# The strip-streaming statistics / emit / reduce passes (k_dws, csrc/frost_dwb.hip) against k_dw3's tile kernels: the statistics are exact integers, the finalize and the
# emit pass pure functions of them -> outputs, observer / BatchNorm state BIT-IDENTICAL; S1 / S2 of the reduce pass are fp32 sums in another grouping (1e-6 on dbeta / dgamma).
@pytest.mark.parametrize("case", [("dw", 32, 32, 3, 1, 112, 16), ("dw", 72, 72, 3, 1, 56, 12), ("dw", 240, 240, 3, 1, 28, 9), ("dw", 40, 40, 3, 1, 30, 5),
                                  ("dw", 96, 96, 3, 2, 112, 10), ("dw", 144, 144, 5, 2, 56, 12), ("dw", 336, 336, 5, 2, 28, 9), ("dw", 40, 40, 3, 2, 30, 5), ("dw", 72, 72, 5, 2, 36, 3)],
                         ids=lambda c: "_".join(str(v) for v in c))
def test_streaming_depthwise_passes_are_exact(case, tmp_path):
    new = run(str(tmp_path), "new", case, {"FROST_SR": "0", "FROST_DW_STREAM": "7", "FROST_DWS_MINW": "8", "DIGEST_CALLS": os.path.join(str(tmp_path), "calls.txt")})
    old = run(str(tmp_path), "old", case, {"FROST_SR": "0", "FROST_DW_STREAM": "0"})
    for k in ("y", "qy", "rm", "rv"):
        assert new[k].tobytes() == old[k].tobytes(), k
    assert relerr(new["dbeta"], old["dbeta"]) <= 1e-5 and relerr(new["dgamma"], old["dgamma"]) <= 1e-4
    a, b = bf16_to_f32(new["dx"]).astype(np.float64), bf16_to_f32(old["dx"]).astype(np.float64)
    assert float((a != b).mean()) <= 2e-3 and relerr(a, b) <= 2e-4


# The conv1 fold of the stride-2 one-sweep kernel (frost_dw_bwd_fused_c1): conv1's reduce pass (S1 / S2) rides on the dx rows of the depthwise backward instead of a pass
# of its own over the stored gradient.  The sums now see the dx values BEFORE their bf16 rounding, so conv1's dc (and what follows from it) moves at the bf16-noise level and
# nothing else moves at all: conv2 / reduce_conv gradients identical, conv1's launch list loses its reduce pass.  16 -> 96 @112 (k3), 24 -> 144 @56 (k5), ragged relatives.
def _block(tmp, tag, case, env):
    out = os.path.join(tmp, f"{tag}.npz")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "block_digest.py"), out] + [str(v) for v in case], check=True, env=dict(os.environ, **env), cwd=ROOT,
                   stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    return np.load(out)


@pytest.mark.parametrize("case", [(16, 96, 112, 3, 24, 6, 2), (24, 144, 56, 5, 40, 5, 2), (24, 72, 30, 3, 24, 9, 2), (16, 72, 44, 5, 16, 3, 2)], ids=lambda c: "_".join(str(v) for v in c))
def test_conv1_reduce_pass_folded_into_the_depthwise_backward(case, tmp_path):
    calls = os.path.join(str(tmp_path), "calls.txt")
    on = _block(str(tmp_path), "on", case, {"FROST_SR": "0", "FROST_DWB_C1": "3", "FROST_DWB_MINW": "8", "DIGEST_CALLS": calls})
    log = open(calls).read().split("\n")
    assert "frost_dw_bwd_fused_c1" in log and log.count("frost_pw_conv_bwd") + log.count("frost_pw_conv_bwd_fused") <= 3, log      # conv1: no reduce pass left (reduce_conv keeps its own)
    off = _block(str(tmp_path), "off", case, {"FROST_SR": "0", "FROST_DWB_C1": "0", "FROST_DWB_MINW": "8"})
    ref = _block(str(tmp_path), "ref", case, {"FROST_GRAD": "fp32"})          # fp32 gradient storage, fp64 sums (csrc/frost_g32.hip): the yardstick
    assert on["y3"].tobytes() == off["y3"].tobytes() == ref["y3"].tobytes()
    for k in ("dw2", "dgamma2", "dbeta2", "dw3", "dgamma3", "dbeta3"):
        assert relerr(on[k], off[k]) <= 1e-4, k
    assert relerr(bf16_to_f32(on["dx"]), bf16_to_f32(off["dx"])) <= 5e-3
    # conv1's gradients against the fp32-gradient mode: the fold must not be further from it than the separate pass is (its sums skip one bf16 rounding), within a 1.25 x
    # allowance for noise on the components that do not depend on S1 / S2 alone
    e_on = {k: relerr(on[k], ref[k]) for k in ("dw1", "dgamma1", "dbeta1")}
    e_off = {k: relerr(off[k], ref[k]) for k in ("dw1", "dgamma1", "dbeta1")}
    print("conv1 vs fp32-gradient mode: fold", e_on, "separate", e_off)
    assert e_on["dbeta1"] <= 1.25 * e_off["dbeta1"] + 1e-4 and e_on["dgamma1"] <= 1.25 * e_off["dgamma1"] + 1e-4 and e_on["dw1"] <= 1.25 * e_off["dw1"] + 1e-4, (e_on, e_off)
    assert relerr(bf16_to_f32(on["dx"]), ref["dx"]) <= 1.25 * relerr(bf16_to_f32(off["dx"]), ref["dx"]) + 1e-4


# The 7 x 7 conv2-emit + reduce_conv kernel with two chunks per iteration on eight waves (k_blk_dw_reduce2, FROST_BLK_B2=1; measured slower, off by default, kept as an A/B
# form): integer GEMM + exact statistics -> every output, record and gradient must be BIT-IDENTICAL to the four-wave kernel -- odd and even chunk counts, partial last chunks,
# k = 3 / 5, 12 and 20 output-channel tiles, several images per launch (the backward then runs on identical saved tensors).
@pytest.mark.parametrize("case", [(192, 1008, 7, 5, 192, 5, 1), (192, 1152, 7, 3, 192, 3, 1), (288, 1728, 7, 5, 320, 2, 1), (96, 328, 7, 5, 56, 9, 1)],
                         ids=lambda c: "_".join(str(v) for v in c))
def test_two_chunk_eight_wave_block_kernel_is_bit_identical(case, tmp_path):
    calls = os.path.join(str(tmp_path), "calls.txt")
    two = _block(str(tmp_path), "two", case, {"FROST_BLK_B2": "1", "DIGEST_CALLS": calls})
    assert "frost_block_dw_reduce" in open(calls).read().split("\n")
    one = _block(str(tmp_path), "one", case, {"FROST_BLK_B2": "0"})
    for k in ("y3", "qy1", "qy2", "qy3"):          # the forward: integer GEMM + exact statistics
        assert two[k].tobytes() == one[k].tobytes(), k
    # the backward does not touch this kernel: it runs on identical saved tensors in both processes and differs only by its own run-to-run noise (float atomics group
    # differently, which moves stochastic-rounding decisions of dc: <= 1.2e-2 of a layer's gradient norm over 36 runs, tests/test_gpu_model.py) -- a sanity bound only
    for k in one.files:
        if k.startswith("d") and k != "dx":
            assert relerr(two[k], one[k]) <= 2e-2, k
    assert relerr(bf16_to_f32(two["dx"]), bf16_to_f32(one["dx"])) <= 2e-2
