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
             FROST_PW_SPEC="0", FROST_DW_SPEC="0", FROST_DGRAD_WIDE="0", FROST_INFER_WIDE="0", FROST_PW_KEEP="0")
CASES = [("pw", 16, 96, 1, 1, 112, 64), ("pw", 32, 16, 1, 1, 112, 64), ("pw", 72, 24, 1, 1, 56, 128), ("pw", 144, 40, 1, 1, 28, 512), ("dw", 72, 72, 3, 1, 56, 64), ("pw", 56, 168, 1, 1, 28, 128), ("pw", 40, 16, 1, 1, 28, 64), ("pw", 56, 336, 1, 1, 28, 512), ("pw", 96, 24, 1, 1, 56, 128), ("pw", 24, 144, 1, 1, 56, 128), ("pw", 240, 1440, 1, 1, 7, 512),
         ("pw", 1728, 320, 1, 1, 7, 256), ("dw", 96, 96, 3, 2, 112, 32), ("dw", 32, 32, 3, 1, 112, 32), ("dw", 1440, 1440, 5, 1, 7, 256),
         ("dw", 144, 144, 5, 2, 56, 64),
         # wide-K reduce layers on the kept-conv-output path (rows of 8 mod 16 bytes, a ragged pixel count, three channel chunks) vs all four passes through k_pw
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
    fast = run(str(tmp_path), "fast", case, {})
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
