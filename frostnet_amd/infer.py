"""bf16 inference of the FLOAT (not fake-quantised) FrostNet on the HIP kernels -- BASELINE.json config c2.

Reference semantics: `model.eval(); model(x)` of a float `frostnet_*` model (`frostnet.py:14-60, 108-121, 271-300`): Conv ->
BatchNorm(running stats) -> ReLU per layer, block wiring squeeze -> cat -> conv1 -> conv2 -> reduce_conv -> (+x), global average
pool -> (dropout is identity in eval) -> 1x1 classifier.  Here BatchNorm is folded into the convolution once per call
(`frost_infer_weight_prep`), activations are NHWC bf16, accumulation is fp32.  Deviation from the reference: bf16 storage of
weights and activations (the c2 configuration asks for bf16); the parity test holds the logits to the fp32 oracle within bf16
tolerance.  No CPU / torch-eager fallback: a missing library raises.
"""
import ctypes as C
import os

import torch

from . import _lib as L
from ._lib import call, ptr, stream


def round_up(a, b):
    return (a + b - 1) // b * b


# Whole-bottleneck fused kernel (csrc/frost_iblock.hip): "1" = every bottleneck the kernel takes (tile from pick_tile), "0" = layer-by-layer launches, "auto"
# (default) = MEASURED per bottleneck: the first forward outside a hipGraph capture times the layer-by-layer launches and the fused launch with each candidate
# tile (3 runs each) and keeps the fastest for that (map size, batch) -- results are bit-identical whichever is chosen (tests/test_gpu_infer.py).
_FUSED = {"0": False, "1": True}.get(os.environ.get("FROST_INFER_FUSED", "auto"), "auto")
# the stem straight from the fp32 image (frost_infer_stem) instead of im2col + GEMM: bit-identical, no 128-byte-per-pixel patch buffer
_STEM_DIRECT = os.environ.get("FROST_INFER_STEM", "direct") != "im2col"
# the wave-per-tile block kernel (csrc/frost_iblockw.hip) as a further candidate of the measured choice for the bottlenecks without a squeeze conv ("0": not offered)
_WAVE = os.environ.get("FROST_INFER_WAVE", "1") != "0"
TILE_CANDIDATES = ((0, 0), (-1, 0), (7, 14), (8, 16), (16, 16), (7, 7), (8, 8), (4, 16), (4, 8))          # (0, 0) = the whole map, (-1, 0) = half of it


def candidate_tiles(lib, h, w, cin, r, cexp, cout, k, stride):
    pad = (k - 1) // 2
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    out = []
    for th, tw in TILE_CANDIDATES:
        th, tw = (ho, wo) if th == 0 else (((ho + 1) // 2, wo) if th < 0 else (min(th, ho), min(tw, wo)))
        if (th, tw) not in out and th * tw <= 256 and lib.frost_infer_block_ok(h, w, cin, r, cexp, cout, k, stride, th, tw) > 0:
            out.append((th, tw))
    return out


def pick_tile(lib, h, w, cin, r, cexp, cout, k, stride):
    """Spatial output tile (th, tw) of one workgroup of frost_infer_block by a cost model (FROST_INFER_FUSED=1; the default mode "auto" measures the candidates
    instead, Bf16Inference._block): region pixels conv1 computes per output pixel plus a fixed per-tile cost, mildly weighted by the residency the tile's LDS
    footprint allows.  None: the fused kernel does not take this geometry."""
    pad = (k - 1) // 2
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    cands = [(ho, wo), ((ho + 1) // 2, wo), (7, 14), (8, 16), (7, 7), (8, 8), (4, 8), (4, 4)]
    if os.environ.get("FROST_IB_TILE"):
        cands = [tuple(int(v) for v in os.environ["FROST_IB_TILE"].split(","))] + cands[-1:]
    best = None
    for th, tw in cands:
        th, tw = min(th, ho), min(tw, wo)
        lds = lib.frost_infer_block_ok(h, w, cin, r, cexp, cout, k, stride, th, tw) if th * tw <= 256 else 0
        if lds <= 0:
            continue
        waste = (-(-ho // th) * th) * (-(-wo // tw) * tw) / float(ho * wo)            # ragged edge tiles
        rp = ((th - 1) * stride + k) * ((tw - 1) * stride + k)                            # region pixels conv1 computes for th * tw outputs (halo included)
        wgs = min(3, (160 * 1024) // lds)                                                 # resident workgroups per CU by LDS
        # cost per output pixel: region pixels plus a fixed per-tile cost (prologue, barriers: ~150 pixel-equivalents), mildly weighted by residency --
        # fitted to tools/prof_iblock.py (104 -> 624 -> 96 @14x14, k5: 7x14 157 us, 7x7 198, 4x8 235, 4x4 475; 16 -> 96 -> 24 @112, k3 s2: 8x8 best)
        score = waste * (rp + 150.0) / (th * tw) * (1.0 + 0.1 * (3 - wgs))
        if best is None or score < best[0] - 1e-9:
            best = (score, th, tw)
    return None if best is None else (best[1], best[2])


class _ILayer:
    __slots__ = ("kind", "k", "stride", "relu", "cout", "cin_g", "cpad", "kpad", "pack", "biasf", "conv", "bn")

    def __init__(self, seq, relu, dev, stem=False):
        conv, bn = seq[0], seq[1]
        if not isinstance(conv, torch.nn.Conv2d) or not isinstance(bn, torch.nn.BatchNorm2d):
            raise RuntimeError("bf16 inference expects the un-fused float model (Conv2d + BatchNorm2d per layer)")
        self.conv, self.bn, self.relu = conv, bn, relu
        self.cout, self.cin_g, self.k = conv.out_channels, conv.in_channels // conv.groups, conv.kernel_size[0]
        self.stride = conv.stride[0]
        self.kind = 2 if stem else (1 if conv.groups > 1 else 0)
        self.cpad = round_up(self.cout, 16)
        if self.kind == 1:
            self.kpad = 0
            self.pack = torch.zeros(self.k * self.k * self.cpad, dtype=torch.float32, device=dev)
        else:
            self.kpad = 64 if stem else round_up(self.cin_g, 32)
            self.pack = torch.zeros((self.cpad // 16) * (self.kpad // 32) * 64 * 8, dtype=torch.int16, device=dev)
        self.biasf = torch.zeros(self.cpad, dtype=torch.float32, device=dev)

    def desc(self):
        bn, w = self.bn, self.conv.weight
        return L.FrostIDesc(w.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                            bn.running_var.data_ptr(), self.pack.data_ptr(), self.biasf.data_ptr(), self.cout, self.cin_g,
                            self.k * self.k, self.kind, self.cpad, self.kpad, 0, 0)


class Bf16Inference:
    """Binds a float FrostNet (eval mode) to the bf16 HIP inference kernels.  `__call__(x)` -> fp32 logits [N, nclass]."""

    def __init__(self, model):
        L.load_library()
        # act='hswish' networks (frostnet.py ConvBNHswish): the layer kernels take the activation as a code (0 none, 1 ReLU, 2 hard-swish); the fused
        # bottleneck kernels (frost_infer_block / _block_w) are written for ReLU, so such a network runs layer by layer
        self.hswish = getattr(model, "act", "relu") == "hswish"
        A = 2 if self.hswish else 1
        p = next(model.parameters())
        if not p.is_cuda:
            raise RuntimeError("Bf16Inference needs the model on the GPU (no CPU fallback on the product path)")
        self.model, self.device = model, p.device
        self.layers = []
        self.stem = self._add(model.conv1.conv, A, stem=True)
        self.blocks = []
        for stage in (model.layer1, model.layer2, model.layer3, model.layer4, model.layer5):
            for blk in stage:
                ent = dict(blk=blk, squeeze=None, conv1=None)
                if blk.expand_ratio != 1:
                    if blk.block_type == "CAS":
                        ent["squeeze"] = self._add(blk.squeeze_conv.conv, A)
                    ent["conv1"] = self._add(blk.conv1.conv, A)
                ent["conv2"] = self._add(blk.conv2.conv, A)
                ent["reduce"] = self._add(blk.reduce_conv.conv, 0)
                self.blocks.append(ent)
        self.last = self._add(model.last_layer.conv, A)
        self.fc = model.classifier[2]
        self._table, self._ptrs, self._versions = None, None, None

    def _prepare_weights(self):
        """BatchNorm folding + fragment packing (frost_infer_weight_prep) once per state of the parameters, as the reference folds once before it evaluates
        (Classification/evaluate.py:131-143): the descriptor table follows re-assigned tensors (data pointers), the packs follow in-place updates (tensor
        version counters).  `refresh()` forces it (for writes the counters do not see, e.g. through a raw pointer)."""
        ts = [t for l in self.layers for t in (l.conv.weight, l.bn.weight, l.bn.bias, l.bn.running_mean, l.bn.running_var)]
        # + the generation of raw-pointer writers (the HIP optimizer step and the training forwards update parameters / running statistics behind torch's back);
        # once such a writer sits in a captured hipGraph its replays are invisible here: prepare on every call then
        ptrs, versions = [t.data_ptr() for t in ts], [t._version for t in ts] + [L.RAW_WRITE_GEN]
        if L.RAW_WRITES_CAPTURED:
            versions = None
        if ptrs != self._ptrs:
            arr = (L.FrostIDesc * len(self.layers))()
            for i, l in enumerate(self.layers):
                arr[i] = l.desc()
            self._table, self._ptrs, self._versions = L.struct_to_tensor(arr, self.device), ptrs, None
        if versions is None or versions != self._versions:
            call("frost_infer_weight_prep", ptr(self._table), len(self.layers), stream())
            self._versions = versions

    def refresh(self):
        self._versions = None

    def _add(self, seq, relu, stem=False):
        l = _ILayer(seq, relu, self.device, stem)
        self.layers.append(l)
        return l

    # ---------------------------------------------------------------------------------------------- kernels
    def _pw(self, l, x, npix, cin):
        y = torch.empty(npix * l.cout + 64, dtype=torch.int16, device=self.device)
        call("frost_infer_pw", ptr(x), ptr(l.pack), ptr(l.biasf), npix, cin, l.cout, int(l.relu), ptr(y), stream())
        return y

    def _dw(self, l, x, n, h, w):
        pad = (l.k - 1) // 2
        ho, wo = (h + 2 * pad - l.k) // l.stride + 1, (w + 2 * pad - l.k) // l.stride + 1
        y = torch.empty(n * ho * wo * l.cout + 64, dtype=torch.int16, device=self.device)
        call("frost_infer_dw", ptr(x), ptr(l.pack), ptr(l.biasf), n, h, w, l.cout, l.k, l.stride, int(l.relu), ptr(y), stream())
        return y, ho, wo

    def _block_plain(self, ent, a, c, n, h, w):
        """One bottleneck as layer-by-layer launches: [squeeze -> cat] -> conv1 -> conv2 -> reduce_conv [-> + x]."""
        x_in, npix = a, n * h * w
        if ent["conv1"] is not None:
            if ent["squeeze"] is not None:
                s = self._pw(ent["squeeze"], a, npix, c)
                cs = ent["squeeze"].cout
                cat = torch.empty(npix * (cs + c) + 64, dtype=torch.int16, device=self.device)
                call("frost_infer_cat", ptr(s), cs, ptr(a), c, npix, ptr(cat), stream())     # cat([squeezed, x], 1)
                a, c = cat, cs + c
            a, c = self._pw(ent["conv1"], a, npix, c), ent["conv1"].cout
        a, h2, w2 = self._dw(ent["conv2"], a, n, h, w)
        npix2 = n * h2 * w2
        a, c = self._pw(ent["reduce"], a, npix2, c), ent["reduce"].cout
        if not ent["blk"].reduction:
            out = torch.empty_like(a)
            call("frost_infer_add", ptr(x_in), ptr(a), npix2 * c, ptr(out), stream())
            a = out
        return a, c, h2, w2

    def _block_fused(self, ent, a, c, n, h, w, tile):
        """The same bottleneck as ONE launch of frost_infer_block with the spatial output tile `tile`."""
        l2, l3, sq, l1 = ent["conv2"], ent["reduce"], ent["squeeze"], ent["conv1"]
        r = sq.cout if sq is not None else 0
        pad = (l2.k - 1) // 2
        h2, w2 = (h + 2 * pad - l2.k) // l2.stride + 1, (w + 2 * pad - l2.k) // l2.stride + 1
        out = torch.empty(n * h2 * w2 * l3.cout + 64, dtype=torch.int16, device=self.device)
        call("frost_infer_block", ptr(a), ptr(sq.pack) if sq is not None else None, ptr(sq.biasf) if sq is not None else None,
             ptr(l1.pack) if l1 is not None else None, ptr(l1.biasf) if l1 is not None else None, ptr(l2.pack), ptr(l2.biasf),
             ptr(l3.pack), ptr(l3.biasf), n, h, w, c, r, l2.cout, l3.cout, l2.k, l2.stride, 0 if ent["blk"].reduction else 1, tile[0], tile[1],
             tile[2] if len(tile) > 2 else 0, tile[3] if len(tile) > 3 else 0, ptr(out), stream())
        return out, l3.cout, h2, w2

    def _block_wave(self, ent, a, c, n, h, w, tile):
        """A bottleneck without a squeeze conv as ONE launch of frost_infer_block_w: one wave per (th, tw) output tile, no workgroup barriers."""
        l2, l3, l1 = ent["conv2"], ent["reduce"], ent["conv1"]
        pad = (l2.k - 1) // 2
        h2, w2 = (h + 2 * pad - l2.k) // l2.stride + 1, (w + 2 * pad - l2.k) // l2.stride + 1
        out = torch.empty(n * h2 * w2 * l3.cout + 64, dtype=torch.int16, device=self.device)
        call("frost_infer_block_w", ptr(a), ptr(l1.pack) if l1 is not None else None, ptr(l1.biasf) if l1 is not None else None, ptr(l2.pack), ptr(l2.biasf),
             ptr(l3.pack), ptr(l3.biasf), n, h, w, c, l2.cout, l3.cout, l2.k, l2.stride, 0 if ent["blk"].reduction else 1, tile[1], tile[2], ptr(out), stream())
        return out, l3.cout, h2, w2

    def _run_choice(self, ent, a, c, n, h, w, choice):
        if choice == "plain":
            return self._block_plain(ent, a, c, n, h, w)
        if choice[0] == "w":
            return self._block_wave(ent, a, c, n, h, w, choice)
        return self._block_fused(ent, a, c, n, h, w, choice)

    def _block(self, ent, a, c, n, h, w):
        if _FUSED is False or self.hswish:
            return self._block_plain(ent, a, c, n, h, w)
        # the measured choice is kept per (resolution, batch BUCKET): a last partial batch or a slightly different batch size reuses its bucket's choice instead of
        # re-running dozens of timed launches inside the caller's forward, and the table stays bounded
        bucket = 1 << max(0, int(n) - 1).bit_length()
        key = ("choice", bucket, h, w, _FUSED)
        choice = ent.get(key)
        if choice is not None:
            ent[key] = ent.pop(key)                   # re-inserted on a hit: the eviction below drops the LEAST RECENTLY USED entry, not the oldest insertion
        if choice is None:
            l2, l3, sq = ent["conv2"], ent["reduce"], ent["squeeze"]
            r = sq.cout if sq is not None else 0
            lib = L.load_library()
            if _FUSED is True:
                choice = pick_tile(lib, h, w, c, r, l2.cout, l3.cout, l2.k, l2.stride) or "plain"
            elif torch.cuda.is_current_stream_capturing():
                return self._block_plain(ent, a, c, n, h, w)          # nothing can be timed inside a capture: run one eager forward first (bench.py does)
            else:
                timed = []
                tiles = candidate_tiles(lib, h, w, c, r, l2.cout, l3.cout, l2.k, l2.stride)
                narrow = -(-(r + c) // 32) <= 2                       # conv1 K <= 64 (the high-resolution blocks): 32-channel chunks are an option
                # every tile with 4 and with 8 waves per workgroup (and, for the narrow blocks, with 64- and 32-channel chunks of the expanded width)
                # bottlenecks without a squeeze conv on small Cin (the high-resolution blocks): the wave-per-tile kernel with its two tiles
                wave = [("w", 4, tw) for tw in (4, 8) if _WAVE and sq is None and
                        lib.frost_infer_block_w_ok(c, 0, l2.cout, l3.cout, l2.k, l2.stride, 1 if ent["conv1"] is not None else 0, 4, tw)]
                # (16 waves: only for whole small maps -- one workgroup per image, so with few images per CU only more waves shorten the serial phases)
                wide16 = [(th, tw, 16, 64) for th, tw in tiles if th * tw <= 64]
                for cand in ["plain"] + [(th, tw, nw, ch) for th, tw in tiles for nw in (4, 8) for ch in ((64, 32) if narrow else (64,))] + wide16 + wave:
                    run = (lambda t=cand: self._run_choice(ent, a, c, n, h, w, t))
                    try:
                        run()
                    except RuntimeError as e:         # only the library's own "this tile / wave count does not fit" refusals mean "not a candidate"
                        if not any(m in str(e) for m in ("too many output tiles per wave", "unsupported geometry / tile", "chunk = 64, or 32", "16 waves take 64-channel chunks only")):
                            raise                     # a genuine launch failure must not be read as "candidate not applicable"
                        continue
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(3):
                        run()
                    e1.record()
                    e1.synchronize()
                    timed.append((e0.elapsed_time(e1) / 3.0, cand))
                choice = min(timed, key=lambda t: t[0])[1]
                ent[("timing", bucket, h, w)] = timed
            stale = [k for k in ent if isinstance(k, tuple) and k[0] == "choice"]
            for k in stale[:-7]:                      # at most 8 (bucket, resolution) entries per bottleneck
                ent.pop(k, None); ent.pop(("timing",) + k[1:4], None)
            ent[key] = choice
        return self._run_choice(ent, a, c, n, h, w, choice)

    @torch.no_grad()
    def __call__(self, x):
        if self.model.training:
            raise RuntimeError("bf16 inference is the eval-mode graph: call model.eval() first")
        if x.dim() != 4 or x.shape[1] != 3 or x.dtype != torch.float32 or x.device != self.device:
            raise ValueError("expected an fp32 (N,3,H,W) tensor on the model's device")
        n, _, h, w = x.shape
        self._prepare_weights()
        ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
        npix = n * ho * wo
        if _STEM_DIRECT and L.load_library().frost_infer_stem_ok(self.stem.cout):
            a = torch.empty(npix * self.stem.cout, dtype=torch.int16, device=self.device)
            call("frost_infer_stem", ptr(x), n, h, w, x.stride(0), x.stride(1), x.stride(2), x.stride(3), ptr(self.stem.pack), ptr(self.stem.biasf),
                 self.stem.cout, int(self.stem.relu), ptr(a), stream())
        else:
            col = torch.empty(npix * 64 + 64, dtype=torch.int16, device=self.device)
            call("frost_infer_stem_im2col", ptr(x), n, h, w, x.stride(0), x.stride(1), x.stride(2), x.stride(3), ptr(col), stream())
            a = self._pw(self.stem, col, npix, 64)
        c, h, w = self.stem.cout, ho, wo
        for ent in self.blocks:
            a, c, h, w = self._block(ent, a, c, n, h, w)
        npix = n * h * w
        a, c = self._pw(self.last, a, npix, c), self.last.cout
        pooled = torch.empty(n, c, dtype=torch.float32, device=self.device)
        call("frost_infer_avgpool", ptr(a), n, h * w, c, ptr(pooled), stream())
        logits = torch.empty(n, self.fc.out_channels, dtype=torch.float32, device=self.device)
        wfc = self.fc.weight.view(self.fc.out_channels, -1)
        call("frost_linear_f32", ptr(pooled), ptr(wfc), ptr(self.fc.bias), n, c, self.fc.out_channels, ptr(logits), stream())
        return logits
