"""Host-side executor of the MI355X FrostNet QAT path: owns the device arenas (qrecords, packed weights, integer
statistics, BN coefficients), sequences the C-ABI kernels of libfrost_hip.so on the current HIP stream and keeps a
tape so the backward pass replays the layers in reverse.  No host synchronisation anywhere: every scalar the
reference keeps in Python/torch buffers (observer min/max, scale, zero-point, BN running stats) lives in device
memory and is produced and consumed by kernels, so a whole fwd+bwd+optimizer step can be captured in a hipGraph.

Mirrors, per call, the reference module graph: frostnet.py:14-145 (ConvBNReLU / ConvBN / CascadePreExBottleneck),
torch nniqat.ConvBn(ReLU)2d._forward_approximate and FakeQuantize."""
import ctypes as C
import math

import os

import torch

from . import _lib as L
from ._lib import call, ptr, stream

SLACK = 256  # bytes of slack after every activation buffer (the MFMA K-tail may over-read, weights there are 0)


def round_up(a, b):
    return (a + b - 1) // b * b


_DW_FUSE = os.environ.get("FROST_DW_FUSE", "1") != "0"     # dev switch for A/B runs
_PROLOGUE3 = os.environ.get("FROST_PROLOGUE3", "1") != "0"  # per-step prologue (sigma snapshot, weight preparation, statistics reset) as three launches over a flat workgroup map (0 = the seven table launches; bit-identical)
_PROLOGUE3_CAP = int(os.environ.get("FROST_PROLOGUE3_CAP", "128"))  # most workgroups one layer gets (32 / 64 / 128: 19.94 / 19.87 / 19.83 ms per step, interleaved)
_DW_BWD_ONE = os.environ.get("FROST_DW_BWD_ONE", "1") != "0"   # depthwise k = 3 stride-1 backward of the tiled (high-resolution) layers: dc + weight gradient + data gradient in one sweep (csrc/frost_dwb.hip)
_DW_ONE_OVER_BLK = os.environ.get("FROST_DWB_OVER_BLK", "0") != "0"
_DW_C1 = os.environ.get("FROST_DWB_C1", "3") != "0"          # ... carrying the reduce pass of the pointwise layer in front of it (stride-2 layers with Cin = 16 / 24)
_DW_FUSE_K5 = os.environ.get("FROST_DW_FUSE_K5", "0") != "0"   # the 5x5 two-images-per-tile fused dc + wgrad variant spills 132 B of scratch: separate kernels are 0.8 % faster end to end (A/B, r2)
_PW_KEEP = os.environ.get("FROST_PW_KEEP", "1") != "0"      # backward of the wide-K pointwise layers: one conv recomputation + element-wise reduce / dc (A/B switch)
_FIN_FOLD = os.environ.get("FROST_FIN_FOLD", "1") != "0"    # dev switch: conv finalize folded into the statistics kernels' last workgroup
_BLOCK_FUSE = os.environ.get("FROST_BLOCK_FUSE", "1") != "0"   # block-boundary folds: the cat's observer update rides in the squeeze's finalize tail
# reduce emit + residual-add range pass in one launch (frost_pw_ew_emit_add), bit-identical to the two launches it replaces.  Slower than them in round 3 (both ended in
# same-address float atomics, one pair per workgroup, ~45 ns each); since the range passes keep per-workgroup slots and the last workgroup folds them
# (range_fold_last, csrc/frost_common.h: add_fwd_minmax 18-39 -> 13-28 us) the fused launch is 19-23 us against 12 + 17: on by default (round 4; -9 launches).
_BLOCK_PAIR = int(os.environ.get("FROST_BLOCK_PAIR", "3"))         # conv1 emit + conv2 statistics at the 14x14 / 7x7 stages: 1 = one launch (k_blk_expand_dw), 2 = chunked emit + image-resident statistics, 3 = whichever measured faster per shape (2 for 5x5 on 14x14: 103 vs 120 us), 0 = layer kernels
_BLOCK_DWRED = os.environ.get("FROST_BLOCK_DWRED", "1") != "0"    # conv2 emit + reduce_conv GEMM / statistics in one launch (same stages)
_BLOCK_DWBWD = int(os.environ.get("FROST_BLOCK_DWBWD", "2"))     # depthwise backward: dc + weight gradient + data gradient in one launch; 1 = 7x7 maps only, 2 = 14x14 too
_BLOCK_DWBRED = os.environ.get("FROST_BLOCK_DWBRED", "1") != "0"  # and its reduce pass in the same image-resident scheme
_BLOCK_C1 = os.environ.get("FROST_BLOCK_C1", "1") != "0"          # ... carrying the reduce pass of the bottleneck's conv1 (frost_block_dw_bwd_c1: conv1's output recomputed on the matrix cores per (image, chunk), S1 / S2 from the fp32 dx values)
_PWC_RED_MAXPIX = int(os.environ.get("FROST_PWC_RED_MAXPIX", "131072"))   # largest pixel count whose reduce pass runs on the chunked kernel (the dc pass always does)
# fused pointwise backward only on maps of at least this many pixels PER IMAGE (28 x 28 and up); below -- the squeeze convs of the 14 x 14 / 7 x 7 stages -- dc + data gradient,
# weight gradient on the second stream (-0.08 ms at B = 512).  A per-image rule: the decision is about which STAGE a layer belongs to and must not flip with the batch size
_PW_FUSE_MINMAP = int(os.environ.get("FROST_PW_FUSE_MINMAP", "400"))
_PWC_EMIT = os.environ.get("FROST_PWC_EMIT", "1") != "0"          # forward emit of wide pointwise layers on the chunked kernel
_BLOCK_EMIT_ADD = os.environ.get("FROST_BLOCK_EMIT_ADD", "1") != "0"
_STEM_FUSED_CVT = os.environ.get("FROST_STEM_CONVERTED", "1") != "0"   # converted inference: QuantStub + stem conv in one launch from the fp32 image (frost_stem_converted), bit-identical
# squeeze_conv statistics + finalize + emit + cat as ONE persistent launch with a device-wide barrier (frost_sq_fwd, csrc/frost_block.hip).  Parity-green and measured: 38 us
# against 27 + 19 us per layer in isolation at 14 x 14, +-0 inside the captured step (20.00 vs 20.01 ms, profiles/r05_persistent_squeeze.txt) -- off by default
_SQ_PERSIST = os.environ.get("FROST_SQ_PERSIST", "0") != "0"
_BLOCK_SQCAT = os.environ.get("FROST_BLOCK_SQCAT", "1") != "0"      # squeeze_conv emit + cat requantisation in one launch (frost_sq_emit_cat), bit-identical to the two it replaces
_PW_FUSE = os.environ.get("FROST_PW_FUSE", "1") != "0"     # dev switch: fused pointwise backward (dc + dgrad + wgrad in one kernel)
_WG_PRIO = int(os.environ.get("FROST_WG_PRIO", "0"))        # priority of the weight-gradient stream(s) (torch: lower = higher priority); A/B switch
_WG_NSTREAMS = int(os.environ.get("FROST_WG_NSTREAMS", "1"))  # weight-gradient streams taken round-robin per fork (single-GPU step only); A/B switch
_WG_STREAM = int(os.environ.get("FROST_WG_STREAM", "1"))   # bit 0: pointwise, bit 1: depthwise weight gradients on a second stream (A/B switch)
# depthwise weight gradients of maps no wider than this also go to the second stream.  Measured (two boxes, 2-3 runs each): 14 -> -0.2 ms, but
# 7, 28, 56 and "only the 14x14 maps" -> +1.1 ms (the captured graph serialises differently): too close to a cliff for a default, stays off
_DW_WG_MAXW = int(os.environ.get("FROST_DW_WG_MAXW", "0"))
# join the weight-gradient stream back into the main stream after this many forked weight gradients (0 = only at the end of the backward).  Inside a captured
# hipGraph ONE long side branch is not scheduled beside the layers it was forked from: the replay trace (profiles/r05_replay_nodes_*.txt) shows all 39 pointwise
# weight gradients starting ~8 ms late, beside the bandwidth-bound 112 x 112 / 56 x 56 backward instead of the latency-bound 14 x 14 / 7 x 7 one
_WG_JOIN = int(os.environ.get("FROST_WG_JOIN", "0"))
# FROST_WG_DEFER = H > 0: the pointwise weight gradients of layers on maps of height <= H are not forked one by one but collected and launched on the side stream
# behind ONE fork when the backward reaches a larger map (their dc / x buffers stay referenced until the join, as before).  Measured (profiles/r05_wgrad_schedule.txt):
# H = 14 is 0.33 ms SLOWER than per-layer forks (21.63 vs 21.30 ms/step) -- in the captured step the per-layer forks already start ~8 ms late, and starting all of
# them 1.8 ms earlier (beside layer3.0 / layer2.1's backward) costs more than the 39 fork gaps of ~4.6 us it removes.  0 = fork per layer (default).
_WG_DEFER = int(os.environ.get("FROST_WG_DEFER", "0"))
# FROST_WG_BATCH = N > 1: one fork per N pointwise weight gradients (each fork costs the main stream a ~4.6 us gap in the captured step): the weight gradients of N
# consecutive layers are launched on the side stream together, right after the dc pass of the N-th (A/B switch; 0 / 1 = fork per layer)
_WG_BATCH = int(os.environ.get("FROST_WG_BATCH", "0"))
# skip_add's backward folded into the element-wise reduce / dc passes of the reduce_conv that produced its second operand (frost_pw_ew_add_bwd): one launch less per
# residual block of the 14 x 14 / 7 x 7 stages and no materialised gout for that layer; bit-identical to the two launches it replaces (A/B switch)
_ADD_BWD_FUSE = os.environ.get("FROST_ADD_BWD_FUSE", "1") != "0"
# quant_cat's backward + the squeeze_conv's backward reduce pass in one launch (frost_sq_bwd_cat, the backward sibling of frost_sq_emit_cat).  Parity-green (the g4 / g4t
# block goldens pass with it) but SLOWER inside the step: 21.31 vs 21.17 ms (interleaved, profiles/r05_fusion_ab.txt) -- the two launches it replaces are 10 - 17 us
# (element-wise, 8 channels per thread) + 13 us (k_pw's reduce instance with DMA-staged tiles); one 256-thread workgroup per 128-pixel tile that recomputes the conv AND
# walks the wide gradient rows at 8-byte granularity does not beat them.  Off by default, kept as the A/B switch.
_SQ_BWD_CAT = os.environ.get("FROST_SQ_BWD_CAT", "0") != "0"


class Act:
    """An NHWC activation held as offset-binary int8 indices plus its qrecord (scale / zero-point on device)."""
    __slots__ = ("buf", "n", "h", "w", "c", "q", "grad", "needs_grad", "cint", "sum_observed", "cat_observed", "kept_next", "cat_done", "add_bwd", "bred_done", "is_col")

    def __init__(self, buf, n, h, w, c, q):
        self.buf, self.n, self.h, self.w, self.c, self.q = buf, n, h, w, c, q
        self.grad = None
        self.needs_grad = True
        self.cint = None          # wide-K pointwise layers: the integer conv output this activation was emitted from, kept for the backward
        self.cat_observed = False  # the cat consuming this activation already had its FakeQuantize record updated (folded into this layer's finalize)
        self.kept_next = None      # set by conv_pair: the integer conv output of the reduce_conv consuming this activation (its statistics pass already ran)
        self.sum_observed = False  # the residual add consuming this activation already had its range pass (fused into this layer's emit)
        self.cat_done = None       # the cat consuming this (squeeze) activation was already written by frost_sq_emit_cat: Engine.cat returns it
        self.bred_done = False     # backward: this (squeeze) layer's reduce pass already ran inside the cat's backward launch (frost_sq_bwd_cat)
        self.add_bwd = None        # backward: (gradient of the add's output, other operand, the add's record, ga, accumulate) -- the add's backward is folded into this activation's producer

    @property
    def npix(self):
        return self.n * self.h * self.w

    @property
    def numel(self):
        return self.npix * self.c

    def dequant(self):
        """fp32 NCHW tensor of the fake-quantised values (tests / feature taps)."""
        out = torch.empty(self.numel, dtype=torch.float32, device=self.buf.device)
        call("frost_dequant_act", ptr(self.buf), self.numel, ptr(self.q), ptr(out), stream())
        return out.view(self.n, self.h, self.w, self.c).permute(0, 3, 1, 2)

    def indices(self):
        """uint8 index tensor, NCHW."""
        v = self.buf[: self.numel].view(self.n, self.h, self.w, self.c).to(torch.int16) + 128
        return v.permute(0, 3, 1, 2).to(torch.uint8)


class QArena:
    """All fake-quantize sites of a model: [nsites, 8] fp32 on device (layout FROST_Q_* in include/frost_hip.h)."""

    def __init__(self, nsites, device):
        self.t = torch.zeros(nsites, L.Q_STRIDE, dtype=torch.float32, device=device)
        self.reset()
        self.next = 0

    def reset(self):
        self.t[:, L.Q_MIN] = float("inf")
        self.t[:, L.Q_MAX] = float("-inf")
        self.t[:, L.Q_SCALE] = 1.0
        self.t[:, L.Q_INV] = 1.0
        self.t[:, L.Q_ZP] = 0.0
        self.t[:, L.Q_FQMIN] = 0.0
        self.t[:, L.Q_FQMAX] = 0.0
        self.t[:, L.Q_QMAX] = 255.0        # activation index range 0..255 (127 under the fbgemm qconfig's reduce_range; set at bind time)
        ti = self.t.view(torch.int32)
        ti[:, L.Q_OBS_EN] = 1          # observer_enabled / fake_quant_enabled (8 bytes each, aliased by the torch buffers)
        ti[:, L.Q_FQ_EN] = 1

    def alloc(self):
        i = self.next
        self.next += 1
        return self.t[i]

    @staticmethod
    def set_qparams(rec, scale, zp):
        rec[L.Q_SCALE] = float(scale)
        rec[L.Q_INV] = 1.0 / float(scale)
        rec.view(torch.int32)[L.Q_ZP] = int(zp)

    @staticmethod
    def get(rec):
        r = rec.detach().cpu()
        return dict(min_val=float(r[L.Q_MIN]), max_val=float(r[L.Q_MAX]), scale=float(r[L.Q_SCALE]),
                    zero_point=int(r.view(torch.int32)[L.Q_ZP]), fq_min=float(r[L.Q_FQMIN]), fq_max=float(r[L.Q_FQMAX]))


class ConvLayer:
    """Device state of one ConvBN(ReLU) (kind pw / dw / stem) or of the classifier conv (kind cls)."""
    KIND_ID = {"pw": 0, "dw": 1, "stem": 2, "cls": 3}

    def __init__(self, name, kind, w, gamma, beta, rmean, rvar, nbt, bias, k, stride, relu, qw, qy):
        self.name, self.kind, self.k, self.stride, self.relu = name, kind, k, stride, relu
        self.w, self.gamma, self.beta, self.rmean, self.rvar, self.nbt, self.bias = w, gamma, beta, rmean, rvar, nbt, bias
        self.cout, self.cin_g = w.shape[0], w.shape[1]
        self.kk = k * k
        self.cpad = round_up(self.cout, 16)
        self.qw, self.qy = qw, qy
        dev = w.device
        if kind == "pw":
            self.kpad = round_up(self.cin_g, 64)
            nb = (self.cpad // 16) * (self.kpad // 64) * 1024
            cit, kb = round_up(self.cin_g, 16) // 16, (self.cpad + 31) // 32
            self.wt_pack = torch.zeros(cit * kb * 64 * 8, dtype=torch.int16, device=dev)
        elif kind == "dw":
            self.kpad, nb, self.wt_pack = 0, self.kk * self.cpad, None
        elif kind == "stem":      # runs as im2col + pointwise with K = 4*kk + 4 = 40 bytes per output pixel
            self.kpad, nb, self.wt_pack = 64, (self.cpad // 16) * 1024, None
        else:
            self.kpad, nb, self.wt_pack = 0, self.cout * self.cin_g, None
        self.wq_pack = torch.zeros(nb + 64, dtype=torch.int8, device=dev)
        self.wsum = torch.zeros(self.cpad, dtype=torch.int32, device=dev)
        self.wscale = torch.ones(self.cpad, dtype=torch.float32, device=dev)     # weight scale per output channel (per-tensor mode: replicated)
        self.wmin = self.wmax = None      # per-channel observer state (MovingAveragePerChannelMinMaxObserver), bound by the runner
        self.per_channel = False
        self.minmax2 = torch.tensor([float("inf"), float("-inf")], dtype=torch.float32, device=dev)
        self.coef = torch.zeros(L.COEF_ROWS, self.cpad, dtype=torch.float32, device=dev)
        self.sigma = torch.ones(self.cout, dtype=torch.float32, device=dev)
        self.stats = None       # view into the engine's stats arena
        self.fin_counter = torch.zeros(L.TICKET_WORDS, dtype=torch.int32, device=dev)     # last-workgroup-done tickets of the statistics pass (two-level, frost_common.h)
        self.bn_mod = None      # the BatchNorm2d module (its .training flag is checked by the runner)
        self.dwq = None         # fp32 scratch for dL/d(fake-quantised weight)

    def desc(self):
        d = L.FrostWDesc()
        d.w, d.gamma, d.rvar = self.w.data_ptr(), (self.gamma.data_ptr() if self.gamma is not None else None), \
            (self.rvar.data_ptr() if self.rvar is not None else None)
        d.qrec, d.wq_pack, d.wsum, d.minmax2 = self.qw.data_ptr(), self.wq_pack.data_ptr(), self.wsum.data_ptr(), \
            self.minmax2.data_ptr()
        d.wt_pack = self.wt_pack.data_ptr() if self.wt_pack is not None else None
        d.wscale = self.wscale.data_ptr()
        d.wmin, d.wmax = (self.wmin.data_ptr(), self.wmax.data_ptr()) if self.per_channel else (None, None)
        d.reserved1 = 1 if self.per_channel else 0
        d.cout, d.cin_g, d.kk, d.kind = self.cout, self.cin_g, self.kk, self.KIND_ID[self.kind]
        d.cpad, d.kpad = self.cpad, self.kpad
        d.reserved0 = 1 if getattr(self, "fold_rsqrt", False) else 0      # convert-time BN fold (gamma * rsqrt) instead of the QAT one
        return d


class Engine:
    def __init__(self, device, rule127=False):
        self.device = torch.device(device)
        self.rule127 = 1 if rule127 else 0
        self.layers = []
        self.tape = []
        self._table = None
        self.on_layer_grads = None     # callback(layer) after a layer's parameter gradients are final (DP overlap)
        self._side, self._keep = None, []
        # fp32-GRADIENT parity mode (csrc/frost_g32.hip; `model.grad_precision = "fp32"` / FROST_GRAD=fp32): activation gradients and dc in fp32, long sums in
        # fp64, plain kernels -- the reference's fp32 autograd precision instead of bf16 storage.  10-30 x slower; for parity statements, not for training runs.
        self.grad_fp32 = os.environ.get("FROST_GRAD", "bf16").lower() == "fp32"
        # FROST_GRAD=mixed / model.grad_precision = "mixed" (VERDICT r5 #3): the fp32-gradient kernels for every layer whose OUTPUT map is at most 14 x 14 (layer3 ... layer5,
        # last_layer, the head: where the bf16 mode's gradient error lives), bf16 storage and the production kernels above that; one fp32 -> bf16 conversion where the two meet
        # (the input gradient of layer3.0's depthwise conv).  Measured cost: profiles/r06_grad_modes.jsonl.
        self.grad_mixed = os.environ.get("FROST_GRAD", "bf16").lower() == "mixed"
        self.mixed_max_hw = int(os.environ.get("FROST_GRAD_MIXED_HW", "14"))

    # ------------------------------------------------------------------------------------------ plan
    def add_layer(self, layer):
        self.layers.append(layer)
        self._table = None
        return layer

    def _ensure_tables(self):
        if self._table is not None:
            return
        n = len(self.layers)
        self._wgmaps = {}
        arr = (L.FrostWDesc * n)()
        for i, l in enumerate(self.layers):
            arr[i] = l.desc()
        self._table = L.struct_to_tensor(arr, self.device)
        self._max_elems = max(l.w.numel() for l in self.layers)
        offs, total = [], 0
        for l in self.layers:
            offs.append(total)
            total += l.cpad * L.STATS_BYTES_PER_CH
        self._stats = torch.zeros(total, dtype=torch.uint8, device=self.device)
        for l, o in zip(self.layers, offs):
            l.stats = self._stats[o: o + l.cpad * L.STATS_BYTES_PER_CH]
        self._cpads = torch.tensor([l.cpad for l in self.layers], dtype=torch.int32, device=self.device)
        self._offs = torch.tensor(offs, dtype=torch.int64, device=self.device)
        ptrs = (C.c_void_p * n)(*[l.sigma.data_ptr() for l in self.layers])
        self._sigma_ptrs = L.struct_to_tensor(ptrs, self.device)

    def begin_step(self, observe=True, part=None):
        """Per-step prologue: BN-fold + weight fake-quant + packing for every layer (3 launches), sigma_r snapshot,
        integer-stat reset.  Must run before the first conv of a forward pass (uses running_var BEFORE its update).
        part = (lo, hi): only the layers [lo, hi) of the table (registration order = forward order) -- the runner prepares the few small layers of the
        high-resolution stages first and the rest (where the parameters are) on a second stream under them; every layer must be covered before it runs."""
        self._ensure_tables()
        n = len(self.layers)
        lo, hi = (0, n) if part is None else (max(0, int(part[0])), min(n, int(part[1])))
        if hi > lo and _PROLOGUE3:
            # three launches over a flat workgroup map (a layer gets workgroups in proportion to its weights) instead of seven launches of nlayers x 8 ... 256 mostly idle ones
            m = hi - lo
            maps = self.__dict__.setdefault("_wgmaps", {})
            if (lo, hi) not in maps:
                rows = []
                for i, l in enumerate(self.layers[lo:hi]):
                    nsl = max(1, min(_PROLOGUE3_CAP, -(-l.w.numel() // 2048)))
                    rows += [[i, sl, nsl, 0] for sl in range(nsl)]
                maps[(lo, hi)] = torch.tensor(rows, dtype=torch.int32, device=self.device)
            wg = maps[(lo, hi)]
            call("frost_step_prologue", C.c_void_p(self._table.data_ptr() + lo * C.sizeof(L.FrostWDesc)), m, ptr(wg), wg.shape[0],
                 C.c_void_p(self._sigma_ptrs.data_ptr() + 8 * lo), ptr(self._stats), C.c_void_p(self._cpads.data_ptr() + 4 * lo),
                 C.c_void_p(self._offs.data_ptr() + 8 * lo), self.rule127, 1 if observe else 0, stream(),
                 prof=("weight_prep", sum(4 * l.w.numel() + l.wq_pack.numel() for l in self.layers[lo:hi])))
        elif hi > lo:
            m = hi - lo
            tab = C.c_void_p(self._table.data_ptr() + lo * C.sizeof(L.FrostWDesc))
            call("frost_save_sigma", tab, C.c_void_p(self._sigma_ptrs.data_ptr() + 8 * lo), m, stream())
            call("frost_weight_prep", tab, m, max(l.w.numel() for l in self.layers[lo:hi]), self.rule127, 1 if observe else 0, stream(),
                 prof=("weight_prep", sum(4 * l.w.numel() + l.wq_pack.numel() for l in self.layers[lo:hi])))
            call("frost_stats_init_table", ptr(self._stats), C.c_void_p(self._cpads.data_ptr() + 4 * lo), C.c_void_p(self._offs.data_ptr() + 8 * lo), m, stream())
        if lo == 0:
            self.tape = []

    # ------------------------------------------------------------------------------------------ helpers
    def new_act(self, n, h, w, c, q):
        buf = torch.empty(n * h * w * c + SLACK, dtype=torch.int8, device=self.device)
        return Act(buf, n, h, w, c, q)

    def act_from_indices(self, idx_nchw, q):
        """Test helper: uint8 NCHW indices -> Act."""
        n, c, h, w = idx_nchw.shape
        a = self.new_act(n, h, w, c, q)
        v = (idx_nchw.to(self.device).permute(0, 2, 3, 1).contiguous().to(torch.int16) - 128).to(torch.int8)
        a.buf[: a.numel].copy_(v.view(-1))
        return a

    def grad_is_fp32(self, a):
        """True iff the gradient of activation `a` is stored in fp32: everywhere in the fp32-gradient mode, on maps of at most mixed_max_hw pixels a side in the mixed mode."""
        return self.grad_fp32 or (self.grad_mixed and a.h <= self.mixed_max_hw)

    def _grad_slot(self, a):
        """Gradient buffer of an Act (bf16 bits, or fp32 in the fp32-gradient / mixed mode): returns (tensor, accumulate_flag)."""
        if a.grad is None:
            a.grad = torch.empty(a.numel + 64, dtype=torch.float32 if self.grad_is_fp32(a) else torch.int16, device=a.buf.device)
            return a.grad, 0
        return a.grad, 1

    # ------------------------------------------------------------------------------------------ forward ops
    def quantize_input(self, x, q, observe=True):
        """QuantStub (frostnet.py:319-320): observer + fake-quantise of the fp32 image; accepts NCHW or channels_last."""
        n, c, h, w = x.shape
        sn, sc, sh, sw = x.stride()
        if observe:
            mm = torch.tensor([float("inf"), float("-inf")], dtype=torch.float32, device=self.device) \
                if not hasattr(self, "_in_mm") else self._in_mm
            self._in_mm = mm
            call("frost_fill_minmax", ptr(mm), 1, stream())
            call("frost_minmax_input", ptr(x), n, c, h, w, sn, sc, sh, sw, ptr(mm), stream())
            call("frost_observer_update", ptr(q), ptr(mm), 0, 0, 1, stream())
        cpad = round_up(c, 4)
        a = self.new_act(n, h, w, cpad, q)
        call("frost_quantize_input", ptr(x), n, c, h, w, sn, sc, sh, sw, ptr(q), ptr(a.buf), cpad, stream())
        a.needs_grad = False
        if getattr(self, "trace", None) is not None:
            self.trace.append(("input", a))
        return a

    def _conv_launch(self, l, x, mode, y):
        """mode 0 = stats pass (reads x), mode 1 = emit pass (reads x, writes y).  Algorithmic bytes of a launch =
        the activation bytes it must move at 1 B/element (+ the packed weights once)."""
        st = ptr(l.stats)
        nb = x.numel + l.wq_pack.numel() + (y.numel if y is not None else 0)
        tag = (f"{l.kind}_fwd_{'emit' if mode else 'stats'}", nb)      # mode 2 / 3 = emit with the converted-inference requantisation (QNNPACK / FBGEMM form)
        if l.kind == "pw" and mode == 1 and _PWC_EMIT and L.load_library().frost_pwc_bwd_ok(x.npix, x.c, l.cout):
            # wide layers outside a conv1 -> conv2 pair (last_layer, the stride-2 blocks' conv1): chunked emit, y leaves as 64-byte row pieces (csrc/frost_pwc.hip)
            call("frost_pwc_conv_fwd_emit", ptr(x.buf), ptr(x.q), ptr(l.wq_pack), ptr(l.wsum), x.npix, x.c, l.cout, ptr(l.coef), ptr(l.qy), ptr(y.buf), stream(), prof=tag)
        elif l.kind in ("pw", "stem"):
            call("frost_pw_conv_fwd", ptr(x.buf), ptr(x.q), ptr(l.wq_pack), ptr(l.wsum), x.npix, x.c, l.cout, mode, st,
                 ptr(l.coef), ptr(l.qy), int(l.relu), ptr(y.buf) if y is not None else None, stream(), prof=tag)
        elif l.kind == "dw":
            call("frost_dw_conv_fwd", ptr(x.buf), ptr(x.q), ptr(l.wq_pack), ptr(l.wsum), x.n, x.h, x.w, x.c, l.k,
                 l.stride, mode, st, ptr(l.coef), ptr(l.qy), int(l.relu), ptr(y.buf) if y is not None else None, stream(),
                 prof=tag)
        else:
            raise ValueError(l.kind)


    def stem_im2col(self, x):
        """The stem's 3 x 3 / stride-2 patches of the quantised NHWC image as rows of 40 bytes (k = tap * 4 + channel): the input of the stem's pointwise
        kernels, kept for its weight gradient.  Needs no weights, so the runner issues it BEFORE joining the weight-preparation stream."""
        ho, wo = (x.h + 2 - 3) // 2 + 1, (x.w + 2 - 3) // 2 + 1
        xc = self.new_act(x.n, ho, wo, 40, x.q)
        call("frost_stem_im2col", ptr(x.buf), ptr(x.q), x.n, x.h, x.w, ptr(xc.buf), stream(), prof=("stem_im2col", x.numel + xc.numel))
        xc.needs_grad = False
        xc.is_col = True
        return xc

    def conv(self, l, x, training=True, observe=True, residual=None, cat=None):
        """ConvBn(ReLU)2d QAT forward + activation fake-quant: stats pass -> finalize -> emit pass (recompute).
        residual = (a, q_sum): this layer is a bottleneck's reduce_conv whose output goes straight into skip_add.add(a, .) with FakeQuantize record
        q_sum -- where the layer keeps its integer conv output, the emit pass and the add's range / observer pass are ONE launch
        (frost_pw_ew_emit_add); the returned activation then carries `sum_observed = True` and Engine.add skips its own range pass.
        cat = (q_b, q_cat): this layer is a squeeze_conv whose output is concatenated with an activation of record q_b under the FakeQuantize record
        q_cat: that record is updated in this layer's finalize tail (`cat_observed = True` on the result; Engine.cat skips frost_cat_observe)."""
        pad = (l.k - 1) // 2
        ho, wo = (x.h + 2 * pad - l.k) // l.stride + 1, (x.w + 2 * pad - l.k) // l.stride + 1
        if getattr(x, "is_col", False):          # the stem's im2col ran already (Engine.stem_im2col: in front of the join with the weight-preparation stream)
            ho, wo = x.h, x.w
        # frozen BatchNorm (`_freeze_stages`, frostnet_features.py:354-359: bn.eval() inside a training forward): this layer normalises with its running
        # statistics (finalize in eval form, nothing updated); its backward is the batch-statistics one minus the S1 / S2 terms (Engine._frozen_after_reduce)
        net_training = training
        l.frozen = bool(training and l.bn_mod is not None and not l.bn_mod.training)
        if l.frozen:
            training = False
        if l.kind == "stem" and not getattr(x, "is_col", False):      # im2col once (kept for the backward wgrad), then the pointwise int8-MFMA kernels
            x = self.stem_im2col(x)
        y = self.new_act(x.n, ho, wo, l.cout, l.qy)
        need_stats = training or observe
        if need_stats and _FIN_FOLD:
            # statistics pass with the finalize folded into its last workgroup (two launches per conv forward instead of three)
            fin = L.FrostFinDesc(l.qw.data_ptr(), l.gamma.data_ptr(), l.beta.data_ptr(), l.rmean.data_ptr(), l.rvar.data_ptr(), l.nbt.data_ptr(),
                                 l.coef.data_ptr(), l.qy.data_ptr(), l.fin_counter.data_ptr(), 1 if training else 0, int(l.relu), 1 if observe else 0, 0,
                                 l.wscale.data_ptr(), None, None)
            cat_fold = cat is not None and _BLOCK_FUSE and l.kind == "pw" and not (_PW_KEEP and x.c > 256 and l.cout < x.c)
            if cat_fold:                  # squeeze_conv: the cat's FakeQuantize record is updated in this layer's finalize tail (no frost_cat_observe launch)
                fin.cat_qrec_b, fin.cat_qrec_y = cat[0].data_ptr(), cat[1].data_ptr()
            nb = x.numel + l.wq_pack.numel()
            if _PW_KEEP and l.kind == "pw" and x.c > 256 and l.cout < x.c:
                # wide-K reduce layer (x rows too long for k_pw's DMA tile, Cout < Cin: the int32 output is smaller than the x re-reads it saves): statistics + finalize on the stand-alone int8 GEMM kernel, which also stores the
                # integer conv output (smaller than x); y is emitted element-wise from it and the backward's reduce / dc need no recomputation
                cint = getattr(x, "kept_next", None)          # conv_pair ran this layer's GEMM / statistics / finalize inside the block kernel
                if cint is None:
                    cint = torch.empty(y.numel + 64, dtype=torch.int32, device=self.device)
                    call("frost_pw_conv_fwd_keep", ptr(x.buf), ptr(x.q), ptr(l.wq_pack), ptr(l.wsum), x.npix, x.c, l.cout, ptr(l.stats), C.byref(fin), ptr(cint),
                         stream(), prof=("pw_fwd_stats", nb + 4 * y.numel))
                x.kept_next = None
                if residual is not None and observe and _BLOCK_EMIT_ADD:
                    a, q_sum = residual
                    call("frost_pw_ew_emit_add", ptr(cint), x.npix, l.cout, ptr(l.coef), ptr(l.qy), int(l.relu), ptr(a.buf), ptr(a.q), ptr(y.buf), ptr(self._add_state()),
                         ptr(q_sum), 1, stream(), prof=("pw_fwd_emit_add", 6 * y.numel))
                    y.sum_observed = True
                else:
                    call("frost_pw_ew", ptr(cint), x.npix, l.cout, ptr(l.coef), ptr(l.qy), int(l.relu), 2, None, ptr(y.buf), stream(),
                         prof=("pw_fwd_emit", 5 * y.numel))
                if getattr(self, "trace", None) is not None:
                    self.trace.append((l.name, y))
                self.tape.append(("conv", l, x, y))
                y.cint = cint if net_training else None
                return y
            if (_SQ_PERSIST and cat_fold and _BLOCK_SQCAT and l.kind == "pw" and l.relu and getattr(l, "hswish", None) is None
                    and L.load_library().frost_sq_fwd_ok(x.npix, x.c, l.cout)):
                # squeeze_conv of a CAS bottleneck as ONE persistent launch: statistics -> device-wide barrier with the finalize inside -> emit + both halves of the cat from
                # the accumulators the workgroup kept (csrc/frost_block.hip k_sq_fwd; bit-identical to frost_pw_conv_fwd_fin + frost_sq_emit_cat)
                ycat = self.new_act(x.n, ho, wo, l.cout + x.c, cat[1])
                need = int(L.load_library().frost_sq_fwd_slot_bytes(x.npix, l.cout))
                if getattr(self, "_sq_slots", None) is None or self._sq_slots.numel() < need:          # per-workgroup statistics rows (one buffer: the squeeze convs run one after another)
                    self._sq_slots = torch.empty(max(need, 4 << 20), dtype=torch.uint8, device=self.device)
                call("frost_sq_fwd", ptr(x.buf), ptr(x.q), ptr(l.wq_pack), ptr(l.wsum), x.npix, x.c, l.cout, ptr(l.stats), C.byref(fin), ptr(self._sq_slots), ptr(y.buf), ptr(ycat.buf),
                     stream(), prof=("sq_fwd", x.numel + l.wq_pack.numel() + y.numel + ycat.numel))
                y.cat_done = ycat
                y.cat_observed = True
                if getattr(self, "trace", None) is not None:
                    self.trace.append((l.name, y))
                self.tape.append(("conv", l, x, y))
                return y
            if l.kind in ("pw", "stem"):
                call("frost_pw_conv_fwd_fin", ptr(x.buf), ptr(x.q), ptr(l.wq_pack), ptr(l.wsum), x.npix, x.c, l.cout, ptr(l.stats), C.byref(fin),
                     stream(), prof=(f"{l.kind}_fwd_stats", nb))
            else:
                call("frost_dw_conv_fwd_fin", ptr(x.buf), ptr(x.q), ptr(l.wq_pack), ptr(l.wsum), x.n, x.h, x.w, x.c, l.k, l.stride, ptr(l.stats),
                     C.byref(fin), stream(), prof=("dw_fwd_stats", nb))
        else:
            if need_stats:
                self._conv_launch(l, x, 0, None)
            call("frost_conv_finalize", ptr(l.stats) if need_stats else None, y.npix, l.cout, ptr(x.q), ptr(l.qw),
                 ptr(l.gamma), ptr(l.beta), ptr(l.rmean), ptr(l.rvar), ptr(l.nbt), 1 if training else 0, int(l.relu),
                 1 if observe else 0, ptr(l.coef), ptr(l.qy), ptr(l.wscale), stream(), prof=("conv_finalize", 48 * l.cout))
        if (need_stats and _FIN_FOLD and cat is not None and cat_fold and _BLOCK_SQCAT and l.kind == "pw" and l.relu and getattr(l, "hswish", None) is None
                and L.load_library().frost_sq_emit_cat_ok(x.c, l.cout)):
            # squeeze_conv of a CAS bottleneck: emit + both halves of the cat from one staged x tile (the cat's record was finalized in the statistics launch's tail)
            ycat = self.new_act(x.n, ho, wo, l.cout + x.c, cat[1])
            call("frost_sq_emit_cat", ptr(x.buf), ptr(x.q), ptr(l.wq_pack), ptr(l.wsum), x.npix, x.c, l.cout, ptr(l.coef), ptr(l.qy), ptr(cat[1]), ptr(y.buf), ptr(ycat.buf),
                 1, stream(), prof=("pw_fwd_emit", x.numel + l.wq_pack.numel() + y.numel + ycat.numel))
            y.cat_done = ycat
        else:
            self._conv_launch(l, x, 1, y)
        if need_stats and _FIN_FOLD and cat is not None and cat_fold:
            y.cat_observed = True
        if getattr(self, "trace", None) is not None:
            self.trace.append((l.name, y))
        self.tape.append(("conv", l, x, y))
        return y

    def pair_fusable(self, l1, l2, x, training, observe):
        """conv1 -> conv2 of a bottleneck at the 14x14 / 7x7 stages: conv1's emit pass and conv2's statistics pass run as ONE launch (csrc/frost_block.hip)."""
        if any(l.bn_mod is not None and not l.bn_mod.training for l in (l1, l2)):
            return False          # a frozen BatchNorm (eval-form finalize) takes the layer path
        return bool(_BLOCK_PAIR and _FIN_FOLD and training and observe and l1.kind == "pw" and l2.kind == "dw" and l1.k == 1
                    and getattr(l1, "hswish", None) is None and getattr(l2, "hswish", None) is None
                    and L.load_library().frost_block_supported(x.h, x.w, l2.k, l2.stride, x.c, l1.cout))

    def reduce_fusable(self, l2, l3, y1):
        """conv2 -> reduce_conv: conv2's emit pass and reduce_conv's GEMM / statistics (kept-output path) as one launch."""
        if l3 is not None and l3.bn_mod is not None and not l3.bn_mod.training:
            return False
        return bool(_BLOCK_DWRED and _PW_KEEP and l3 is not None and l3.kind == "pw" and l3.k == 1 and l2.cout > 256 and l3.cout < l2.cout
                    and getattr(l3, "hswish", None) is None
                    and L.load_library().frost_block_dw_reduce_supported(y1.h, y1.w, l2.k, l2.stride, l2.cout, l3.cout))

    def conv_pair(self, l1, l2, x, training=True, observe=True, l3=None):
        """`conv2(conv1(x))` of CascadePreExBottleneck.forward (frostnet.py:134-137) with the expanded tensor kept on chip between conv1's activation
        FakeQuantize and conv2's batch statistics: conv1 statistics + finalize (k_pw) -> frost_block_expand_dw_stats -> conv2 emit (k_dw3).
        Tape and saved tensors are those of two Engine.conv calls: the backward is unchanged."""
        l1.frozen = l2.frozen = False
        if l3 is not None:
            l3.frozen = False
        y1 = self.new_act(x.n, x.h, x.w, l1.cout, l1.qy)
        fin1 = L.FrostFinDesc(l1.qw.data_ptr(), l1.gamma.data_ptr(), l1.beta.data_ptr(), l1.rmean.data_ptr(), l1.rvar.data_ptr(), l1.nbt.data_ptr(),
                              l1.coef.data_ptr(), l1.qy.data_ptr(), l1.fin_counter.data_ptr(), 1, int(l1.relu), 1, 0, l1.wscale.data_ptr(), None, None)
        call("frost_pw_conv_fwd_fin", ptr(x.buf), ptr(x.q), ptr(l1.wq_pack), ptr(l1.wsum), x.npix, x.c, l1.cout, ptr(l1.stats), C.byref(fin1),
             stream(), prof=("pw_fwd_stats", x.numel + l1.wq_pack.numel()))
        fin2 = L.FrostFinDesc(l2.qw.data_ptr(), l2.gamma.data_ptr(), l2.beta.data_ptr(), l2.rmean.data_ptr(), l2.rvar.data_ptr(), l2.nbt.data_ptr(),
                              l2.coef.data_ptr(), l2.qy.data_ptr(), l2.fin_counter.data_ptr(), 1, int(l2.relu), 1, 0, l2.wscale.data_ptr(), None, None)
        if _BLOCK_PAIR == 2 or (_BLOCK_PAIR == 3 and x.h == 14 and l2.k == 5):
            self._conv_launch(l1, x, 1, y1)
            call("frost_block_dw_stats", ptr(y1.buf), ptr(y1.q), ptr(l2.wq_pack), ptr(l2.wsum), x.n, x.h, x.w, l2.cout, l2.k, ptr(l2.stats), C.byref(fin2), stream(),
                 prof=("blk_dw_stats", y1.numel))
        else:
            call("frost_block_expand_dw_stats", ptr(x.buf), ptr(x.q), ptr(l1.wq_pack), ptr(l1.wsum), ptr(l1.coef), ptr(l1.qy), ptr(y1.buf), x.n, x.h, x.w, x.c,
                 l1.cout, ptr(l2.wq_pack), ptr(l2.wsum), l2.k, ptr(l2.stats), C.byref(fin2), stream(),
                 prof=("blk_expand_dw", x.numel + l1.wq_pack.numel() + y1.numel))
        y2 = self.new_act(x.n, x.h, x.w, l2.cout, l2.qy)
        if self.reduce_fusable(l2, l3, y1):
            # the block's third boundary: y2 goes from the depthwise stencil through LDS into reduce_conv's K-split GEMM; Engine.conv(l3, y2, kept=...) emits
            cint = torch.empty(x.npix * l3.cout + 64, dtype=torch.int32, device=self.device)
            fin3 = L.FrostFinDesc(l3.qw.data_ptr(), l3.gamma.data_ptr(), l3.beta.data_ptr(), l3.rmean.data_ptr(), l3.rvar.data_ptr(), l3.nbt.data_ptr(),
                                  l3.coef.data_ptr(), l3.qy.data_ptr(), l3.fin_counter.data_ptr(), 1, int(l3.relu), 1, 0, l3.wscale.data_ptr(), None, None)
            call("frost_block_dw_reduce", ptr(y1.buf), ptr(y1.q), ptr(l2.wq_pack), ptr(l2.wsum), ptr(l2.coef), ptr(l2.qy), int(l2.relu), ptr(y2.buf), x.n, x.h, x.w,
                 l2.cout, l2.k, ptr(l3.wq_pack), ptr(l3.wsum), l3.cout, ptr(cint), ptr(l3.stats), C.byref(fin3), stream(),
                 prof=("blk_dw_reduce", y1.numel + y2.numel + l3.wq_pack.numel() + 4 * x.npix * l3.cout))
            y2.kept_next = cint
        else:
            self._conv_launch(l2, y1, 1, y2)
        if getattr(self, "trace", None) is not None:
            self.trace.append((l1.name, y1)); self.trace.append((l2.name, y2))
        self.tape.append(("conv", l1, x, y1))
        self.tape.append(("conv", l2, y1, y2))
        return y2

    def cat(self, a, b, q, observe=True):
        """FloatFunctional.cat + its FakeQuantize (frostnet.py:129)."""
        y = getattr(a, "cat_done", None)
        if y is not None:                  # written by the squeeze's emit launch (frost_sq_emit_cat)
            a.cat_done = None
        else:
            if not getattr(a, "cat_observed", False):
                call("frost_cat_observe", ptr(a.q), ptr(b.q), ptr(q), 1 if observe else 0, stream())
            y = self.new_act(a.n, a.h, a.w, a.c + b.c, q)
            call("frost_cat_requant", ptr(a.buf), ptr(a.q), a.c, ptr(b.buf), ptr(b.q), b.c, a.npix, ptr(q), ptr(y.buf), stream(),
                 prof=("cat_fwd", 2 * y.numel))
        if getattr(self, "trace", None) is not None:
            self.trace.append(("cat", y))
        self.tape.append(("cat", a, b, y))
        return y

    def _add_state(self):
        if not hasattr(self, "_add_mm"):      # {2 unused, arrival ticket, per-workgroup range slots}: zeroed once, every launch leaves the ticket zeroed again
            self._add_mm = torch.zeros(L.load_library().frost_add_state_floats(), dtype=torch.float32, device=self.device)
        return self._add_mm

    def add(self, a, b, q, observe=True):
        """FloatFunctional.add + its FakeQuantize (frostnet.py:142)."""
        if observe and not getattr(b, "sum_observed", False):       # range of the sum + observer update in one launch (the update runs in the last workgroup)
            call("frost_add_minmax_observe", ptr(a.buf), ptr(a.q), ptr(b.buf), ptr(b.q), a.numel, ptr(self._add_state()), ptr(q), 1, stream(),
                 prof=("add_fwd_minmax", 2 * a.numel))
        y = self.new_act(a.n, a.h, a.w, a.c, q)
        call("frost_add_requant", ptr(a.buf), ptr(a.q), ptr(b.buf), ptr(b.q), a.numel, ptr(q), ptr(y.buf), stream(),
             prof=("add_fwd_emit", 3 * a.numel))
        if getattr(self, "trace", None) is not None:
            self.trace.append(("add", y))
        self.tape.append(("add", a, b, y))
        return y

    def hswish(self, x, q_relu6, q_site, q_out, observe=True):
        """Quantizable hard-swish on a fake-quantised activation (reference `_Hswish`, mobilenetv3.py:43-56): q_relu6 / q_site = the FakeQuantize
        records of `relu6` and `quant_mul1` (both observed), q_out = the record of the result after mul_scalar(1/6).  Table-driven: see csrc/frost_convert.hip."""
        if not hasattr(self, "_hsw_present"):
            self._hsw_present = torch.zeros(8, dtype=torch.int32, device=self.device)
        lut = torch.empty(256 + 1024, dtype=torch.uint8, device=self.device)
        y = self.new_act(x.n, x.h, x.w, x.c, q_out)
        call("frost_hswish_fwd", ptr(x.buf), ptr(x.q), x.numel, ptr(self._hsw_present), ptr(q_relu6), ptr(q_site), ptr(q_out), 1 if observe else 0, ptr(lut),
             ptr(y.buf), stream(), prof=("hswish_fwd", 3 * x.numel))
        self.tape.append(("hswish", x, y, lut))
        return y

    def hswish_converted(self, x, q_site, q_out, owner=None):
        """The converted model's hard-swish on a quint8 activation (QFunctional.add_scalar -> nnq.ReLU6 -> QFunctional.mul at quant_mul1's frozen record -> mul_scalar):
        a 256-entry table of the input index built on the device from the two records (csrc/frost_convert.hip k_hsw_cvt_lut), then one pass.  `owner` (the conv layer):
        the records are frozen after convert(), so the table is built once per (layer, records' version) and later forwards run the table pass alone."""
        key = (x.q.data_ptr(), x.q._version, q_site._version)
        cached = owner is not None and getattr(owner, "_hsw_cvt", None) is not None and owner._hsw_cvt[0] == key
        lut = owner._hsw_cvt[1] if cached else torch.empty(256, dtype=torch.uint8, device=self.device)
        y = self.new_act(x.n, x.h, x.w, x.c, q_out)
        call("frost_hswish_converted", ptr(x.buf), ptr(x.q), x.numel, None if cached else ptr(q_site), ptr(q_out), ptr(lut), ptr(y.buf), stream(),
             prof=("hswish_converted", 2 * x.numel))
        if owner is not None and not cached:
            owner._hsw_cvt = (key, lut)
        return y

    def head(self, l, x, drop_mask=None, observe=True):
        """AdaptiveAvgPool2d(1) -> Dropout -> nnqat.Conv2d(1280,nclass,1) + activation FQ (frostnet.py:295-299)."""
        n, c = x.n, x.c
        pooled = torch.empty(n, c, dtype=torch.float32, device=self.device)
        call("frost_avgpool", ptr(x.buf), ptr(x.q), n, x.h * x.w, c, ptr(drop_mask), ptr(pooled), stream())
        raw = torch.empty(n, l.cout, dtype=torch.float32, device=self.device)
        call("frost_classifier_fwd", ptr(pooled), ptr(l.wq_pack), ptr(l.qw), ptr(l.bias), n, c, l.cout, ptr(raw),
             ptr(l.wscale) if l.per_channel else None, stream())
        if observe:
            if not hasattr(self, "_head_mm"):
                self._head_mm = torch.empty(2, dtype=torch.float32, device=self.device)
            call("frost_fill_minmax", ptr(self._head_mm), 1, stream())
            call("frost_minmax_f32", ptr(raw), raw.numel(), ptr(self._head_mm), stream())
            call("frost_observer_update", ptr(l.qy), ptr(self._head_mm), 0, 0, 1, stream())
        logits = torch.empty_like(raw)
        self.last_raw = raw              # pre-fake-quant classifier output (tests: the north-star 1e-3 comparison point)
        call("frost_fake_quant_f32", ptr(raw), raw.numel(), ptr(l.qy), 0, getattr(self, "act_qmax", 255), ptr(logits), None, stream())
        self.tape.append(("head", l, x, pooled, raw, drop_mask))
        return logits

    # ------------------------------------------------------------------------------------------ converted int8 inference
    def prepare_converted(self, observe=True):
        """torch.quantization.convert: fold BN with the running statistics (gamma * rsqrt(var + eps)), run every weight FakeQuantize once
        more on the folded weight (the reference converts with observers still enabled, Classification/evaluate.py:130) and keep the
        resulting int8 packs.  After this the weights are frozen: converted forwards do not touch them again."""
        for l in self.layers:
            l.fold_rsqrt = True
            l._cfin_key = None
            l._hsw_cvt = None
        self._table = None
        self._ensure_tables()
        call("frost_weight_prep", ptr(self._table), len(self.layers), self._max_elems, self.rule127, 1 if observe else 0, stream())

    def conv_converted(self, l, x, fb=False, cat=None):
        """quantized::conv2d(_relu) of the converted model: integer bias + fp32 requantisation in the emit epilogue (mode 2); fb = the FBGEMM engine's form
        for a per-channel ('fbgemm' qconfig) model: float bias + per-channel multipliers (mode 3)."""
        pad = (l.k - 1) // 2
        ho, wo = (x.h + 2 * pad - l.k) // l.stride + 1, (x.w + 2 * pad - l.k) // l.stride + 1
        if l.kind == "stem":
            xc = self.new_act(x.n, ho, wo, 40, x.q)
            call("frost_stem_im2col", ptr(x.buf), ptr(x.q), x.n, x.h, x.w, ptr(xc.buf), stream())
            x = xc
        y = self.new_act(x.n, ho, wo, l.cout, l.qy)
        self._converted_coef(l, x.q, fb)
        hsw = getattr(l, "hswish", None)
        if hsw is None and cat is not None and _BLOCK_SQCAT and l.kind == "pw" and L.load_library().frost_sq_emit_cat_ok(x.c, l.cout):
            # squeeze_conv of a CAS bottleneck: the converted emit and both halves of the cat (records frozen) in one launch
            ycat = self.new_act(x.n, ho, wo, l.cout + x.c, cat)
            call("frost_sq_emit_cat", ptr(x.buf), ptr(x.q), ptr(l.wq_pack), ptr(l.wsum), x.npix, x.c, l.cout, ptr(l.coef), ptr(l.qy), ptr(cat), ptr(y.buf), ptr(ycat.buf),
                 3 if fb else 2, stream())
            if getattr(self, "trace", None) is not None:
                self.trace.append((l.name, y)); self.trace.append(("cat", ycat))
            return ycat
        self._conv_launch(l, x, 3 if fb else 2, y)
        if getattr(self, "trace", None) is not None:
            self.trace.append((l.name, y))
        if hsw is not None:                       # ConvBNHswish: the conv emitted linearly at its own record; the hard-swish is a table pass
            y = self.hswish_converted(y, hsw[1], hsw[2], owner=l)
            if getattr(self, "trace", None) is not None:
                self.trace.append((l.name + ".act", y))
        return y

    def stem_converted(self, l, x, q_in, fb=False):
        """QuantStub (frozen record) + the converted stem conv from the fp32 image in ONE launch; None when the fused launch does not apply (a site trace
        wants the quantised image, the switch is off, an unusual stem width) -- the caller then runs quantize_input + conv_converted."""
        if not _STEM_FUSED_CVT or getattr(self, "trace", None) is not None or l.kind != "stem" or x.shape[1] != 3 \
                or not L.load_library().frost_stem_converted_ok(l.cout):
            return None
        n, _, h, w = x.shape
        ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
        self._converted_coef(l, q_in, fb)
        y = self.new_act(n, ho, wo, l.cout, l.qy)
        call("frost_stem_converted", ptr(x), n, h, w, *x.stride(), ptr(q_in), ptr(l.wq_pack), ptr(l.wsum), ptr(l.coef), ptr(l.qy), l.cout, 3 if fb else 2,
             ptr(y.buf), stream())
        if getattr(l, "hswish", None) is not None:
            y = self.hswish_converted(y, l.hswish[1], l.hswish[2], owner=l)
        return y

    def _converted_coef(self, l, qx, fb):
        """Requantisation coefficients of a converted conv (integer bias at scale s_x * s_w, multiplier s_x * s_w / s_y): functions of records that are
        frozen after convert(), so they are computed ONCE per (layer, input record) -- not per forward (70 launches of a few microseconds each)."""
        # (the key carries torch's version counters of everything the coefficients are computed from: load_state_dict / in-place edits of the records -- all rows
        # of the qrecord arena share one counter -- or of the BatchNorm tensors of a converted model invalidate it.  A hipGraph captured AFTER an eager forward
        # holds no finalize launch: re-capture after such an edit)
        key = (qx.data_ptr(), bool(fb), qx._version, l.qy._version) + tuple(t._version for t in (l.gamma, l.beta, l.rmean, l.rvar, l.bias) if t is not None)
        if getattr(l, "_cfin_key", None) == key:
            return
        if getattr(l, "_cfin_key", None) is not None and getattr(self, "_cfin_captured", False) and not getattr(self, "_cfin_warned", False):
            # a hipGraph captured earlier replays the kernels with the OLD coefficient rows only until this launch rewrites them in place (same buffer), but a graph
            # captured after an eager forward holds no finalize launch at all: say so once instead of leaving it to a comment
            import warnings
            warnings.warn("frostnet_amd: a FakeQuantize record / BatchNorm tensor of a converted model changed after a forward was captured into a hipGraph; "
                          "the coefficient rows are recomputed now -- re-capture the graph if it was captured after an eager forward", RuntimeWarning)
            self._cfin_warned = True
        if fb:
            call("frost_conv_finalize_converted_fb", ptr(qx), ptr(l.wscale), ptr(l.gamma), ptr(l.beta), ptr(l.rmean), ptr(l.rvar), l.cout, ptr(l.coef),
                 ptr(l.qy), stream())
        else:
            call("frost_conv_finalize_converted", ptr(qx), ptr(l.qw), ptr(l.gamma), ptr(l.beta), ptr(l.rmean), ptr(l.rvar), l.cout, ptr(l.coef),
                 ptr(l.qy), stream())
        if not torch.cuda.is_current_stream_capturing():
            l._cfin_key = key
        else:
            self._cfin_captured = True

    def add_converted(self, a, b, q):
        y = self.new_act(a.n, a.h, a.w, a.c, q)
        call("frost_add_qnnpack", ptr(a.buf), ptr(a.q), ptr(b.buf), ptr(b.q), a.numel, ptr(q), ptr(y.buf), stream())
        return y

    def head_converted(self, l, x, fb=False):
        n, c = x.n, x.c
        pooled = torch.empty(n, c, dtype=torch.int32, device=self.device)
        call("frost_avgpool_q", ptr(x.buf), n, x.h * x.w, c, ptr(pooled), stream())
        if fb:
            call("frost_conv_finalize_converted_fb", ptr(x.q), ptr(l.wscale), None, ptr(l.bias), None, None, l.cout, ptr(l.coef), ptr(l.qy), stream())
        else:
            call("frost_conv_finalize_converted", ptr(x.q), ptr(l.qw), None, ptr(l.bias), None, None, l.cout, ptr(l.coef), ptr(l.qy), stream())
        logits = torch.empty(n, l.cout, dtype=torch.float32, device=self.device)
        self.last_logit_idx = torch.empty(n, l.cout, dtype=torch.uint8, device=self.device)
        call("frost_classifier_q_fb" if fb else "frost_classifier_q", ptr(pooled), ptr(x.q), ptr(l.wq_pack), ptr(l.coef), n, c, l.cout, ptr(l.qy), ptr(logits),
             ptr(self.last_logit_idx), stream())
        return logits

    # ------------------------------------------------------------------------------------------ backward
    def backward(self, dlogits=None, out_grads=None, boundaries=None, on_bucket=None):
        """Replay the tape in reverse. dlogits: fp32 [n, nclass] gradient of the loss w.r.t. the logits.
        Parameter gradients are written (=, not +=) into l.w.grad / l.gamma.grad / l.beta.grad / l.bias.grad.
        boundaries / on_bucket (data parallel, frostnet_amd.parallel.SegmentedStep): `boundaries` maps id(layer) -> bucket index;
        once that layer's backward is done every gradient of the bucket is final: the side stream is joined, the bucket's weight
        gradients are finalized with one table launch and on_bucket(index) is called (it ends a hipGraph segment / starts the
        bucket's all-reduce) while the rest of the backward is still to run."""
        self._prepare_dwq()
        self._pending = []          # conv layers whose weight-gradient finalize is deferred to one table launch
        self._frozen_stashed = []   # frozen-BatchNorm layers whose S1 / S2 rows are set aside right now
        try:
            if self.grad_fp32:
                return self._backward_g32(dlogits, boundaries, on_bucket)
            return self._backward_bf16(dlogits, boundaries, on_bucket)
        except BaseException:
            for l in self._frozen_stashed:              # an aborted backward must not leave zeroed coefficient rows behind
                if getattr(l, "_s12", None) is not None:
                    l.coef[L.COEF_S1: L.COEF_S2 + 1].copy_(l._s12)
                    l._s12 = None
            raise

    def _backward_bf16(self, dlogits, boundaries, on_bucket):
        # Pointwise weight gradients run on a second stream: nothing downstream needs them until the finalize at the end of the
        # backward, and the short low-resolution kernels leave launch gaps and tails that an independent kernel can fill.
        # Off when per-layer gradients are awaited (data parallel) and while the per-kernel profiler times the main stream.
        self._side = None
        if _WG_STREAM and self.on_layer_grads is None and L.PROFILER is None and self.device.type == "cuda":
            if getattr(self, "_wg_stream", None) is None:
                self._wg_stream = torch.cuda.Stream(device=self.device, priority=_WG_PRIO)
                self._wg_extra = [torch.cuda.Stream(device=self.device, priority=_WG_PRIO) for _ in range(max(0, _WG_NSTREAMS - 1))]
            # several weight-gradient streams only in the single-GPU step: a bucketed backward joins ONE side stream per bucket (decided per backward, not
            # once at the first one -- an eager single-GPU backward followed by a bucketed one must not keep round-robining)
            self._wg_more = self._wg_extra if boundaries is None else []
            self._side = self._wg_stream
            self._side.wait_stream(torch.cuda.current_stream())          # after the dwq arena fill
            for st in self._wg_more:
                st.wait_stream(torch.cuda.current_stream())
            self._keep = []
            self._forks = 0
            self._deferred = []
        rtape = list(reversed(self.tape))
        for ti, entry in enumerate(rtape):
            kind = entry[0]
            if self.grad_mixed and self._mixed_entry(entry):          # a deep-stage node: the fp32-gradient kernels (csrc/frost_g32.hip), on the main stream
                if kind == "conv":
                    if self.on_layer_grads is not None:
                        self.on_layer_grads(entry[1])
                    if boundaries is not None and id(entry[1]) in boundaries:
                        self._close_bucket(boundaries[id(entry[1])], on_bucket)
                continue
            if kind == "head":
                _, l, x, pooled, raw, drop = entry
                g = torch.empty_like(raw)
                call("frost_mask_logits", ptr(dlogits.contiguous()), ptr(raw), ptr(l.qy), raw.numel(), ptr(g), stream())
                dwq = self._dwq_buf(l.cout * l.cin_g)
                gx, _ = self._grad_slot(x)
                dpool = torch.empty_like(pooled)
                self._ensure_grad(l)
                if self.grad_is_fp32(x):          # mixed mode: the pooled gradient spreads over the map in fp32 (the kernel's bf16 output goes to a scratch buffer)
                    scratch = torch.empty(x.numel + 64, dtype=torch.int16, device=self.device)
                    call("frost_head_bwd", ptr(g), ptr(pooled), ptr(l.wq_pack), ptr(l.qw), x.n, x.c, l.cout, x.h * x.w,
                         ptr(drop), ptr(dwq), ptr(l.bias.grad), ptr(scratch), ptr(dpool), ptr(l.wscale) if l.per_channel else None, stream())
                    call("frost_g32_pool_bwd", ptr(dpool), ptr(drop), x.n, x.h * x.w, x.c, ptr(gx), stream())
                else:
                    call("frost_head_bwd", ptr(g), ptr(pooled), ptr(l.wq_pack), ptr(l.qw), x.n, x.c, l.cout, x.h * x.w,
                         ptr(drop), ptr(dwq), ptr(l.bias.grad), ptr(gx), ptr(dpool), ptr(l.wscale) if l.per_channel else None, stream())
                call("frost_weight_grad_finalize", ptr(dwq), ptr(l.w), None, None, ptr(l.qw), ptr(l.coef), l.cout,
                     l.cin_g, 1, l.cpad, ptr(l.w.grad), None, None, 0, ptr(l.wscale), stream())
                if self.on_layer_grads is not None:
                    self.on_layer_grads(l)
                if boundaries is not None and id(l) in boundaries:
                    self._close_bucket(boundaries[id(l)], on_bucket)
            elif kind == "conv":
                _, l, x, y = entry
                self._conv_backward(l, x, y, rtape[ti + 1] if ti + 1 < len(rtape) else None)
                if self.on_layer_grads is not None:
                    self.on_layer_grads(l)
                if boundaries is not None and id(l) in boundaries:
                    self._close_bucket(boundaries[id(l)], on_bucket)
            elif kind == "cat":
                _, a, b, y = entry
                ga, fa = self._grad_slot(a)
                gb, fb = self._grad_slot(b)
                nxt = rtape[ti + 1] if ti + 1 < len(rtape) else None
                sq = nxt[1] if (nxt is not None and nxt[0] == "conv" and nxt[3] is a and nxt[2] is b) else None
                if (_SQ_BWD_CAT and sq is not None and sq.kind == "pw" and getattr(sq, "hswish", None) is None and not sq.per_channel
                        and L.load_library().frost_sq_bwd_cat_ok(b.c, a.c)):
                    # the next tape entry is the squeeze_conv that produced `a` from `b`: its reduce pass (S1 / S2) rides in the cat's backward launch
                    call("frost_sq_bwd_cat", ptr(b.buf), ptr(b.q), ptr(sq.wq_pack), ptr(sq.wsum), b.npix, b.c, a.c, ptr(sq.coef), ptr(sq.qy), int(sq.relu), ptr(y.q),
                         ptr(y.grad), ptr(ga), fa, ptr(gb), fb, stream(), prof=("cat_bwd", 5 * y.numel + b.numel))
                    a.bred_done = True
                else:
                    call("frost_cat_bwd", ptr(y.grad), ptr(a.buf), ptr(a.q), a.c, ptr(b.buf), ptr(b.q), b.c, a.npix, ptr(y.q),
                         ptr(ga), fa, ptr(gb), fb, stream(), prof=("cat_bwd", 5 * y.numel))
                y.grad = None
            elif kind == "hswish":
                _, x, y, lut = entry
                gx, acc = self._grad_slot(x)
                call("frost_hswish_bwd", ptr(y.grad), ptr(x.buf), x.numel, ptr(lut), ptr(gx), acc, stream(), prof=("hswish_bwd", 5 * x.numel))
                y.grad = None
            elif kind == "add":
                _, a, b, y = entry
                nxt = rtape[ti + 1] if ti + 1 < len(rtape) else None
                if (_ADD_BWD_FUSE and b.grad is None and getattr(b, "cint", None) is not None and nxt is not None and nxt[0] == "conv" and nxt[3] is b
                        and self._ew_backward(nxt[1], nxt[2])):
                    # the next tape entry is the reduce_conv that produced b, and its backward runs element-wise over its kept integer conv output: those two
                    # passes apply the add's STE window themselves (frost_pw_ew_add_bwd) -- nothing is launched here, b's gradient is never materialised
                    ga, fa = self._grad_slot(a)
                    b.add_bwd = (y.grad, a, y.q, ga, fa)
                    b.grad = y.grad                       # (placeholder: the fused passes read the add's gradient)
                else:
                    ga, fa = self._grad_slot(a)
                    gb, fb = self._grad_slot(b)
                    call("frost_add_bwd", ptr(y.grad), ptr(a.buf), ptr(a.q), ptr(b.buf), ptr(b.q), a.numel, ptr(y.q), ptr(ga),
                         fa, ptr(gb), fb, stream(), prof=("add_bwd", 8 * a.numel))
                y.grad = None
        self._flush_deferred_wgrads()
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._wg_stream)          # join: every weight gradient is accumulated
            for st in getattr(self, "_wg_more", []):
                torch.cuda.current_stream().wait_stream(st)
            self._side = None
        self._finalize_pending()
        self._keep = []
        self.tape = []

    def _mixed_entry(self, entry):
        """Mixed gradient mode: run one tape entry with the fp32-gradient kernels if its OUTPUT gradient is an fp32 one (grad_is_fp32); returns False for the others."""
        kind = entry[0]
        s = stream()
        if kind == "head":
            return False          # (the head's own GEMMs are shared by both modes; its input gradient follows grad_is_fp32(x) in the caller)
        if kind == "conv":
            _, l, x, y = entry
            if not self.grad_is_fp32(y):
                return False
            self._conv_backward_g32(l, x, y)
            return True
        if kind == "cat":
            _, a, b, y = entry
            if not self.grad_is_fp32(y):
                return False
            ga, fa = self._grad_slot(a)
            gb, fb = self._grad_slot(b)
            call("frost_g32_cat_bwd", ptr(y.grad), ptr(a.buf), ptr(a.q), a.c, ptr(b.buf), ptr(b.q), b.c, a.npix, ptr(y.q), ptr(ga), fa, ptr(gb), fb, s)
            y.grad = None
            return True
        if kind == "add":
            _, a, b, y = entry
            if not self.grad_is_fp32(y):
                return False
            ga, fa = self._grad_slot(a)
            gb, fb = self._grad_slot(b)
            call("frost_g32_add_bwd", ptr(y.grad), ptr(a.buf), ptr(a.q), ptr(b.buf), ptr(b.q), a.numel, ptr(y.q), ptr(ga), fa, ptr(gb), fb, s)
            y.grad = None
            return True
        return False

    # ------------------------------------------------------------------------------------------ fp32-gradient parity mode
    def _backward_g32(self, dlogits, boundaries, on_bucket):
        """The tape replayed with the plain fp32 kernels of csrc/frost_g32.hip (same formulas, fp32 storage, fp64 sums)."""
        self._side = None
        s = stream()
        for entry in reversed(self.tape):
            kind = entry[0]
            if kind == "head":
                _, l, x, pooled, raw, drop = entry
                g = torch.empty_like(raw)
                call("frost_mask_logits", ptr(dlogits.contiguous()), ptr(raw), ptr(l.qy), raw.numel(), ptr(g), s)
                dwq = self._dwq_buf(l.cout * l.cin_g)
                gx, _ = self._grad_slot(x)
                dpool = torch.empty_like(pooled)
                scratch = torch.empty(x.numel + 64, dtype=torch.int16, device=self.device)          # (the fp32 GEMMs of the head are shared; its bf16 gx is discarded)
                self._ensure_grad(l)
                call("frost_head_bwd", ptr(g), ptr(pooled), ptr(l.wq_pack), ptr(l.qw), x.n, x.c, l.cout, x.h * x.w, ptr(drop), ptr(dwq), ptr(l.bias.grad),
                     ptr(scratch), ptr(dpool), ptr(l.wscale) if l.per_channel else None, s)
                call("frost_g32_pool_bwd", ptr(dpool), ptr(drop), x.n, x.h * x.w, x.c, ptr(gx), s)
                call("frost_weight_grad_finalize", ptr(dwq), ptr(l.w), None, None, ptr(l.qw), ptr(l.coef), l.cout, l.cin_g, 1, l.cpad, ptr(l.w.grad), None, None, 0,
                     ptr(l.wscale), s)
            elif kind == "conv":
                _, l, x, y = entry
                self._conv_backward_g32(l, x, y)
            elif kind == "cat":
                _, a, b, y = entry
                ga, fa = self._grad_slot(a)
                gb, fb = self._grad_slot(b)
                call("frost_g32_cat_bwd", ptr(y.grad), ptr(a.buf), ptr(a.q), a.c, ptr(b.buf), ptr(b.q), b.c, a.npix, ptr(y.q), ptr(ga), fa, ptr(gb), fb, s)
                y.grad = None
            elif kind == "add":
                _, a, b, y = entry
                ga, fa = self._grad_slot(a)
                gb, fb = self._grad_slot(b)
                call("frost_g32_add_bwd", ptr(y.grad), ptr(a.buf), ptr(a.q), ptr(b.buf), ptr(b.q), a.numel, ptr(y.q), ptr(ga), fa, ptr(gb), fb, s)
                y.grad = None
            else:
                raise NotImplementedError(f"fp32-gradient mode: no plain kernel for the '{kind}' node")
            if kind in ("head", "conv"):
                l = entry[1]
                if self.on_layer_grads is not None:
                    self.on_layer_grads(l)
                if boundaries is not None and id(l) in boundaries:
                    self._close_bucket(boundaries[id(l)], on_bucket)
        self._finalize_pending()
        self.tape = []

    def _conv_backward_g32(self, l, x, y):
        self._ensure_grad(l)
        s = stream()
        kind = {"pw": 0, "dw": 1, "stem": 2}[l.kind]
        per = l.cin_g * l.kk
        if getattr(l, "_g32_qw", None) is None:
            l._g32_qw = torch.empty(l.cout * per + 64, dtype=torch.int8, device=self.device)
        qw = l._g32_qw
        call("frost_g32_wq", ptr(l.w), ptr(l.gamma), ptr(l.sigma), ptr(l.qw), ptr(l.wscale) if l.per_channel else None, l.cout, per, ptr(qw), s)
        cin_g = 1 if kind == 1 else l.cin_g
        geo = (kind, x.n, x.h, x.w, x.c, cin_g, l.cout, l.k, l.stride)
        if getattr(self, "_g32_scratch", None) is None:          # partial sums / partial tiles of the two-stage reductions (deterministic, fp64 second stage)
            self._g32_scratch = torch.empty(int(L.load_library().frost_g32_scratch_bytes()), dtype=torch.uint8, device=self.device)
        scr = self._g32_scratch
        dc = torch.empty(y.numel + 64, dtype=torch.float32, device=self.device)
        if kind != 1 and L.load_library().frost_g32_x_ok(kind, y.npix, x.c, cin_g, l.cout):
            # pointwise / stem: the reduce and dc passes recompute the integer conv output on the int8 MFMA -- no int32 tensor is written or re-read
            gx_ = (kind, x.n, x.h, x.w, x.c, cin_g, l.cout)
            call("frost_g32_reduce_x", ptr(x.buf), ptr(x.q), ptr(qw), *gx_, ptr(l.coef), ptr(l.qy), int(l.relu), ptr(y.grad), ptr(scr), s, prof=("g32_reduce", x.numel + 4 * y.numel))
            self._frozen_after_reduce(l)
            call("frost_g32_dc_x", ptr(x.buf), ptr(x.q), ptr(qw), *gx_, ptr(l.coef), ptr(l.qy), int(l.relu), ptr(y.grad), ptr(dc), s, prof=("g32_dc", x.numel + 8 * y.numel))
        else:
            acc = torch.empty(y.numel + 64, dtype=torch.int32, device=self.device)
            call("frost_g32_conv_acc", ptr(x.buf), ptr(x.q), ptr(qw), *geo, ptr(acc), s, prof=("g32_conv_acc", x.numel + 4 * y.numel))
            call("frost_g32_reduce", ptr(acc), y.npix, l.cout, ptr(l.coef), ptr(l.qy), int(l.relu), ptr(y.grad), ptr(scr), s, prof=("g32_reduce", 8 * y.numel))
            self._frozen_after_reduce(l)
            call("frost_g32_dc", ptr(acc), y.npix, l.cout, ptr(l.coef), ptr(l.qy), int(l.relu), ptr(y.grad), ptr(dc), s, prof=("g32_dc", 12 * y.numel))
        if getattr(self, "_dbg", False):
            self._last_dc = dc
        if x.needs_grad and not self.grad_is_fp32(x):
            # mixed mode, the one place where the two storage forms meet (an fp32-gradient layer whose input map is above the threshold): fp32 data gradient into a scratch
            # buffer, rounded to bf16 into x's slot (added to what is already there)
            tmp = torch.empty(x.numel + 64, dtype=torch.float32, device=self.device)
            call("frost_g32_dgrad", ptr(dc), ptr(qw), ptr(l.qw), ptr(l.wscale) if l.per_channel else None, *geo, ptr(tmp), 0, s, prof=("g32_dgrad", 4 * y.numel + 4 * x.numel))
            if x.grad is not None:
                tmp[: x.numel].add_(x.grad[: x.numel].view(torch.bfloat16).float())
            x.grad = torch.empty(x.numel + 64, dtype=torch.int16, device=self.device)
            x.grad[x.numel:].zero_()
            x.grad[: x.numel].copy_(tmp[: x.numel].to(torch.bfloat16).view(torch.int16))
        elif x.needs_grad:
            gx, accf = self._grad_slot(x)
            call("frost_g32_dgrad", ptr(dc), ptr(qw), ptr(l.qw), ptr(l.wscale) if l.per_channel else None, *geo, ptr(gx), accf, s, prof=("g32_dgrad", 4 * y.numel + 4 * x.numel))
        call("frost_g32_wgrad", ptr(dc), ptr(x.buf), ptr(x.q), *geo, ptr(l.dwq), ptr(scr), s, prof=("g32_wgrad", 4 * y.numel + x.numel))
        self._after_conv_backward(l, s)
        y.grad = None

    @staticmethod
    def _dwq_elems(numel):
        """Floats of a raw weight-gradient buffer for a layer of `numel` weights: layers of at most FROST_DWQ_SPREAD_MAX weights keep FROST_DWQ_NC copies at a stride of
        round_up(numel, 64) -- the persistent backward kernels spread their flush atomics over them, the parameter-gradient finalize adds them up (csrc/frost_common.h)."""
        return numel if numel > L.DWQ_SPREAD_MAX else L.DWQ_NC * round_up(numel, 64)

    def _dwq_buf(self, numel):
        """A raw weight-gradient buffer outside the arena (the classifier head): zeroed, sized by the same rule (its kernels write copy 0)."""
        t = torch.zeros(self._dwq_elems(numel), dtype=torch.float32, device=self.device) if numel <= L.DWQ_SPREAD_MAX \
            else torch.empty(numel, dtype=torch.float32, device=self.device)
        return t

    def _prepare_dwq(self):
        """One fp32 scratch arena for every layer's dL/d(fake-quantised weight) (the wgrad kernels accumulate with atomics):
        one fill per step instead of one per layer."""
        if getattr(self, "_dwq_arena", None) is None or self._dwq_layers != len(self.layers):
            sizes = [self._dwq_elems(l.w.numel()) + (self._dwq_elems(l.cout * 40) if l.kind == "stem" else 0) for l in self.layers]
            self._dwq_arena = torch.empty(sum(sizes), dtype=torch.float32, device=self.device)
            o = 0
            for l, sz in zip(self.layers, sizes):
                l.dwq = self._dwq_arena[o: o + l.w.numel()]                 # copy 0 (the views the kernels get are base pointers: the copies follow at dwq_stride)
                if l.kind == "stem":
                    o2 = o + self._dwq_elems(l.w.numel())
                    l.dwq_col = self._dwq_arena[o2: o2 + l.cout * 40]
                o += sz
            self._dwq_layers = len(self.layers)
            self._gtables = {}
        self._dwq_arena.zero_()

    def _flush_deferred_wgrads(self, fork=True):
        """Launch the collected pointwise weight gradients of the low-resolution stages on the side stream behind one fork (`fork=False`: the caller forks right after)."""
        todo, self._deferred = getattr(self, "_deferred", None) or [], []
        if not todo or self._side is None:
            return
        ev = torch.cuda.Event()                    # (fork=False: the caller forks again right after -- these launches still need their own dependency on the main stream)
        ev.record()
        self._side.wait_event(ev)
        sw = C.c_void_p(self._side.cuda_stream)
        for dc, x, l, ynum in todo:
            call("frost_pw_wgrad", ptr(dc), ptr(x.buf), ptr(x.q), x.npix, x.c, l.cout, ptr(l.dwq), sw, prof=("pw_wgrad", 2 * ynum + x.numel))

    def _close_bucket(self, index, on_bucket):
        """Every layer of gradient bucket `index` has run its backward: join the weight-gradient stream, finalize, notify."""
        self._flush_deferred_wgrads()
        if self._side is not None:
            for st in [self._wg_stream] + list(getattr(self, "_wg_more", [])):
                torch.cuda.current_stream().wait_stream(st)
            self._keep = []
        self._finalize_pending(slot=index)
        if on_bucket is not None:
            on_bucket(index)

    def _finalize_pending(self, slot=-1):
        """One table launch finalizes the weight gradients of every pending layer; the device tables are cached per slot
        (slot = gradient bucket, -1 = the whole network)."""
        if not self._pending:
            return
        for l in self._pending:
            self._ensure_grad(l)
        key = tuple((id(l), l.dwq.data_ptr(), l.w.grad.data_ptr(), l.gamma.grad.data_ptr(), l.beta.grad.data_ptr()) for l in self._pending)
        cache = self.__dict__.setdefault("_gtables", {})
        if slot not in cache or cache[slot][0] != key:
            arr = (L.FrostGDesc * len(self._pending))()
            for i, l in enumerate(self._pending):
                arr[i] = L.FrostGDesc(l.dwq.data_ptr(), l.w.data_ptr(), l.gamma.data_ptr(), l.sigma.data_ptr(), l.qw.data_ptr(),
                                      l.coef.data_ptr(), l.w.grad.data_ptr(), l.gamma.grad.data_ptr(), l.beta.grad.data_ptr(),
                                      l.cout, l.cin_g * l.kk, l.cpad, 0, l.wscale.data_ptr())
            cache[slot] = (key, L.struct_to_tensor(arr, self.device))
        for l in self._pending:
            self._frozen_restore(l)
        call("frost_weight_grad_finalize_table", ptr(cache[slot][1]), len(self._pending), stream())
        for l in self._pending:
            self._frozen_fix_dgamma(l)
        self._pending = []

    @staticmethod
    def _ensure_grad(l):
        for p in (l.w, l.gamma, l.beta, l.bias):
            if p is not None and p.grad is None:
                p.grad = torch.zeros_like(p)

    def _ew_backward(self, l, x):
        """True iff this layer's backward runs its reduce / dc passes element-wise over the kept integer conv output (the rule of _conv_backward)."""
        fused = l.kind in ("pw", "stem") and _PW_FUSE and bool(L.load_library().frost_pw_bwd_fused_ok(x.npix, x.c, l.cout)) and (l.kind == "stem" or x.h * x.w >= _PW_FUSE_MINMAP)
        return bool(_PW_KEEP and not fused and l.kind == "pw" and x.c > 256 and l.cout < x.c and not getattr(l, "frozen", False))

    def _conv_backward(self, l, x, y, nxt=None):
        self._ensure_grad(l)
        gout = y.grad
        s = stream()
        if _WG_DEFER and getattr(self, "_deferred", None) and x.h > _WG_DEFER:
            self._flush_deferred_wgrads()
        if _WG_JOIN and self._side is not None and getattr(self, "_forks", 0) >= _WG_JOIN:
            for st in [self._wg_stream] + list(getattr(self, "_wg_more", [])):      # short side branches: the weight gradients forked so far (on every side stream) must be done before this layer starts
                torch.cuda.current_stream().wait_stream(st)
            self._forks = 0
        fused = l.kind in ("pw", "stem") and _PW_FUSE and bool(L.load_library().frost_pw_bwd_fused_ok(x.npix, x.c, l.cout)) and (l.kind == "stem" or x.h * x.w >= _PW_FUSE_MINMAP)
        blk_dw = (l.kind == "dw" and _BLOCK_DWBWD and (x.h <= 7 or _BLOCK_DWBWD >= 2) and (x.grad is None or not x.needs_grad)
                  and bool(L.load_library().frost_block_dw_bwd_supported(x.h, x.w, l.k, l.stride, x.c)))      # dc stays in LDS there: no buffer
        if blk_dw and _DW_ONE_OVER_BLK and x.needs_grad and x.grad is None and L.load_library().frost_dw_bwd_fused_ok(x.h, x.w, x.c, l.k, l.stride):
            blk_dw = False          # A/B: the strip-streaming kernel instead of the image-resident one where both have an instance (FROST_DWB_MINW lets it take small maps)
        dw_one = (l.kind == "dw" and not blk_dw and _DW_BWD_ONE and x.needs_grad and x.grad is None
                  and bool(L.load_library().frost_dw_bwd_fused_ok(x.h, x.w, x.c, l.k, l.stride)))                 # dc stays in registers there: no buffer
        dc = None if (fused or blk_dw or dw_one) else torch.empty(y.numel + 64, dtype=torch.int16, device=self.device)
        if getattr(self, "_dbg", False):
            self._last_dc = dc
        if l.kind in ("pw", "stem"):
            dwq_final = l.dwq
            if l.kind == "stem":
                l.dwq, dwq_final = l.dwq_col, l.dwq
            args = (ptr(x.buf), ptr(x.q), ptr(l.wq_pack), ptr(l.wsum), ptr(l.wt_pack), ptr(l.qw), x.npix, x.c, l.cout)
            # wide-K layers (Cin > 256: x rows too long for k_pw's DMA tile): ONE recomputation of the integer conv output on the stand-alone GEMM kernel,
            # then the reduce and dc passes element-wise over it (N << K here: the int32 output is smaller than x) instead of two chunked k_pw passes
            cint = None
            if _PW_KEEP and not fused and l.kind == "pw" and x.c > 256 and l.cout < x.c:
                cint = getattr(y, "cint", None)          # kept by the training forward; otherwise (forward without statistics) one recomputation here
                if cint is None:
                    cint = torch.empty(y.numel + 64, dtype=torch.int32, device=self.device)
                    call("frost_pw_conv_int", ptr(x.buf), ptr(x.q), ptr(l.wq_pack), ptr(l.wsum), x.npix, x.c, l.cout, ptr(cint), s,
                         prof=("pw_bwd_reduce", x.numel + 4 * y.numel))
                y.cint = None
                ab = getattr(y, "add_bwd", None)
                if ab is not None:          # + skip_add's backward: g = the add's gradient inside the add's STE window; the residual branch's gradient leaves here
                    call("frost_pw_ew_add_bwd", ptr(cint), x.npix, l.cout, ptr(l.coef), ptr(l.qy), int(l.relu), 0, ptr(ab[0]), ptr(ab[1].buf), ptr(ab[1].q), ptr(y.buf),
                         ptr(ab[2]), ptr(ab[3]), ab[4], None, s, prof=("pw_bwd_reduce", 10 * y.numel))
                else:
                    call("frost_pw_ew", ptr(cint), x.npix, l.cout, ptr(l.coef), ptr(l.qy), int(l.relu), 0, ptr(gout), None, s,
                         prof=("pw_bwd_reduce", 6 * y.numel))
            else:
                # algorithmic bytes: x 1 B/el, gradients 2 B/el (bf16)
                pwc = (not fused) and l.kind == "pw" and bool(L.load_library().frost_pwc_bwd_ok(x.npix, x.c, l.cout))      # wide layers: chunked kernel, full-line gout / dc I/O
                if getattr(y, "bred_done", False):          # squeeze_conv: S1 / S2 were accumulated by the cat's backward launch (frost_sq_bwd_cat)
                    y.bred_done = False
                elif pwc and x.npix <= _PWC_RED_MAXPIX:       # (per 64-pixel tile a pair of float atomics per channel: above ~100 k pixels k_pw's fatter tiles win -- measured 98 vs 186 us at 28 x 28)
                    call("frost_pwc_conv_bwd", ptr(x.buf), ptr(x.q), ptr(l.wq_pack), ptr(l.wsum), x.npix, x.c, l.cout, 0, ptr(l.coef), ptr(l.qy), int(l.relu), ptr(gout), None, s,
                         prof=("pw_bwd_reduce", x.numel + 2 * y.numel))
                else:
                    call("frost_pw_conv_bwd", *args, 0, ptr(l.coef), ptr(l.qy), int(l.relu), ptr(gout), None, None, 0, s,
                         prof=("pw_bwd_reduce", x.numel + 2 * y.numel))
            self._frozen_after_reduce(l)
            if fused:
                # dc pass + data gradient + weight gradient in one kernel: the dc tile never leaves LDS (layers with Cout*Cin <= ~19 k)
                gx, acc = self._grad_slot(x) if x.needs_grad else (None, 0)
                call("frost_pw_conv_bwd_fused", *args, ptr(l.coef), ptr(l.qy), int(l.relu), ptr(gout), ptr(dc), ptr(gx), acc, ptr(l.dwq), s,
                     prof=("pw_bwd_fused", x.numel + 2 * y.numel + (2 * x.numel if x.needs_grad else 0)))
                if l.kind == "stem":
                    l.dwq = dwq_final
                    call("frost_stem_wgrad_remap", ptr(l.dwq_col), l.cout, l.cin_g, ptr(l.dwq), s)
                self._after_conv_backward(l, s)
                y.grad = None
                return
            if cint is not None and getattr(y, "add_bwd", None) is not None:
                ab = y.add_bwd
                call("frost_pw_ew_add_bwd", ptr(cint), x.npix, l.cout, ptr(l.coef), ptr(l.qy), int(l.relu), 1, ptr(ab[0]), ptr(ab[1].buf), ptr(ab[1].q), ptr(y.buf),
                     ptr(ab[2]), None, 0, ptr(dc), s, prof=("pw_bwd_dc", 10 * y.numel))
                y.add_bwd = None
            elif cint is not None:
                call("frost_pw_ew", ptr(cint), x.npix, l.cout, ptr(l.coef), ptr(l.qy), int(l.relu), 1, ptr(gout), ptr(dc), s,
                     prof=("pw_bwd_dc", 8 * y.numel))
            elif pwc:
                call("frost_pwc_conv_bwd", ptr(x.buf), ptr(x.q), ptr(l.wq_pack), ptr(l.wsum), x.npix, x.c, l.cout, 1, ptr(l.coef), ptr(l.qy), int(l.relu), ptr(gout), ptr(dc), s,
                     prof=("pw_bwd_dc", x.numel + 4 * y.numel))
            else:
                call("frost_pw_conv_bwd", *args, 1, ptr(l.coef), ptr(l.qy), int(l.relu), ptr(gout), ptr(dc), None, 0, s,
                     prof=("pw_bwd_dc", x.numel + 4 * y.numel))
            defer_wg = bool(_WG_DEFER and self._side is not None and (_WG_STREAM & 1) and l.kind == "pw" and x.h <= _WG_DEFER)
            if _WG_BATCH > 1 and self._side is not None and (_WG_STREAM & 1) and l.kind == "pw":
                defer_wg = True
                if len(self._deferred) + 1 >= _WG_BATCH:          # this layer completes a batch: its dc is done, flush the earlier ones now and this one below
                    self._wg_flush_after = True
            if self._side is not None and (_WG_STREAM & 1) and not defer_wg:      # fork right after the dc pass: the weight gradient runs beside dgrad and what follows
                self._flush_deferred_wgrads(fork=False)
                ev = torch.cuda.Event()
                ev.record()
                self._forks = getattr(self, "_forks", 0) + 1
                if getattr(self, "_wg_more", None):          # several weight-gradient streams: the next one takes this fork
                    allst = [self._wg_stream] + self._wg_more
                    self._side = allst[self._forks % len(allst)]
                self._side.wait_event(ev)
            if x.needs_grad:
                gx, acc = self._grad_slot(x)
                if l.kind == "pw" and L.load_library().frost_pw_dgrad_wide_ok(x.npix, x.c, l.cout):
                    # long dc rows (Cout > 128): a plain bf16 GEMM kernel of its own, dc fragments straight from memory, weight stages through LDS
                    call("frost_pw_dgrad_wide", ptr(dc), ptr(l.wt_pack), ptr(l.qw), x.npix, x.c, l.cout, ptr(gx), acc, s,
                         prof=("pw_dgrad", 2 * y.numel + 2 * x.numel))
                else:
                    call("frost_pw_conv_bwd", *args, 2, ptr(l.coef), ptr(l.qy), int(l.relu), ptr(gout), ptr(dc), ptr(gx), acc, s,
                         prof=("pw_dgrad", 2 * y.numel + 2 * x.numel))
            sw = s
            if self._side is not None and (_WG_STREAM & 1):      # dc (and x) stay referenced until the join
                self._keep.append((dc, x.buf))
                sw = C.c_void_p(self._side.cuda_stream)
            if defer_wg:
                self._deferred.append((dc, x, l, y.numel))
                if getattr(self, "_wg_flush_after", False):
                    self._wg_flush_after = False
                    self._flush_deferred_wgrads()
            else:
                call("frost_pw_wgrad", ptr(dc), ptr(x.buf), ptr(x.q), x.npix, x.c, l.cout, ptr(l.dwq), sw,
                     prof=("pw_wgrad", 2 * y.numel + x.numel))
            if l.kind == "stem":
                l.dwq = dwq_final
                call("frost_stem_wgrad_remap", ptr(l.dwq_col), l.cout, l.cin_g, ptr(l.dwq), sw)
        elif l.kind == "dw":
            args = (ptr(x.buf), ptr(x.q), ptr(l.wq_pack), ptr(l.wsum), ptr(l.qw), x.n, x.h, x.w, x.c, l.k, l.stride)
            blk = _BLOCK_DWBWD and (x.h <= 7 or _BLOCK_DWBWD >= 2) and L.load_library().frost_block_dw_bwd_supported(x.h, x.w, l.k, l.stride, x.c)
            blk_reduce = blk          # (the image-resident reduce pass stays even when the strip-streaming kernel takes the rest: FROST_DWB_OVER_BLK)
            blk = blk and (blk_dw or not dw_one)
            if blk_reduce and _BLOCK_DWBRED:
                call("frost_block_dw_bwd_reduce", ptr(x.buf), ptr(x.q), ptr(l.wq_pack), ptr(l.wsum), x.n, x.h, x.w, x.c, l.k, ptr(l.coef), ptr(l.qy), int(l.relu),
                     ptr(gout), s, prof=("blk_dw_bred", x.numel + 2 * y.numel))
            else:
                call("frost_dw_conv_bwd", *args, 0, ptr(l.coef), ptr(l.qy), int(l.relu), ptr(gout), None, s,
                     prof=("dw_bwd_reduce", x.numel + 2 * y.numel))
            self._frozen_after_reduce(l)
            gslot = self._grad_slot(x) if x.needs_grad else (None, 0)
            c1b = nxt[1] if (blk and not gslot[1] and _BLOCK_C1 and x.needs_grad and nxt is not None and nxt[0] == "conv" and nxt[3] is x) else None
            if (c1b is not None and c1b.kind == "pw" and c1b.k == 1 and not getattr(c1b, "frozen", False) and not c1b.per_channel and getattr(c1b, "hswish", None) is None
                    and L.load_library().frost_block_dw_bwd_c1_ok(x.h, x.w, l.k, l.stride, x.c, nxt[2].c)):
                # ... and the launch carries the reduce pass of the pointwise layer that produced x (conv1 of the bottleneck): conv1's integer output is recomputed per
                # (image, chunk) on the matrix cores and its S1 / S2 accumulate from the fp32 dx values, so conv1's backward starts at its dc pass (x.bred_done)
                x0 = nxt[2]
                call("frost_block_dw_bwd_c1", ptr(x.buf), ptr(x.q), ptr(l.wq_pack), ptr(l.wsum), ptr(l.qw), ptr(l.wscale) if l.per_channel else None, x.n, x.h, x.w, x.c,
                     l.k, ptr(l.coef), ptr(l.qy), int(l.relu), ptr(gout), ptr(gslot[0]), ptr(l.dwq),
                     ptr(x0.buf), ptr(x0.q), ptr(c1b.wq_pack), ptr(c1b.wsum), ptr(c1b.coef), x0.c, int(c1b.relu), s,
                     prof=("blk_dw_bwd", x.numel + 2 * y.numel + 2 * x.numel + x0.numel))
                x.bred_done = True
                self._after_conv_backward(l, s)
                y.grad = None
                return
            if blk and not gslot[1]:
                # 14x14 / 7x7 maps: the image's dc lives in an LDS plane; weight gradient and data gradient come from it (csrc/frost_block.hip)
                call("frost_block_dw_bwd", ptr(x.buf), ptr(x.q), ptr(l.wq_pack), ptr(l.wsum), ptr(l.qw), ptr(l.wscale) if l.per_channel else None, x.n, x.h, x.w, x.c,
                     l.k, ptr(l.coef), ptr(l.qy), int(l.relu), ptr(gout), ptr(gslot[0]) if x.needs_grad else None, ptr(l.dwq), s,
                     prof=("blk_dw_bwd", x.numel + 2 * y.numel + (2 * x.numel if x.needs_grad else 0)))
                self._after_conv_backward(l, s)
                y.grad = None
                return
            c1 = nxt[1] if (dw_one and _DW_C1 and nxt is not None and nxt[0] == "conv" and nxt[3] is x) else None
            if (c1 is not None and c1.kind == "pw" and not getattr(c1, "frozen", False) and not c1.per_channel and getattr(c1, "hswish", None) is None and not gslot[1]
                    and L.load_library().frost_dw_bwd_fused_c1_ok(x.h, x.w, x.c, l.k, l.stride, nxt[2].c)):
                # ... and the sweep carries the reduce pass of the pointwise layer that produced x (conv1 of the bottleneck): its S1 / S2 accumulate from the dx values as they
                # are formed, so conv1's backward starts at its dc pass (x.bred_done) and the gradient tensor is read once instead of twice
                x0 = nxt[2]
                call("frost_dw_bwd_fused_c1", ptr(x.buf), ptr(x.q), ptr(l.wq_pack), ptr(l.wsum), ptr(l.qw), ptr(l.wscale) if l.per_channel else None, x.n, x.h, x.w, x.c,
                     l.k, l.stride, ptr(l.coef), ptr(l.qy), int(l.relu), ptr(gout), ptr(gslot[0]), ptr(l.dwq),
                     ptr(x0.buf), ptr(x0.q), ptr(c1.wq_pack), ptr(c1.wsum), ptr(c1.coef), x0.c, int(c1.relu), s,
                     prof=("dw_bwd_one", x.numel + 2 * y.numel + 2 * x.numel + x0.numel))
                x.bred_done = True
                self._after_conv_backward(l, s)
                y.grad = None
                return
            if dw_one and not gslot[1]:
                # dc never leaves registers: the strip-streaming kernel computes weight gradient and data gradient from it in the same sweep
                call("frost_dw_bwd_fused", ptr(x.buf), ptr(x.q), ptr(l.wq_pack), ptr(l.wsum), ptr(l.qw), ptr(l.wscale) if l.per_channel else None, x.n, x.h, x.w, x.c,
                     l.k, l.stride, ptr(l.coef), ptr(l.qy), int(l.relu), ptr(gout), ptr(gslot[0]), ptr(l.dwq), s,
                     prof=("dw_bwd_one", x.numel + 2 * y.numel + 2 * x.numel))
                self._after_conv_backward(l, s)
                y.grad = None
                return
            # dc pass + weight gradient in one sweep where the kernel's register state allows it (the library applies the same
            # rule and would otherwise run the two kernels itself; calling them separately keeps the profiler tags per kernel)
            if _DW_FUSE and l.stride == 1 and (l.k == 3 or (l.k == 5 and y.w <= 8 and _DW_FUSE_K5)):
                call("frost_dw_conv_bwd_dc_wgrad", *args, ptr(l.coef), ptr(l.qy), int(l.relu), ptr(gout), ptr(dc), ptr(l.dwq), s,
                     prof=("dw_bwd_dc", 2 * x.numel + 4 * y.numel))
            else:
                call("frost_dw_conv_bwd", *args, 1, ptr(l.coef), ptr(l.qy), int(l.relu), ptr(gout), ptr(dc), s,
                     prof=("dw_bwd_dc", x.numel + 4 * y.numel))
                # (stays on the main stream by default: beside the pointwise weight gradients it slows everything down -- measured -5 %)
                sw = s
                if self._side is not None and ((_WG_STREAM & 2) or y.w <= _DW_WG_MAXW):
                    ev = torch.cuda.Event()
                    ev.record()
                    self._side.wait_event(ev)
                    self._keep.append((dc, x.buf))
                    sw = C.c_void_p(self._side.cuda_stream)
                call("frost_dw_wgrad", ptr(dc), ptr(x.buf), ptr(x.q), x.n, x.h, x.w, x.c, l.k, l.stride, ptr(l.dwq), sw,
                     prof=("dw_wgrad", 2 * y.numel + x.numel))
            if x.needs_grad:
                gx, acc = gslot
                call("frost_dw_dgrad", ptr(dc), ptr(l.wq_pack), ptr(l.qw), x.n, x.h, x.w, x.c, l.k, l.stride, ptr(gx), acc,
                     ptr(l.wscale) if l.per_channel else None, s,
                     prof=("dw_dgrad", 2 * y.numel + 2 * x.numel))
        self._after_conv_backward(l, s)
        y.grad = None

    def _frozen_after_reduce(self, l):
        """Backward of a layer whose BatchNorm is frozen (eval form, running statistics): y = c + (beta - rm * gamma / sigma_r), so dc = gy -- the
        batch-statistics expression dc = K1 (gy - S1/n - xhat S2/n) with K1 = 1 (what the eval-form finalize wrote) and without the S1 / S2 terms.  The dc
        kernels (every variant) read those terms from the coefficient rows, so the rows are set aside and zeroed between the reduce pass and the dc pass;
        _frozen_restore puts them back for the parameter-gradient finalize (dbeta = S1) and corrects dgamma."""
        if not getattr(l, "frozen", False):
            return
        # (the reduce passes spread S1 / S2 over four row copies -- rows 5 / 6 and 8 .. 13: folded into rows 5 / 6 here, so that the stash below holds the totals)
        extra = l.coef[L.COEF_ROWS_NAMED:]
        l.coef[L.COEF_S1].add_(extra[0::2].sum(0))
        l.coef[L.COEF_S2].add_(extra[1::2].sum(0))
        extra.zero_()
        rows = l.coef[L.COEF_S1: L.COEF_S2 + 1]
        if getattr(l, "_s12_buf", None) is None or l._s12_buf.shape != rows.shape:
            l._s12_buf = torch.empty_like(rows)          # persistent stash: no allocation per step (a captured step keeps its address)
        l._s12_buf.copy_(rows)
        l._s12 = l._s12_buf
        rows.zero_()
        self._frozen_stashed.append(l)

    def _frozen_restore(self, l):
        if getattr(l, "frozen", False) and getattr(l, "_s12", None) is not None:
            l.coef[L.COEF_S1: L.COEF_S2 + 1].copy_(l._s12)

    def _frozen_fix_dgamma(self, l):
        """Nothing to correct: the parameter-gradient finalize evaluates dgamma = S2 + (quantisation-residual sum) / sigma_r, which is the frozen-BatchNorm
        gradient as well as the batch-statistics one (csrc/frost_head.hip, k_wgrad_finalize); only the stash is dropped."""
        if getattr(l, "frozen", False):
            l._s12 = None

    def _after_conv_backward(self, l, s):
        if self.on_layer_grads is None:
            self._pending.append(l)        # single GPU: all layers finalized by one table launch at the end of the backward
        else:
            self._frozen_restore(l)
            call("frost_weight_grad_finalize", ptr(l.dwq), ptr(l.w), ptr(l.gamma), ptr(l.sigma), ptr(l.qw), ptr(l.coef), l.cout,
                 l.cin_g, l.kk, l.cpad, ptr(l.w.grad), ptr(l.gamma.grad), ptr(l.beta.grad), 0, ptr(l.wscale), s)
            self._frozen_fix_dgamma(l)


def grad_to_float(g, n, h, w, c):
    """gradient buffer (bf16 bits, or fp32 in the fp32-gradient mode) -> fp32 NCHW (tests)."""
    g = g[: n * h * w * c]
    g = g if g.dtype == torch.float32 else g.view(torch.bfloat16).float()
    return g.view(n, h, w, c).permute(0, 3, 1, 2)


def float_to_grad(t_nchw, fp32=False):
    """fp32 NCHW (any strides) -> NHWC gradient buffer with 64 elements of slack: bf16 bits, or fp32 (`fp32=True`: the fp32-gradient mode, Engine.grad_fp32).  One strided
    copy with the conversion folded in (+ a 64-element fill of the slack), whatever the layout of the incoming gradient -- the detector's map gradients arrive as
    strided slices of the loss's [N, P, 4] / [N, P, C] gradient tensors."""
    n, c, h, w = t_nchw.shape
    ne = n * h * w * c
    out = torch.empty(ne + 64, dtype=torch.float32 if fp32 else torch.bfloat16, device=t_nchw.device)
    out[ne:].zero_()
    out[:ne].view(n, h, w, c).copy_(t_nchw.permute(0, 2, 3, 1))
    return out if fp32 else out.view(torch.int16)
