"""The FLOAT (not fake-quantised) FrostNet on the HIP kernels: train-mode forward + backward and eval-mode forward.

This is the StatAssist warm-up phase of the reference's schedule (`Classification/train.py:149-165`: the float model is trained
for `FP_epoch` epochs with the same GradBoost optimizer, `is_warmup=True`, before `prepare_qat`), i.e. stock
`Conv2d(bias=False) -> BatchNorm2d -> ReLU` modules in the `frostnet.py:14-145` topology, cross-entropy on the logits.

Reference semantics kept: BatchNorm2d training semantics (batch mean / biased variance to normalise, running statistics updated
with momentum 0.1 and the unbiased variance, `num_batches_tracked += 1`), eval mode uses the running statistics, dropout before the
classifier, parameter gradients accumulated into `p.grad`.  Deviation from the reference (stated tolerance in
`tests/test_gpu_float.py`): by default activations and activation gradients are stored as NHWC bf16 (fp32 accumulation everywhere,
fp32 parameters / statistics / weight gradients).  `precision="fp32"` (FloatRunner argument, `model.float_precision`, or the environment
variable FROST_FLOAT_PRECISION) keeps them in fp32 like the reference and runs the products on the fp32 MFMA: the reference's FP32-train
end-to-end gate applies to that mode (tests/test_gpu_float.py::test_fp32_mode_*).

The module tree stays the owner of every Parameter and buffer; parameter gradients live in one flat fp32 arena (views assigned to
`p.grad`), the same contract as the fake-quant runner, so the multi-tensor GradBoost step and the data-parallel all-reduce work
on it unchanged and `statassist_qat_switch` keeps the optimizer state attached.  No CPU / torch-eager fallback: a missing library raises.
"""
import ctypes as C
import os

import torch

from . import _lib as L
from ._lib import call, ptr, stream

STATS, EMIT, BRED, BDC = 0, 1, 2, 3
DESC_BYTES = C.sizeof(L.FrostFDesc)


def round_up(a, b):
    return (a + b - 1) // b * b


class FAct:
    """NHWC activation, bf16 (int16 storage) or fp32; 64 elements of slack for the 16-byte tail loads."""
    __slots__ = ("buf", "n", "h", "w", "c", "src")

    def __init__(self, buf, n, h, w, c, src=None):
        self.buf, self.n, self.h, self.w, self.c = buf, n, h, w, c
        self.src = src          # (descriptor pointer, relu) of the producing layer when `buf` holds its CONV OUTPUT (the consumer applies BN + ReLU on load)

    @property
    def npix(self):
        return self.n * self.h * self.w

    def float(self):
        v = self.buf[: self.npix * self.c]
        v = v.view(torch.bfloat16).float() if v.dtype == torch.int16 else v
        return v.view(self.n, self.h, self.w, self.c).permute(0, 3, 1, 2)


_KEEP_CONV = os.environ.get("FROST_FLOAT_KEEP_CONV", "1") != "0"        # training keeps each layer's conv output for element-wise passes (A/B knob)
_WGRAD_SIDE = os.environ.get("FROST_FLOAT_WGRAD_SIDE", "1") != "0"      # weight gradients on a second stream (A/B knob)
_LAZY_EMIT = os.environ.get("FROST_FLOAT_LAZY_EMIT", "1") != "0"        # training: conv1's activation is never written -- the depthwise kernels apply BN + ReLU to its kept conv output on load (A/B knob)


def _act_code(mod):
    """Activation code of the float kernels' `relu` argument (csrc/frost_float.hip: 0 none, 1 ReLU, 2 hard-swish) for an activated layer module."""
    from .frostnet import ConvBNHswish
    return 2 if isinstance(mod, ConvBNHswish) else 1


class _FLayer:
    def __init__(self, name, seq, relu, dev, stem=False, fp32=False):
        conv, bn = seq[0], seq[1]
        if not isinstance(conv, torch.nn.Conv2d) or not isinstance(bn, torch.nn.BatchNorm2d):
            raise RuntimeError("the float device path expects the un-fused float model (Conv2d + BatchNorm2d per layer)")
        self.name, self.conv, self.bn, self.relu = name, conv, bn, relu
        self.cout, self.cin_g, self.k = conv.out_channels, conv.in_channels // conv.groups, conv.kernel_size[0]
        self.stride = conv.stride[0]
        self.kind = 2 if stem else (1 if conv.groups > 1 else 0)
        self.cpad = round_up(self.cout, 16)
        self.pack_t, self.kpad_t, self.fp32 = None, 0, fp32
        kq = 16 if fp32 else 32                 # elements per 64-byte K step of the MFMA A-fragment pack
        if self.kind == 1:
            self.kpad = 0
            self.pack = torch.zeros(self.k * self.k * self.cpad, dtype=torch.float32, device=dev)
        else:
            self.kpad = 64 if stem else round_up(self.cin_g, kq)
            self.pack = torch.zeros((self.cpad // 16) * (self.kpad // kq) * 1024, dtype=torch.uint8, device=dev)
            if self.kind == 0:
                self.kpad_t = round_up(self.cout, kq)
                self.pack_t = torch.zeros((round_up(self.cin_g, 16) // 16) * (self.kpad_t // kq) * 1024, dtype=torch.uint8, device=dev)
        self.stat = torch.zeros(8 * 4 * self.cpad, dtype=torch.float64, device=dev)       # FST_SLOTS copies of [sum, sum of squares, S1, S2]
        self.coef = torch.zeros(8 * self.cpad, dtype=torch.float32, device=dev)
        self.x = None          # saved input of the last recorded forward
        self.c = None          # kept conv output of the last recorded training forward (element-wise backward passes)
        self.desc_ptr = None   # device address of this layer's FrostFDesc

    def desc(self, gviews):
        bn = self.bn
        return L.FrostFDesc(self.conv.weight.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                            bn.running_var.data_ptr(), bn.num_batches_tracked.data_ptr(), self.pack.data_ptr(),
                            self.pack_t.data_ptr() if self.pack_t is not None else None, self.stat.data_ptr(), self.coef.data_ptr(),
                            gviews[id(bn.weight)].data_ptr(), gviews[id(bn.bias)].data_ptr(), self.cout, self.cin_g,
                            self.k * self.k, self.kind, self.cpad, self.kpad, self.kpad_t, 1 if self.fp32 else 0)


class _FloatFunction(torch.autograd.Function):
    """autograd boundary: image -> logits; backward() runs the hand-written backward pass, which writes the parameter gradients
    into the gradient arena (p.grad views) -- nothing is returned through autograd."""

    @staticmethod
    def forward(ctx, anchor, x, runner):
        ctx.runner = runner
        return runner._forward_impl(x, record=True)

    @staticmethod
    def backward(ctx, dlogits):
        ctx.runner._backward_impl(dlogits)
        return None, None, None


class _FloatFeatFunction(torch.autograd.Function):
    """image -> the four feature maps [x1, x2, x3, x5] (fp32 NCHW) of the features backbone (frostnet_features.py:342-352)."""

    @staticmethod
    def forward(ctx, anchor, x, runner):
        acts = runner._features_impl(x, record=True)
        ctx.runner, ctx.acts = runner, acts
        return tuple(a.float().contiguous() for a in acts)

    @staticmethod
    def backward(ctx, *grads):
        ctx.runner._backward_features(ctx.acts, grads)
        return None, None, None


class FloatRunner:
    """Binds a float FrostNet (classification model or features backbone) to the float HIP kernels."""

    def __init__(self, model, precision=None):
        L.load_library()
        self._set_precision(precision or getattr(model, "float_precision", None))
        params = list(model.parameters())
        if not params[0].is_cuda:
            raise RuntimeError("FloatRunner needs the model on the GPU (no CPU fallback on the product path)")
        if model._is_qat_prepared():
            raise RuntimeError("FloatRunner binds the float model; a QAT-prepared model runs through FrostRunner")
        self.model, self.device = model, params[0].device
        self._bind_params(params)
        self.layers = []
        self.stem = self._add("conv1", model.conv1.conv, _act_code(model.conv1), stem=True)
        self.blocks, self.stage_ends = [], []
        for lname in ("layer1", "layer2", "layer3", "layer4", "layer5"):
            for bi, blk in enumerate(getattr(model, lname)):
                self.blocks.append(self._bind_block(f"{lname}.{bi}", blk))
            self.stage_ends.append(len(self.blocks) - 1)
        self.last = self._add("last_layer", model.last_layer.conv, _act_code(model.last_layer)) if hasattr(model, "last_layer") else None
        self.fc = model.classifier[2] if hasattr(model, "classifier") else None
        self.drop_rate = float(model.classifier[1].p) if self.fc is not None else 0.0
        self._finish()
        self._stem_tmp = torch.zeros(self.stem.cout * 64, dtype=torch.float32, device=self.device)

    def _set_precision(self, precision):
        precision = precision or os.environ.get("FROST_FLOAT_PRECISION", "bf16")
        if precision not in ("bf16", "fp32"):
            raise ValueError("float path precision is 'bf16' or 'fp32'")
        self.precision, self.fp32 = precision, precision == "fp32"
        self._adt = torch.float32 if self.fp32 else torch.int16          # activation / activation-gradient storage
        sfx = "_f32" if self.fp32 else ""
        self._fn = {n: n + sfx for n in ("frost_float_pw", "frost_float_dw", "frost_float_ew", "frost_float_dw_dgrad", "frost_float_dw_wgrad", "frost_float_pw_wgrad", "frost_float_dw_src", "frost_float_dw_wgrad_src",
                                         "frost_float_grad_merge", "frost_float_avgpool", "frost_float_head_bwd")}
        self._fn["cat"] = "frost_float_cat_f32" if self.fp32 else "frost_infer_cat"
        self._fn["add"] = "frost_float_add_f32" if self.fp32 else "frost_infer_add"
        self._fn["im2col"] = "frost_float_stem_im2col_f32" if self.fp32 else "frost_infer_stem_im2col"

    def _store(self, dst, src_nhwc):
        """fp32 NHWC tensor -> the activation storage type."""
        dst.copy_(src_nhwc.reshape(-1) if self.fp32 else src_nhwc.to(torch.bfloat16).view(torch.int16).reshape(-1))

    @classmethod
    def for_block(cls, block, precision=None):
        """Bind a single float CascadePreExBottleneck (teacher-forced block tests: `block_step`)."""
        L.load_library()
        r = cls.__new__(cls)
        r._set_precision(precision)
        r.model, r.device = block, next(block.parameters()).device
        r._bind_params(list(block.parameters()))
        r.layers, r.stem, r.last, r.fc = [], None, None, None
        r.blocks = [r._bind_block("B", block)]
        r._finish()
        return r

    def _bind_params(self, params):
        self._params = params
        self.grad_arena = torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device=self.device)
        self._grad_views, self._gv, off = [], {}, 0
        for p in params:
            v = self.grad_arena[off: off + p.numel()].view_as(p)
            self._grad_views.append(v)
            self._gv[id(p)] = v
            off += p.numel()

    def _bind_block(self, pre, blk):
        ent = dict(blk=blk, squeeze=None, conv1=None)
        if blk.expand_ratio != 1:
            if blk.block_type == "CAS":
                ent["squeeze"] = self._add(pre + ".squeeze_conv", blk.squeeze_conv.conv, _act_code(blk.squeeze_conv))
            ent["conv1"] = self._add(pre + ".conv1", blk.conv1.conv, _act_code(blk.conv1))
        ent["conv2"] = self._add(pre + ".conv2", blk.conv2.conv, _act_code(blk.conv2))
        ent["reduce"] = self._add(pre + ".reduce_conv", blk.reduce_conv.conv, 0)
        return ent

    def _finish(self):
        arr = (L.FrostFDesc * len(self.layers))()
        for i, l in enumerate(self.layers):
            arr[i] = l.desc(self._gv)
        self._table = L.struct_to_tensor(arr, self.device)
        for i, l in enumerate(self.layers):
            l.desc_ptr = C.c_void_p(self._table.data_ptr() + i * DESC_BYTES)
        m = self.model
        self._sig = tuple(p.data_ptr() for p in m.parameters()) + tuple(b.data_ptr() for b in m.buffers())
        self.on_grads_ready = None      # data-parallel hook: called once when the backward has written every gradient

    def to_act(self, x):
        """fp32 (N,C,H,W) -> NHWC bf16 activation."""
        n, c, h, w = x.shape
        a = self._new(n, h, w, c)
        self._store(a.buf[: n * h * w * c], x.permute(0, 2, 3, 1).contiguous())
        return a

    def block_step(self, x, gy):
        """One teacher-forced train-mode forward + backward of the bound block: returns (y, dx) as fp32 NCHW; p.grad is written."""
        call("frost_float_weight_prep", ptr(self._table), len(self.layers), stream())
        y = self._block(self.blocks[0], self.to_act(x), True, True)
        yf = y.float().contiguous()
        self._grads_written = False              # a test entry: p.grad is overwritten, not accumulated
        self._begin_backward()
        dx = self._block_bwd(self.blocks[0], self.to_act(gy))
        self._end_backward()
        return yf, dx.float().contiguous()

    def _add(self, name, seq, relu, stem=False):
        l = _FLayer(name, seq, relu, self.device, stem, self.fp32)
        self.layers.append(l)
        return l

    def rng_state(self):
        """Dropout Philox stream {seed, draws}: the float warm-up shares the device generator of the fake-quant runner (runner.dropout_mask),
        so harness.save_checkpoint / load_checkpoint resume it the same way."""
        from .runner import rng_state
        return rng_state(self)

    def set_rng_state(self, state):
        from .runner import set_rng_state
        set_rng_state(self, state)

    def still_valid(self):
        m = self.model
        return self._sig == tuple(p.data_ptr() for p in m.parameters()) + tuple(b.data_ptr() for b in m.buffers())

    def enable_data_parallel(self, nbuckets=1, group=None):
        """Data-parallel warm-up: the same one exchange step as the fake-quant path (frostnet_amd.parallel.GradSync over the flat
        gradient arena); the float backward hands the whole arena over when it ends, `grad_sync.finish()` waits and averages."""
        from .parallel import GradSync
        offs, off = [], 0
        for p in self._params:
            offs.append(off)
            off += p.numel()
        self.grad_sync = GradSync(self.grad_arena, offs, nbuckets, group)
        self.on_grads_ready = lambda: self.grad_sync.ready(0)
        return self.grad_sync

    def bind_grads(self):
        for p, v in zip(self._params, self._grad_views):
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                p.grad = v

    def _new(self, n, h, w, c):
        return FAct(torch.empty(n * h * w * c + 64, dtype=self._adt, device=self.device), n, h, w, c)

    # ------------------------------------------------------------------------------------------ forward
    def _conv(self, l, a, training, record, out=None, ldy=None, lazy=False):
        """Conv -> BN -> [ReLU] (frostnet.py:14-60).  `out`/`ldy`: write into a slice of a wider buffer (the cat).
        lazy (training, kept conv output): no emit pass -- the returned activation carries the conv output and `src`, its only consumer (the depthwise layer)
        applies BN + ReLU on load (frost_float_dw_src / frost_float_dw_wgrad_src)."""
        if l.kind == 1:
            pad = (l.k - 1) // 2
            ho, wo = (a.h + 2 * pad - l.k) // l.stride + 1, (a.w + 2 * pad - l.k) // l.stride + 1
        else:
            ho, wo = a.h, a.w
        lazy = bool(lazy and out is None and training and _KEEP_CONV)
        y = self._new(a.n, ho, wo, l.cout) if (out is None and not lazy) else None
        npix_o = a.n * ho * wo
        dst, ld = (ptr(y.buf) if y is not None else None, l.cout) if out is None else (out, ldy)
        if training and _KEEP_CONV:
            # training: ONE convolution per layer.  The statistics pass stores the conv output c; y = [relu](c*scale + bias) is an element-wise pass
            # over c, and so are the backward's statistics and dc (frost_float_ew) -- c is kept until the layer's backward
            cbuf = torch.empty(npix_o * l.cout + 64, dtype=self._adt, device=self.device)
            if l.kind == 1:
                if a.src is not None:
                    call(self._fn["frost_float_dw_src"], l.desc_ptr, ptr(a.buf), a.src[0], a.src[1], a.n, a.h, a.w, a.c, l.k, l.stride, int(l.relu), STATS, ptr(cbuf), stream())
                else:
                    call(self._fn["frost_float_dw"], l.desc_ptr, ptr(a.buf), a.n, a.h, a.w, a.c, l.k, l.stride, int(l.relu), STATS, None, ptr(cbuf), stream())
            else:
                call(self._fn["frost_float_pw"], l.desc_ptr, ptr(a.buf), ptr(l.pack), npix_o, a.c, l.cout, int(l.relu), STATS, None, 0, ptr(cbuf), l.cout, stream())
            call("frost_float_bn_finalize", l.desc_ptr, l.cout, npix_o, stream())
            L.note_raw_write()          # running_mean / running_var change through raw pointers: the bf16-inference weight cache must not trust torch's version counters
            if lazy:
                y = FAct(cbuf, a.n, ho, wo, l.cout, src=(l.desc_ptr, int(l.relu)))
            else:
                call(self._fn["frost_float_ew"], l.desc_ptr, ptr(cbuf), npix_o, l.cout, int(l.relu), EMIT, None, 0, dst, ld, stream())
            if record:
                l.c = cbuf
        elif l.kind == 1:
            if training:
                call(self._fn["frost_float_dw"], l.desc_ptr, ptr(a.buf), a.n, a.h, a.w, a.c, l.k, l.stride, int(l.relu), STATS, None, None, stream())
                call("frost_float_bn_finalize", l.desc_ptr, l.cout, npix_o, stream())
                L.note_raw_write()
            call(self._fn["frost_float_dw"], l.desc_ptr, ptr(a.buf), a.n, a.h, a.w, a.c, l.k, l.stride, int(l.relu), EMIT, None, dst, stream())
        else:
            if training:
                call(self._fn["frost_float_pw"], l.desc_ptr, ptr(a.buf), ptr(l.pack), npix_o, a.c, l.cout, int(l.relu), STATS, None, 0, None, 0, stream())
                call("frost_float_bn_finalize", l.desc_ptr, l.cout, npix_o, stream())
                L.note_raw_write()
            call(self._fn["frost_float_pw"], l.desc_ptr, ptr(a.buf), ptr(l.pack), npix_o, a.c, l.cout, int(l.relu), EMIT, None, 0, dst, ld, stream())
        if record:
            l.x = a
        return y

    def _block(self, ent, a, training, record):
        """CascadePreExBottleneck.forward (frostnet.py:124-145)."""
        blk, inp = ent["blk"], a
        if ent["conv1"] is not None:
            if ent["squeeze"] is not None:
                cs = ent["squeeze"].cout
                cat = self._new(a.n, a.h, a.w, cs + a.c)
                sq = self._conv(ent["squeeze"], a, training, record)
                call(self._fn["cat"], ptr(sq.buf), cs, ptr(a.buf), a.c, a.npix, ptr(cat.buf), stream())     # cat([squeezed, x], 1)
                a = cat
            c2 = ent["conv2"]
            a = self._conv(ent["conv1"], a, training, record,
                           lazy=bool(training and _KEEP_CONV and _LAZY_EMIT and c2.kind == 1 and c2.k in (3, 5) and c2.stride in (1, 2)))
        a = self._conv(ent["conv2"], a, training, record)
        a = self._conv(ent["reduce"], a, training, record)
        if not blk.reduction:
            out = self._new(a.n, a.h, a.w, a.c)
            call(self._fn["add"], ptr(inp.buf), ptr(a.buf), a.npix * a.c, ptr(out.buf), stream())
            a = out
        if record:
            ent["inp"], ent["out"] = inp, a
        return a

    def _trunk(self, x, training, record):
        if x.dim() != 4 or x.shape[1] != 3 or not x.is_cuda:
            raise ValueError("expected an (N,3,H,W) tensor on the model's device")
        if x.numel() == 0:
            raise ValueError("empty batch")
        if x.dtype != torch.float32:
            x = x.float()
        n, _, h, w = x.shape
        if training:
            for l in self.layers:
                if not l.bn.training:
                    raise NotImplementedError(f"{l.name}: BatchNorm in eval mode inside a training forward of the FLOAT model (_freeze_stages): the float HIP "
                                              "backward implements the batch-statistics gradient only; the QAT-prepared model supports frozen BatchNorm")
        call("frost_float_weight_prep", ptr(self._table), len(self.layers), stream())
        if not training:
            call("frost_float_bn_eval", ptr(self._table), len(self.layers), stream())
        ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
        col = self._new(n, ho, wo, 64)
        call(self._fn["im2col"], ptr(x), n, h, w, x.stride(0), x.stride(1), x.stride(2), x.stride(3), ptr(col.buf), stream())
        a = self._conv(self.stem, col, training, record)
        outs = []
        for ent in self.blocks:
            a = self._block(ent, a, training, record)
            outs.append(a)
        return a, outs

    def forward(self, x):
        if self.model.training and torch.is_grad_enabled():
            return _FloatFunction.apply(self._params[0], x, self)
        return self._forward_impl(x, record=False)

    def _forward_impl(self, x, record):
        training = self.model.training
        a, _ = self._trunk(x, training, record)
        a = self._conv(self.last, a, training, record)
        drop = None
        if training and self.drop_rate > 0.0:
            keep = 1.0 - self.drop_rate
            from .runner import dropout_mask          # the same device Philox stream as the fake-quant runner (seed from torch + rank, resumable)
            drop = dropout_mask(self, a.n * a.c, keep).view(a.n, a.c)
        pooled = torch.empty(a.n, a.c, dtype=torch.float32, device=self.device)
        call(self._fn["frost_float_avgpool"], ptr(a.buf), a.n, a.h * a.w, a.c, ptr(drop), ptr(pooled), stream())
        nclass = self.fc.out_channels
        logits = torch.empty(a.n, nclass, dtype=torch.float32, device=self.device)
        call("frost_linear_f32", ptr(pooled), ptr(self.fc.weight), ptr(self.fc.bias), a.n, a.c, nclass, ptr(logits), stream())
        if record:
            self._head = (a, pooled, drop)
        return logits

    def forward_features(self, x):
        if self.model.training and torch.is_grad_enabled():
            return list(_FloatFeatFunction.apply(self._params[0], x, self))
        return [a.float().contiguous() for a in self._features_impl(x, record=False)]

    def _features_impl(self, x, record):
        _, outs = self._trunk(x, self.model.training, record)
        e = self.stage_ends
        return [outs[e[0]], outs[e[1]], outs[e[2]], outs[e[4]]]          # x4 is skipped (frostnet_features.py:350)

    # ------------------------------------------------------------------------------------------ backward
    def _conv_bwd(self, l, gy, ldg, need_dx):
        """gy: device pointer (c_void_p) of the bf16 output gradient, rows of ldg elements.  Returns dx (FAct) or None."""
        a = l.x
        if l.kind == 1:
            pad = (l.k - 1) // 2
            ho, wo = (a.h + 2 * pad - l.k) // l.stride + 1, (a.w + 2 * pad - l.k) // l.stride + 1
        else:
            ho, wo = a.h, a.w
        npix_o = a.n * ho * wo
        dc = torch.empty(npix_o * l.cout + 64, dtype=self._adt, device=self.device)
        gw = self._gv[id(l.conv.weight)]
        dx = None
        kept = getattr(l, "c", None)
        if kept is not None:          # the conv output of the forward is at hand: statistics and dc are element-wise
            call(self._fn["frost_float_ew"], l.desc_ptr, ptr(kept), npix_o, l.cout, int(l.relu), BRED, gy, ldg, None, 0, stream())
            call("frost_float_bwd_finalize", l.desc_ptr, l.cout, npix_o, stream())
            call(self._fn["frost_float_ew"], l.desc_ptr, ptr(kept), npix_o, l.cout, int(l.relu), BDC, gy, ldg, ptr(dc), l.cout, stream())
            l.c = None
        if l.kind == 1:
            if kept is None:
                if ldg != l.cout:
                    raise RuntimeError("depthwise gradients are dense")
                call(self._fn["frost_float_dw"], l.desc_ptr, ptr(a.buf), a.n, a.h, a.w, a.c, l.k, l.stride, int(l.relu), BRED, gy, None, stream())
                call("frost_float_bwd_finalize", l.desc_ptr, l.cout, npix_o, stream())
                call(self._fn["frost_float_dw"], l.desc_ptr, ptr(a.buf), a.n, a.h, a.w, a.c, l.k, l.stride, int(l.relu), BDC, gy, ptr(dc), stream())
            if need_dx:
                dx = self._new(a.n, a.h, a.w, a.c)
                call(self._fn["frost_float_dw_dgrad"], l.desc_ptr, ptr(dc), a.n, a.h, a.w, a.c, l.k, l.stride, ptr(dx.buf), stream())
            if a.src is not None:
                self._on_side(lambda: call(self._fn["frost_float_dw_wgrad_src"], ptr(dc), ptr(a.buf), a.src[0], a.src[1], a.n, a.h, a.w, a.c, l.k, l.stride, ptr(gw), stream()), dc, a)
            else:
                self._on_side(lambda: call(self._fn["frost_float_dw_wgrad"], ptr(dc), ptr(a.buf), a.n, a.h, a.w, a.c, l.k, l.stride, ptr(gw), stream()), dc, a)
        else:
            if kept is None:
                call(self._fn["frost_float_pw"], l.desc_ptr, ptr(a.buf), ptr(l.pack), npix_o, a.c, l.cout, int(l.relu), BRED, gy, ldg, None, 0, stream())
                call("frost_float_bwd_finalize", l.desc_ptr, l.cout, npix_o, stream())
                call(self._fn["frost_float_pw"], l.desc_ptr, ptr(a.buf), ptr(l.pack), npix_o, a.c, l.cout, int(l.relu), BDC, gy, ldg, ptr(dc), l.cout, stream())
            if need_dx:
                dx = self._new(a.n, a.h, a.w, a.c)
                # data gradient = a plain bf16 GEMM with the transposed pack: the tuned pointwise skeleton (DMA double-buffered tiles,
                # resident weights, LDS-staged output) that also serves the fake-quant dgrad and the bf16 inference layers
                if self.fp32:     # the same GEMM on the fp32 MFMA (mode 4 of the float pointwise kernel)
                    call("frost_float_pw_f32", None, ptr(dc), ptr(l.pack_t), npix_o, l.cout, a.c, 0, 4, None, 0, ptr(dx.buf), a.c, stream())
                elif l.kind == 0 and L.load_library().frost_pw_dgrad_wide_ok(npix_o, a.c, l.cout):
                    call("frost_pw_dgrad_wide", ptr(dc), ptr(l.pack_t), None, npix_o, a.c, l.cout, ptr(dx.buf), 0, stream())      # the stand-alone bf16 GEMM (frost_wgrad.hip)
                else:
                    call("frost_infer_pw", ptr(dc), ptr(l.pack_t), None, npix_o, l.cout, a.c, 0, ptr(dx.buf), stream())
            if l.kind == 2:
                self._stem_tmp.zero_()
                call(self._fn["frost_float_pw_wgrad"], ptr(dc), ptr(a.buf), npix_o, 64, 64, l.cout, ptr(self._stem_tmp), 64, stream())
                call("frost_float_stem_wscatter", ptr(self._stem_tmp), l.cout, ptr(gw), stream())
            else:
                self._on_side(lambda: call(self._fn["frost_float_pw_wgrad"], ptr(dc), ptr(a.buf), npix_o, a.c, a.c, l.cout, ptr(gw), a.c, stream()), dc, a)
        l.x = None
        return dx

    def _block_bwd(self, ent, g, need_dx=True):
        """g: FAct gradient of the block output.  Returns the gradient of the block input."""
        blk, inp = ent["blk"], ent["inp"]
        res = g if not blk.reduction else None            # add: the gradient reaches both branches unchanged
        d = self._conv_bwd(ent["reduce"], ptr(g.buf), g.c, True)
        d = self._conv_bwd(ent["conv2"], ptr(d.buf), d.c, ent["conv1"] is not None or need_dx or res is not None)
        cat, cs, sq = None, 0, None
        if ent["conv1"] is not None:
            d = self._conv_bwd(ent["conv1"], ptr(d.buf), d.c, True)
            if ent["squeeze"] is not None:
                cs = ent["squeeze"].cout
                cat = d                                       # gradient of cat([squeezed, x]): columns [0, cs) -> squeeze, [cs, ..) -> x
                sq = self._conv_bwd(ent["squeeze"], ptr(cat.buf), cat.c, True)
                d = None
        ent["inp"] = ent["out"] = None
        if res is None and cat is None:
            return d
        out = self._new(inp.n, inp.h, inp.w, inp.c)
        direct = d if cat is None else sq                     # dense gradient w.r.t. the block input from the conv chain
        call(self._fn["frost_float_grad_merge"], ptr(res.buf) if res is not None else None, ptr(cat.buf) if cat is not None else None, cs,
             cat.c if cat is not None else 8, ptr(direct.buf) if direct is not None else None, inp.npix, inp.c, ptr(out.buf), stream())
        return out

    def _begin_backward(self):
        """torch semantics: backward() accumulates into p.grad until zero_grad().  Gradients the caller has not cleared are set aside (one copy of
        the arena) and added back by _end_backward; in the usual loop (zero_grad() -> p.grad is None) nothing is copied."""
        prev = None
        live = [getattr(self, "_grads_written", False) and p.grad is not None and p.grad.data_ptr() == v.data_ptr()
                for p, v in zip(self._params, self._grad_views)]
        if any(live):
            if getattr(self, "grad_sync", None) is not None:
                raise RuntimeError("gradient accumulation over several backward passes is not supported together with the data-parallel exchange")
            prev = self.grad_arena.clone()
            for ok, v in zip(live, self._grad_views):
                if not ok:
                    prev[v.storage_offset(): v.storage_offset() + v.numel()].zero_()
        self.bind_grads()
        self.grad_arena.zero_()
        self._carry = prev
        self._keep = []

    def _on_side(self, launch, *keep):
        """Weight gradients are off the backward's critical path (nothing reads them before the optimizer): they run on a second stream under
        the next layers' passes, which at the 14x14 / 7x7 stages leave most of the GPU idle.  `keep`: the buffers the launch reads -- held until
        the join in _end_backward, so the caching allocator cannot hand them to the main stream while the side stream still reads them."""
        if not _WGRAD_SIDE:
            return launch()
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.device)
        self._side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._side):
            launch()
        self._keep.extend(keep)

    def _end_backward(self):
        if getattr(self, "_side", None) is not None and self._keep:
            torch.cuda.current_stream().wait_stream(self._side)
        self._keep = []
        if getattr(self, "_carry", None) is not None:
            self.grad_arena.add_(self._carry)
        self._carry = None
        self._grads_written = True

    def _trunk_bwd(self, g, taps=None):
        for i in range(len(self.blocks) - 1, -1, -1):
            if taps is not None and i in taps:
                t = taps[i]
                if g is None:
                    g = t
                else:
                    s = self._new(g.n, g.h, g.w, g.c)
                    call(self._fn["add"], ptr(g.buf), ptr(t.buf), g.npix * g.c, ptr(s.buf), stream())
                    g = s
            if g is None:
                continue
            g = self._block_bwd(self.blocks[i], g)
        self._conv_bwd(self.stem, ptr(g.buf), g.c, False)
        self._end_backward()
        if self.on_grads_ready is not None:
            self.on_grads_ready()

    def _backward_impl(self, dlogits):
        self._begin_backward()
        a, pooled, drop = self._head
        self._head = None
        dl = dlogits.contiguous().float()
        n, nclass = dl.shape
        g = self._new(a.n, a.h, a.w, a.c)
        scratch = torch.empty(n, a.c, dtype=torch.float32, device=self.device)
        call(self._fn["frost_float_head_bwd"], ptr(dl), ptr(pooled), ptr(self.fc.weight), n, a.c, nclass, a.h * a.w, ptr(drop),
             ptr(self._gv[id(self.fc.weight)]), ptr(self._gv[id(self.fc.bias)]), ptr(g.buf), ptr(scratch), stream())
        g = self._conv_bwd(self.last, ptr(g.buf), g.c, True)
        self._trunk_bwd(g)

    def _backward_features(self, acts, grads):
        self._begin_backward()
        e = self.stage_ends
        taps = {}
        for bi, a, gr in zip((e[0], e[1], e[2], e[4]), acts, grads):
            if gr is None:
                continue
            t = self._new(a.n, a.h, a.w, a.c)
            self._store(t.buf[: a.npix * a.c], gr.float().permute(0, 2, 3, 1).contiguous())
            taps[bi] = t
        # blocks after the last tap (x4 -> x5 are all used; nothing is dead in the reference's backbone)
        self._trunk_bwd(None, taps)
