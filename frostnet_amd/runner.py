"""Binds a QAT-prepared FrostNet module tree to the HIP engine.

The module tree (stock torch / torch.ao modules, reference state_dict layout -- SURVEY.md Appendix C) stays the owner
of every Parameter; its observer / fake-quant buffers (`scale`, `zero_point`, `min_val`, `max_val`) are re-pointed at
rows of the engine's qrecord arena, so `state_dict()` / `load_state_dict()` keep working while the kernels update
those scalars on-device.  Parameter gradients live in ONE flat fp32 arena (views assigned to `p.grad`), which is what
the multi-tensor optimizer and the data-parallel all-reduce operate on.
"""
import os

import torch

from . import _lib as L
from .engine import ConvLayer, Engine, QArena


_PREP_SPLIT = os.environ.get("FROST_PREP_SPLIT", "0") != "0"    # ... in two parts: the high-resolution stages' layers first, the rest joined in front of layer3 (A/B switch; measured +-0 inside the captured step, profiles/r06_prologue_ab.txt: off)
_STEM_COL_EARLY = os.environ.get("FROST_STEM_COL_EARLY", "1") != "0"      # the stem's im2col in front of the join with the weight-preparation stream (A/B)
_PREP_SIDE = os.environ.get("FROST_PREP_SIDE", "1") != "0"      # training: the per-step weight preparation on a second stream beside the QuantStub passes

class _QATFunction(torch.autograd.Function):
    """autograd boundary: image -> logits.  backward() runs the hand-written backward pass, which writes the parameter
    gradients straight into the gradient arena (p.grad views) -- nothing is returned through autograd."""

    @staticmethod
    def forward(ctx, anchor, x, runner):
        ctx.runner = runner
        out = runner._forward_impl(x, record=True)
        ctx.gen = runner._new_generation()
        return out

    @staticmethod
    def backward(ctx, dlogits):
        ctx.runner._check_generation(ctx.gen)
        ctx.runner._backward_impl(dlogits)
        return None, None, None


class _QATFeatFunction(torch.autograd.Function):
    """image -> the four fake-quantised feature maps [x1, x2, x3, x5] (dequantised fp32 NCHW); DeQuantStub is identity."""

    @staticmethod
    def forward(ctx, anchor, x, runner):
        acts = runner._features_impl(x, record=True)
        ctx.runner, ctx.acts = runner, acts
        ctx.gen = runner._new_generation()
        return tuple(a.dequant().contiguous() for a in acts)

    @staticmethod
    def backward(ctx, *grads):
        from .engine import float_to_grad
        ctx.runner._check_generation(ctx.gen)
        for a, g in zip(ctx.acts, grads):     # tap gradients first; later layers' dgrads accumulate on top
            if g is None:
                g = torch.zeros(a.n, a.c, a.h, a.w, device=a.buf.device)
            a.grad = float_to_grad(g, fp32=ctx.runner.E.grad_is_fp32(a))
        ctx.runner._backward_impl(None)
        return None, None, None


def dropout_seed():
    """Philox key of the dropout masks: torch's seed (torch.manual_seed is honoured, Classification/train.py:38-39) mixed with the data-parallel
    rank, so that the ranks of one job draw different masks for their shards."""
    import os
    rank = 0
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        rank = torch.distributed.get_rank()
    else:
        rank = int(os.environ.get("RANK", "0"))
    return (torch.initial_seed() + 0x9E3779B97F4A7C15 * (rank + 1)) & 0xFFFFFFFFFFFFFFFF


def dropout_mask(runner, n, keep):
    """nn.Dropout's mask (frostnet.py:297) for `n` pooled features: {0, 1/keep} floats drawn on the device (Philox4x32-10, counter = (index, draw)).
    The draw counter lives in device memory (a captured hipGraph draws a fresh mask per replay) and, with the seed, is part of the runner's
    `rng_state()` so that a resumed run continues the stream.  Shared by the fake-quant and the float runner."""
    if getattr(runner, "_drop_ctr", None) is None:
        runner._drop_ctr = torch.zeros(2, dtype=torch.int64, device=runner.device)        # {draw, arrival ticket}
        runner._drop_seed = dropout_seed()
    out = torch.empty(n, dtype=torch.float32, device=runner.device)
    if torch.cuda.is_current_stream_capturing():
        runner._drop_captured = True
    L.call("frost_dropout_mask", L.ptr(runner._drop_ctr), runner._drop_seed, n, keep, L.ptr(out), L.stream())
    return out


def rng_state(runner):
    """{seed, draws} of the dropout stream (checkpointed by harness.save_checkpoint beside the optimizer's Philox offset = its step count)."""
    if getattr(runner, "_drop_ctr", None) is None:
        return {"seed": dropout_seed(), "draws": 0}
    return {"seed": int(runner._drop_seed), "draws": int(runner._drop_ctr[0].item())}


def set_rng_state(runner, state, discard_captured=False):
    """Restore the dropout stream.  The draw counter is written IN PLACE: a hipGraph captured earlier holds the counter's address and keeps drawing from
    the restored position (a replaced tensor would leave the graph advancing the old one).  The seed is a launch argument baked into a captured graph:
    restoring a DIFFERENT seed while such a graph exists raises before anything is changed (no partial restore); `discard_captured=True` says the
    caller drops that graph and re-captures the step (the mark is cleared; the next capture bakes the new seed)."""
    seed = int(state["seed"]) & 0xFFFFFFFFFFFFFFFF
    if discard_captured:
        runner._drop_captured = False
    if getattr(runner, "_drop_seed", None) is not None and int(runner._drop_seed) != seed and getattr(runner, "_drop_captured", False):
        raise RuntimeError("set_rng_state: the dropout seed is a launch argument baked into the captured hipGraph; drop that graph, call "
                           "set_rng_state(state, discard_captured=True) and re-capture the step to restore a different seed")
    new = torch.tensor([int(state["draws"]), 0], dtype=torch.int64, device=runner.device)
    if getattr(runner, "_drop_ctr", None) is None:
        runner._drop_ctr = new
    else:
        runner._drop_ctr.copy_(new)
    runner._drop_seed = seed


def _is_fused_fq(fq):
    return type(fq).__name__ == "FusedMovingAvgObsFakeQuantize"


class FrostRunner:
    def __init__(self, model):
        L.load_library()   # raises if the HIP library is missing: no CPU fallback on the device path
        self.model = model
        self.flags_dirty = True      # host summary of the per-site enable flags: re-read before the next forward (see _observe_hint)
        if not model._is_qat_prepared():
            raise NotImplementedError(
                "FrostRunner binds the fake-quantised (QAT-prepared) FrostNet; the float model runs through "
                "frostnet_amd.float_train.FloatRunner (model.hip_runner() picks the right one).")
        params = list(model.parameters())
        if not params:
            raise RuntimeError("this module has no parameters of its own (an nn.DataParallel replica?): the HIP path is one process per "
                               "GPU -- see frostnet_amd.parallel / `bench.py --gpus N`")
        self.device = params[0].device
        if self.device.type != "cuda":
            raise RuntimeError(f"FrostRunner needs the model on the HIP device (parameters are on {self.device})")
        self.E = Engine(self.device)
        self._sig = tuple(p.data_ptr() for p in params) + tuple(b.data_ptr() for b in model.buffers())
        self._build()

    @classmethod
    def for_block(cls, block):
        """Bind a single QAT-prepared CascadePreExBottleneck (teacher-forced block tests)."""
        L.load_library()
        r = cls.__new__(cls)
        r.model, r.device = block, next(block.parameters()).device
        r.E, r.qa, r.rule127 = Engine(r.device), QArena(32, r.device), False
        r.block = r._bind_block("B", block)
        r.E.rule127 = 1 if r.rule127 else 0
        r._params = list(block.parameters())
        r.grad_arena = torch.zeros(sum(p.numel() for p in r._params), dtype=torch.float32, device=r.device)
        r._grad_views, off = [], 0
        for p in r._params:
            r._grad_views.append(r.grad_arena[off: off + p.numel()].view_as(p))
            off += p.numel()
        return r

    def read_flags(self):
        """ONE device -> host read of the enable-flag columns of the qrecord arena (the torch `observer_enabled` / `fake_quant_enabled` buffers are
        views into it): refuses a disabled fake-quantizer, returns True iff any site's observer is enabled."""
        flags = self.qa.t.view(torch.int32)[: self.qa.next][:, [L.Q_OBS_EN, L.Q_FQ_EN]].cpu()
        if bool((flags[:, 1] == 0).any()):
            raise NotImplementedError("fake_quant_enabled == 0 (torch.quantization.disable_fake_quant): the int8 engine has no float "
                                      "activation mode; run the float model (FloatRunner) instead")
        return bool((flags[:, 0] != 0).any())

    @property
    def observe(self):
        """True iff any site's observer is enabled (reads the device flags now)."""
        self._obs_cached = self.read_flags()
        self.flags_dirty = False
        return self._obs_cached

    def _observe_hint(self, training):
        """What the host passes as `observe`: 1 = run the statistics passes and let every site's own device flag decide.  Training needs the
        statistics anyway (BatchNorm); a fully frozen network in eval skips them.  The flags live on the device (per-site switches without a host
        round trip); the host keeps a summary and re-reads it only when it may have changed: after any `.apply(...)` (how
        torch.quantization.enable/disable_observer/fake_quant are applied) on the model OR ANY OF ITS SUB-MODULES (frostnet._FLAG_EPOCH), after
        `model.train()` / `.eval()`, or after `runner.flags_dirty = True`.  A disabled fake-quantizer is refused in both modes.  A flag written straight
        into a FakeQuantize buffer without any of these is honoured per site on the device at once; it reaches the summary (which only decides whether
        a fully frozen eval forward may skip its statistics passes) at the next of those events -- set `flags_dirty` after such a write."""
        from . import frostnet as _F
        if not torch.cuda.is_current_stream_capturing():
            # no per-forward device -> host read: the summary is re-read when an `.apply()` ran anywhere in a FrostNet tree since the last read (`_FLAG_EPOCH`: root or
            # any sub-module), after `model.train()` / `.eval()`, or when `flags_dirty` was set by hand
            if getattr(self, "flags_dirty", True) or getattr(self, "_flag_epoch", -1) != _F._FLAG_EPOCH[0]:
                self._obs_cached = self.read_flags()
                self.flags_dirty = False
                self._flag_epoch = _F._FLAG_EPOCH[0]
        return True if training else getattr(self, "_obs_cached", True)

    def rng_state(self):
        return rng_state(self)

    def set_rng_state(self, state):
        set_rng_state(self, state)

    def still_valid(self):
        m = self.model
        sig = tuple(p.data_ptr() for p in m.parameters()) + tuple(b.data_ptr() for b in m.buffers())
        return sig == self._sig

    # ------------------------------------------------------------------------------------------ binding
    @staticmethod
    def _is_per_channel(fq):
        return getattr(fq, "qscheme", None) in (torch.per_channel_symmetric, torch.per_channel_affine)

    def _bind_weight_per_channel(self, fq, rec, layer):
        """Weight FakeQuantize with a MovingAveragePerChannelMinMaxObserver (qint8, per_channel_symmetric, ch_axis 0 -- the reference's
        'fbgemm' qconfig, Classification/latency_check.py:221-226): min_val / max_val / scale are [cout] vectors, aliased onto the layer's
        device arrays; zero_point is all zeros; the enable flags live in the qrecord like everywhere else."""
        obs = fq.activation_post_process
        if getattr(obs, "ch_axis", 0) != 0:
            raise NotImplementedError("per-channel weight quantisation along ch_axis != 0")
        dev, cout = layer.w.device, layer.cout
        layer.per_channel = True
        layer.wmin = torch.full((cout,), float("inf"), dtype=torch.float32, device=dev)
        layer.wmax = torch.full((cout,), float("-inf"), dtype=torch.float32, device=dev)
        with torch.no_grad():
            if obs.min_val.numel() == cout:
                layer.wmin.copy_(obs.min_val.float())
                layer.wmax.copy_(obs.max_val.float())
            if fq.scale.numel() == cout:
                layer.wscale[:cout].copy_(fq.scale.float())
                rec[L.Q_SCALE] = fq.scale.float().max()
        self._alias_flags(fq, rec)
        fq._buffers["scale"] = layer.wscale[:cout]
        fq._buffers["zero_point"] = torch.zeros(cout, dtype=torch.int32, device=dev)
        obs._buffers["min_val"], obs._buffers["max_val"] = layer.wmin, layer.wmax
        return rec

    def _alias_flags(self, fq, rec):
        for name, slot in (("observer_enabled", L.Q_OBS_EN), ("fake_quant_enabled", L.Q_FQ_EN)):
            buf = fq._buffers[name]
            ri = rec.view(torch.int32)
            with torch.no_grad():
                ri[slot] = (buf.reshape(-1)[0] != 0).to(torch.int32)
                ri[slot + 1] = 0
            if buf.dtype == torch.int64:                 # FusedMovingAvgObsFakeQuantize (qconfig version 1)
                fq._buffers[name] = rec[slot:slot + 2].view(torch.int64)
            elif buf.dtype == torch.uint8:               # FakeQuantize (version 0)
                fq._buffers[name] = rec.view(torch.uint8)[4 * slot:4 * slot + 1]
            else:
                fq._buffers[name] = ri[slot:slot + 1].view(buf.dtype) if buf.element_size() == 4 else buf

    def _bind_fq(self, fq, rec):
        """Copy the module's current observer/qparam values into the qrecord, then alias the buffers onto it."""
        obs = fq.activation_post_process
        if fq.dtype == torch.quint8:                     # activation site: index range 0..quant_max (127 with reduce_range)
            with torch.no_grad():
                rec[L.Q_QMAX] = float(fq.quant_max)
            self.E.act_qmax = int(fq.quant_max)
        with torch.no_grad():
            rec[L.Q_MIN] = obs.min_val.reshape(-1)[0].float() if obs.min_val.numel() else float("inf")
            rec[L.Q_MAX] = obs.max_val.reshape(-1)[0].float() if obs.max_val.numel() else float("-inf")
            rec[L.Q_SCALE] = fq.scale.reshape(-1)[0].float()
            rec[L.Q_INV] = 1.0 / fq.scale.reshape(-1)[0].float()
            rec.view(torch.int32)[L.Q_ZP] = fq.zero_point.reshape(-1)[0].to(torch.int32)
        # torch.quantization.disable_observer / enable_observer / disable_fake_quant (the helpers at Classification/train.py:27-33,
        # evaluate.py:131-143) write 0/1 into the module's `observer_enabled` / `fake_quant_enabled` buffers.  Those buffers are
        # aliased onto the qrecord like scale / zero_point, so the write lands in device memory and the kernels test the site's own
        # flag: per-module granularity, no instance patching (deepcopy / pickling of the model stay intact), no host round trip.
        self._alias_flags(fq, rec)
        fq._buffers["scale"] = rec[L.Q_SCALE:L.Q_SCALE + 1]
        fq._buffers["zero_point"] = rec.view(torch.int32)[L.Q_ZP:L.Q_ZP + 1]
        obs._buffers["min_val"] = rec[L.Q_MIN]
        obs._buffers["max_val"] = rec[L.Q_MAX]
        return rec

    def _conv_layer(self, name, blk, kind):
        m = blk.conv[0]   # nniqat.ConvBnReLU2d / ConvBn2d
        relu = type(m).__name__ == "ConvBnReLU2d"
        pc = self._is_per_channel(m.weight_fake_quant)
        qw = self.qa.alloc() if pc else self._bind_fq(m.weight_fake_quant, self.qa.alloc())
        qy = self._bind_fq(m.activation_post_process, self.qa.alloc())
        self.rule127 = self.rule127 or _is_fused_fq(m.weight_fake_quant)
        l = ConvLayer(name, kind, m.weight, m.bn.weight, m.bn.bias, m.bn.running_mean, m.bn.running_var,
                      m.bn.num_batches_tracked, None, m.kernel_size[0], m.stride[0], relu, qw, qy)
        if pc:
            self._bind_weight_per_channel(m.weight_fake_quant, qw, l)
        l.bn_mod = m.bn
        l.qmod = m                # the QAT module this layer executes (export_converted walks the module tree)
        l.hswish = None
        act = getattr(blk, "act", None)
        if act is not None and type(act).__name__ == "Hswish":
            # ConvBNHswish (frostnet.py surface of the reference's `_ConvBNHswish`): the conv emits as a linear ConvBn2d with its own output FakeQuantize, then
            # the quantizable hard-swish -- FakeQuantize records of nn.ReLU6 and of quant_mul1 (both observed), plus the derived record of the result
            # after mul_scalar(1/6) (same indices, scale / 6: not a module buffer)
            l.hswish = (self._bind_fq(act.relu6.activation_post_process, self.qa.alloc()),
                        self._bind_fq(act.quant_mul1.activation_post_process, self.qa.alloc()), self.qa.alloc())
        return self.E.add_layer(l)

    def _conv(self, l, x, training, obs):
        """ConvBN(ReLU) / ConvBNHswish forward on the engine."""
        y = self.E.conv(l, x, training, obs)
        if l.hswish is not None:
            y = self.E.hswish(y, l.hswish[0], l.hswish[1], l.hswish[2], obs)
        return y

    def _bind_block(self, pre, b):
        d = dict(mod=b, squeeze=None, conv1=None, q_cat=None, q_add=None)
        if b.expand_ratio != 1:
            if b.block_type == "CAS":
                d["squeeze"] = self._conv_layer(pre + ".squeeze_conv", b.squeeze_conv, "pw")
                d["q_cat"] = self._bind_fq(b.quant_cat.activation_post_process, self.qa.alloc())
            d["conv1"] = self._conv_layer(pre + ".conv1", b.conv1, "pw")
        d["conv2"] = self._conv_layer(pre + ".conv2", b.conv2, "dw")
        d["reduce"] = self._conv_layer(pre + ".reduce_conv", b.reduce_conv, "pw")
        if not b.reduction:
            d["q_add"] = self._bind_fq(b.skip_add.activation_post_process, self.qa.alloc())
        return d

    def block_forward(self, d, inp, training, obs):
        """CascadePreExBottleneck.forward (frostnet.py:124-145) on the engine."""
        E = self.E
        out = inp
        if d["conv1"] is not None:
            if d["squeeze"] is not None:
                if d["squeeze"].hswish is None:
                    sq = E.conv(d["squeeze"], inp, training, obs, cat=(inp.q, d["q_cat"]))       # the cat's observer update rides in the squeeze's finalize tail
                else:
                    sq = self._conv(d["squeeze"], inp, training, obs)
                out = E.cat(sq, inp, d["q_cat"], obs)
            if E.pair_fusable(d["conv1"], d["conv2"], out, training, obs):
                out = E.conv_pair(d["conv1"], d["conv2"], out, training, obs, l3=d["reduce"])
            else:
                out = self._conv(d["conv2"], self._conv(d["conv1"], out, training, obs), training, obs)
        else:
            out = self._conv(d["conv2"], out, training, obs)
        if d["q_add"] is not None:
            out = E.conv(d["reduce"], out, training, obs, residual=(inp, d["q_add"]))      # reduce_conv is linear (never hard-swish): straight to the engine
            out = E.add(inp, out, d["q_add"], obs)
        else:
            out = self._conv(d["reduce"], out, training, obs)
        return out

    def _build(self):
        m = self.model
        blocks = [b for layer in (m.layer1, m.layer2, m.layer3, m.layer4, m.layer5) for b in layer]
        nsites = sum(1 for mod in m.modules() if hasattr(mod, "observer_enabled")) + 16     # every FakeQuantize of the tree + slack
        nsites += sum(1 for mod in m.modules() if type(mod).__name__ == "Hswish")              # + the derived output record of every hard-swish
        self.qa = QArena(nsites, self.device)
        self.rule127 = False
        self.q_in = self._bind_fq(m.quant.activation_post_process, self.qa.alloc())
        self.stem = self._conv_layer("conv1", m.conv1, "stem")
        self.blocks = []
        for lname in ("layer1", "layer2", "layer3", "layer4", "layer5"):
            for bi, b in enumerate(getattr(m, lname)):
                self.blocks.append(self._bind_block(f"{lname}.{bi}", b))
        self.last = self._conv_layer("last_layer", m.last_layer, "pw") if hasattr(m, "last_layer") else None
        self.cls = None
        if hasattr(m, "classifier"):
            c = m.classifier[2]
            pc = self._is_per_channel(c.weight_fake_quant)
            qw = self.qa.alloc() if pc else self._bind_fq(c.weight_fake_quant, self.qa.alloc())
            qy = self._bind_fq(c.activation_post_process, self.qa.alloc())
            self.cls = self.E.add_layer(ConvLayer("classifier.2", "cls", c.weight, None, None, None, None, None, c.bias,
                                                  1, 1, False, qw, qy))
            if pc:
                self._bind_weight_per_channel(c.weight_fake_quant, qw, self.cls)
            self.cls.qmod = c
            self.drop_rate = float(m.classifier[1].p)
        self._bind_extra()
        self.E.rule127 = 1 if self.rule127 else 0
        # buffers were re-pointed: refresh the validity signature
        self._sig = tuple(p.data_ptr() for p in m.parameters()) + tuple(b.data_ptr() for b in m.buffers())
        # flat gradient arena, parameter order = model.parameters() (the reference's registration order)
        params = list(m.parameters())
        self.grad_arena = torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device=self.device)
        self._grad_views, off = [], 0
        for p in params:
            self._grad_views.append(self.grad_arena[off: off + p.numel()].view_as(p))
            off += p.numel()
        self._params = params

    def _bind_extra(self):
        """Subclasses bind further layers here (SSDRunner: extras + prediction heads)."""

    def enable_data_parallel(self, nbuckets=4, group=None):
        """Attach the bucketed, backward-overlapped gradient all-reduce (frostnet_amd.parallel.GradSync)."""
        from .parallel import GradSync
        offs, off = {}, 0
        for p in self._params:
            offs[p.data_ptr()] = off
            off += p.numel()
        self.grad_sync = GradSync(self.grad_arena, list(offs.values()), nbuckets, group)
        self.E.on_layer_grads = lambda l: self.grad_sync.ready(offs[l.w.data_ptr()])
        return self.grad_sync

    def bind_grads(self):
        for p, v in zip(self._params, self._grad_views):
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                p.grad = v

    def _carry_over(self):
        """torch semantics: `backward()` ACCUMULATES into p.grad until `zero_grad()`.  The kernels write (=) into the flat arena, so the
        gradients a caller has not cleared are set aside here and added back after the backward pass.  Returns None in the usual loop
        (`zero_grad()` -> p.grad is None, or no backward has run yet): no copy, no extra launch."""
        if not getattr(self, "_grads_written", False):
            return None
        live = [p.grad is not None and p.grad.data_ptr() == v.data_ptr() for p, v in zip(self._params, self._grad_views)]
        if not any(live):
            return None
        if getattr(self, "grad_sync", None) is not None:
            raise RuntimeError("gradient accumulation over several backward passes is not supported together with the data-parallel exchange: "
                               "call zero_grad() (set_to_none=True) between steps")
        prev = self.grad_arena.clone()
        for ok, v in zip(live, self._grad_views):
            if not ok:                       # this parameter's gradient was cleared (or replaced): nothing to carry
                prev[v.storage_offset(): v.storage_offset() + v.numel()].zero_()
        return prev

    # ------------------------------------------------------------------------------------------ execution
    def _new_generation(self):
        """The saved activations of a training forward live on the engine's tape, not on the autograd node: one forward, one backward."""
        self._gen = getattr(self, "_gen", 0) + 1
        return self._gen

    def _check_generation(self, gen):
        if gen != getattr(self, "_gen", 0) or not self.E.tape:
            raise RuntimeError("backward through a FrostNet forward whose tape is gone: the HIP path keeps ONE recorded forward per model "
                               "(a later forward replaced it, or backward already ran). Run forward -> backward pairs one at a time.")

    # ------------------------------------------------------------------------------------------ converted int8 inference
    def convert(self):
        """`torch.quantization.convert(model.eval())` on the device (Classification/evaluate.py:130-134): from now on forward() computes what
        the converted model computes on the QNNPACK engine -- int8 weights frozen at convert time, integer bias, fp32 requantisation,
        QNNPACK's fixed-point add, rounding average pool -- instead of the fake-quant eval graph (whose observers keep moving and whose
        pooled features are not re-quantised; the two differ by 8-25 % of the activations' indices, see tests/test_gpu_convert.py)."""
        if getattr(self, "converted", False):
            return self                          # idempotent: a second convert() must not move the weight observers again
        pcs = [l.per_channel for l in self.E.layers]
        if any(pcs) and not all(pcs):
            raise NotImplementedError("convert(): mixed per-tensor / per-channel weight quantisation")
        # a per-channel model (the reference's 'fbgemm' qconfig, Classification/latency_check.py:221-226) converts to the FBGEMM engine's kernels:
        # float-bias requantisation with per-channel multipliers and the float add; activations no longer clamp at the observers' 7-bit range
        # (quantize_per_tensor / requantisation saturate at 255) -- the qrecords' index range is widened accordingly
        self.converted_fb = all(pcs)
        with torch.cuda.device(self.device):
            self.E.prepare_converted(observe=True)         # per-site observer flags still decide on the device
            if self.converted_fb:
                self.qa.t[:, L.Q_QMAX] = 255.0
        self.converted = True
        return self

    def _trunk_converted(self, x, taps=None):
        """QuantStub + stem + every bottleneck of the converted model; returns (last activation, [block outputs])."""
        E, fb = self.E, getattr(self, "converted_fb", False)
        if x.dtype != torch.float32:
            x = x.float()
        a = E.stem_converted(self.stem, x, self.q_in, fb)
        if a is None:
            a = E.quantize_input(x, self.q_in, observe=False)
            a = E.conv_converted(self.stem, a, fb)
        feats = []
        for d in self.blocks:
            inp, out = a, a
            if d["conv1"] is not None:
                if d["squeeze"] is not None:
                    out = E.conv_converted(d["squeeze"], inp, fb, cat=d["q_cat"])          # returns the cat when the fused launch applies ...
                    if out.c != d["squeeze"].cout + inp.c:                                  # ... else the squeezed activation
                        out = E.cat(out, inp, d["q_cat"], observe=False)
                out = E.conv_converted(d["conv1"], out, fb)
            out = E.conv_converted(d["conv2"], out, fb)
            out = E.conv_converted(d["reduce"], out, fb)
            if d["q_add"] is not None:
                # QNNPACK: integer fixed-point q8add; FBGEMM: dequantise, add in fp32, quantise -- the fake-quant graph's own add arithmetic
                out = E.add(inp, out, d["q_add"], observe=False) if fb else E.add_converted(inp, out, d["q_add"])
            a = out
            feats.append(a)
            if taps is not None:
                taps.append(a)
        return a, feats

    def _stage_ends(self, feats):
        ends, i = [], 0
        for lname in ("layer1", "layer2", "layer3", "layer4", "layer5"):
            i += len(getattr(self.model, lname))
            ends.append(feats[i - 1])
        return ends

    def _forward_converted(self, x, taps=None):
        a, _ = self._trunk_converted(x, taps)
        fb = getattr(self, "converted_fb", False)
        a = self.E.conv_converted(self.last, a, fb)
        logits = self.E.head_converted(self.cls, a, fb)
        self.E.tape = []
        return logits

    def _features_converted(self, x):
        """The converted features backbone (frostnet_features.py:350: [x1, x2, x3, x5], DeQuantStub on each)."""
        _, feats = self._trunk_converted(x)
        ends = self._stage_ends(feats)
        self.E.tape = []
        return [ends[0], ends[1], ends[2], ends[4]]

    def export_converted(self):
        """The state_dict of `torch.quantization.convert(model.eval())` (what Classification/evaluate.py:140-143 and Object_Detection/qeval_convert.py save as the
        deployment artefact), assembled from the device state of the converted model: per quantized conv `weight` (the int8 values the device convolves with --
        frost_export_wq -- as a qint8 per-tensor / per-channel quantized tensor), `bias` (fuse_conv_bn_weights' fp32 expression, evaluated with torch on the
        host), `scale` / `zero_point` (the layer's output record); `scale` / `zero_point` of the QuantStub and of every FloatFunctional.  Keys, shapes and
        dtypes are those of stock torch, so the CPU model `torch.quantization.convert(prepare_qat(fuse_model(FrostNet(...))).eval())` loads it with
        load_state_dict(strict=True) and computes the device's logits bit for bit (tests/test_gpu_convert.py)."""
        from collections import OrderedDict
        if not getattr(self, "converted", False):
            raise RuntimeError("export_converted: call hip_convert() first (torch.quantization.convert precedes state_dict() in the reference)")
        E = self.E
        E._ensure_tables()
        by_mod = {id(l.qmod): (i, l) for i, l in enumerate(E.layers) if getattr(l, "qmod", None) is not None}
        sd = OrderedDict()

        def qparams(fq, shape):
            sc = fq.scale.detach().float().reshape(-1)[:1].cpu()
            zp = fq.zero_point.detach().reshape(-1)[:1].to(torch.int64).cpu()
            return sc.reshape(shape).clone(), zp.reshape(shape).clone()
        with torch.cuda.device(self.device):
            for name, mod in self.model.named_modules():
                pre = name + "." if name else ""
                if id(mod) in by_mod:
                    i, l = by_mod[id(mod)]
                    n = l.cout * l.cin_g * l.kk
                    q8 = torch.empty(n, dtype=torch.int8, device=self.device)
                    L.call("frost_export_wq", L.ptr(E._table), i, n, L.ptr(q8), L.stream())
                    q8 = q8.cpu().view(l.cout, l.cin_g, l.k, l.k)
                    if l.per_channel:
                        w = torch._make_per_channel_quantized_tensor(q8, l.wscale[: l.cout].detach().double().cpu(), torch.zeros(l.cout, dtype=torch.int64), 0)
                    else:
                        w = torch._make_per_tensor_quantized_tensor(q8, float(l.qw[L.Q_SCALE]), 0)
                    sd[pre + "weight"] = w
                    if l.gamma is not None:             # torch.nn.utils.fusion.fuse_conv_bn_weights with conv bias None (nniqat.ConvBn2d.to_float)
                        g, b, rm, rv = (t.detach().float().cpu() for t in (l.gamma, l.beta, l.rmean, l.rvar))
                        sd[pre + "bias"] = (torch.zeros_like(rm) - rm) * torch.rsqrt(rv + l.bn_mod.eps) * g + b
                    else:
                        sd[pre + "bias"] = l.bias.detach().float().cpu().clone()
                    sd[pre + "scale"], sd[pre + "zero_point"] = qparams(mod.activation_post_process, ())
                elif isinstance(mod, torch.ao.quantization.QuantStub) and hasattr(mod, "activation_post_process"):
                    sd[pre + "scale"], sd[pre + "zero_point"] = qparams(mod.activation_post_process, (1,))
                elif type(mod).__name__ == "FloatFunctional" and hasattr(mod.activation_post_process, "scale"):
                    sd[pre + "scale"], sd[pre + "zero_point"] = qparams(mod.activation_post_process, ())
        return sd

    def _check_input(self, x):
        if x.device != self.device:
            raise RuntimeError(f"input is on {x.device} but the model's parameters (and its HIP runner) are on {self.device}")
        if x.requires_grad:
            raise NotImplementedError("the HIP path does not produce a gradient w.r.t. the input image (QuantStub output has none in "
                                      "the reference's training loops either); detach the input")

    def forward(self, x):
        self._check_input(x)
        training = self.model.training
        if getattr(self, "converted", False):
            if training:
                raise RuntimeError("this model was converted to int8 inference (hip_convert): it cannot be trained; rebuild the QAT model")
            with torch.cuda.device(self.device):
                return self._forward_converted(x)
        with torch.cuda.device(self.device):          # kernels launch on the CURRENT device's stream: make it the model's
            if training and torch.is_grad_enabled():
                return _QATFunction.apply(self._params[0], x, self)
            return self._forward_impl(x, record=False)

    def _trunk(self, x, training):
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("expected an (N,3,H,W) tensor")
        if x.numel() == 0:
            raise ValueError("empty batch")
        # a BatchNorm in eval mode inside a training forward (`_freeze_stages`, frostnet_features.py:354-359; mmdet's norm_eval) is honoured per layer:
        # Engine.conv normalises with the running statistics and the backward drops the batch-statistics terms (Engine._frozen_after_reduce)
        E, obs = self.E, self._observe_hint(training)
        self._obs = obs
        gp = getattr(self.model, "grad_precision", None)          # "fp32": the fp32-gradient parity mode of the backward (csrc/frost_g32.hip); default bf16 storage
        if gp is not None:
            if gp not in ("bf16", "fp32", "mixed"):
                raise ValueError("model.grad_precision must be 'bf16', 'fp32' or 'mixed'")
            E.grad_fp32, E.grad_mixed = gp == "fp32", gp == "mixed"
        late_at = None
        if _PREP_SIDE and training and not E.grad_fp32:
            # the per-step weight preparation (BN fold + weight fake-quant + packing of all 70 layers: a handful of latency-bound launches, ~165 us) has nothing to do
            # with the image: it runs on a second stream beside the QuantStub's passes over the input (range, observer, quantise: ~115 us of bandwidth), joined before conv1
            if getattr(E, "_prep_stream", None) is None:
                E._prep_stream = torch.cuda.Stream(device=E.device)
            # ... and in TWO parts (round 6): the stem + layer1 + layer2 (a few thousand weights) first; the rest -- layer3 on, where the parameters are: the launch is as
            # long as its largest layers -- stays on the second stream under the high-resolution forward and is joined in front of the first block that needs it.  In the
            # captured step the whole preparation cost 0.28 ms (FROST_ABL_SKIP, profiles/r06_pricing.txt) although it ran beside the QuantStub.
            cur = torch.cuda.current_stream()
            E._prep_stream.wait_stream(cur)
            split = self._prep_split() if _PREP_SPLIT else None
            with torch.cuda.stream(E._prep_stream):
                E.begin_step(observe=obs, part=None if split is None else (0, split[0]))
                if split is not None:
                    early = torch.cuda.Event()
                    early.record()
                    E.begin_step(observe=obs, part=(split[0], len(E.layers)))
            a = E.quantize_input(x, self.q_in, observe=obs)
            if _STEM_COL_EARLY and self.stem.kind == "stem" and getattr(E, "trace", None) is None:
                a = E.stem_im2col(a)          # needs no weights: in front of the join (the join used to hold it back by the tail of the weight preparation)
            if split is None:
                cur.wait_stream(E._prep_stream)
            else:
                cur.wait_event(early)
                late_at = split[1]
        else:
            E.begin_step(observe=obs)
            a = E.quantize_input(x, self.q_in, observe=obs)
        a = self._conv(self.stem, a, training, obs)
        feats = []
        for bi, d in enumerate(self.blocks):
            if late_at is not None and bi == late_at:
                torch.cuda.current_stream().wait_stream(E._prep_stream)
                late_at = None
            a = self.block_forward(d, a, training, obs)
            feats.append(a)
        if late_at is not None:
            torch.cuda.current_stream().wait_stream(E._prep_stream)
        return a, feats

    def _prep_split(self):
        """(layer index, block index) at which the late part of the weight preparation starts: the first block of layer3 (None: no such block / too few layers)."""
        got = getattr(self, "_prep_split_cache", None)
        if got is None:
            got = (None,)
            for bi, d in enumerate(self.blocks):
                first = d["squeeze"] or d["conv1"] or d["conv2"]
                if first.name.startswith("layer3."):
                    idx = self.E.layers.index(first)
                    if 0 < idx < len(self.E.layers):
                        got = ((idx, bi),)
                    break
            self._prep_split_cache = got
        return got[0]

    def forward_features(self, x):
        self._check_input(x)
        if getattr(self, "converted", False):
            if self.model.training:
                raise RuntimeError("this model was converted to int8 inference (hip_convert): it cannot be trained; rebuild the QAT model")
            with torch.cuda.device(self.device):
                return [a.dequant().contiguous() for a in self._features_converted(x)]
        with torch.cuda.device(self.device):
            if self.model.training and torch.is_grad_enabled():
                return list(_QATFeatFunction.apply(self._params[0], x, self))
            acts = self._features_impl(x, record=False)
            return [a.dequant().contiguous() for a in acts]

    def _features_impl(self, x, record):
        if x.dtype != torch.float32:
            x = x.float()
        _, feats = self._trunk(x, self.model.training)
        ends, i = [], 0
        for lname in ("layer1", "layer2", "layer3", "layer4", "layer5"):
            i += len(getattr(self.model, lname))
            ends.append(feats[i - 1])
        if not record:
            self.E.tape = []
        return [ends[0], ends[1], ends[2], ends[4]]          # x4 is skipped (frostnet_features.py:350)

    def _forward_impl(self, x, record):
        training = self.model.training
        if x.dtype != torch.float32:
            x = x.float()
        a, _ = self._trunk(x, training)
        a = self._conv(self.last, a, training, self._obs)
        drop = None
        drop_rate = float(self.model.classifier[1].p)
        if training and drop_rate > 0.0:
            keep = 1.0 - drop_rate
            drop = dropout_mask(self, a.n * a.c, keep).view(a.n, a.c)
        logits = self.E.head(self.cls, a, drop, self._obs)
        if not record:
            self.E.tape = []
        return logits

    def _backward_impl(self, dlogits):
        with torch.cuda.device(self.device):
            prev = self._carry_over()
            self.bind_grads()
            self.E.backward(dlogits)
            if prev is not None:
                self.grad_arena.add_(prev)
            self._grads_written = True


class _QATMapsFunction(torch.autograd.Function):
    """image -> a list of fake-quantised maps, dequantised to fp32 NCHW (one DeQuantStub per map); backward feeds their gradients to the tape."""

    @staticmethod
    def forward(ctx, anchor, x, runner):
        acts = runner._maps_impl(x, record=True)
        ctx.runner, ctx.acts = runner, acts
        ctx.gen = runner._new_generation()
        # NCHW-shaped views of NHWC memory (channels_last): the detector's `_assemble` permutes them straight back to NHWC, so a contiguous NCHW copy here would be
        # a transposing pass per map each way
        return tuple(a.dequant() for a in acts)

    @staticmethod
    def backward(ctx, *grads):
        from .engine import float_to_grad
        ctx.runner._check_generation(ctx.gen)
        with torch.cuda.device(ctx.runner.device):
            for a, g in zip(ctx.acts, grads):
                if g is None:
                    g = torch.zeros(a.n, a.c, a.h, a.w, device=a.buf.device)
                a.grad = float_to_grad(g, fp32=ctx.runner.E.grad_is_fp32(a))
            ctx.runner._backward_impl(None)
        return None, None, None


class SSDRunner(FrostRunner):
    """frostnet_amd.ssdlite.SSDLiteFrostNet on the engine: backbone (sources at strides 8/16/32) + SSDLite extras + separable prediction heads,
    every layer a fake-quantised ConvBN(ReLU) on the int8-MFMA / LDS depthwise kernels; twelve dequantised maps come back."""

    def _bind_extra(self):
        m = self.model
        self.extras = [[self._conv_layer(f"extras.{i}.pw1", e.pw1, "pw"), self._conv_layer(f"extras.{i}.dw", e.dw, "dw"),
                        self._conv_layer(f"extras.{i}.pw2", e.pw2, "pw")] for i, e in enumerate(m.extras)]
        self.heads = []
        for i, (l, c) in enumerate(zip(m.loc, m.conf)):
            self.heads.append((self._conv_layer(f"loc.{i}.dw", l.dw, "dw"), self._conv_layer(f"loc.{i}.pw", l.pw, "pw"),
                               self._conv_layer(f"conf.{i}.dw", c.dw, "dw"), self._conv_layer(f"conf.{i}.pw", c.pw, "pw")))

    def _maps_converted(self, x):
        """The converted detector (Object_Detection/qeval_convert.py: torch.quantization.convert of the QAT SSD): backbone, extras and the separable
        prediction heads as quantized convs with the requantisation epilogue; twelve maps."""
        E, fb = self.E, getattr(self, "converted_fb", False)
        a, feats = self._trunk_converted(x)
        ends = self._stage_ends(feats)
        sources = [ends[1], ends[2], ends[4]]
        for pw1, dw, pw2 in self.extras:
            a = E.conv_converted(pw2, E.conv_converted(dw, E.conv_converted(pw1, a, fb), fb), fb)
            sources.append(a)
        maps = []
        for s, (ldw, lpw, cdw, cpw) in zip(sources, self.heads):
            maps.append(E.conv_converted(lpw, E.conv_converted(ldw, s, fb), fb))
            maps.append(E.conv_converted(cpw, E.conv_converted(cdw, s, fb), fb))
        E.tape = []
        return maps

    def forward_maps(self, x):
        self._check_input(x)
        if getattr(self, "converted", False):
            if self.model.training:
                raise RuntimeError("this model was converted to int8 inference (hip_convert): it cannot be trained; rebuild the QAT model")
            with torch.cuda.device(self.device):
                return [a.dequant().contiguous() for a in self._maps_converted(x)]
        with torch.cuda.device(self.device):
            if self.model.training and torch.is_grad_enabled():
                return list(_QATMapsFunction.apply(self._params[0], x, self))
            return [a.dequant().contiguous() for a in self._maps_impl(x, record=False)]

    def _maps_impl(self, x, record):
        training = self.model.training
        if x.dtype != torch.float32:
            x = x.float()
        a, feats = self._trunk(x, training)
        obs, E = self._obs, self.E
        ends, i = [], 0
        for lname in ("layer1", "layer2", "layer3", "layer4", "layer5"):
            i += len(getattr(self.model, lname))
            ends.append(feats[i - 1])
        sources = [ends[1], ends[2], ends[4]]
        for pw1, dw, pw2 in self.extras:
            a = E.conv(pw2, E.conv(dw, E.conv(pw1, a, training, obs), training, obs), training, obs)
            sources.append(a)
        maps = []
        for s, (ldw, lpw, cdw, cpw) in zip(sources, self.heads):
            maps.append(E.conv(lpw, E.conv(ldw, s, training, obs), training, obs))
            maps.append(E.conv(cpw, E.conv(cdw, s, training, obs), training, obs))
        if not record:
            E.tape = []
        return maps
