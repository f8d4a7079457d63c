"""FrostNet classifier -- the reference's nn.Module surface (frostnet.py:14-451) over the MI355X HIP engine.

Same classes, constructor signatures, attribute names and state_dict keys as the reference (`ConvBNReLU`, `ConvBN`,
`CascadePreExBottleneck`, `FrostNet`, 30 `frostnet[_quant]_{large,base,small}_{1_25,1_0,0_75,0_5,0_35}` factories), so
`Classification/train.py`-style drivers, reference checkpoints and `torch.quantization.prepare_qat` work unchanged.

The module tree holds stock torch modules (that is what defines the state_dict layout).  Execution:
  * tensors on the HIP device + model switched to QAT (`fuse_model()` + `prepare_qat`) -> the hand-written gfx950
    path (frostnet_amd.engine / libfrost_hip.so); missing library => RuntimeError, never a silent eager fallback;
  * tensors on the CPU -> the stock module graph, i.e. the reference's own definition (config c1 plumbing only).
"""
import torch
import torch.nn as nn

try:  # soft dependency, exactly what frostnet.py:4-5 needs
    from timm.models.registry import register_model
except Exception:  # pragma: no cover
    def register_model(fn):
        MODEL_REGISTRY[fn.__name__] = fn
        return fn

MODEL_REGISTRY = {}
_FLAG_EPOCH = [0]      # bumped by every `.apply(fn)` on a module of this file: the runners compare it with the epoch of their host-side summary of the per-site enable flags


class _FlagNotify(nn.Module):
    """`model.layer3.apply(torch.quantization.disable_observer)` (or on a single block / ConvBNReLU) writes the per-site `observer_enabled` /
    `fake_quant_enabled` flags -- device memory on the HIP path.  The write itself is honoured per site at once; the host-side summary a runner keeps of
    them (frostnet_amd.runner.FrostRunner._observe_hint) is invalidated here, whichever sub-module the call was made on."""

    def apply(self, fn):
        out = super().apply(fn)
        _FLAG_EPOCH[0] += 1
        return out


def _fuse(seq, names):
    """frostnet.py:28,60 call torch.quantization.fuse_modules; torch>=1.11 split the train-mode variant out."""
    import torch.ao.quantization as aoq
    if seq.training:
        aoq.fuse_modules_qat(seq, names, inplace=True)
    else:
        aoq.fuse_modules(seq, names, inplace=True)


class ConvBNReLU(_FlagNotify):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1):
        super(ConvBNReLU, self).__init__()
        self.conv = nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                                            bias=False),
                                  nn.BatchNorm2d(out_channels), nn.ReLU(False))

    def forward(self, x):
        return self.conv(x)

    def fuse_model(self):
        _fuse(self.conv, ['0', '1', '2'])


class ConvBN(_FlagNotify):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1):
        super(ConvBN, self).__init__()
        self.conv = nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                                            bias=False),
                                  nn.BatchNorm2d(out_channels))

    def forward(self, x):
        return self.conv(x)

    def fuse_model(self):
        _fuse(self.conv, ['0', '1'])


class Hswish(nn.Module):
    """Quantizable hard-swish: the reference's `_Hswish` (Classification/models/imagenet/mobilenetv3.py:43-56) -- x * relu6(x + 3) / 6 written with
    FloatFunctionals so that prepare_qat hangs FakeQuantize sites on it (two execute: nn.ReLU6's and quant_mul1's; add_scalar / mul_scalar are not
    observed).  Same attribute names, so its state_dict keys are the reference's.  FrostNet itself uses ReLU (frostnet.py:21); this is the optional
    activation BASELINE.json's north_star names, reachable through `act="hswish"`.  On the HIP device the op is `Engine.hswish`
    (csrc/frost_convert.hip: presence bitmap -> observers -> 256-entry tables), bit-exact against the reference golden G10."""

    def __init__(self, inplace=True):
        super(Hswish, self).__init__()
        self.relu6 = nn.ReLU6(inplace)
        self.quant_mul1 = nn.quantized.FloatFunctional()
        self.quant_mul2 = nn.quantized.FloatFunctional()
        self.quant_add = nn.quantized.FloatFunctional()

    def forward(self, x):
        out = self.quant_add.add_scalar(x, 3.0)
        out = self.relu6(out)
        out = self.quant_mul1.mul(x, out)
        out = self.quant_mul2.mul_scalar(out, 1 / 6)
        return out


class ConvBNHswish(_FlagNotify):
    """Conv -> BatchNorm -> hard-swish: the reference's `_ConvBNHswish` (mobilenetv3.py:72-88) in FrostNet's layer vocabulary (`conv` = the fusable
    Conv2d + BatchNorm2d pair, `act` = Hswish)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1):
        super(ConvBNHswish, self).__init__()
        self.conv = nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias=False),
                                  nn.BatchNorm2d(out_channels))
        self.act = Hswish(True)

    def forward(self, x):
        return self.act(self.conv(x))

    def fuse_model(self):
        _fuse(self.conv, ['0', '1'])


def _make_divisible(v, divisor=8, min_value=None):
    """frostnet.py:62-79."""
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


class CascadePreExBottleneck(_FlagNotify):
    """The Frost bottleneck (frostnet.py:81-145)."""

    def __init__(self, in_channels, out_channels, quantized=False, kernel_size=3, stride=1, dilation=1, expand_ratio=6,
                 reduce_factor=4, block_type='CAS', act='relu'):
        super(CascadePreExBottleneck, self).__init__()
        if act not in ('relu', 'hswish'):
            raise ValueError("act must be 'relu' (the reference's FrostNet) or 'hswish'")
        ConvBNReLU = {'relu': globals()['ConvBNReLU'], 'hswish': ConvBNHswish}[act]      # the activated-layer class of this block
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = kernel_size
        self.stride = stride
        self.expand_ratio = expand_ratio
        self.quantized = quantized
        if in_channels // reduce_factor < 8:
            block_type = 'MB'
        self.block_type = block_type
        r_channels = _make_divisible(in_channels // reduce_factor)
        self.reduction = not (stride == 1 and in_channels == out_channels)
        if self.expand_ratio == 1:
            self.squeeze_conv = None
            self.conv1 = None
            n_channels = in_channels
        else:
            if block_type == 'CAS':
                self.squeeze_conv = ConvBNReLU(in_channels, r_channels, 1)
                n_channels = r_channels + in_channels
            else:
                n_channels = in_channels
            self.conv1 = ConvBNReLU(n_channels, n_channels * expand_ratio, 1)
        # NB: dilation is accepted and ignored, exactly like the reference (frostnet.py:116-118)
        self.conv2 = ConvBNReLU(n_channels * expand_ratio, n_channels * expand_ratio, kernel_size, stride,
                                (kernel_size - 1) // 2, 1, groups=n_channels * expand_ratio)
        self.reduce_conv = ConvBN(n_channels * expand_ratio, out_channels, 1)
        if self.quantized:
            self.skip_add = nn.quantized.FloatFunctional()
            self.quant_cat = nn.quantized.FloatFunctional()

    def forward(self, x):
        if not self.expand_ratio == 1:
            if self.block_type == 'CAS':
                squeezed = self.squeeze_conv(x)
                out = self.quant_cat.cat([squeezed, x], 1) if self.quantized else torch.cat([squeezed, x], 1)
            else:
                out = x
            out = self.conv1(out)
        else:
            out = x
        out = self.conv2(out)
        out = self.reduce_conv(out)
        if not self.reduction:
            out = self.skip_add.add(x, out) if self.quantized else torch.add(x, out)
        return out


# stage tables, rows = [kernel_size, c, e, r, s]  (frostnet.py:156-269)
_SETTINGS = {
    'large': ([[3, 16, 1, 1, 1], [3, 24, 6, 4, 2], [3, 24, 3, 4, 1]],
              [[5, 40, 6, 4, 2], [3, 40, 3, 4, 1]],
              [[5, 80, 6, 4, 2], [5, 80, 3, 4, 1], [5, 80, 3, 4, 1], [5, 96, 6, 4, 1], [5, 96, 3, 4, 1], [3, 96, 3, 4, 1],
               [3, 96, 3, 4, 1]],
              [[5, 192, 6, 2, 2], [5, 192, 6, 4, 1], [5, 192, 6, 4, 1], [5, 192, 3, 4, 1], [5, 192, 3, 4, 1]],
              [[5, 320, 6, 2, 1]]),
    'base': ([[3, 16, 1, 1, 1], [5, 24, 6, 4, 2], [3, 24, 3, 4, 1]],
             [[5, 40, 3, 4, 2], [5, 40, 3, 4, 1]],
             [[5, 80, 3, 4, 2], [3, 80, 3, 4, 1], [5, 96, 3, 2, 1], [3, 96, 3, 4, 1], [5, 96, 3, 4, 1], [5, 96, 3, 4, 1]],
             [[5, 192, 6, 2, 2], [5, 192, 3, 2, 1], [5, 192, 3, 2, 1], [5, 192, 3, 2, 1]],
             [[5, 320, 6, 2, 1]]),
    'small': ([[3, 16, 1, 1, 1], [5, 24, 3, 4, 2], [3, 24, 3, 4, 1]],
              [[5, 40, 3, 4, 2]],
              [[5, 80, 3, 4, 2], [5, 80, 3, 4, 1], [3, 80, 3, 4, 1], [5, 96, 3, 2, 1], [5, 96, 3, 4, 1], [5, 96, 3, 4, 1]],
              [[5, 192, 6, 4, 2], [5, 192, 6, 4, 1], [5, 192, 6, 4, 1]],
              [[5, 320, 6, 2, 1]]),
}


class _FrostBase(_FlagNotify):
    def _make_layer(self, block, block_setting, width_mult, dilation=1):
        layers = list()
        for k, c, e, r, s in block_setting:
            out_channels = _make_divisible(int(c * width_mult))
            extra = {} if getattr(self, "act", "relu") == "relu" else {"act": self.act}
            layers.append(block(self.in_channels, out_channels, quantized=self.quantized, kernel_size=k, stride=s,
                                dilation=dilation, expand_ratio=e, reduce_factor=r, **extra))
            self.in_channels = out_channels
        return nn.Sequential(*layers)

    def _init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out')
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def fuse_model(self):
        for m in self.modules():
            if type(m) in [ConvBNReLU, ConvBN, ConvBNHswish]:
                m.fuse_model()

    # ---- HIP execution --------------------------------------------------------------------------------
    def state_dict(self, *args, **kwargs):
        """As nn.Module.state_dict; the observer / qparam buffers that the HIP runner aliased onto its qrecord arena (float, int32, uint8
        and int64 views of ONE storage -- torch.save refuses such a set) are returned as detached copies, so
        `torch.save({'state_dict': model.state_dict(), ...})` (Classification/train.py:210-218) works on the bound model."""
        sd = super().state_dict(*args, **kwargs)
        r = self.__dict__.get("_hip_runner")
        arena = getattr(getattr(r, "qa", None), "t", None)
        if arena is not None:
            base = arena.untyped_storage().data_ptr()
            for k, v in list(sd.items()):
                if torch.is_tensor(v) and v.device == arena.device and v.untyped_storage().data_ptr() == base:
                    sd[k] = v.detach().clone()
        return sd

    def __getstate__(self):
        """copy.deepcopy / pickle (EMA or best-model snapshots, torch.save(model)): the device executors are per-instance caches
        and stay behind; the copy builds its own on first use.  The observer / qparam buffers a bound model aliases onto the runner's qrecord
        arena (float, int32, uint8 and int64 views of ONE storage: torch.save refuses such a set) are replaced by detached copies in the
        pickled state, module by module, so whole-module pickling works like state_dict() does."""
        d = self.__dict__.copy()
        r = d.pop("_hip_runner", None)
        d.pop("_bf16_infer", None)
        # `_hip_converted` stays: a copy of a converted model has no frozen int8 weights; its forward raises (hip_runner) rather than computing the QAT eval graph
        arena = getattr(getattr(r, "qa", None), "t", None)
        if arena is not None:
            import copy as _copy
            base = arena.untyped_storage().data_ptr()

            def detach_aliases(mod):
                new = _copy.copy(mod)                                   # shallow: parameters stay shared with the live module (deepcopy / pickle copy them)
                new._buffers = {k: (v.detach().clone() if torch.is_tensor(v) and v.device == arena.device and v.untyped_storage().data_ptr() == base else v)
                                for k, v in mod._buffers.items()}
                new._modules = {k: (detach_aliases(c) if c is not None else None) for k, c in mod._modules.items()}
                return new
            d["_modules"] = {k: (detach_aliases(c) if c is not None else None) for k, c in self._modules.items()}
            d["_buffers"] = {k: (v.detach().clone() if torch.is_tensor(v) and v.device == arena.device and v.untyped_storage().data_ptr() == base else v)
                             for k, v in self._buffers.items()}
        return d

    def _is_qat_prepared(self):
        return hasattr(self.conv1.conv[0], "weight_fake_quant")

    def _flags_changed(self):
        r = self.__dict__.get("_hip_runner")
        if r is not None and hasattr(r, "flags_dirty"):
            r.flags_dirty = True

    def apply(self, fn):
        """nn.Module.apply; `model.apply(torch.quantization.disable_observer)` and friends write the per-site enable flags on the device, so the
        runner's host-side summary of them is re-read before the next forward (no per-forward device -> host read otherwise)."""
        out = super().apply(fn)
        self._flags_changed()
        return out

    def train(self, mode=True):
        out = super().train(mode)
        self._flags_changed()
        return out

    def hip_runner(self):
        """The device executor bound to this module tree (built lazily, rebuilt if parameters were moved)."""
        if getattr(self, "_is_replica", False):
            raise RuntimeError("nn.DataParallel replicas are not supported by the HIP path (replicas share one runner and its device-0 "
                               "pointers): data parallelism is one process per GPU -- frostnet_amd.parallel, `bench.py --gpus N`")
        qat = self._is_qat_prepared()
        r = self.__dict__.get("_hip_runner")
        want = getattr(self, "float_precision", None)        # 'bf16' (default) | 'fp32': activation storage of the float (warm-up) path
        if r is not None and not qat and want is not None and getattr(r, "precision", want) != want:
            r = None
        if r is None or r.model is not self or r.is_qat != qat or not r.still_valid():
            if qat and self.__dict__.get("_hip_converted", False):
                # the converted state (int8 packs frozen at convert time) lived on the runner that is gone (parameters or buffers moved, or this is a
                # deepcopy / unpickled copy of a converted model): converting again would move the weight observers a second time, silently running the
                # fake-quant eval graph would be a different model.  Refuse -- on EVERY call: no runner is cached while the flag is set.
                self.__dict__.pop("_hip_runner", None)
                raise RuntimeError("this model was hip_convert()-ed and its device executor is gone (parameters or buffers moved, or the model was copied / "
                                   "unpickled): the frozen int8 weights went with it; rebuild the QAT model, load its state and call hip_convert() again")
            if qat:
                from .runner import FrostRunner
                r = FrostRunner(self)
            else:                                   # the float model: StatAssist warm-up training / float eval (float_train.py)
                from .float_train import FloatRunner
                r = FloatRunner(self)
            r.is_qat = qat
            self.__dict__["_hip_runner"] = r
        return r


    def hip_convert(self):
        """Device counterpart of `torch.quantization.convert(model.eval(), inplace=True)` (Classification/evaluate.py:130): the QAT-prepared
        model on the HIP device switches to converted int8 inference semantics (frostnet_amd.runner.FrostRunner.convert)."""
        if not self._is_qat_prepared():
            raise RuntimeError("hip_convert needs the QAT-prepared model (fuse_model + prepare_qat), like torch.quantization.convert")
        self.eval()
        self.hip_runner().convert()
        # marks the module tree: if the runner holding the frozen int8 weights is ever lost (model.to(), a moved parameter, a deepcopy / pickle of the
        # model -- __getstate__ keeps this flag and drops the runner), hip_runner() raises on every call instead of running the fake-quant eval graph
        self.__dict__["_hip_converted"] = True
        return self

    def hip_export_converted(self):
        """`model.state_dict()` of the converted model (Classification/evaluate.py:140-143 saves exactly that as `quantized_<name>.pth`): a mapping that the
        stock-torch CPU model `torch.quantization.convert(<the same QAT-prepared architecture>.eval())` loads with load_state_dict(strict=True) and that
        then computes, on the QNNPACK / FBGEMM CPU engine, the logits (maps) the device computes.  frostnet_amd.runner.FrostRunner.export_converted."""
        if not self.__dict__.get("_hip_converted", False):
            raise RuntimeError("hip_export_converted: call hip_convert() first")
        return self.hip_runner().export_converted()


class FrostNet(_FrostBase):
    def __init__(self, nclass=1000, mode='large', width_mult=1.0, quantized=False, bottleneck=CascadePreExBottleneck,
                 drop_rate=0.2, dilated=False, act='relu', **kwargs):
        super(FrostNet, self).__init__()
        self.quantized = quantized
        if mode not in _SETTINGS:
            raise ValueError('Unknown mode.')
        if act not in ('relu', 'hswish'):
            raise ValueError("act must be 'relu' (the reference's FrostNet, frostnet.py:21) or 'hswish' (the quantizable hard-swish of its MobileNetV3 zoo)")
        self.act = act                          # 'hswish': every ConvBNReLU of the network becomes ConvBNHswish (SURVEY N4 / north_star wording)
        ConvBNReLU = {'relu': globals()['ConvBNReLU'], 'hswish': ConvBNHswish}[act]
        l1, l2, l3, l4, l5 = _SETTINGS[mode]
        self.in_channels = _make_divisible(int(32 * min(1.0, width_mult)))
        self.conv1 = ConvBNReLU(3, self.in_channels, 3, 2, 1)
        self.layer1 = self._make_layer(bottleneck, l1, width_mult, 1)
        self.layer2 = self._make_layer(bottleneck, l2, width_mult, 1)
        self.layer3 = self._make_layer(bottleneck, l3, width_mult, 1)
        dilation = 2 if dilated else 1          # no effect, like the reference
        self.layer4 = self._make_layer(bottleneck, l4, width_mult, dilation)
        self.layer5 = self._make_layer(bottleneck, l5, width_mult, dilation)
        last_in_channels = self.in_channels
        self.last_layer = ConvBNReLU(last_in_channels, 1280, 1)
        self.classifier = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Dropout(drop_rate), nn.Conv2d(1280, nclass, 1))
        self.mode = mode
        self._init_weights()
        if self.quantized:
            self.quant = torch.quantization.QuantStub()
            self.dequant = torch.quantization.DeQuantStub()

    def hip_infer_bf16(self, x):
        """bf16 inference of the float (un-fused, not QAT-prepared) model on the HIP kernels (BASELINE.json config c2):
        eval-mode BatchNorm folded, NHWC bf16 activations, fp32 accumulation.  See frostnet_amd/infer.py."""
        if self._is_qat_prepared():
            raise RuntimeError("hip_infer_bf16 is the float model's inference path; a QAT-prepared model runs model(x)")
        inf = self.__dict__.get("_bf16_infer")
        if inf is None or inf.device != next(self.parameters()).device:
            from .infer import Bf16Inference
            inf = Bf16Inference(self)
            self.__dict__["_bf16_infer"] = inf
        return inf(x)

    def forward(self, x):
        if x.is_cuda:           # device tensors always take the HIP path (fake-quant or float runner); the stock modules below are the CPU definition
            return self.hip_runner().forward(x)
        if self.quantized:
            x = self.quant(x)
        x = self.conv1(x)
        x = self.layer1(x)
        x = self.layer2(x)
        x = self.layer3(x)
        x = self.layer4(x)
        x = self.layer5(x)
        x = self.last_layer(x)
        x = self.classifier(x)
        if self.quantized:
            x = self.dequant(x)
        return x.view(x.size(0), x.size(1))


def _factory(mode, width, quantized):
    def fn(**kwargs):
        return FrostNet(nclass=1000, mode=mode, width_mult=width, quantized=quantized, bottleneck=CascadePreExBottleneck,
                        **kwargs)
    return fn


for _q in (True, False):
    for _mode in ('large', 'base', 'small'):
        for _wm, _tag in ((1.25, '1_25'), (1.0, '1_0'), (0.75, '0_75'), (0.5, '0_5'), (0.35, '0_35')):
            _name = f"frostnet_{'quant_' if _q else ''}{_mode}_{_tag}"
            _fn = _factory(_mode, _wm, _q)
            _fn.__name__ = _name
            _fn.__qualname__ = _name
            globals()[_name] = register_model(_fn)
            MODEL_REGISTRY.setdefault(_name, globals()[_name])


def create_model(name, pretrained=False, **kwargs):
    """timm-style entry point: `num_classes`/`pretrained` are accepted and ignored like the reference."""
    return MODEL_REGISTRY[name](**kwargs)


def qat_prepare(model, version=0, backend="qnnpack"):
    """Classification/train.py:166-173: fuse -> qconfig -> prepare_qat, parameters keep their identity."""
    from torch.ao.quantization import get_default_qat_qconfig, prepare_qat
    model.train()
    model.fuse_model()
    model.qconfig = get_default_qat_qconfig(backend, version=version)
    prepare_qat(model, inplace=True)
    return model
