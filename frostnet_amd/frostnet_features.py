"""FrostNet feature backbone -- the reference's `frostnet_features.py` surface (class `FrostNet(mode, width_mult, bottleneck,
quantized, pretrained)`, `forward -> [x1, x2, x3, x5]` at strides 4/8/16/32, `init_weights`, `_freeze_stages`,
`load_state_dict` / `load_checkpoint` helpers, mmdet `BACKBONES` registration when mmdet is importable).

Reference facts kept: same stem + layer1..5 tables as the classifier, no last_layer/classifier, x4 is skipped
(frostnet_features.py:342-352).  Extension for BASELINE.json config 5 (QAT detection; the reference's features class is FP
only): with `quantized=True` the module also carries `quant`/`dequant` stubs and `fuse_model()` following the SSDLite-MobileNetV2
convention of the reference (Object_Detection/ssd_qmv2.py:209-215,249-252: QuantStub at the input, DeQuantStub per source),
so `fuse_model(); prepare_qat` works and the HIP engine serves the four fake-quantised feature maps (dequantised to fp32).
"""
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from .frostnet import _SETTINGS, CascadePreExBottleneck, ConvBN, ConvBNReLU, _FrostBase, _make_divisible  # noqa: F401

try:  # pragma: no cover - mmdet is not in this image
    from mmdet.models.builder import BACKBONES
    _register = BACKBONES.register_module()
except Exception:
    def _register(cls):
        return cls


def _strip_parallel_prefix(key):
    """`module.` in front of a key = the checkpoint was written from a DataParallel / DDP wrapper."""
    return key[len("module."):] if key.startswith("module.") else key


def load_state_dict(checkpoint_path, use_ema=False):
    """Read a timm-style training checkpoint and return a plain name -> tensor mapping for `Module.load_state_dict`.

    Behaviour of the reference's ingest helper (frostnet_features.py:10-30): the file may be a bare state_dict or a dict that wraps
    one under `state_dict` (and the EMA shadow weights under `state_dict_ema`, preferred when `use_ema`); wrapper prefixes are
    removed.  A missing file raises FileNotFoundError."""
    if not checkpoint_path or not os.path.isfile(checkpoint_path):
        raise FileNotFoundError(f"no checkpoint at '{checkpoint_path}'")
    blob = torch.load(checkpoint_path, map_location="cpu")
    picked = None
    if isinstance(blob, dict):
        for key in (("state_dict_ema",) if use_ema else ()) + ("state_dict",):
            if isinstance(blob.get(key), dict):
                picked = key
                break
    weights = blob[picked] if picked is not None else blob
    return OrderedDict((_strip_parallel_prefix(k), v) for k, v in weights.items())


def load_checkpoint(model, checkpoint_path, use_ema=False, strict=True):
    """frostnet_features.py:33-35.  Returns torch's (missing_keys, unexpected_keys) record."""
    return model.load_state_dict(load_state_dict(checkpoint_path, use_ema), strict=strict)


@_register
class FrostNet(_FrostBase):
    def __init__(self, mode='large', width_mult=1.0, bottleneck=CascadePreExBottleneck, quantized=False, pretrained='', **kwargs):
        super(FrostNet, self).__init__()
        self.quantized = quantized
        if mode not in _SETTINGS:
            raise ValueError('Unknown mode.')
        l1, l2, l3, l4, l5 = _SETTINGS[mode]
        self.in_channels = _make_divisible(int(32 * min(1.0, width_mult)))
        self.conv1 = ConvBNReLU(3, self.in_channels, 3, 2, 1)
        self.layer1 = self._make_layer(bottleneck, l1, width_mult, 1)
        self.layer2 = self._make_layer(bottleneck, l2, width_mult, 1)
        self.layer3 = self._make_layer(bottleneck, l3, width_mult, 1)
        self.layer4 = self._make_layer(bottleneck, l4, width_mult, 1)
        self.layer5 = self._make_layer(bottleneck, l5, width_mult, 1)
        self.mode = mode
        if self.quantized:
            self.quant = torch.quantization.QuantStub()
            self.dequant = torch.quantization.DeQuantStub()

    def init_weights(self, pretrained=''):
        """mmdet backbone hook (frostnet_features.py:314-319): EMA weights of a timm checkpoint when a path is given (non-strict: the
        classifier head of the checkpoint has no counterpart here), else the Kaiming / unit-BN initialisation."""
        if pretrained:
            return load_checkpoint(self, pretrained, use_ema=True, strict=False)
        self._init_weights()
        return None

    def forward(self, x):
        if x.is_cuda:
            return self.hip_runner().forward_features(x)
        if self.quantized:
            x = self.quant(x)
        x = self.conv1(x)
        x1 = self.layer1(x)
        x2 = self.layer2(x1)
        x3 = self.layer3(x2)
        x4 = self.layer4(x3)
        x5 = self.layer5(x4)
        feats = [x1, x2, x3, x5]
        if self.quantized:
            feats = [self.dequant(f) for f in feats]
        return feats

    def _freeze_stages(self):
        """frostnet_features.py:354-359: every BatchNorm2d (also the `.bn` of a fused QAT conv) goes to eval mode, i.e. normalises
        with its running statistics and stops updating them.  The QAT-prepared model honours this per layer on the HIP path (training forward in eval
        form + the frozen-statistics backward, engine.Engine._frozen_after_reduce; tests/test_gpu_round4.py); the float HIP path refuses it."""
        for bn in (m for m in self.modules() if isinstance(m, nn.BatchNorm2d)):
            bn.eval()
