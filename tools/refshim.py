"""Dev-container-only loader for the upstream reference (never shipped, never imported by the product).

Imports `/root/reference/{frostnet,frostnet_features,optimizer}.py` *in place* (no copy) behind the
four shims SURVEY.md Appendix D describes, so `tools/gen_golden.py` can run the reference and write
golden input/output vectors into `tests/golden/`.  Nothing under `frostnet_amd/`, `oracle/`, `tests/`,
`bench.py` or `__graft_entry__.py` imports this file; the GPU box has no `/root/reference`.
"""
import importlib.util
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import torch

REF = "/root/reference"
_REGISTRY = {}


def _install_timm_shim():
    if "timm" in sys.modules:
        return
    timm = types.ModuleType("timm")
    data = types.ModuleType("timm.data")
    data.IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
    data.IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)
    data.IMAGENET_INCEPTION_MEAN = (0.5, 0.5, 0.5)
    data.IMAGENET_INCEPTION_STD = (0.5, 0.5, 0.5)
    models = types.ModuleType("timm.models")
    registry = types.ModuleType("timm.models.registry")

    def register_model(fn):
        _REGISTRY[fn.__name__] = fn
        return fn

    registry.register_model = register_model
    timm.data, timm.models, models.registry = data, models, registry
    sys.modules.update({"timm": timm, "timm.data": data, "timm.models": models,
                        "timm.models.registry": registry})


def _install_fuse_shim():
    """torch<=1.10 `fuse_modules` produced QAT-fusable modules in train mode; modern torch asserts."""
    import torch.ao.quantization as aoq
    if getattr(torch.quantization, "_frost_shim", False):
        return
    orig = aoq.fuse_modules

    def fuse_modules(model, names, inplace=False, **kw):
        if model.training:
            return aoq.fuse_modules_qat(model, names, inplace=inplace, **kw)
        return orig(model, names, inplace=inplace, **kw)

    torch.quantization.fuse_modules = fuse_modules
    torch.quantization._frost_shim = True


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_frostnet():
    _install_timm_shim()
    _install_fuse_shim()
    if "ref_frostnet" in sys.modules:
        return sys.modules["ref_frostnet"], _REGISTRY
    return _load("ref_frostnet", f"{REF}/frostnet.py"), _REGISTRY


def load_features():
    _install_fuse_shim()
    if "refpkg.backbones.frostnet_features" in sys.modules:
        return sys.modules["refpkg.backbones.frostnet_features"]
    pkg = types.ModuleType("refpkg"); pkg.__path__ = []
    bb = types.ModuleType("refpkg.backbones"); bb.__path__ = []
    builder = types.ModuleType("refpkg.builder")

    class _Reg:
        def register_module(self):
            return lambda cls: cls

    builder.BACKBONES = _Reg()
    sys.modules.update({"refpkg": pkg, "refpkg.backbones": bb, "refpkg.builder": builder})
    return _load("refpkg.backbones.frostnet_features", f"{REF}/frostnet_features.py")


def load_mobilenetv3():
    """The reference's MobileNetV3 baseline file: only its quantizable `_Hswish` module is used."""
    if "ref_mobilenetv3" in sys.modules:
        return sys.modules["ref_mobilenetv3"]
    return _load("ref_mobilenetv3", f"{REF}/Classification/models/imagenet/mobilenetv3.py")


def load_detection():
    """The reference's SSD loss pieces (Object_Detection/layers): PriorBox and MultiBoxLoss.  cv2 / torchvision are absent here and only
    imported at module level by files we do not execute: stub them."""
    if "layers" not in sys.modules:
        sys.modules.setdefault("cv2", types.ModuleType("cv2"))
        if "torchvision" not in sys.modules:
            tv, tvo = types.ModuleType("torchvision"), types.ModuleType("torchvision.ops")
            tv.ops, tvo.nms = tvo, None
            sys.modules["torchvision"], sys.modules["torchvision.ops"] = tv, tvo
        sys.path.insert(0, f"{REF}/Object_Detection")
    from layers.functions.prior_box import PriorBox
    from layers.modules.multibox_loss import MultiBoxLoss
    return PriorBox, MultiBoxLoss


def load_optimizer():
    if "ref_optimizer" in sys.modules:
        return sys.modules["ref_optimizer"]
    torch.Tensor.cuda = lambda self, *a, **k: self  # optimizer.py:180 `.cuda()` on a CPU box
    return _load("ref_optimizer", f"{REF}/optimizer.py")


def load_helpers():
    if "ref_helpers" in sys.modules:
        return sys.modules["ref_helpers"]
    return _load("ref_helpers", f"{REF}/Classification/utils/helper_functions.py")


def qat_prepare(model, version=0, backend="qnnpack"):
    """Classification/train.py:171-173 with the qconfig version pinned (SURVEY H-3)."""
    from torch.ao.quantization import get_default_qat_qconfig, prepare_qat
    model.train()
    model.fuse_model()
    model.qconfig = get_default_qat_qconfig(backend, version=version)
    prepare_qat(model, inplace=True)
    return model
