"""Caller-side pieces of the hot path that the reference keeps in its (non-importable) training scripts, restated so a
`Classification/train.py`-style driver runs on the HIP path unchanged in behaviour:

  * `make_param_groups`  -- Classification/train.py:121-137 (one group per tensor; depthwise wd 0, other 4-D wd, rest 0.01*wd)
  * `adjust_learning_rate_cosine` -- Classification/utils/helper_functions.py:231-261 (per-iteration cosine + linear warm-up)
  * `statassist_qat_switch` -- Classification/train.py:149-173 (StatAssist FP epoch(s) with the SAME optimizer, flip
    `is_warmup`, then fuse + prepare_qat keeping Parameter identity so optimizer state survives)
  * `train_one_iter` / `accuracy` -- helper_functions.py:32-46,118-155 (accuracy() fixed for torch>=1.7: reshape, not view)
  * `train` / `val` -- the epoch loops, helper_functions.py:99-163 / 306-350: mean over the iterations of the per-batch loss / top-1 / top-5.  `val`
    only switches to `model.eval()` -- as in the reference the observers stay live and keep moving with validation data.
    The running sums stay on the device: ONE host read per epoch instead of the reference's three `.item()` per iteration.
  * `checkpoint_state` / `save_checkpoint` / `load_checkpoint` -- Classification/train.py:193-223, helper_functions.py:400-407: the reference's
    dictionary keys, plus `hip_rng` (dropout Philox seed + draw count; the GradBoost stream resumes from the optimizer's step count).
"""
import math

import torch


class _CEFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        from ._lib import call, ptr, stream
        n, c = logits.shape
        x = logits.contiguous().float()
        loss = torch.zeros((), dtype=torch.float32, device=x.device)
        dlogits = torch.empty_like(x)
        call("frost_softmax_ce", ptr(x), ptr(target.contiguous()), n, c, 1.0 / n, ptr(loss), ptr(dlogits), stream())
        ctx.save_for_backward(dlogits)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        return dlogits * g, None


class CrossEntropyLoss(torch.nn.Module):
    """`nn.CrossEntropyLoss()` (mean reduction; Classification/train.py:147) with forward and backward in ONE hand-written kernel on the
    device (`frost_softmax_ce`): loss and dlogits come out of the same pass, no host synchronisation (capturable).  Every target must be a
    valid class index (the reference's loaders never produce ignore_index); CPU tensors take torch's implementation."""

    def forward(self, logits, target):
        if not logits.is_cuda:
            return torch.nn.functional.cross_entropy(logits, target)
        if target.dtype != torch.int64 or target.dim() != 1 or logits.dim() != 2:
            raise ValueError("CrossEntropyLoss expects (N, C) logits and (N,) int64 class indices")
        return _CEFunction.apply(logits, target)


def make_param_groups(model, weight_decay):
    groups = []
    others = weight_decay * 0.01
    for _, value in model.named_parameters():
        if len(value.shape) == 4:
            groups.append({"params": [value], "weight_decay": 0.0 if value.shape[1] == 1 else weight_decay})
        else:
            groups.append({"params": [value], "weight_decay": others})
    return groups


def cosine_lr(lr, warmup_lr, warmup_epochs, epochs, epoch, it, dataset_len):
    total_iter = (epochs - warmup_epochs) * dataset_len
    current_iter = it + (epoch - warmup_epochs) * dataset_len
    if epoch < warmup_epochs:
        return warmup_lr + (lr - warmup_lr) * float(it + epoch * dataset_len) / (warmup_epochs * dataset_len)
    return lr / 2 * (math.cos(math.pi * current_iter / total_iter) + 1)


def adjust_learning_rate_cosine(optimizer, epoch, it, dataset_len, args):
    """args: .anneal, .restart_epochs, .epochs, .warmup_epochs, .warmup_lr, .lr (same attribute names as the reference)."""
    if getattr(args, "anneal", False):
        epoch = epoch % args.restart_epochs
        epochs = args.restart_epochs
    else:
        epochs = args.epochs
    lr = cosine_lr(args.lr, args.warmup_lr, args.warmup_epochs, epochs, epoch, it, dataset_len)
    for group in optimizer.param_groups:
        group["lr"] = lr
    return lr


def accuracy(output, target, topk=(1,)):
    with torch.no_grad():
        maxk = max(topk)
        _, pred = output.topk(maxk, 1, True, True)
        correct = pred.t().eq(target.view(1, -1).expand(maxk, -1))
        return [correct[:k].reshape(-1).float().sum(0, keepdim=True).mul_(100.0 / target.size(0)) for k in topk]


def train_one_iter(model, criterion, optimizer, x, target, grad_sync=None):
    """zero_grad -> forward -> loss -> backward -> (DP all-reduce) -> step   (helper_functions.py:139-143)."""
    optimizer.zero_grad(set_to_none=True)
    out = model(x)
    loss = criterion(out, target)
    loss.backward()
    if grad_sync is not None:
        grad_sync.finish()
    optimizer.step()
    return loss, out


def statassist_qat_switch(model, optimizer, qconfig_version=0, backend="qnnpack"):
    """After the FP warm-up epoch(s): turn GradBoost noise on and switch the model to fake-quant mode.  prepare_qat reuses
    the same Parameter objects, so the optimizer's moments / exp_max statistics stay attached."""
    from .frostnet import qat_prepare
    if hasattr(optimizer, "is_warmup"):
        optimizer.is_warmup = False
    before = [id(p) for p in model.parameters()]
    qat_prepare(model, version=qconfig_version, backend=backend)
    assert before == [id(p) for p in model.parameters()], "prepare_qat must keep parameter identity"
    return model


def _device_of(model):
    return next(model.parameters()).device


def _epoch(loader, model, criterion, optimizer, epoch, args, grad_sync, train_mode):
    dev = _device_of(model)
    sums, n = torch.zeros(3, dtype=torch.float64, device=dev), 0
    for i, (inp, target) in enumerate(loader):
        inp, target = inp.to(dev, non_blocking=True), target.to(dev, non_blocking=True)
        if train_mode:
            if args is not None and getattr(args, "lrsch", "cos_lr") == "cos_lr" and hasattr(args, "dataset_len"):
                adjust_learning_rate_cosine(optimizer, epoch, i, args.dataset_len, args)
            loss, out = train_one_iter(model, criterion, optimizer, inp, target, grad_sync)
        else:
            with torch.no_grad():
                out = model(inp)
                loss = criterion(out, target)
        acc1, acc5 = accuracy(out, target, topk=(1, min(5, out.shape[1])))
        sums += torch.stack([loss.detach().double().reshape(()), acc1[0].double(), acc5[0].double()])
        n += 1
    if n == 0:
        raise ValueError("empty loader")
    loss, a1, a5 = (sums / n).tolist()           # the epoch's single device -> host read
    return loss, a1, a5


def train(train_loader, model, criterion, optimizer, epoch, total_ep=None, args=None, grad_sync=None):
    """One training epoch (helper_functions.py:99-163): returns (mean loss, mean top-1, mean top-5) over the iterations."""
    model.train()
    return _epoch(train_loader, model, criterion, optimizer, epoch, args, grad_sync, True)


def val(val_loader, model, criterion):
    """One validation pass (helper_functions.py:306-350): `model.eval()` only -- BatchNorm uses its running statistics while every FakeQuantize
    observer that has not been disabled explicitly keeps updating, exactly as the reference's val() leaves them."""
    model.eval()
    return _epoch(val_loader, model, criterion, None, 0, None, None, False)


def checkpoint_state(model, optimizer, epoch, **scalars):
    """The dictionary Classification/train.py:208-218 saves every epoch (same keys; `scalars` = lossTr, lossVal, acc1Tr, ... , lr, Max_name ...)."""
    state = {"epoch": epoch + 1, "arch": str(type(model).__name__), "state_dict": model.state_dict(), "optimizer": optimizer.state_dict()}
    state.update(scalars)
    runner = getattr(model, "_hip_runner", None)
    if runner is not None and hasattr(runner, "rng_state"):
        state["hip_rng"] = runner.rng_state()
    return state


def save_checkpoint(state, filenameCheckpoint="checkpoint.pth.tar"):
    torch.save(state, filenameCheckpoint)


def load_checkpoint(model, optimizer, filenameCheckpoint="checkpoint.pth.tar", map_location=None):
    """Resume: parameters / buffers (BN statistics, observer ranges, qparams), optimizer state (GradBoost statistics, step count = Philox offset)
    and the dropout stream.  Returns the checkpoint dictionary (epoch, best accuracies ...)."""
    ck = torch.load(filenameCheckpoint, map_location=map_location, weights_only=False)
    model.load_state_dict(ck["state_dict"])
    if optimizer is not None and "optimizer" in ck:
        optimizer.load_state_dict(ck["optimizer"])
    if "hip_rng" in ck and hasattr(model, "hip_runner"):
        dev = _device_of(model)
        if dev.type == "cuda":
            r = model.hip_runner()
            if hasattr(r, "set_rng_state"):
                r.set_rng_state(ck["hip_rng"])
    return ck
