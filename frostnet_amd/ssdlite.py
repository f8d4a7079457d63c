"""SSDLite detector on the FrostNet feature backbone -- BASELINE.json config c5 ("SSDLite-FrostNet backbone 512x512 QAT detection").

The reference never wired FrostNet into its detector (Object_Detection/ssd_qmv2.py hard-codes MobileNetV2 @300), so this is the
composition SURVEY N3 asks for, following the reference's conventions:
  * one QuantStub at the image, one DeQuantStub per prediction map (ssd_qmv2.py:205-216,249-252);
  * sources = backbone stages at strides 8 / 16 / 32 (x2, x3, x5 of frostnet_features.py:342-352) + extra stages at 64 / 128 / 256;
  * SSDLite form (MobileNetV2 paper, sec. 6.3): every regular conv of the SSD extras / prediction layers becomes depthwise 3x3 + pointwise 1x1 --
    exactly the two conv kinds the Frost bottleneck is made of, so the whole detector runs on the fake-quantised HIP engine
    (ConvBNReLU / ConvBN of frostnet.py:14-60, BN in every layer like the reference's `ConvBN` head layers, ssd_qmv2.py:56-77);
  * PriorBox (layers/functions/prior_box.py:28-55) and MultiBoxLoss (layers/modules/multibox_loss.py:48-117, layers/box_utils.py:71-139)
    restated in vectorised, device-agnostic torch (they are the caller's loss, not conv math) and pinned to the reference by goldens.
Class-score maps are padded to a multiple of 8 channels (the HIP backward's channel granularity); the padding is sliced off before the loss."""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .frostnet import _SETTINGS, CascadePreExBottleneck, ConvBN, ConvBNReLU, _FrostBase, _make_divisible

# 512x512 analogue of the reference's SSD300 VOC table (data/config.py:18-34): same aspect ratios / variances, sizes scaled by 512/300
SSD512_VOC = dict(num_classes=21, min_dim=512, feature_maps=[64, 32, 16, 8, 4, 2], steps=[8, 16, 32, 64, 128, 256],
                  min_sizes=[51, 102, 189, 276, 363, 450], max_sizes=[102, 189, 276, 363, 450, 537],
                  aspect_ratios=[[2], [2, 3], [2, 3], [2, 3], [2], [2]], variance=[0.1, 0.2], clip=True, name="VOC")


def ssd_cfg_for(res):
    """SSD512_VOC rescaled to a square input of `res` pixels (tests / rehearsals at a smaller resolution): the feature maps follow the network's
    stride-8 ... stride-256 sources (ceil division: every stride-2 conv has padding 1), the box sizes keep their fraction of the image."""
    if res == 512:
        return dict(SSD512_VOC)
    cfg = dict(SSD512_VOC, min_dim=res, feature_maps=[-(-res // s) for s in SSD512_VOC["steps"]])
    cfg["min_sizes"] = [v * res / 512.0 for v in SSD512_VOC["min_sizes"]]
    cfg["max_sizes"] = [v * res / 512.0 for v in SSD512_VOC["max_sizes"]]
    return cfg


def prior_boxes(cfg):
    """PriorBox.get_prior (layers/functions/prior_box.py:28-55): [sum_k f_k^2 * (2 + 2*len(ar_k)), 4] boxes (cx, cy, w, h) in [0, 1]."""
    size = cfg["min_dim"]
    out = []
    for k, f in enumerate(cfg["feature_maps"]):
        f_k = size / cfg["steps"][k]
        s_k = cfg["min_sizes"][k] / size
        s_kp = math.sqrt(s_k * (cfg["max_sizes"][k] / size))
        wh = [(s_k, s_k), (s_kp, s_kp)]
        for ar in cfg["aspect_ratios"][k]:
            r = math.sqrt(ar)
            wh += [(s_k * r, s_k / r), (s_k / r, s_k * r)]
        c = (torch.arange(f, dtype=torch.float64) + 0.5) / f_k
        cy, cx = torch.meshgrid(c, c, indexing="ij")                        # row-major: i (y) outer, j (x) inner, like product(range(f), repeat=2)
        whs = torch.tensor(wh, dtype=torch.float64)
        box = torch.cat([cx.reshape(-1, 1, 1).expand(-1, len(wh), 1), cy.reshape(-1, 1, 1).expand(-1, len(wh), 1),
                         whs.unsqueeze(0).expand(f * f, -1, -1)], 2)
        out.append(box.reshape(-1, 4))
    out = torch.cat(out).to(torch.float32)
    return out.clamp_(0, 1) if cfg["clip"] else out


def pad_targets(targets, device):
    """Ragged ground truth (list of [n_i, 5] rows x1, y1, x2, y2, label) -> (boxes [N, K, 5], valid [N, K]) with K = max n_i.  The row counts are
    host knowledge (tensor shapes), so this is one concatenation and one gather on the device -- no per-image work, no synchronisation."""
    lens = [int(t.size(0)) for t in targets]
    k = max(1, max(lens))
    flat = torch.cat([t.reshape(-1, 5).to(device) for t in targets] + [torch.zeros(1, 5, device=device)])     # last row = the padding row
    idx = torch.full((len(lens), k), flat.size(0) - 1, dtype=torch.long)
    off = 0
    for i, n in enumerate(lens):
        idx[i, :n] = torch.arange(off, off + n)
        off += n
    valid = torch.tensor([[j < n for j in range(k)] for n in lens], dtype=torch.bool)
    return flat[idx.to(device)], valid.to(device)


def match_priors(threshold, truths, valid, priors, variances, labels):
    """layers/box_utils.py:71-139 `match` + `encode`, for the whole batch at once: every ground-truth box keeps its best prior, every prior takes
    its best ground truth; overlap < threshold -> background (label 0).  truths [N,K,4], valid [N,K], labels [N,K] -> (loc_t [N,P,4], conf_t [N,P]).
    Padding rows overlap nothing (-1) and write to a spare column P."""
    n, k, P = truths.size(0), truths.size(1), priors.size(0)
    pp = torch.cat([priors[:, :2] - priors[:, 2:] / 2, priors[:, :2] + priors[:, 2:] / 2], 1)                 # point_form
    lt = torch.max(truths[:, :, None, :2], pp[None, None, :, :2])
    rb = torch.min(truths[:, :, None, 2:], pp[None, None, :, 2:])
    inter = (rb - lt).clamp(min=0).prod(3)
    area_t = ((truths[..., 2] - truths[..., 0]) * (truths[..., 3] - truths[..., 1]))[:, :, None]
    area_p = ((pp[:, 2] - pp[:, 0]) * (pp[:, 3] - pp[:, 1]))[None, None, :]
    ov = torch.where(valid[:, :, None], inter / (area_t + area_p - inter), inter.new_full((), -1.0))          # jaccard [N,K,P]
    best_prior_idx = torch.where(valid, ov.argmax(2), torch.full_like(ov[:, :, 0], P, dtype=torch.long))      # [N,K]
    bto, bti = ov.max(1)                                                                                      # [N,P]
    bto = torch.cat([bto, bto.new_zeros(n, 1)], 1).scatter_(1, best_prior_idx, 2.0)
    bti = torch.cat([bti, bti.new_zeros(n, 1)], 1)
    for j in range(k):                                      # (sequential on purpose: a later ground truth wins a shared prior, as in the reference)
        bti.scatter_(1, best_prior_idx[:, j:j + 1], j)
    bto, bti = bto[:, :P], bti[:, :P]
    matched = truths.gather(1, bti[:, :, None].expand(-1, -1, 4))
    conf = torch.where(bto < threshold, torch.zeros_like(bti), labels.gather(1, bti).long() + 1)
    g_cxcy = ((matched[..., :2] + matched[..., 2:]) / 2 - priors[None, :, :2]) / (variances[0] * priors[None, :, 2:])
    g_wh = torch.log((matched[..., 2:] - matched[..., :2]) / priors[None, :, 2:]) / variances[1]
    return torch.cat([g_cxcy, g_wh], 2), conf


_MBOX_HIP = os.environ.get("FROST_MBOX_HIP", "1") != "0"          # dev switch: the loss as HIP kernels on the device (default) or as torch stages


class _MBoxFunction(torch.autograd.Function):
    """MultiBoxLoss forward + backward on the device (csrc/frost_mbox.hip): (loss_l, loss_c) = f(loc, conf); nothing synchronises with the host."""

    @staticmethod
    def forward(ctx, loc, conf, priors, boxes, valid, threshold, negpos, variance):
        from ._lib import call, load_library, ptr, stream
        lib = load_library()
        n, p, c, k = loc.size(0), loc.size(1), conf.size(2), boxes.size(1)
        dev = loc.device
        loc_c, conf_c = loc.detach().contiguous().float(), conf.detach().contiguous().float()
        pri = priors.detach().contiguous().float()
        bx = boxes.detach().contiguous().float()
        vd = valid.detach().contiguous().to(torch.uint8) if valid.dtype != torch.bool else valid.detach().contiguous()
        bto = torch.empty(n, p, dtype=torch.float32, device=dev)
        bti = torch.empty(n, p, dtype=torch.int32, device=dev)
        loc_t = torch.empty(n, p, 4, dtype=torch.float32, device=dev)
        conf_t = torch.empty(n, p, dtype=torch.int32, device=dev)
        lc = torch.empty(n, p, dtype=torch.float32, device=dev)
        sel = torch.empty(n, p, dtype=torch.uint8, device=dev)
        num_pos = torch.empty(n, dtype=torch.int32, device=dev)
        out = torch.zeros(lib.frost_mbox_workspace_floats(), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            call("frost_mbox_forward", ptr(loc_c), ptr(conf_c), ptr(pri), ptr(bx), ptr(vd), n, p, c, k, float(threshold), int(negpos), float(variance[0]), float(variance[1]),
                 ptr(bto), ptr(bti), ptr(loc_t), ptr(conf_t), ptr(lc), ptr(sel), ptr(num_pos), ptr(out), stream())
        ctx.save_for_backward(loc_c, conf_c, loc_t, conf_t, sel, out)
        ctx.conf_t, ctx.num_pos = conf_t, num_pos          # (tests read the matching)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_l, g_c):
        from ._lib import call, ptr, stream
        loc, conf, loc_t, conf_t, sel, out = ctx.saved_tensors
        n, p, c = conf.shape
        dloc, dconf = torch.empty_like(loc), torch.empty_like(conf)
        gl = g_l.contiguous().float() if g_l is not None else None
        gc = g_c.contiguous().float() if g_c is not None else None
        with torch.cuda.device(loc.device):
            call("frost_mbox_backward", ptr(loc), ptr(conf), ptr(loc_t), ptr(conf_t), ptr(sel), ptr(out), ptr(gl), ptr(gc), n, p, c, ptr(dloc), ptr(dconf), stream())
        return dloc, dconf, None, None, None, None, None, None


class MultiBoxLoss(nn.Module):
    """SSD loss (layers/modules/multibox_loss.py:48-117): smooth-L1 on the matched priors + cross-entropy on positives and the hardest
    negatives (3:1), both divided by the number of positives.  forward((loc [N,P,4], conf [N,P,C], priors [P,4]), targets) with targets the
    reference's list of [n_i,5], or pad_targets(...) of it made ahead of time.  Fixed shapes and no host synchronisation throughout (masked sums
    instead of boolean gathers), so the loss records into a HIP graph with the rest of the step."""

    def __init__(self, num_classes, overlap_thresh=0.5, neg_pos=3, variance=(0.1, 0.2)):
        super().__init__()
        self.num_classes, self.threshold, self.negpos_ratio, self.variance = num_classes, overlap_thresh, neg_pos, variance

    def forward(self, predictions, targets):
        loc_data, conf_data, priors = predictions
        num, num_priors = loc_data.size(0), loc_data.size(1)
        priors = priors[:num_priors].to(loc_data.device)
        boxes, valid = targets if isinstance(targets, tuple) else pad_targets(targets, loc_data.device)
        if loc_data.is_cuda and _MBOX_HIP:
            # the device path: matching, per-prior loss, hard negative mining and both gradients as four HIP kernels (csrc/frost_mbox.hip); the torch stages below
            # are the CPU definition (and the yardstick of tests/test_gpu_detect.py)
            return _MBoxFunction.apply(loc_data, conf_data, priors, boxes, valid, self.threshold, self.negpos_ratio, self.variance)
        return self.forward_torch(loc_data, conf_data, priors, boxes, valid)

    def forward_torch(self, loc_data, conf_data, priors, boxes, valid):
        """The loss as batched torch stages (the restatement of multibox_loss.py:48-117 pinned to the reference's goldens, tests/test_oracle_golden.py G11)."""
        with torch.no_grad():
            loc_t, conf_t = match_priors(self.threshold, boxes[..., :4], valid, priors, self.variance, boxes[..., 4])
            pos = conf_t > 0
            num_pos = pos.long().sum(1, keepdim=True)
        zero = loc_data.new_zeros(())
        loss_l = torch.where(pos[:, :, None], F.smooth_l1_loss(loc_data, loc_t, reduction="none"), zero).sum()
        # per-prior classification loss: log-sum-exp minus the target's score (== cross-entropy; the reference's max-shifted log_sum_exp)
        lc = torch.logsumexp(conf_data, 2) - conf_data.gather(2, conf_t[:, :, None]).squeeze(2)
        with torch.no_grad():                                   # hard negative mining: rank the non-positive priors by their loss
            _, loss_idx = lc.detach().masked_fill(pos, 0).sort(1, descending=True)
            _, idx_rank = loss_idx.sort(1)
            num_neg = torch.clamp(self.negpos_ratio * num_pos, max=pos.size(1) - 1)
            sel = pos | (idx_rank < num_neg)
        loss_c = torch.where(sel, lc, zero).sum()
        n = num_pos.sum().to(loss_l.dtype)
        return loss_l / n, loss_c / n


class ExtraBlock(nn.Module):
    """SSDLite extra stage: 1x1 (-> c/2) -> depthwise 3x3 stride 2 -> 1x1 (-> c), ReLU after each (the reference applies F.relu after every
    extra layer, ssd_qmv2.py:240-243)."""

    def __init__(self, cin, cout):
        super().__init__()
        mid = cout // 2
        self.pw1 = ConvBNReLU(cin, mid, 1)
        self.dw = ConvBNReLU(mid, mid, 3, 2, 1, 1, groups=mid)
        self.pw2 = ConvBNReLU(mid, cout, 1)

    def forward(self, x):
        return self.pw2(self.dw(self.pw1(x)))


class SepHead(nn.Module):
    """SSDLite prediction layer: depthwise 3x3 + BN + ReLU, then 1x1 + BN (linear), like the reference's ConvBN head layers but separable."""

    def __init__(self, cin, cout):
        super().__init__()
        self.dw = ConvBNReLU(cin, cin, 3, 1, 1, 1, groups=cin)
        self.pw = ConvBN(cin, cout, 1)

    def forward(self, x):
        return self.pw(self.dw(x))


class SSDLiteFrostNet(_FrostBase):
    ANCHORS = [4, 6, 6, 6, 4, 4]
    EXTRAS = [512, 256, 256]

    def __init__(self, num_classes=21, mode="large", width_mult=1.0, cfg=None, bottleneck=CascadePreExBottleneck):
        super().__init__()
        self.quantized = True
        self.num_classes, self.cfg = num_classes, dict(cfg or SSD512_VOC)
        l1, l2, l3, l4, l5 = _SETTINGS[mode]
        self.in_channels = _make_divisible(int(32 * min(1.0, width_mult)))
        self.conv1 = ConvBNReLU(3, self.in_channels, 3, 2, 1)
        chans = []
        for i, st in enumerate((l1, l2, l3, l4, l5)):
            setattr(self, f"layer{i + 1}", self._make_layer(bottleneck, st, width_mult, 1))
            chans.append(self.in_channels)
        src = [chans[1], chans[2], chans[4]]
        self.extras = nn.ModuleList()
        cin = chans[4]
        for c in self.EXTRAS:
            self.extras.append(ExtraBlock(cin, c))
            src.append(c)
            cin = c
        self.source_channels = src
        self.conf_pad = [(a * num_classes + 7) // 8 * 8 for a in self.ANCHORS]
        self.loc = nn.ModuleList([SepHead(c, a * 4) for c, a in zip(src, self.ANCHORS)])
        self.conf = nn.ModuleList([SepHead(c, p) for c, p in zip(src, self.conf_pad)])
        self.quant = torch.quantization.QuantStub()
        self.dequant = torch.quantization.DeQuantStub()
        self._init_weights()
        self.register_buffer("priors", prior_boxes(self.cfg), persistent=False)

    def _assemble(self, maps):
        """12 dequantised NCHW maps [loc0, conf0, loc1, ...] -> (loc [N,P,4], conf [N,P,C], priors)."""
        n = maps[0].size(0)
        loc = torch.cat([m.permute(0, 2, 3, 1).reshape(n, -1) for m in maps[0::2]], 1).view(n, -1, 4)
        conf = torch.cat([m.permute(0, 2, 3, 1)[..., : a * self.num_classes].reshape(n, -1) for m, a in zip(maps[1::2], self.ANCHORS)], 1)
        return loc, conf.view(n, -1, self.num_classes), self.priors

    def forward(self, x):
        if x.is_cuda:
            return self._assemble(self.hip_runner().forward_maps(x))
        x = self.conv1(self.quant(x))
        sources = []
        for i in range(5):
            x = getattr(self, f"layer{i + 1}")(x)
            if i in (1, 2, 4):
                sources.append(x)
        for e in self.extras:
            x = e(x)
            sources.append(x)
        maps = []
        for s, l, c in zip(sources, self.loc, self.conf):
            maps += [self.dequant(l(s)), self.dequant(c(s))]
        return self._assemble(maps)

    def hip_runner(self):
        r = self.__dict__.get("_hip_runner")
        if r is None or r.model is not self or not r.still_valid():
            if not self._is_qat_prepared():
                raise NotImplementedError("the SSDLite detector runs on the HIP device in fake-quant (QAT-prepared) mode: fuse_model() + prepare_qat first")
            from .runner import SSDRunner
            r = SSDRunner(self)
            r.is_qat = True
            self.__dict__["_hip_runner"] = r
        return r
