"""v8 detection loss on the GPU (reference: ultralytics/utils/loss.py:152-255, utils/tal.py:14-327,
utils/metrics.py:74-134).

Same mathematics as the reference, re-expressed with dense masks instead of boolean indexing and
without the per-image Python loop of `preprocess` (loss.py:185-188), so every tensor has a static
shape given the maximum number of boxes per image.  It works on [B, 8400, .] tensors - small,
latency-bound work that SURVEY.md section 8 (A10) keeps in PyTorch ops on ROCm; the maps it consumes
and the gradients it returns flow through the HIP kernels.
"""
import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F

from ..nn.modules.head import dist2bbox, make_anchors

DEFAULT_HYP = SimpleNamespace(box=7.5, cls=0.5, dfl=1.5)  # reference cfg/default.yaml:98-100


class SplitPreds:
    """train-mode Detect output before the channel concat: box[i] [B, 64, H, W], cls[i] [B, nc, H, W]."""

    def __init__(self, box, cls):
        self.box, self.cls = box, cls


def bbox_ciou(b1, b2, eps=1e-7):
    """CIoU for xyxy boxes broadcast over leading dims (reference metrics.py:74-134, xywh=False, CIoU=True)."""
    x1, y1, x2, y2 = b1.unbind(-1)
    X1, Y1, X2, Y2 = b2.unbind(-1)
    w1, h1 = x2 - x1, y2 - y1 + eps
    w2, h2 = X2 - X1, Y2 - Y1 + eps
    inter = (torch.minimum(x2, X2) - torch.maximum(x1, X1)).clamp_(0) * (torch.minimum(y2, Y2) - torch.maximum(y1, Y1)).clamp_(0)
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    cw = torch.maximum(x2, X2) - torch.minimum(x1, X1)
    ch = torch.maximum(y2, Y2) - torch.minimum(y1, Y1)
    c2 = cw.pow(2) + ch.pow(2) + eps
    rho2 = ((X1 + X2 - x1 - x2).pow(2) + (Y1 + Y2 - y1 - y2).pow(2)) / 4
    v = (4 / math.pi**2) * ((w2 / h2).atan() - (w1 / h1).atan()).pow(2)
    with torch.no_grad():
        alpha = v / (v - iou + (1 + eps))
    return iou - (rho2 / c2 + v * alpha)


class TaskAlignedAssigner:
    """reference tal.py:14-327 with topk / alpha / beta as loss.py:169 constructs it."""

    def __init__(self, topk=10, num_classes=80, alpha=0.5, beta=6.0, eps=1e-9):
        self.topk, self.nc, self.alpha, self.beta, self.eps = topk, num_classes, alpha, beta, eps

    @torch.no_grad()
    def __call__(self, pd_scores, pd_bboxes, anc_points, gt_labels, gt_bboxes, mask_gt):
        bs, na, _ = pd_scores.shape
        nmax = gt_bboxes.shape[1]
        if nmax == 0:
            return torch.zeros_like(pd_bboxes), torch.zeros_like(pd_scores), torch.zeros(bs, na, dtype=torch.bool, device=pd_scores.device)
        lt, rb = gt_bboxes[:, :, None, :2], gt_bboxes[:, :, None, 2:]
        ap = anc_points[None, None]
        mask_in = torch.minimum((ap - lt).amin(-1), (rb - ap).amin(-1)).gt(self.eps).to(pd_scores.dtype)  # [B, G, A]
        m = mask_in * mask_gt
        lab = gt_labels.squeeze(-1).long().clamp_(0, self.nc - 1)
        scores = pd_scores.transpose(1, 2).gather(1, lab[:, :, None].expand(-1, -1, na)) * m
        overlaps = bbox_ciou(gt_bboxes[:, :, None, :], pd_bboxes[:, None, :, :]).clamp_(0) * m
        align = scores.pow(self.alpha) * overlaps.pow(self.beta)
        # top-k anchors per gt (tal.py:198-229); invalid gts point all their picks at anchor 0, which the
        # ">1 hits" rule then discards exactly as the reference does
        _, idx = torch.topk(align, self.topk, dim=-1)
        idx = idx.masked_fill(~mask_gt.bool().expand(-1, -1, self.topk), 0)
        count = torch.zeros_like(align, dtype=torch.int32).scatter_add_(-1, idx, torch.ones_like(idx, dtype=torch.int32))
        count = torch.where(count > 1, torch.zeros_like(count), count)
        mask_pos = count.to(align.dtype) * mask_in * mask_gt
        # an anchor claimed by several gts goes to the one with the highest overlap (tal.py:305-327)
        fg = mask_pos.sum(-2)
        multi = (fg[:, None, :] > 1).expand(-1, nmax, -1)
        is_best = F.one_hot(overlaps.argmax(1), nmax).permute(0, 2, 1).to(mask_pos.dtype)
        mask_pos = torch.where(multi, is_best, mask_pos)
        fg = mask_pos.sum(-2)
        gt_idx = mask_pos.argmax(-2)
        labels = lab.gather(1, gt_idx)
        target_bboxes = gt_bboxes.gather(1, gt_idx[:, :, None].expand(-1, -1, 4))
        target_scores = F.one_hot(labels, self.nc).to(pd_scores.dtype) * (fg > 0)[:, :, None]
        align = align * mask_pos
        pos_align = align.amax(-1, keepdim=True)
        pos_ov = (overlaps * mask_pos).amax(-1, keepdim=True)
        norm = (align * pos_ov / (pos_align + self.eps)).amax(-2).unsqueeze(-1)
        return target_bboxes, target_scores * norm, fg > 0


class v8DetectionLoss:
    """criterion(preds, batch) -> (loss * batch_size [3], loss.detach() [3]) as reference loss.py:201-255."""

    def __init__(self, model, tal_topk=10):
        det = model.model[-1]
        self.hyp = getattr(model, "args", None) or DEFAULT_HYP
        self.stride = det.stride
        self.nc = det.nc
        self.reg_max = det.reg_max
        self.no = det.nc + det.reg_max * 4
        self.device = next(model.parameters()).device
        self.assigner = TaskAlignedAssigner(topk=tal_topk, num_classes=self.nc, alpha=0.5, beta=6.0)
        self.proj = torch.arange(det.reg_max, dtype=torch.float, device=self.device)
        self.gains = torch.tensor([self.hyp.box, self.hyp.cls, self.hyp.dfl], dtype=torch.float, device=self.device)
        self.stride_list = [float(v) for v in det.stride]  # host copy: no device->host reads on the step path
        self._geom = {}  # (level shapes) -> (anchor_points, stride_tensor, xyxy scale); built once, graph-capture safe

    def geometry(self, feats):
        key = tuple(tuple(f.shape[2:]) for f in feats)
        g = self._geom.get(key)
        if g is None:
            anchor_points, stride_tensor = make_anchors(feats, self.stride_list, 0.5)
            h, w = feats[0].shape[2:]
            s0 = self.stride_list[0]
            scale = torch.tensor([w * s0, h * s0, w * s0, h * s0], dtype=torch.float, device=self.device)
            g = self._geom[key] = (anchor_points, stride_tensor, scale)
        return g

    def preprocess(self, batch_idx, cls, bboxes, batch_size, scale, max_boxes=None):
        """ragged (img, cls, xywh-normalised) rows -> dense [B, nmax, 5] (cls, xyxy pixels), no Python loop."""
        dev = self.device
        n = batch_idx.numel()
        if n == 0:
            return torch.zeros(batch_size, 0, 5, device=dev)
        img = batch_idx.to(dev).long().view(-1)
        counts = torch.zeros(batch_size, dtype=torch.long, device=dev).scatter_add_(0, img, torch.ones_like(img))
        nmax = int(counts.max()) if max_boxes is None else int(max_boxes)  # pass batch["max_boxes"] to avoid this host sync
        order = torch.argsort(img, stable=True)
        starts = torch.cumsum(counts, 0) - counts
        pos = torch.arange(n, device=dev) - starts[img[order]]
        rows = torch.cat((cls.to(dev).float().view(-1, 1), bboxes.to(dev).float()), 1)[order]
        out = torch.zeros(batch_size, nmax, 5, device=dev)
        out[img[order], pos] = rows
        box = out[..., 1:5] * scale
        out[..., 1:5] = torch.cat((box[..., :2] - box[..., 2:] / 2, box[..., :2] + box[..., 2:] / 2), -1)
        return out

    def __call__(self, preds, batch):
        dev = self.device
        if isinstance(preds, SplitPreds):
            # fast path from Detect.forward_split: per-level box / class maps in NHWC memory -> [B, A, .] are
            # (almost) free views; gradients come back in the layout the conv backward kernels consume
            feats = preds.box
            B = feats[0].shape[0]
            pred_distri = torch.cat([b.permute(0, 2, 3, 1).reshape(B, -1, self.reg_max * 4) for b in preds.box], 1).float()
            pred_scores = torch.cat([c.permute(0, 2, 3, 1).reshape(B, -1, self.nc) for c in preds.cls], 1).float()
        else:
            feats = preds[1] if isinstance(preds, tuple) else preds
            B = feats[0].shape[0]
            cat = torch.cat([f.float().reshape(B, self.no, -1) for f in feats], 2)
            pred_distri, pred_scores = cat.split((self.reg_max * 4, self.nc), 1)
            pred_scores = pred_scores.permute(0, 2, 1).contiguous()
            pred_distri = pred_distri.permute(0, 2, 1).contiguous()
        anchor_points, stride_tensor, scale = self.geometry(feats)

        targets = self.preprocess(batch["batch_idx"], batch["cls"], batch["bboxes"], B, scale, batch.get("max_boxes"))
        gt_labels, gt_bboxes = targets.split((1, 4), 2)
        mask_gt = gt_bboxes.sum(2, keepdim=True).gt(0.0).float()

        b, a, c = pred_distri.shape
        dist = (pred_distri.view(b, a, 4, c // 4).softmax(3) * self.proj).sum(-1)  # expectation (reference: .matmul(proj))
        pred_bboxes = dist2bbox(dist, anchor_points, xywh=False)

        target_bboxes, target_scores, fg = self.assigner(
            pred_scores.detach().sigmoid(), pred_bboxes.detach() * stride_tensor, anchor_points * stride_tensor, gt_labels, gt_bboxes, mask_gt
        )
        tss = target_scores.sum().clamp(min=1.0)
        loss_cls = F.binary_cross_entropy_with_logits(pred_scores, target_scores, reduction="none").sum() / tss

        target_bboxes = target_bboxes / stride_tensor
        weight = target_scores.sum(-1)  # [B, A]; zero on background anchors
        fgf = fg.to(weight.dtype)
        iou = bbox_ciou(pred_bboxes, target_bboxes)
        loss_box = (torch.where(fg, 1.0 - iou, torch.zeros_like(iou)) * weight).sum() / tss
        # DFL (loss.py:65-83,101-104) on every anchor, masked by the foreground weight
        ltrb = torch.cat((anchor_points - target_bboxes[..., :2], target_bboxes[..., 2:] - anchor_points), -1).clamp(0, self.reg_max - 1 - 0.01)
        tl = ltrb.long()
        wl = (tl + 1).to(ltrb.dtype) - ltrb
        logp = pred_distri.view(b, a, 4, self.reg_max).log_softmax(-1)
        ce_l = -logp.gather(-1, tl.unsqueeze(-1)).squeeze(-1)
        ce_r = -logp.gather(-1, (tl + 1).unsqueeze(-1)).squeeze(-1)
        dfl = (ce_l * wl + ce_r * (1 - wl)).mean(-1)
        loss_dfl = (dfl * weight * fgf).sum() / tss

        loss = torch.stack((loss_box, loss_cls, loss_dfl)) * self.gains
        return loss * B, loss.detach()
