"""Oracle detection loss (CPU, fp32).  TEST INFRASTRUCTURE ONLY.

Restates utils/loss.py (v8DetectionLoss :152-255, BboxLoss :86-108, DFLoss :65-83),
utils/tal.py (TaskAlignedAssigner :14-327, bbox2dist :391) and utils/metrics.py (bbox_iou :74-134)
of the reference, in the straightforward boolean-mask form the reference uses.
"""
import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F

from .modules import dist2bbox, make_anchors

DEFAULT_HYP = SimpleNamespace(box=7.5, cls=0.5, dfl=1.5)  # cfg/default.yaml:98-100


def bbox_ciou(b1, b2, eps=1e-7):
    """CIoU of xyxy boxes, last dim 4.  utils/metrics.py:74-134 with xywh=False, CIoU=True."""
    x1, y1, x2, y2 = b1.chunk(4, -1)
    X1, Y1, X2, Y2 = b2.chunk(4, -1)
    w1, h1 = x2 - x1, y2 - y1 + eps
    w2, h2 = X2 - X1, Y2 - Y1 + eps
    inter = (torch.minimum(x2, X2) - torch.maximum(x1, X1)).clamp(min=0) * (
        torch.minimum(y2, Y2) - torch.maximum(y1, Y1)
    ).clamp(min=0)
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    cw = torch.maximum(x2, X2) - torch.minimum(x1, X1)
    chh = torch.maximum(y2, Y2) - torch.minimum(y1, Y1)
    c2 = cw**2 + chh**2 + eps
    rho2 = ((X1 + X2 - x1 - x2) ** 2 + (Y1 + Y2 - y1 - y2) ** 2) / 4
    v = (4 / math.pi**2) * (torch.atan(w2 / h2) - torch.atan(w1 / h1)) ** 2
    with torch.no_grad():
        alpha = v / (v - iou + (1 + eps))
    return iou - (rho2 / c2 + v * alpha)


def bbox2dist(anchor_points, bbox, reg_max):
    """utils/tal.py:391-394."""
    x1y1, x2y2 = bbox.chunk(2, -1)
    return torch.cat((anchor_points - x1y1, x2y2 - anchor_points), -1).clamp(0, reg_max - 0.01)


class TaskAlignedAssigner:
    """utils/tal.py:14-327 (topk=10, alpha=0.5, beta=6.0 as constructed by loss.py:169)."""

    def __init__(self, topk=10, num_classes=80, alpha=0.5, beta=6.0, eps=1e-9):
        self.topk, self.nc, self.alpha, self.beta, self.eps = topk, num_classes, alpha, beta, eps

    @torch.no_grad()
    def __call__(self, pd_scores, pd_bboxes, anc_points, gt_labels, gt_bboxes, mask_gt):
        bs, na = pd_scores.shape[:2]
        nmax = gt_bboxes.shape[1]
        if nmax == 0:  # tal.py:66-73
            z = torch.zeros_like(pd_scores[..., 0])
            return torch.zeros_like(pd_bboxes), torch.zeros_like(pd_scores), z.bool()

        # anchors inside each gt box: tal.py:285-302
        lt, rb = gt_bboxes.view(-1, 1, 4).chunk(2, 2)
        deltas = torch.cat((anc_points[None] - lt, rb - anc_points[None]), dim=2).view(bs, nmax, na, -1)
        mask_in_gts = deltas.amin(3).gt(self.eps).to(pd_scores.dtype)

        # alignment metric: tal.py:147-183
        m = (mask_in_gts * mask_gt).bool()
        overlaps = torch.zeros(bs, nmax, na, dtype=pd_bboxes.dtype)
        bbox_scores = torch.zeros(bs, nmax, na, dtype=pd_scores.dtype)
        bi = torch.arange(bs).view(-1, 1).expand(-1, nmax)
        ci = gt_labels.squeeze(-1).long()
        bbox_scores[m] = pd_scores[bi, :, ci][m]
        pb = pd_bboxes.unsqueeze(1).expand(-1, nmax, -1, -1)[m]
        gb = gt_bboxes.unsqueeze(2).expand(-1, -1, na, -1)[m]
        overlaps[m] = bbox_ciou(gb, pb).squeeze(-1).clamp(min=0)
        align = bbox_scores.pow(self.alpha) * overlaps.pow(self.beta)

        # top-k per gt: tal.py:198-229
        topk_mask = mask_gt.expand(-1, -1, self.topk).bool()
        _, idx = torch.topk(align, self.topk, dim=-1, largest=True)
        idx = idx.masked_fill(~topk_mask, 0)
        count = torch.zeros(align.shape, dtype=torch.int8)
        ones = torch.ones_like(idx[:, :, :1], dtype=torch.int8)
        for k in range(self.topk):
            count.scatter_add_(-1, idx[:, :, k : k + 1], ones)
        count.masked_fill_(count > 1, 0)
        mask_pos = count.to(align.dtype) * mask_in_gts * mask_gt

        # one gt per anchor: tal.py:305-327
        fg = mask_pos.sum(-2)
        if fg.max() > 1:
            multi = (fg.unsqueeze(1) > 1).expand(-1, nmax, -1)
            best = overlaps.argmax(1)
            is_best = torch.zeros_like(mask_pos)
            is_best.scatter_(1, best.unsqueeze(1), 1)
            mask_pos = torch.where(multi, is_best, mask_pos).float()
            fg = mask_pos.sum(-2)
        gt_idx = mask_pos.argmax(-2)

        # targets: tal.py:231-283
        flat = gt_idx + torch.arange(bs)[:, None] * nmax
        labels = gt_labels.long().flatten()[flat].clamp(min=0)
        target_bboxes = gt_bboxes.view(-1, 4)[flat]
        target_scores = F.one_hot(labels, self.nc).to(pd_scores.dtype)
        target_scores = torch.where(fg[:, :, None] > 0, target_scores, torch.zeros_like(target_scores))

        # normalise: tal.py:114-120
        align = align * mask_pos
        pos_align = align.amax(dim=-1, keepdim=True)
        pos_ov = (overlaps * mask_pos).amax(dim=-1, keepdim=True)
        norm = (align * pos_ov / (pos_align + self.eps)).amax(-2).unsqueeze(-1)
        return target_bboxes, target_scores * norm, fg.bool()


class v8DetectionLoss:
    """utils/loss.py:152-255.  `model` needs .model[-1] (Detect) with stride/nc/reg_max."""

    def __init__(self, model, hyp=DEFAULT_HYP, tal_topk=10):
        det = model.model[-1]
        self.hyp = hyp
        self.stride = det.stride
        self.nc = det.nc
        self.reg_max = det.reg_max
        self.no = det.nc + det.reg_max * 4
        self.assigner = TaskAlignedAssigner(topk=tal_topk, num_classes=self.nc, alpha=0.5, beta=6.0)
        self.proj = torch.arange(det.reg_max, dtype=torch.float)

    def preprocess(self, targets, batch_size, scale):
        """loss.py:174-190: ragged [n,6] (img, cls, xywh normalised) -> dense [B, nmax, 5] (cls, xyxy pixels)."""
        nl, ne = targets.shape
        if nl == 0:
            return torch.zeros(batch_size, 0, ne - 1)
        img = targets[:, 0]
        counts = torch.stack([(img == j).sum() for j in range(batch_size)])
        out = torch.zeros(batch_size, int(counts.max()), ne - 1)
        for j in range(batch_size):
            sel = img == j
            n = int(sel.sum())
            if n:
                out[j, :n] = targets[sel, 1:]
        xy, wh = out[..., 1:3] * scale[:2], out[..., 3:5] * scale[2:]
        out[..., 1:5] = torch.cat((xy - wh / 2, xy + wh / 2), -1)  # xywh2xyxy, utils/ops.py:432-449
        return out

    def __call__(self, preds, batch):
        feats = preds[1] if isinstance(preds, tuple) else preds
        B = feats[0].shape[0]
        cat = torch.cat([f.reshape(B, self.no, -1) for f in feats], 2)
        pred_distri, pred_scores = cat.split((self.reg_max * 4, self.nc), 1)
        pred_scores = pred_scores.permute(0, 2, 1).contiguous()
        pred_distri = pred_distri.permute(0, 2, 1).contiguous()
        dtype = pred_scores.dtype
        imgsz = torch.tensor(feats[0].shape[2:], dtype=dtype) * self.stride[0]
        anchor_points, stride_tensor = make_anchors(feats, self.stride, 0.5)

        targets = torch.cat((batch["batch_idx"].view(-1, 1), batch["cls"].view(-1, 1), batch["bboxes"]), 1)
        targets = self.preprocess(targets.float(), B, imgsz[[1, 0, 1, 0]])
        gt_labels, gt_bboxes = targets.split((1, 4), 2)
        mask_gt = gt_bboxes.sum(2, keepdim=True).gt(0.0).to(dtype)

        # DFL decode: loss.py:192-199
        b, a, c = pred_distri.shape
        dist = pred_distri.view(b, a, 4, c // 4).softmax(3).matmul(self.proj.to(dtype))
        pred_bboxes = dist2bbox(dist, anchor_points, xywh=False)

        target_bboxes, target_scores, fg_mask = self.assigner(
            pred_scores.detach().sigmoid(),
            (pred_bboxes.detach() * stride_tensor).to(gt_bboxes.dtype),
            anchor_points * stride_tensor,
            gt_labels,
            gt_bboxes,
            mask_gt,
        )
        tss = max(target_scores.sum(), 1)
        loss = torch.zeros(3)
        loss[1] = F.binary_cross_entropy_with_logits(pred_scores, target_scores.to(dtype), reduction="none").sum() / tss
        if fg_mask.sum():
            target_bboxes = target_bboxes / stride_tensor
            weight = target_scores.sum(-1)[fg_mask].unsqueeze(-1)
            iou = bbox_ciou(pred_bboxes[fg_mask], target_bboxes[fg_mask])
            loss[0] = ((1.0 - iou) * weight).sum() / tss
            # DFL: loss.py:65-83,101-104
            tgt = bbox2dist(anchor_points, target_bboxes, self.reg_max - 1)[fg_mask]
            pd = pred_distri[fg_mask].view(-1, self.reg_max)
            tgt = tgt.clamp(0, self.reg_max - 1 - 0.01)
            tl = tgt.long()
            tr = tl + 1
            wl = tr - tgt
            wr = 1 - wl
            dfl = (
                F.cross_entropy(pd, tl.view(-1), reduction="none").view(tl.shape) * wl
                + F.cross_entropy(pd, tr.view(-1), reduction="none").view(tl.shape) * wr
            ).mean(-1, keepdim=True)
            loss[2] = (dfl * weight).sum() / tss
        loss = loss * torch.tensor([self.hyp.box, self.hyp.cls, self.hyp.dfl])
        return loss * B, loss.detach()
