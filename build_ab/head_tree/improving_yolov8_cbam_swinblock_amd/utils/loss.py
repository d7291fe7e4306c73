"""v8 detection loss (reference: ultralytics/utils/loss.py:152-255 v8DetectionLoss, utils/tal.py:14-327
TaskAlignedAssigner, utils/metrics.py:74-134 CIoU).

The mathematics lives in csrc/loss.hip: DFL decode, task-aligned assignment, CIoU / DFL / BCE sums and their
gradients run as ten small HIP launches on the NHWC maps the Detect convolutions write (no [B, 8400, .] copies,
no boolean indexing, every shape static given the maximum number of boxes per image, so the whole step can be
captured in a HIP graph).  This module only mirrors the reference's criterion interface.
"""
from types import SimpleNamespace

import torch

from .. import ops

DEFAULT_HYP = SimpleNamespace(box=7.5, cls=0.5, dfl=1.5)  # reference cfg/default.yaml:98-100


class SplitPreds:
    """train-mode Detect output before the channel concat: box[i] [B, 64, H, W], cls[i] [B, nc, H, W]."""

    def __init__(self, box, cls):
        self.box, self.cls = box, cls


def _nhwc_map(t, box):
    """the tensor itself when the kernels can read it: NHWC memory (a channel slice of a wider, padded buffer is fine: the kernels
    take a pixel stride), 16-byte-aligned rows for box maps; else a dense NHWC copy (differentiable)."""
    if ops.is_nhwc(t) and (not box or (t.data_ptr() % 16 == 0 and (ops.as_ymi(t).ld * t.element_size()) % 16 == 0)):
        return t
    return t.contiguous(memory_format=torch.channels_last)


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
        self.topk, self.alpha, self.beta = tal_topk, 0.5, 6.0  # reference loss.py:169
        self.gains = torch.tensor([self.hyp.box, self.hyp.cls, self.hyp.dfl], dtype=torch.float, device=self.device)
        self.stride_list = [float(v) for v in det.stride]  # host copy: no device->host reads on the step path
        self._scales = {}  # batch size -> device [6] = (gains * B, gains)

    def max_boxes(self, batch_idx, batch_size):
        """largest number of labels in one image (one device->host read; pass batch["max_boxes"] to avoid it)."""
        if batch_idx.numel() == 0:
            return 0
        return int(torch.bincount(batch_idx.reshape(-1).long(), minlength=batch_size).max())

    def __call__(self, preds, batch):
        if isinstance(preds, SplitPreds):
            box, cls = list(preds.box), list(preds.cls)
        else:  # list of [B, no, H, W] maps (reference layout): split the channels again (loss.py:205-207)
            feats = preds[1] if isinstance(preds, tuple) else preds
            box = [f[:, : self.reg_max * 4] for f in feats]
            cls = [f[:, self.reg_max * 4 :] for f in feats]
        if not box[0].is_cuda:
            raise RuntimeError("v8DetectionLoss runs in libyolo_mi355 kernels: predictions must be on the MI355X (cuda) device; there is no CPU path")
        box = [_nhwc_map(t, True) for t in box]
        cls = [_nhwc_map(t if t.dtype == box[0].dtype else t.to(box[0].dtype), False) for t in cls]
        B = box[0].shape[0]
        h, w = box[0].shape[2:]
        s0 = self.stride_list[0]
        g = batch.get("max_boxes")
        if g is None:
            g = self.max_boxes(batch["batch_idx"], B)
        g = max(int(g), 1)  # an all-background batch still needs one (empty) slot per image
        targets = ops.detect_targets(batch["batch_idx"], batch["cls"], batch["bboxes"], B, g, w * s0, h * s0, box[0].device)
        # both results leave the last loss kernel: loss * gains * B (differentiable) and loss * gains (reference loss.py:250-255)
        scale = self._scales.get(B)
        if scale is None:
            gh = [float(self.hyp.box), float(self.hyp.cls), float(self.hyp.dfl)]
            scale = self._scales[B] = torch.tensor([g * B for g in gh] + gh, dtype=torch.float32, device=box[0].device)
        return ops.detect_loss(box, cls, self.stride_list, targets, scale, self.topk, self.alpha, self.beta)
