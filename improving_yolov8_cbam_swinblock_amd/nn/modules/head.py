"""Detect head (reference: ultralytics/nn/modules/head.py:23-183), legacy (v8) class branch."""
import math
import os

import torch
import torch.nn as nn

from ... import ops
from .block import DFL
from .conv import Conv


def make_anchors(feats, strides, grid_cell_offset=0.5):
    """anchor centres and strides per level (reference utils/tal.py:364-376)."""
    pts, st = [], []
    dtype, device = torch.float32, feats[0].device
    for f, s in zip(feats, strides):
        h, w = f.shape[2:]
        sx = torch.arange(w, device=device, dtype=dtype) + grid_cell_offset
        sy = torch.arange(h, device=device, dtype=dtype) + grid_cell_offset
        gy, gx = torch.meshgrid(sy, sx, indexing="ij")
        pts.append(torch.stack((gx, gy), -1).view(-1, 2))
        st.append(torch.full((h * w, 1), float(s), dtype=dtype, device=device))
    return torch.cat(pts), torch.cat(st)


def dist2bbox(distance, anchor_points, xywh=True, dim=-1):
    """reference utils/tal.py:379-388."""
    lt, rb = distance.chunk(2, dim)
    x1y1, x2y2 = anchor_points - lt, anchor_points + rb
    if xywh:
        return torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), dim)
    return torch.cat((x1y1, x2y2), dim)


class _Out1x1(nn.Conv2d):
    """Detect's biased 1x1 output conv: an nn.Conv2d (same state-dict keys) whose forward is the MFMA kernel."""

    def forward(self, x):
        return ops.conv_affine_act(ops.to_internal(x), self.weight, None, self.bias, 1, ops.ACT_NONE, None, pad_out=True)


class Detect(nn.Module):
    """per level: cat(cv2[i](x), cv3[i](x)); train -> list of [B, 4*16+nc, H, W]; eval -> (decoded, list)."""

    dynamic = False
    export = False
    format = None
    end2end = False
    max_det = 300
    shape = None
    anchors = torch.empty(0)
    strides = torch.empty(0)
    legacy = True  # parse_model sets True for v8 YAMLs (reference tasks.py:1355,1488); only that branch is built

    def __init__(self, nc=80, ch=()):
        super().__init__()
        if not self.legacy:
            raise NotImplementedError("only the legacy (YOLOv8) class branch of Detect is on the hot path")
        self.nc = nc
        self.nl = len(ch)
        self.reg_max = 16
        self.no = nc + self.reg_max * 4
        self.stride = torch.zeros(self.nl)
        c2, c3 = max((16, ch[0] // 4, self.reg_max * 4)), max(ch[0], min(self.nc, 100))
        self.cv2 = nn.ModuleList(nn.Sequential(Conv(x, c2, 3), Conv(c2, c2, 3), _Out1x1(c2, 4 * self.reg_max, 1)) for x in ch)
        self.cv3 = nn.ModuleList(nn.Sequential(Conv(x, c3, 3), Conv(c3, c3, 3), _Out1x1(c3, self.nc, 1)) for x in ch)
        self.dfl = DFL(self.reg_max) if self.reg_max > 1 else nn.Identity()

    def _branch(self, seq, xi):
        """Conv -> Conv -> biased 1x1 (head.py:45-59)."""
        h = xi
        for m in seq:
            h = m(h)
        return h

    @property
    def pair_ok(self):
        """the first convolutions of the two branches of a level read the same input (reference head.py:71-72): in training they run as ONE
        convolution with c2 + c3 output channels (ops.conv_bn_act_pair).  A static property of the module (the model graph counts Detect as
        ONE consumer of each input when it holds): both are unfused Conv blocks with SiLU and widths in whole 16-byte bf16 chunks."""
        a, b = self.cv2[0][0], self.cv3[0][0]
        if not ops.HOOKS["detect_pair"]:  # test / A-B hook: the two branches as separate convolutions
            return False
        return (hasattr(a, "bn") and hasattr(b, "bn") and isinstance(a.act, nn.SiLU) and isinstance(b.act, nn.SiLU)
                and a.conv.out_channels % 8 == 0 and b.conv.out_channels % 8 == 0 and a.conv.kernel_size == b.conv.kernel_size)

    def _level(self, i, xi):
        """(box map, class map) of level i from the internal tensor xi."""
        a, b = self.cv2[i], self.cv3[i]
        if self.training and torch.is_grad_enabled() and self.pair_ok:
            h = ops.conv_bn_act_pair(xi, a[0].conv, a[0].bn, b[0].conv, b[0].bn)
            ha, hb = ops.chan_split2(h, a[0].conv.out_channels)
            for m in a[1:]:
                ha = m(ha)
            for m in b[1:]:
                hb = m(hb)
            return ha, hb
        return self._branch(a, xi), self._branch(b, xi)

    def _maps(self, x):
        """-> ([box map], [class map]) of all levels.  Training: the stages of the three levels in lockstep, one GEMM launch per stage
        (ops.detect_train) when the branch shapes allow; else level by level as the reference loops (head.py:70-72)."""
        xs = [ops.to_internal(xi) for xi in x]
        if self.training and torch.is_grad_enabled() and self.pair_ok:
            levels = [(a[0], b[0], a[1], b[1], a[2], b[2]) for a, b in zip(self.cv2, self.cv3)]
            if ops.detect_train_ok(levels, xs[0].dtype) and all(xi.dtype == xs[0].dtype for xi in xs):
                return ops.detect_train(xs, levels)
        box, cls = [], []
        for i in range(self.nl):
            bi, ci = self._level(i, xs[i])
            box.append(bi)
            cls.append(ci)
        return box, cls

    def forward(self, x):
        x = list(x)
        box, cls = self._maps(x)
        for i in range(self.nl):
            x[i] = ops.concat([box[i], cls[i]])
        if self.training:
            return x
        y = self._inference(box, cls)  # decoded from the branch outputs (16-byte aligned rows), not from the odd-width concat
        self.shape = x[0].shape
        return y if self.export else (y, x)

    def forward_split(self, x):
        """train-mode maps WITHOUT the per-level concat (the loss splits them again, reference loss.py:205-207):
        -> (box list [B, 64, H, W], cls list [B, nc, H, W]).  Used by DetectionModel.loss."""
        return self._maps(x)

    def _inference(self, box, cls=None):
        """reference head.py:103-142, non-export branch: DFL expectation, anchor decode, stride scale and class sigmoid in
        one HIP launch over the Detect maps (csrc/loss.hip infer_decode_kernel) -> [B, 4+nc, A] float32.
        box / cls: per-level branch outputs; or, as in the reference, one list of concatenated [B, no, H, W] maps."""
        if self.reg_max != 16:
            raise NotImplementedError("the decode kernel is built for reg_max = 16")
        if cls is None:  # reference signature: split the concatenated maps (a copy when their rows are not 16-byte aligned)
            maps = [ops.to_internal(t) for t in box]
            box = [t[:, : self.reg_max * 4].contiguous(memory_format=torch.channels_last) for t in maps]
            cls = [t[:, self.reg_max * 4 :].contiguous(memory_format=torch.channels_last) for t in maps]
        key = (self.stride.data_ptr(), self.stride._version, self.stride.device)  # re-read after `stride` is assigned, edited or moved
        if getattr(self, "_stride_key", None) != key:
            self._stride_host = [float(s) for s in self.stride]  # one device->host read, not one per call
            self._stride_key = key
        return ops.detect_decode(box, cls, self._stride_host)

    def bias_init(self):
        """reference head.py:144-155."""
        for a, b, s in zip(self.cv2, self.cv3, self.stride):
            a[-1].bias.data[:] = 1.0
            b[-1].bias.data[: self.nc] = math.log(5 / self.nc / (640 / s) ** 2)
