"""Bottleneck / C2f / SPPF / DFL (reference: ultralytics/nn/modules/block.py)."""
import torch
import torch.nn as nn

from ... import ops
from .conv import Conv


class DFL(nn.Module):
    """expectation over the 16-bin box distribution (reference block.py:58-77); frozen weights 0..15.
    Inference-side decode on [B, 4*c1, A] tensors: tiny, stays in torch ops."""

    def __init__(self, c1=16):
        super().__init__()
        self.conv = nn.Conv2d(c1, 1, 1, bias=False).requires_grad_(False)
        self.conv.weight.data[:] = torch.arange(c1, dtype=torch.float).view(1, c1, 1, 1)
        self.c1 = c1

    def forward(self, x):
        b, _, a = x.shape
        p = x.view(b, 4, self.c1, a).float().softmax(2)
        return (p * self.conv.weight.view(1, 1, self.c1, 1).float()).sum(2)


class Bottleneck(nn.Module):
    """x + cv2(cv1(x)) when shortcut and c1 == c2 (reference block.py:479-488); the add rides in
    cv2's BN/SiLU kernel."""

    def __init__(self, c1, c2, shortcut=True, g=1, k=(3, 3), e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, k[0], 1)
        self.cv2 = Conv(c_, c2, k[1], 1, g=g)
        self.add = shortcut and c1 == c2

    def forward(self, x, out=None):
        xi = ops.to_internal(x)
        # the shortcut rides in cv2's BatchNorm + SiLU kernel when cv1's operand IS the tensor to add.  A width that is not a
        # multiple of the 16-byte chunk reaches cv1 as a zero-padded copy (more channels than cv2 produces): the add is then a
        # separate launch on the caller's tensor and autograd forms the gradient sum
        fused_add = self.add and xi.shape[1] == self.cv2.conv.out_channels
        if fused_add and self.training and ops.join_of(xi) is None:
            ops.mark_join(xi, 2)  # consumers of x here: cv1 and the shortcut; their gradient sum forms in cv1's data-gradient epilogue
        h = self.cv1(xi)
        if fused_add or not self.add:
            return self.cv2(h, residual=xi if self.add else None, out=out)
        return ops.add_residual(self.cv2(h), x, out)


class C2f(nn.Module):
    """cv1 -> 2 chunks -> n chained Bottlenecks -> concat -> cv2 (reference block.py:279-304).
    The chunks are channel slices of cv1's NHWC output (no copy)."""

    def __init__(self, c1, c2, n=1, shortcut=False, g=1, e=0.5):
        super().__init__()
        self.c = int(c2 * e)
        self.cv1 = Conv(c1, 2 * self.c, 1, 1)
        self.cv2 = Conv((2 + n) * self.c, c2, 1)
        self.m = nn.ModuleList(Bottleneck(self.c, self.c, shortcut, g, k=((3, 3), (3, 3)), e=1.0) for _ in range(n))

    def forward(self, x, out=None):
        """out: optional ops.OutSlot for the block's result (a slice of a later Concat's buffer; train mode)."""
        x = ops.to_internal(x)
        buf, slot = None, (lambda j: None)
        dt = x.dtype
        dense = self.c % ops.chunk_elems(dt) == 0
        if self.training and hasattr(self.cv1, "bn") and dense:
            # train mode: cv1 and every Bottleneck write straight into their slice of the concat buffer
            n, _, h, w = x.shape
            buf = ops.empty_nhwc(n, (2 + len(self.m)) * self.c, h, w, dt, x.device)
            slot = lambda j: ops.OutSlot(buf, j * self.c)  # noqa: E731
        t, last = ops.c2f_split(self.cv1(x, out=slot(0)), self.c)  # (both chunks, second chunk): channel slices, no copy
        ys = [t]  # both chunks go into the concat at once: they are adjacent in memory
        first_join = None
        for j, m in enumerate(self.m):
            # consumers of a Bottleneck's input: its cv1, its shortcut (if any) and - for j >= 1, where the input is the
            # previous Bottleneck's output - the concat; the right half of t reaches the concat through c2f_split instead
            # Marked whenever the Bottleneck consumes `last` itself (a width in whole 16-byte chunks: to_internal is the identity),
            # with or without the concat buffer - ops.concat is a join-aware consumer either way.  Other widths reach the
            # Bottleneck as padded copies: autograd sums their gradients and nothing is marked.
            if dense:
                ops.mark_join(last, 1 + int(m.add) + int(j >= 1), force=(j == 0))
                if j == 0:
                    first_join = ops.join_of(last)
            last = m(last, out=slot(2 + j))
            ys.append(last)
        # the concat's gradient for the right half of t and the first Bottleneck's input gradient are summed in the latter's epilogue
        return self.cv2(ops.concat(ys, buf, split_join=(first_join, self.c) if first_join is not None else None), out=out)

    forward_split = forward


class SPPF(nn.Module):
    """cv1 -> three chained k x k max-pools -> concat(4) -> cv2 (reference block.py:201-226); the
    pools and the concat are one LDS-staged kernel (csrc/pool.hip)."""

    def __init__(self, c1, c2, k=5):
        super().__init__()
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_ * 4, c2, 1, 1)
        self.m = nn.MaxPool2d(kernel_size=k, stride=1, padding=k // 2)  # attribute kept for parity with the reference

    def forward(self, x, out=None):
        x = ops.to_internal(x)
        c_ = self.cv1.conv.out_channels
        cat = None
        if self.training and hasattr(self.cv1, "bn") and c_ % ops.chunk_elems(x.dtype) == 0:
            n, _, h, w = x.shape
            cat = ops.empty_nhwc(n, 4 * c_, h, w, x.dtype, x.device)  # cv1 writes y0 straight into slice 0 of the concat buffer
        y0 = self.cv1(x, out=ops.OutSlot(cat, 0) if cat is not None else None)
        return self.cv2(ops.sppf_pool_cat(y0, self.m.kernel_size, cat), out=out)
