"""CBAM of the fork (reference: ultralytics/nn/modules/cbam.py)."""
import torch.nn as nn

from ... import ops


class ChannelAttention(nn.Module):
    """parameter container + lazy MLP creation rule of reference cbam.py:5-38."""

    def __init__(self, in_planes=None, ratio=16):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.max_pool = nn.AdaptiveMaxPool2d(1)
        self.in_planes = in_planes
        self.ratio = ratio
        self.shared_MLP = None
        if in_planes is not None:
            self.create_mlp(in_planes)

    def create_mlp(self, in_planes):
        reduced = max(1, in_planes // self.ratio)
        self.shared_MLP = nn.Sequential(
            nn.Conv2d(in_planes, reduced, 1, bias=False), nn.ReLU(), nn.Conv2d(reduced, in_planes, 1, bias=False)
        )


class SpatialAttention(nn.Module):
    """parameter container of reference cbam.py:40-53."""

    def __init__(self, kernel_size=7):
        super().__init__()
        assert kernel_size in (3, 7), "kernel size must be 3 or 7"
        self.conv = nn.Conv2d(2, 1, kernel_size, padding=3 if kernel_size == 7 else 1, bias=False)


class CBAM(nn.Module):
    """x * ca(x) * sa(x * ca(x)) (reference cbam.py:55-71), `CBAM()` creating its MLP from the first
    input's channel count with ratio 16 (cbam.py:31-33,59) - created on the input's device here."""

    def __init__(self, channels=None):
        super().__init__()
        self.ca = ChannelAttention(channels, ratio=8 if channels and channels < 128 else 16)
        self.sa = SpatialAttention(kernel_size=7)

    def forward(self, x, out=None):
        x = ops.to_internal(x)
        if self.ca.shared_MLP is None:
            self.ca.create_mlp(x.shape[1])
            self.ca.shared_MLP.to(x.device)
        mlp = self.ca.shared_MLP
        return ops.cbam(x, mlp[0].weight, mlp[2].weight, self.sa.conv.weight, out)
