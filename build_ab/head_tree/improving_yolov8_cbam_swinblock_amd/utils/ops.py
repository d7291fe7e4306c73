"""small helpers with the reference's names (ultralytics/utils/ops.py)."""
import math

import torch


def make_divisible(x, divisor):
    """reference utils/ops.py:130-143."""
    if isinstance(divisor, torch.Tensor):
        divisor = int(divisor.max())
    return math.ceil(x / divisor) * divisor


def xywh2xyxy(x):
    """reference utils/ops.py:432-449."""
    xy, half = x[..., :2], x[..., 2:] / 2
    return torch.cat((xy - half, xy + half), -1)
