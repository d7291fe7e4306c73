"""Operator modules of the hot path, same names as the reference's ultralytics/nn/modules/__init__.py."""
from .block import C2f, SPPF, Bottleneck, DFL  # noqa: F401
from .cbam import CBAM, ChannelAttention, SpatialAttention  # noqa: F401
from .conv import Concat, Conv, Upsample, autopad  # noqa: F401
from .head import Detect  # noqa: F401
from .swin_block import SwinBlock, window_partition, window_reverse  # noqa: F401
