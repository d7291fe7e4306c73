"""Conv / Concat / Upsample operator modules (reference: ultralytics/nn/modules/conv.py).

Same constructor signatures, attribute names (`conv`, `bn`, `act`) and state-dict keys as the
reference, so its checkpoints load and `BaseModel.fuse()` semantics carry over; `self.conv` / `self.bn`
are parameter containers only: the arithmetic runs in libyolo_mi355.so (implicit-GEMM MFMA conv with
BatchNorm statistics in the epilogue, see csrc/igemm.hip).
"""
import torch
import torch.nn as nn

from ... import ops


def autopad(k, p=None, d=1):
    """'same' padding (reference conv.py:28-34)."""
    if d > 1:
        k = d * (k - 1) + 1 if isinstance(k, int) else [d * (x - 1) + 1 for x in k]
    if p is None:
        p = k // 2 if isinstance(k, int) else [x // 2 for x in k]
    return p


class Conv(nn.Module):
    """conv2d(bias-free) -> BatchNorm2d -> SiLU.  Reference conv.py:37-91.

    forward(x): x is any [N, C, H, W] cuda tensor (NCHW float32 from the caller, or the NHWC
    bf16/f32 tensors these modules hand to each other); returns a logical [N, C2, H', W'] tensor in
    NHWC memory.  An optional `residual` implements Bottleneck's add in the same kernel chain.
    """

    default_act = nn.SiLU()

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, d=1, act=True):
        super().__init__()
        if g != 1 or d != 1:
            raise NotImplementedError("libyolo_mi355 implements the groups=1, dilation=1 convolutions the YOLOv8 graph uses")
        if isinstance(k, (tuple, list)):
            if k[0] != k[1]:
                raise NotImplementedError("square kernels only")
            k = k[0]
        if k not in (1, 3) or s not in (1, 2) or autopad(k, p, d) != k // 2:
            raise NotImplementedError(f"Conv(k={k}, s={s}, p={p}): kernels are built for k in (1,3), s in (1,2), 'same' padding")
        self.conv = nn.Conv2d(c1, c2, k, s, autopad(k, p, d), groups=g, dilation=d, bias=False)
        self.bn = nn.BatchNorm2d(c2)
        self.act = self.default_act if act is True else act if isinstance(act, nn.Module) else nn.Identity()

    def _act_code(self):
        if isinstance(self.act, nn.SiLU):
            return ops.ACT_SILU
        if isinstance(self.act, nn.Identity):
            return ops.ACT_NONE
        raise NotImplementedError(f"activation {type(self.act).__name__} has no fused epilogue")

    def forward(self, x, residual=None, out=None):
        """out: optional ops.OutSlot (train mode only): write the result into a slice of a concat buffer."""
        if self.training and hasattr(self, "bn") and ops.first_conv_ok(x, self.conv, residual, out) and isinstance(self.act, nn.SiLU):
            # the model's first layer on the caller's float32 NCHW image: direct kernel, no layout pass (csrc/first_conv.hip)
            return ops.first_conv_bn_act(x, self.conv.weight, self.bn, self._act_code())
        x = ops.to_internal(x)
        k, s = self.conv.kernel_size[0], self.conv.stride[0]
        if not hasattr(self, "bn"):
            return self.forward_fuse(x, residual)
        if self.training:
            return ops.conv_bn_act(x, self.conv.weight, self.bn, s, self._act_code(), residual, out)
        # eval: y = act(conv * scale + shift) with the running statistics, one kernel
        bn = self.bn
        scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        shift = bn.bias - bn.running_mean * scale
        return ops.conv_affine_act(x, self.conv.weight, scale.float(), shift.float(), s, self._act_code(), residual)

    def forward_fuse(self, x, residual=None, out=None):
        """after fuse(): conv (with bias) -> act.  Reference conv.py:81-91.  (`out` slots are a train-mode feature.)"""
        x = ops.to_internal(x)
        return ops.conv_affine_act(x, self.conv.weight, None, self.conv.bias, self.conv.stride[0], self._act_code(), residual)


class Concat(nn.Module):
    """channel concat (reference conv.py:655-683)."""

    def __init__(self, dimension=1):
        super().__init__()
        if dimension != 1:
            raise NotImplementedError("Concat along channels only")
        self.d = dimension

    def forward(self, x, buf=None):
        """buf: the pre-allocated buffer whose slices the producers already wrote (model graph, train mode): no copies."""
        dt = ops.compute_dtype(x[0])
        return ops.concat([ops.to_internal(t, dt) for t in x], buf)


class Upsample(nn.Module):
    """nn.Upsample(None, 2, 'nearest') as used by the YOLOv8 head (reference yolov8.yaml:759,764).
    parse_model maps the YAML name `nn.Upsample` here; same constructor arguments."""

    def __init__(self, size=None, scale_factor=None, mode="nearest"):
        super().__init__()
        if size is not None or scale_factor not in (2, 2.0) or mode != "nearest":
            raise NotImplementedError("only nearest 2x upsampling is on the hot path")
        self.size, self.scale_factor, self.mode = size, scale_factor, mode

    def forward(self, x, out=None):
        return ops.upsample2x(ops.to_internal(x), out)
