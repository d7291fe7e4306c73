"""Oracle operator modules (CPU, fp32, NCHW).  TEST INFRASTRUCTURE ONLY - see oracle/__init__.py.

Each class keeps the reference's constructor signature and state-dict keys so that a reference
`state_dict()` loads with `strict=True`, but the arithmetic is written out explicitly (batch
statistics, attention, GELU, pooling windows) instead of delegating to the fused torch modules,
so the oracle states the algorithm the HIP kernels have to reproduce.

Citations are relative to /root/reference/ultralytics/.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .quant import st, stf  # storage-precision mode (identity unless oracle.quant.storage(dtype) is active)


def autopad(k, p=None, d=1):
    """'same' padding for kernel k / dilation d.  nn/modules/conv.py:28-34."""
    if d > 1:
        k = d * (k - 1) + 1 if isinstance(k, int) else [d * (v - 1) + 1 for v in k]
    if p is None:
        p = k // 2 if isinstance(k, int) else [v // 2 for v in k]
    return p


def silu(x):
    return x * torch.sigmoid(x)


def batchnorm2d(x, bn, training):
    """BatchNorm2d written out.  Semantics of nn.BatchNorm2d as used by nn/modules/conv.py:66,79
    with eps / momentum from utils/torch_utils.py:468-470 (1e-3 / 0.03).

    train: normalise with the biased batch variance; running_var is updated with the unbiased one.
    """
    if training:
        n = x.numel() // x.shape[1]
        mean = x.mean(dim=(0, 2, 3))
        var = ((x - mean[None, :, None, None]) ** 2).mean(dim=(0, 2, 3))
        with torch.no_grad():
            m = bn.momentum
            bn.running_mean.mul_(1 - m).add_(mean.detach(), alpha=m)
            bn.running_var.mul_(1 - m).add_(var.detach() * (n / max(n - 1, 1)), alpha=m)
            bn.num_batches_tracked += 1
    else:
        mean, var = bn.running_mean, bn.running_var
    inv = torch.rsqrt(var + bn.eps)
    return (x - mean[None, :, None, None]) * (inv * bn.weight)[None, :, None, None] + bn.bias[None, :, None, None]


class Conv(nn.Module):
    """conv2d(bias-free) -> BatchNorm2d -> SiLU.  nn/modules/conv.py:37-91."""

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, d=1, act=True):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, k, s, autopad(k, p, d), groups=g, dilation=d, bias=False)
        self.bn = nn.BatchNorm2d(c2)
        self.has_act = act is True

    def forward(self, x, residual=None):
        """residual: Bottleneck's shortcut (block.py:488), added here so that storage-precision mode rounds the sum once,
        as the product's fused BatchNorm + SiLU + add kernel does; plain float32 arithmetic is unchanged by it."""
        c = self.conv
        y = F.conv2d(x, c.weight, c.bias, c.stride, c.padding, c.dilation, c.groups)
        if hasattr(self, "bn"):  # conv.py:79 ; after fuse() (tasks.py:219-225) the bn is gone: conv.py:81-91
            y = batchnorm2d(st(y), self.bn, self.training)  # the raw convolution output is a stored tensor (BatchNorm backward reads it)
        y = silu(y) if self.has_act else y
        if residual is not None:
            y = residual + y
        return st(y)


def fuse_conv_and_bn(conv, bn):
    """Fold BN running statistics into the conv.  utils/torch_utils.py:240-271."""
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    fused = nn.Conv2d(
        conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding, conv.dilation, conv.groups, bias=True
    ).requires_grad_(False)
    fused.weight.copy_(conv.weight * scale[:, None, None, None])
    b0 = torch.zeros_like(bn.running_mean) if conv.bias is None else conv.bias
    fused.bias.copy_((b0 - bn.running_mean) * scale + bn.bias)
    return fused


class Concat(nn.Module):
    """torch.cat along `dimension`.  nn/modules/conv.py:655-683."""

    def __init__(self, dimension=1):
        super().__init__()
        self.d = dimension

    def forward(self, xs):
        return torch.cat(xs, self.d)


class Bottleneck(nn.Module):
    """x + cv2(cv1(x)) when shortcut and c1 == c2.  nn/modules/block.py:479-488."""

    def __init__(self, c1, c2, shortcut=True, g=1, k=(3, 3), e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, k[0], 1)
        self.cv2 = Conv(c_, c2, k[1], 1, g=g)
        self.add = shortcut and c1 == c2

    def forward(self, x):
        return self.cv2(self.cv1(x), residual=x if self.add else None)


class C2f(nn.Module):
    """cv1 -> split in two -> n chained Bottlenecks -> concat all -> cv2.  nn/modules/block.py:279-304."""

    def __init__(self, c1, c2, n=1, shortcut=False, g=1, e=0.5):
        super().__init__()
        self.c = int(c2 * e)
        self.cv1 = Conv(c1, 2 * self.c, 1, 1)
        self.cv2 = Conv((2 + n) * self.c, c2, 1)
        self.m = nn.ModuleList(Bottleneck(self.c, self.c, shortcut, g, k=((3, 3), (3, 3)), e=1.0) for _ in range(n))

    def forward(self, x):
        t = self.cv1(x)
        ys = [t[:, : self.c], t[:, self.c :]]
        for b in self.m:
            ys.append(b(ys[-1]))
        return self.cv2(torch.cat(ys, 1))


def first_max(x, dim):
    """max over `dim` whose gradient goes to the FIRST maximal element in index order - the rule of every max the
    reference uses on this path (nn.MaxPool2d, nn.AdaptiveMaxPool2d, torch.max(dim): ATen keeps the running maximum and
    replaces it only by a strictly greater value).  `amax` / `torch.maximum` would split the gradient between tied
    elements instead.  Pinned by the reference-generated fixtures `*_ties` (tests/golden/make_golden.py ties)."""
    idx = x.argmax(dim=dim, keepdim=True)  # "the indices of the first maximal value are returned"
    return x.gather(dim, idx).squeeze(dim)


def maxpool_same(x, k):
    """MaxPool2d(k, stride 1, pad k//2) with -inf padding (block.py:220): the k*k window positions in the order ATen scans
    them (rows, then columns), first maximum wins."""
    p = k // 2
    xp = F.pad(x, (p, p, p, p), value=float("-inf"))
    H, W = x.shape[-2:]
    cols = torch.stack([xp[..., dy : dy + H, dx : dx + W] for dy in range(k) for dx in range(k)], dim=-1)
    return first_max(cols, -1)


class SPPF(nn.Module):
    """cv1 -> three chained k x k max-pools -> concat(4) -> cv2.  nn/modules/block.py:201-226."""

    def __init__(self, c1, c2, k=5):
        super().__init__()
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_ * 4, c2, 1, 1)
        self.k = k

    def forward(self, x):
        y0 = self.cv1(x)
        y1 = maxpool_same(y0, self.k)
        y2 = maxpool_same(y1, self.k)
        y3 = maxpool_same(y2, self.k)
        return self.cv2(torch.cat((y0, y1, y2, y3), 1))


class ChannelAttention(nn.Module):
    """sigmoid(MLP(avgpool x) + MLP(maxpool x)), shared bias-free 1x1-conv MLP.  nn/modules/cbam.py:5-38.

    With in_planes=None the MLP is created on the first forward from the input's channel count
    (cbam.py:31-33) with the ratio given at construction.
    """

    def __init__(self, in_planes=None, ratio=16):
        super().__init__()
        self.in_planes = in_planes
        self.ratio = ratio
        self.shared_MLP = None
        if in_planes is not None:
            self.create_mlp(in_planes)

    def create_mlp(self, in_planes):
        hidden = max(1, in_planes // self.ratio)  # cbam.py:22
        self.shared_MLP = nn.Sequential(
            nn.Conv2d(in_planes, hidden, 1, bias=False), nn.ReLU(), nn.Conv2d(hidden, in_planes, 1, bias=False)
        )

    def forward(self, x):
        if self.shared_MLP is None:
            self.create_mlp(x.shape[1])
        w1 = self.shared_MLP[0].weight.flatten(1)  # [hidden, C]
        w2 = self.shared_MLP[2].weight.flatten(1)  # [C, hidden]
        avg = x.mean(dim=(2, 3))  # [B, C]
        mx = first_max(x.flatten(2), -1)  # nn.AdaptiveMaxPool2d(1), cbam.py:9,36: first maximum in row-major pixel order
        z = torch.relu(avg @ w1.t()) @ w2.t() + torch.relu(mx @ w1.t()) @ w2.t()
        return torch.sigmoid(z)[:, :, None, None]


class SpatialAttention(nn.Module):
    """sigmoid(conv_kxk(cat[mean_C x, max_C x])), bias-free.  nn/modules/cbam.py:40-53."""

    def __init__(self, kernel_size=7):
        super().__init__()
        assert kernel_size in (3, 7)
        self.conv = nn.Conv2d(2, 1, kernel_size, padding=3 if kernel_size == 7 else 1, bias=False)

    def forward(self, x):
        s = torch.cat((x.mean(dim=1, keepdim=True), first_max(x, 1).unsqueeze(1)), 1)  # torch.max(x, dim=1), cbam.py:50: first maximal channel
        return torch.sigmoid(F.conv2d(s, self.conv.weight, None, 1, self.conv.padding))


class CBAM(nn.Module):
    """x1 = x * ca(x); out = x1 * sa(x1).  nn/modules/cbam.py:55-71 (ratio rule :59)."""

    def __init__(self, channels=None):
        super().__init__()
        self.ca = ChannelAttention(channels, ratio=8 if channels and channels < 128 else 16)
        self.sa = SpatialAttention(kernel_size=7)

    def forward(self, x):
        x = x * self.ca(x)
        return st(x * self.sa(x))  # (the product keeps x * ca in registers: only the result is a stored tensor)


def window_partition(x, ws):
    """[B,H,W,C] -> [B*nW, ws*ws, C].  nn/modules/swin_block.py:8-13."""
    B, H, W, C = x.shape
    x = x.view(B, H // ws, ws, W // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)


def window_reverse(windows, ws, H, W):
    """[B*nW, ws*ws, C] -> [B,H,W,C].  nn/modules/swin_block.py:15-20."""
    B = int(windows.shape[0] / (H * W / ws / ws))
    x = windows.view(B, H // ws, W // ws, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, -1)


def window_token_index(B, H, W, ws):
    """Integer restatement of window_partition: for every (window, token) the flat index
    b*H*W + h*W + w of the source pixel in the padded [B,H,W] grid.  swin_block.py:8-13."""
    idx = torch.arange(B * H * W, dtype=torch.int64).view(B, H, W, 1)
    return window_partition(idx, ws).squeeze(-1)


def layernorm(x, ln):
    """LayerNorm over the last dim, eps 1e-5, biased variance.  swin_block.py:27,30."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + ln.eps) * ln.weight + ln.bias


def gelu_erf(x):
    """Exact (erf) GELU = nn.GELU() default.  swin_block.py:33."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def multihead_self_attention(t, mha):
    """nn.MultiheadAttention(C, heads, batch_first=True)(t, t, t) written out.  swin_block.py:29,51.

    packed in_proj -> split q,k,v -> heads -> q/sqrt(hd) -> softmax over ALL keys of the window
    (pad tokens are ordinary keys: the reference passes no mask) -> PV -> merge heads -> out_proj.
    """
    Bn, L, C = t.shape
    h = mha.num_heads
    hd = C // h
    qkv = st(t @ mha.in_proj_weight.t() + mha.in_proj_bias)
    q, k, v = qkv.split(C, dim=-1)
    q = q.view(Bn, L, h, hd).transpose(1, 2) * (1.0 / math.sqrt(hd))
    k = k.view(Bn, L, h, hd).transpose(1, 2)
    v = v.view(Bn, L, h, hd).transpose(1, 2)
    p = torch.softmax(q @ k.transpose(-1, -2), dim=-1)
    o = st((stf(p) @ v).transpose(1, 2).reshape(Bn, L, C))
    return o @ mha.out_proj.weight.t() + mha.out_proj.bias


class SwinBlock(nn.Module):
    """Windowed MSA block of the fork.  nn/modules/swin_block.py:23-58.

    pad right/bottom to a multiple of ws (zeros) -> NHWC -> windows -> t = LN1(t) ->
    t = t + MHA(t) (skip from the NORMALISED tokens, :50-52) -> t = t + MLP(LN2(t)) -> reverse -> crop.
    """

    def __init__(self, dim, num_heads=2, window_size=7):
        super().__init__()
        self.dim = dim
        self.window_size = window_size
        self.norm1 = nn.LayerNorm(dim)
        self.attn = nn.MultiheadAttention(embed_dim=dim, num_heads=num_heads, batch_first=True)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = nn.Sequential(nn.Linear(dim, dim * 4), nn.GELU(), nn.Linear(dim * 4, dim))

    def forward(self, x):
        B, C, H, W = x.shape
        ws = self.window_size
        ph = (ws - H % ws) % ws
        pw = (ws - W % ws) % ws
        x = F.pad(x, (0, pw, 0, ph))
        Hp, Wp = H + ph, W + pw
        t = window_partition(x.permute(0, 2, 3, 1).contiguous(), ws)
        t = st(layernorm(t, self.norm1))
        t = st(t + multihead_self_attention(t, self.attn))
        u = st(layernorm(t, self.norm2))
        h = st(gelu_erf(st(u @ self.mlp[0].weight.t() + self.mlp[0].bias)))
        t = st(t + (h @ self.mlp[2].weight.t() + self.mlp[2].bias))
        x = window_reverse(t, ws, Hp, Wp).permute(0, 3, 1, 2)
        return x[:, :, :H, :W]


class DFL(nn.Module):
    """Expectation of the 16-bin distribution (frozen 1x1 conv with weights 0..15).  block.py:58-77."""

    def __init__(self, c1=16):
        super().__init__()
        self.conv = nn.Conv2d(c1, 1, 1, bias=False).requires_grad_(False)
        self.conv.weight.data[:] = torch.arange(c1, dtype=torch.float).view(1, c1, 1, 1)
        self.c1 = c1

    def forward(self, x):
        b, _, a = x.shape
        p = x.view(b, 4, self.c1, a).softmax(2)
        return (p * self.conv.weight.view(1, 1, self.c1, 1)).sum(2)


def make_anchors(feats, strides, offset=0.5):
    """Anchor centres + per-anchor stride for a list of feature maps.  utils/tal.py:364-376."""
    pts, st = [], []
    for f, s in zip(feats, strides):
        h, w = f.shape[2:]
        sx = torch.arange(w, dtype=f.dtype, device=f.device) + offset
        sy = torch.arange(h, dtype=f.dtype, device=f.device) + offset
        gy, gx = torch.meshgrid(sy, sx, indexing="ij")
        pts.append(torch.stack((gx, gy), -1).view(-1, 2))
        st.append(torch.full((h * w, 1), float(s), dtype=f.dtype, device=f.device))
    return torch.cat(pts), torch.cat(st)


def dist2bbox(distance, anchor_points, xywh=True, dim=-1):
    """ltrb distances -> box.  utils/tal.py:379-388."""
    lt, rb = distance.chunk(2, dim)
    x1y1 = anchor_points - lt
    x2y2 = anchor_points + rb
    if xywh:
        return torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), dim)
    return torch.cat((x1y1, x2y2), dim)


class Detect(nn.Module):
    """YOLOv8 detect head, legacy (v8) class branch.  nn/modules/head.py:23-155.

    parse_model sets `legacy=True` for v8 YAMLs (tasks.py:1355,1488), so cv3 is the plain
    Conv3x3-Conv3x3-Conv2d1x1 stack (head.py:49).
    """

    legacy = True

    def __init__(self, nc=80, ch=()):
        super().__init__()
        self.nc = nc
        self.nl = len(ch)
        self.reg_max = 16
        self.no = nc + self.reg_max * 4
        self.stride = torch.zeros(self.nl)
        c2, c3 = max((16, ch[0] // 4, self.reg_max * 4)), max(ch[0], min(self.nc, 100))
        self.cv2 = nn.ModuleList(
            nn.Sequential(Conv(x, c2, 3), Conv(c2, c2, 3), nn.Conv2d(c2, 4 * self.reg_max, 1)) for x in ch
        )
        self.cv3 = nn.ModuleList(nn.Sequential(Conv(x, c3, 3), Conv(c3, c3, 3), nn.Conv2d(c3, self.nc, 1)) for x in ch)
        self.dfl = DFL(self.reg_max)

    def forward(self, x):
        x = [st(torch.cat((self.cv2[i](x[i]), self.cv3[i](x[i])), 1)) for i in range(self.nl)]  # head.py:71-72 (the biased 1x1 outputs are stored tensors)
        if self.training:
            return x
        return self._inference(x), x

    def _inference(self, x):
        """head.py:103-142 (non-export branch)."""
        b = x[0].shape[0]
        x_cat = torch.cat([xi.reshape(b, self.no, -1) for xi in x], 2)
        anchors, strides = (t.transpose(0, 1) for t in make_anchors(x, self.stride, 0.5))
        box, cls = x_cat.split((self.reg_max * 4, self.nc), 1)
        dbox = dist2bbox(self.dfl(box), anchors.unsqueeze(0), xywh=True, dim=1) * strides
        return torch.cat((dbox, cls.sigmoid()), 1)

    def bias_init(self):
        """head.py:144-155."""
        for a, b, s in zip(self.cv2, self.cv3, self.stride):
            a[-1].bias.data[:] = 1.0
            b[-1].bias.data[: self.nc] = math.log(5 / self.nc / (640 / s) ** 2)
