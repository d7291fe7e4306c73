"""The fork's blocks and the glue ops: SwinBlock pieces (LayerNorm, window attention, fused / unfused MLP, window partition / reverse), CBAM, SPPF pools,
concat / split / upsample / residual, the detection loss and decode (reference: nn/modules/swin_block.py, cbam.py, block.py:201-226, utils/loss.py:152-255)."""
import ctypes
import os

import torch

from .. import _lib
from .._lib import ACT_GELU, ACT_NONE, ACT_SILU, ConvProblem, DgradProblem, as_ymi, check, chunk_elems, empty_nhwc, is_nhwc, ptr, stream_ptr, workspace, ymi_dtype
from .base import (  # noqa: F401
    HOOKS, L, _accumulate, _byref, _dense_ok, _join_plain, _note_use, _prep_adds, grad_nhwc, join_of, round_up,
)
from .weights import (  # noqa: F401
    _wgrad_maybe_async, pack_conv_dgrad, pack_conv_fwd,
)
from .conv import (  # noqa: F401
    padded_grad_like,
)

class _SwinMlp(torch.autograd.Function):
    """out = fc2(gelu(fc1(u))) + residual on token matrices (swin_block.py:33,53) through ymi_swin_mlp_fwd / _bwd_data: GELU rides
    in fc1's epilogue (second output) and its derivative in fc2's data-gradient epilogue, so the [T, 4C] activation and its
    gradient are never passed through stand-alone activation kernels."""

    @staticmethod
    def forward(ctx, u, w1, b1, w2, b2, residual, join, res_join):
        dtype = u.dtype
        t, c = u.shape
        hidden = w1.shape[0]
        dev = u.device
        pre = torch.empty((t, hidden), dtype=dtype, device=dev)
        post = torch.empty((t, hidden), dtype=dtype, device=dev)
        out = torch.empty((t, w2.shape[0]), dtype=dtype, device=dev)
        _note_use(w1, w2)
        w1p = pack_conv_fwd(w1, c, dtype)
        w2p = pack_conv_fwd(w2, hidden, dtype)
        check(
            L().ymi_swin_mlp_fwd(_byref(as_ymi(u)), ptr(w1p), ptr(b1), hidden, ptr(w2p), ptr(b2), _byref(as_ymi(residual)) if residual is not None else None,
                                 _byref(as_ymi(pre)), _byref(as_ymi(post)), _byref(as_ymi(out)), stream_ptr()),
            "swin_mlp_fwd",
        )
        ctx.save_for_backward(u, w1, w2, pre, post)
        ctx.biases = (b1, b2)
        ctx.cfg = (b1 is not None, b2 is not None, residual is not None)
        ctx.joins = (join, res_join)
        return out

    @staticmethod
    def backward(ctx, dout):
        u, w1, w2, pre, post = ctx.saved_tensors
        has_b1, has_b2, has_res = ctx.cfg
        dtype = u.dtype
        dout = grad_nhwc(dout, dtype)
        join, res_join = ctx.joins
        dres = _join_plain(res_join, dout) if (has_res and ctx.needs_input_grad[5]) else None
        t, c = u.shape
        hidden = w1.shape[0]
        dpre = torch.empty_like(pre)
        need_du = ctx.needs_input_grad[0]
        adds = (join.arrive() if join is not None else []) if need_du else []
        fa = _prep_adds(adds, dtype, False)
        du = torch.empty((t, c), dtype=dtype, device=u.device) if need_du else None
        w2d = pack_conv_dgrad(w2, w2.shape[0], 1, dtype)
        w1d = pack_conv_dgrad(w1, hidden, 1, dtype) if need_du else None
        check(
            L().ymi_swin_mlp_bwd_data(_byref(as_ymi(dout)), ptr(w2d), _byref(as_ymi(pre)), _byref(as_ymi(dpre)), ptr(w1d) if need_du else None,
                                      _byref(as_ymi(fa[0])) if len(fa) > 0 else None, _byref(as_ymi(fa[1])) if len(fa) > 1 else None,
                                      _byref(as_ymi(du)) if need_du else None, stream_ptr()),
            "swin_mlp_bwd_data",
        )
        if need_du:
            if len(fa) > 2:
                _accumulate(du, fa[2:])
            if adds is None:
                join.deposit(du)
                du = None
        nig = ctx.needs_input_grad  # (u, w1, b1, w2, b2, ...): frozen parameters get no GEMM and no deferred record
        dw1 = db1 = dw2 = db2 = None
        if nig[3] or (has_b2 and nig[4]):
            dw2, db2 = _wgrad_maybe_async(post, dout, w2.shape[0], hidden, 1, 1, has_b2, (w2, ctx.biases[1]))
            dw2, db2 = (dw2.view(w2.shape) if nig[3] else None), (db2 if nig[4] else None)
        if nig[1] or (has_b1 and nig[2]):
            dw1, db1 = _wgrad_maybe_async(u, dpre, hidden, c, 1, 1, has_b1, (w1, ctx.biases[0]))
            dw1, db1 = (dw1.view(w1.shape) if nig[1] else None), (db1 if nig[2] else None)
        return du, dw1, db1, dw2, db2, dres, None, None


def swin_mlp(u, fc1, fc2, residual=None):
    """fc2(gelu(fc1(u))) + residual for the two nn.Linear of SwinBlock.mlp (exact-erf GELU)."""
    return _SwinMlp.apply(u, fc1.weight, fc1.bias, fc2.weight, fc2.bias, residual, join_of(u), join_of(residual) if residual is not None else None)


def swin_ln_mlp_ok(x, fc1):
    """SwinBlock's second half as the fused kernels of csrc/swin_mlp.hip: bfloat16 tokens of 256 channels (HOOKS["fused_swin_mlp"]: test / A-B hook)."""
    return (HOOKS["fused_swin_mlp"] and x.dim() == 2 and x.is_cuda and x.dtype == torch.bfloat16 and x.stride(1) == 1 and x.stride(0) % 8 == 0
            and bool(L().ymi_swin_ln_mlp_supported(x.shape[1], fc1.weight.shape[0], ymi_dtype(x.dtype))))



class _SwinLnMlp(torch.autograd.Function):
    """out = x + fc2(gelu(fc1(LayerNorm(x)))) on a token matrix - swin_block.py:53 with norm2 and mlp of swin_block.py:30-35 - as ONE forward
    kernel (the [T, 4C] activations stay in registers; training stores the bf16 pre-activations once, in the kernel's own order) and ONE
    data-gradient kernel (d_pre = (d_out W2) * gelu'(pre), d_u = d_pre W1; it writes gelu(pre) and d_pre row-major for the two weight-gradient
    GEMMs), then LayerNorm's backward with the skip's gradient as its addend.  x has no other consumer: no GradJoin."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, w1, b1, w2, b2):
        t, c = x.shape
        hidden = w1.shape[0]
        dev = x.device
        lib = L()
        _note_use(w1, w2)
        packed = torch.empty(lib.ymi_swin_ln_mlp_pack_elems(c, hidden), dtype=torch.bfloat16, device=dev)
        check(lib.ymi_swin_ln_mlp_pack(ptr(w1.detach()), ptr(w2.detach()), c, hidden, ptr(packed), stream_ptr()), "swin_ln_mlp_pack")
        train = any(ctx.needs_input_grad)
        out = torch.empty((t, c), dtype=x.dtype, device=dev)
        u = torch.empty((t, c), dtype=x.dtype, device=dev) if train else None
        stats = torch.empty((2, t), dtype=torch.float32, device=dev) if train else None
        pre = torch.empty(lib.ymi_swin_ln_mlp_pre_elems(t, hidden), dtype=x.dtype, device=dev) if train else None
        check(
            lib.ymi_swin_ln_mlp_fwd(_byref(as_ymi(x)), ptr(gamma), ptr(beta), eps, ptr(packed), ptr(b1), ptr(b2), hidden, _byref(as_ymi(u)) if train else None,
                                    ptr(stats[0]) if train else None, ptr(stats[1]) if train else None, ptr(pre), _byref(as_ymi(out)), stream_ptr()),
            "swin_ln_mlp_fwd",
        )
        if train:
            ctx.save_for_backward(x, gamma, w1, w2, u, stats, pre, packed)
            ctx.biases = (b1, b2)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, gamma, w1, w2, u, stats, pre, packed = ctx.saved_tensors
        b1, b2 = ctx.biases
        dtype = x.dtype
        dev = x.device
        t, c = x.shape
        hidden = w1.shape[0]
        dout = grad_nhwc(dout, dtype)
        # (the kernel stores whole 256-token tiles: post / dpre are the first t rows of padded buffers)
        cap = L().ymi_swin_ln_mlp_pre_elems(t, hidden)
        post = torch.empty(cap, dtype=dtype, device=dev).view(-1, hidden)[:t]
        dpre = torch.empty(cap, dtype=dtype, device=dev).view(-1, hidden)[:t]
        du = torch.empty((t, c), dtype=dtype, device=dev)
        check(L().ymi_swin_ln_mlp_bwd_data(_byref(as_ymi(dout)), ptr(packed), ptr(pre), hidden, _byref(as_ymi(post)), _byref(as_ymi(dpre)), _byref(as_ymi(du)),
                                           stream_ptr()), "swin_ln_mlp_bwd_data")
        nig = ctx.needs_input_grad  # (x, gamma, beta, eps, w1, b1, w2, b2)
        dw1 = db1 = dw2 = db2 = None
        if nig[6] or nig[7]:
            dw2, db2 = _wgrad_maybe_async(post, dout, c, hidden, 1, 1, True, (w2, b2))
            dw2, db2 = (dw2.view(w2.shape) if nig[6] else None), (db2 if nig[7] else None)
        if nig[4] or nig[5]:
            dw1, db1 = _wgrad_maybe_async(u, dpre, hidden, c, 1, 1, True, (w1, b1))
            dw1, db1 = (dw1.view(w1.shape) if nig[4] else None), (db1 if nig[5] else None)
        # LayerNorm's backward on d_u; the skip's gradient (d_out itself) is its addend: dx = LN'(d_u) + d_out
        dx = torch.empty_like(x)
        dgb = torch.empty((2, c), dtype=torch.float32, device=dev)
        wsb = workspace(2048 * 2 * c * 4 + 256, dev, "ln")
        check(
            L().ymi_layernorm_bwd_add(_byref(as_ymi(x)), 0, _byref(as_ymi(du)), ptr(gamma), ptr(stats[0]), ptr(stats[1]), _byref(as_ymi(dout)), _byref(as_ymi(dx)),
                                      ptr(dgb[0]), ptr(dgb[1]), ptr(wsb), wsb.numel(), stream_ptr()),
            "layernorm_bwd",
        )
        return dx, dgb[0], dgb[1], None, dw1, db1, dw2, db2


def swin_ln_mlp(x, ln, fc1, fc2):
    """x + fc2(gelu(fc1(ln(x)))) (swin_block.py:53) through the fused kernels; callers check swin_ln_mlp_ok first."""
    return _SwinLnMlp.apply(x, ln.weight, ln.bias, float(ln.eps), fc1.weight, fc1.bias, fc2.weight, fc2.bias)


class _Act(torch.autograd.Function):
    """elementwise activation on a token matrix (exact-erf GELU of swin_block.py:33)."""

    @staticmethod
    def forward(ctx, pre, act):
        out = torch.empty_like(pre)
        check(L().ymi_scale_shift_act(_byref(as_ymi(pre)), None, None, act, None, _byref(as_ymi(out)), stream_ptr()), "scale_shift_act")
        ctx.save_for_backward(pre)
        ctx.act = act
        return out

    @staticmethod
    def backward(ctx, dy):
        (pre,) = ctx.saved_tensors
        if ctx.act != ACT_GELU:
            raise NotImplementedError("only GELU has a stand-alone backward")
        dy = grad_nhwc(dy, pre.dtype)
        dx = torch.empty_like(pre)
        check(L().ymi_gelu_bwd(_byref(as_ymi(pre)), _byref(as_ymi(dy)), _byref(as_ymi(dx)), stream_ptr()), "gelu_bwd")
        return dx, None


def gelu(pre):
    return _Act.apply(pre, ACT_GELU)


class _AddResidual(torch.autograd.Function):
    """y + r as its own launches (Bottleneck shortcut, block.py:488, for widths whose shortcut cannot ride in the BatchNorm + SiLU
    kernel); both gradients are the incoming one."""

    @staticmethod
    def forward(ctx, y, r, slot=None):
        n, c, h, w = y.shape
        out = slot.view(n, c, h, w, y.dtype) if slot is not None else empty_nhwc(n, c, h, w, y.dtype, y.device)
        check(L().ymi_copy(_byref(as_ymi(y)), _byref(as_ymi(out)), stream_ptr()), "copy")
        check(L().ymi_add_inplace(_byref(as_ymi(r)), _byref(as_ymi(out)), stream_ptr()), "add_inplace")
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g, None


def add_residual(y, r, slot=None):
    """y: internal tensor; r: any [N, C, H, W] cuda tensor of the same channel count (brought to y's dtype and NHWC memory)."""
    if r.dtype != y.dtype or not is_nhwc(r):
        buf = empty_nhwc(*r.shape, y.dtype, r.device)
        buf.copy_(r)
        r = buf
    return _AddResidual.apply(y, r, slot)


# ------------------------------------------------------------------------ concat / upsample
class _Concat(torch.autograd.Function):
    """channel concat by strided copies into one NHWC buffer (conv.py:683, block.py:226,304)."""

    @staticmethod
    def forward(ctx, buf, joins, split_join, *xs):
        n, _, h, w = xs[0].shape
        cs = [t.shape[1] for t in xs]
        out = buf if buf is not None else empty_nhwc(n, sum(cs), h, w, xs[0].dtype, xs[0].device)
        if out.shape[1] != sum(cs) or out.dtype != xs[0].dtype:
            raise RuntimeError("concat buffer does not match its inputs")
        off = 0
        for t, c in zip(xs, cs):
            dst = out[:, off : off + c]
            if not (t.data_ptr() == dst.data_ptr() and t.stride() == dst.stride()):  # producers given an OutSlot already wrote here
                check(L().ymi_copy(_byref(as_ymi(t)), _byref(as_ymi(dst)), stream_ptr()), "copy")
            off += c
        ctx.cs = cs
        ctx.joins = joins
        ctx.split_join = split_join
        return out if buf is None else out[:, :]

    @staticmethod
    def backward(ctx, g):
        outs, off = [], 0
        for c, j in zip(ctx.cs, ctx.joins):
            outs.append(_join_plain(j, g[:, off : off + c]))  # inputs with other consumers: the slice is deposited for the last of them
            off += c
        if ctx.split_join is not None and outs[0] is not None:
            # C2f: the right half of the first input is ALSO the first Bottleneck's input.  Tell that tensor's join where this half of
            # the gradient lies: the Bottleneck's data gradient adds it in its epilogue and writes the sum back in place.
            join, c0 = ctx.split_join
            join.out = outs[0][:, c0:]
        return (None, None, None, *outs)


def concat(xs, buf=None, split_join=None):
    """channel concat; buf: optional pre-allocated NHWC buffer whose slices some inputs already alias (OutSlot).
    split_join: (GradJoin of the tensor that is the channel slice [c0:] of the FIRST input, c0) - see _Concat.backward."""
    return _Concat.apply(buf, tuple(join_of(t) for t in xs), split_join, *xs)


class _C2fSplit(torch.autograd.Function):
    """C2f's `chunk(2, 1)` (block.py:302) without autograd's slice bookkeeping: returns (t, right half of t) as
    views.  Backward receives the gradient of the whole tensor (from the concat) and of the right half (from the
    first Bottleneck) in ONE call and adds the latter into the former's right half in place with one kernel,
    instead of zero-fill + copy + strided add (three passes over the tensor)."""

    @staticmethod
    def forward(ctx, t, c):
        ctx.c = c
        ctx.shape = t.shape
        return t.view_as(t), t[:, c:]

    @staticmethod
    def backward(ctx, g_full, g_right):
        c = ctx.c
        if g_full is None:
            n, c2, h, w = ctx.shape
            g_full = empty_nhwc(n, c2, h, w, g_right.dtype, g_right.device)
            g_full.zero_()
        if g_right is not None and g_right.data_ptr() == g_full[:, c:].data_ptr() and g_right.stride() == g_full.stride() and g_right.dtype == g_full.dtype:
            pass  # the first Bottleneck's data gradient already summed into the right half in place (GradJoin.out)
        elif g_right is not None:
            dt = g_full.dtype
            if not _dense_ok(g_full, dt):
                g_full = grad_nhwc(g_full, dt)
            g_right = grad_nhwc(g_right, dt)
            # g_full is the (privately owned) output of the consumer conv's data-gradient kernel
            check(L().ymi_add_inplace(_byref(as_ymi(g_right)), _byref(as_ymi(g_full[:, c:])), stream_ptr()), "add_inplace")
        return g_full, None


def c2f_split(t, c):
    return _C2fSplit.apply(t, int(c))


class _Upsample2x(torch.autograd.Function):
    """nn.Upsample(None, 2, 'nearest') (yolov8.yaml:759,764) and its adjoint (2x2 block sums).  slot: write into a slice
    of the consuming Concat's buffer; join: the input has other consumers, whose gradient the adjoint accumulates onto."""

    @staticmethod
    def forward(ctx, x, slot=None, join=None):
        n, c, h, w = x.shape
        out = slot.view(n, c, 2 * h, 2 * w, x.dtype) if slot is not None else empty_nhwc(n, c, 2 * h, 2 * w, x.dtype, x.device)
        check(L().ymi_upsample2x(_byref(as_ymi(x)), _byref(as_ymi(out)), stream_ptr()), "upsample2x")
        ctx.join = join
        return out

    @staticmethod
    def backward(ctx, g):
        g = grad_nhwc(g, g.dtype)
        n, c, h, w = g.shape
        adds = ctx.join.arrive() if ctx.join is not None else []
        if adds:
            # the other consumers' gradient (e.g. the slice a Concat's consumer wrote for this tensor) is private to this
            # join: accumulate the block sums onto it in place
            dx = grad_nhwc(adds[0], g.dtype)
            check(L().ymi_upsample2x_bwd_acc(_byref(as_ymi(g)), _byref(as_ymi(dx)), stream_ptr()), "upsample2x_bwd_acc")
            if len(adds) > 1:
                _accumulate(dx, adds[1:])
            return dx, None, None
        dx = empty_nhwc(n, c, h // 2, w // 2, g.dtype, g.device)
        check(L().ymi_upsample2x_bwd(_byref(as_ymi(g)), _byref(as_ymi(dx)), stream_ptr()), "upsample2x_bwd")
        if adds is None:
            ctx.join.deposit(dx)
            return None, None, None
        return dx, None, None


def upsample2x(x, slot=None):
    return _Upsample2x.apply(x, slot, join_of(x))


# ----------------------------------------------------------------------------------- SPPF pools
class _SppfPool(torch.autograd.Function):
    """cat[y0, mp(y0), mp(mp(y0)), mp(mp(mp(y0)))] in one buffer: block.py:222-226."""

    @staticmethod
    def forward(ctx, y0, k, cat=None):
        n, c, h, w = y0.shape
        if cat is None:
            cat = empty_nhwc(n, 4 * c, h, w, y0.dtype, y0.device)
        sl = [cat[:, i * c : (i + 1) * c] for i in range(4)]
        if not (y0.data_ptr() == sl[0].data_ptr() and y0.stride() == sl[0].stride()):  # (the producer given the slot already wrote it)
            check(L().ymi_copy(_byref(as_ymi(y0)), _byref(as_ymi(sl[0])), stream_ptr()), "copy")
        check(L().ymi_sppf_pool3_fwd(_byref(as_ymi(y0)), k, _byref(as_ymi(sl[1])), _byref(as_ymi(sl[2])), _byref(as_ymi(sl[3])), stream_ptr()), "sppf_pool3_fwd")
        ctx.save_for_backward(cat)
        ctx.k, ctx.c = k, c
        return cat

    @staticmethod
    def backward(ctx, g):
        (cat,) = ctx.saved_tensors
        c, k = ctx.c, ctx.k
        n, _, h, w = cat.shape
        g = grad_nhwc(g, cat.dtype)
        y = [cat[:, i * c : (i + 1) * c] for i in range(3)]
        d = [g[:, i * c : (i + 1) * c] for i in range(4)]
        dx = empty_nhwc(n, c, h, w, cat.dtype, cat.device)
        nws = int(L().ymi_sppf_pool3_bwd_workspace(n, h, w, c, ymi_dtype(cat.dtype)))
        ws = torch.empty(nws, dtype=torch.uint8, device=cat.device) if nws else None
        check(
            L().ymi_sppf_pool3_bwd(_byref(as_ymi(y[0])), _byref(as_ymi(y[1])), _byref(as_ymi(y[2])), k, _byref(as_ymi(d[0])), _byref(as_ymi(d[1])),
                                   _byref(as_ymi(d[2])), _byref(as_ymi(d[3])), _byref(as_ymi(dx)), ptr(ws) if ws is not None else None, nws, stream_ptr()),
            "sppf_pool3_bwd",
        )
        return dx, None, None


def sppf_pool_cat(y0, k, cat=None):
    """cat: optional concat buffer whose slice 0 y0 already is (SPPF.cv1 wrote it there)."""
    return _SppfPool.apply(y0, int(k), cat)


# ----------------------------------------------------------------------------------------- CBAM
class _Cbam(torch.autograd.Function):
    """cbam.py:62-71 (channel attention :29-38, spatial attention :48-53)."""

    @staticmethod
    def forward(ctx, x, w1, w2, wsa, slot=None):
        n, c, h, w = x.shape
        hidden = w1.shape[0]
        ksa = wsa.shape[-1]
        dev = x.device
        out = slot.view(n, c, h, w, x.dtype) if slot is not None else empty_nhwc(n, c, h, w, x.dtype, dev)
        f32 = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        ca = torch.empty((n, c), **f32)
        pooled = torch.empty((n, 2, c), **f32)
        pool_arg = torch.empty((n, c), **i32)
        smap = torch.empty((n, h, w, 2), **f32)
        smap_arg = torch.empty((n, h, w), **i32)
        sa = torch.empty((n, h, w), **f32)
        w1c, w2c, wsc = w1.detach().reshape(hidden, c).contiguous(), w2.detach().reshape(c, hidden).contiguous(), wsa.detach().reshape(2, ksa, ksa).contiguous()
        check(
            L().ymi_cbam_fwd(_byref(as_ymi(x)), ptr(w1c), ptr(w2c), hidden, ptr(wsc), ksa, _byref(as_ymi(out)), ptr(ca), ptr(pooled), ptr(pool_arg),
                             ptr(smap), ptr(smap_arg), ptr(sa), stream_ptr()),
            "cbam_fwd",
        )
        ctx.save_for_backward(x, w1, w2, wsa, ca, pooled, pool_arg, smap, smap_arg, sa)
        return out

    @staticmethod
    def backward(ctx, g):
        x, w1, w2, wsa, ca, pooled, pool_arg, smap, smap_arg, sa = ctx.saved_tensors
        n, c, h, w = x.shape
        hidden, ksa = w1.shape[0], wsa.shape[-1]
        dev = x.device
        g = grad_nhwc(g, x.dtype)
        dx = empty_nhwc(n, c, h, w, x.dtype, dev)
        dw1 = torch.empty((hidden, c), dtype=torch.float32, device=dev)
        dw2 = torch.empty((c, hidden), dtype=torch.float32, device=dev)
        dwsa = torch.empty((2, ksa, ksa), dtype=torch.float32, device=dev)
        w1c, w2c, wsc = w1.detach().reshape(hidden, c).contiguous(), w2.detach().reshape(c, hidden).contiguous(), wsa.detach().reshape(2, ksa, ksa).contiguous()
        ws = workspace(L().ymi_cbam_bwd_workspace(n, h, w, c, hidden), dev, "cbam")
        check(
            L().ymi_cbam_bwd(_byref(as_ymi(x)), _byref(as_ymi(g)), ptr(w1c), ptr(w2c), hidden, ptr(wsc), ksa, ptr(ca), ptr(pooled), ptr(pool_arg),
                             ptr(smap), ptr(smap_arg), ptr(sa), _byref(as_ymi(dx)), ptr(dw1), ptr(dw2), ptr(dwsa), ptr(ws), ws.numel(), stream_ptr()),
            "cbam_bwd",
        )
        return dx, dw1.view(w1.shape), dw2.view(w2.shape), dwsa.view(wsa.shape), None


def cbam(x, w1, w2, wsa, slot=None):
    return _Cbam.apply(x, w1, w2, wsa, slot)


# ----------------------------------------------------------------------------------- SwinBlock
def window_pad(h, w, ws):
    return round_up(h, ws), round_up(w, ws)


class _LayerNorm(torch.autograd.Function):
    """LayerNorm over channels.  ws > 0: x is the NHWC image and rows are gathered through the window map
    (pad + rearrange + window_partition + norm1, swin_block.py:41-50); ws == 0: x is a token matrix (norm2)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, ws, join=None):
        dev = x.device
        if ws > 0:
            n, c, h, w = x.shape
            hp, wp = window_pad(h, w, ws)
            t = n * hp * wp
        else:
            t, c = x.shape
        out = torch.empty((t, c), dtype=x.dtype, device=dev)
        stats = torch.empty((2, t), dtype=torch.float32, device=dev)
        check(L().ymi_layernorm_fwd(_byref(as_ymi(x)), ws, ptr(gamma), ptr(beta), eps, _byref(as_ymi(out)), ptr(stats[0]), ptr(stats[1]), stream_ptr()), "layernorm_fwd")
        ctx.save_for_backward(x, gamma, stats)
        ctx.ws = ws
        ctx.join = join
        return out

    @staticmethod
    def backward(ctx, g):
        x, gamma, stats = ctx.saved_tensors
        ws = ctx.ws
        dev = x.device
        g = grad_nhwc(g, x.dtype)
        c = x.shape[1]
        dx = empty_nhwc(*x.shape, x.dtype, dev) if ws > 0 else torch.empty_like(x)
        dgb = torch.empty((2, c), dtype=torch.float32, device=dev)
        wsb = workspace(2048 * 2 * c * 4 + 256, dev, "ln")
        adds = ctx.join.arrive() if ctx.join is not None else []
        fa = _prep_adds(adds, x.dtype, ws > 0)
        if ws > 0 and fa and fa[0].shape != x.shape:
            raise RuntimeError("layernorm backward: addend shape")
        check(
            L().ymi_layernorm_bwd_add(_byref(as_ymi(x)), ws, _byref(as_ymi(g)), ptr(gamma), ptr(stats[0]), ptr(stats[1]),
                                      _byref(as_ymi(fa[0])) if fa else None, _byref(as_ymi(dx)), ptr(dgb[0]), ptr(dgb[1]), ptr(wsb), wsb.numel(), stream_ptr()),
            "layernorm_bwd",
        )
        if len(fa) > 1:
            _accumulate(dx, fa[1:])
        if adds is None:
            ctx.join.deposit(dx)
            dx = None
        return dx, dgb[0], dgb[1], None, None, None


def layernorm(x, ln, ws=0):
    return _LayerNorm.apply(x, ln.weight, ln.bias, float(ln.eps), int(ws), join_of(x))


class _WindowAttention(torch.autograd.Function):
    """softmax(q k^T / sqrt(hd)) v per (window, head) on packed qkv tokens: the core of
    nn.MultiheadAttention as called at swin_block.py:51 (no mask: pad tokens are ordinary keys)."""

    @staticmethod
    def forward(ctx, qkv, wlen, heads):
        t, c3 = qkv.shape
        c = c3 // 3
        out = torch.empty((t, c), dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty((t, heads), dtype=torch.float32, device=qkv.device)
        check(L().ymi_window_attention_fwd(_byref(as_ymi(qkv)), wlen, heads, _byref(as_ymi(out)), ptr(lse), stream_ptr()), "window_attention_fwd")
        ctx.save_for_backward(qkv, out, lse)
        ctx.cfg = (wlen, heads)
        return out

    @staticmethod
    def backward(ctx, g):
        qkv, out, lse = ctx.saved_tensors
        wlen, heads = ctx.cfg
        g = grad_nhwc(g, qkv.dtype)
        dqkv = torch.empty_like(qkv)
        check(
            L().ymi_window_attention_bwd(_byref(as_ymi(qkv)), _byref(as_ymi(out)), _byref(as_ymi(g)), ptr(lse), wlen, heads, _byref(as_ymi(dqkv)), stream_ptr()),
            "window_attention_bwd",
        )
        return dqkv, None, None


def window_attention(qkv, wlen, heads):
    return _WindowAttention.apply(qkv, int(wlen), int(heads))


class _WindowReverse(torch.autograd.Function):
    """tokens -> NHWC image with the padding cropped (window_reverse + rearrange + crop, swin_block.py:55-58)."""

    @staticmethod
    def forward(ctx, tokens, n, h, w, ws, slot=None):
        c = tokens.shape[1]
        out = slot.view(n, c, h, w, tokens.dtype) if slot is not None else empty_nhwc(n, c, h, w, tokens.dtype, tokens.device)
        check(L().ymi_window_reverse(_byref(as_ymi(tokens)), ws, _byref(as_ymi(out)), stream_ptr()), "window_reverse")
        ctx.cfg = (ws, tokens.shape[0])
        return out

    @staticmethod
    def backward(ctx, g):
        ws, t = ctx.cfg
        g = grad_nhwc(g, g.dtype)
        d = torch.empty((t, g.shape[1]), dtype=g.dtype, device=g.device)
        check(L().ymi_window_partition(_byref(as_ymi(g)), ws, _byref(as_ymi(d)), stream_ptr()), "window_partition")
        return d, None, None, None, None, None


def window_reverse(tokens, n, h, w, ws, slot=None):
    return _WindowReverse.apply(tokens, int(n), int(h), int(w), int(ws), slot)


def window_partition_index(n, hp, wp, ws, device):
    idx = torch.empty(n * hp * wp, dtype=torch.int32, device=device)
    check(L().ymi_window_partition_index(n, hp, wp, ws, ptr(idx), stream_ptr()), "window_partition_index")
    return idx


def window_partition(x, ws):
    """stand-alone copy form (tests): NHWC image -> [T, C] tokens with zero padding."""
    n, c, h, w = x.shape
    hp, wp = window_pad(h, w, ws)
    out = torch.empty((n * hp * wp, c), dtype=x.dtype, device=x.device)
    check(L().ymi_window_partition(_byref(as_ymi(x)), ws, _byref(as_ymi(out)), stream_ptr()), "window_partition")
    return out


# ---- v8 detection loss on the Detect maps (csrc/loss.hip) -------------------------------------------------------
def detect_targets(batch_idx, cls, bboxes, batch_size, max_boxes, img_w, img_h, device):
    """ragged label rows -> dense [B, max_boxes, 5] (class, xyxy pixels); reference loss.py:176-191 preprocess."""
    n = int(batch_idx.numel())
    out = torch.empty(batch_size, max_boxes, 5, dtype=torch.float32, device=device)
    bi = batch_idx.to(device=device, dtype=torch.float32).reshape(-1).contiguous()
    cl = cls.to(device=device, dtype=torch.float32).reshape(-1).contiguous()
    bb = bboxes.to(device=device, dtype=torch.float32).reshape(-1, 4).contiguous()
    check(L().ymi_detect_targets(ptr(bi) if n else None, ptr(cl) if n else None, ptr(bb) if n else None, n, batch_size, max_boxes, float(img_w),
                                 float(img_h), ptr(out), stream_ptr()), "detect_targets")
    return out


def _map_array(maps):
    return (_lib.YmiTensor * len(maps))(*[as_ymi(t) for t in maps])


class _DetectLoss(torch.autograd.Function):
    """(box maps, class maps) -> (loss [3] * scale[:3], items [3] = loss * scale[3:]) with loss = (box, cls, dfl) sums / max(sum of
    target scores, 1): both results of the criterion (reference loss.py:250-255) leave the last loss kernel, no elementwise launches."""

    @staticmethod
    def forward(ctx, targets, strides, topk, alpha, beta, scale6, *maps):
        nl = len(maps) // 2
        box, cls = maps[:nl], maps[nl:]
        dev = box[0].device
        b = box[0].shape[0]
        anchors = sum(int(t.shape[2] * t.shape[3]) for t in box)
        g = int(targets.shape[1])
        sb, wb = ctypes.c_size_t(0), ctypes.c_size_t(0)
        check(L().ymi_detect_loss_sizes(b, anchors, g, _byref(sb), _byref(wb)), "detect_loss_sizes")
        state = torch.empty(sb.value, dtype=torch.uint8, device=dev)
        ws = workspace(wb.value, dev, "detloss")
        out = torch.empty(6, dtype=torch.float32, device=dev)
        st = (ctypes.c_float * nl)(*[float(s) for s in strides])
        check(
            L().ymi_detect_loss_fwd(nl, _map_array(box), _map_array(cls), st, ptr(targets), g, int(topk), float(alpha), float(beta), ptr(scale6), ptr(out),
                                    ptr(state), state.numel(), ptr(ws), ws.numel(), stream_ptr()),
            "detect_loss_fwd",
        )
        ctx.save_for_backward(state, scale6, *maps)
        ctx.strides = st
        loss, items = out[:3], out[3:]
        ctx.mark_non_differentiable(items)
        ctx.set_materialize_grads(False)  # (no zero tensor for `items`: it was a fill launch per step)
        return loss, items

    @staticmethod
    def backward(ctx, gl, _gitems):
        state, scale6, *maps = ctx.saved_tensors
        nl = len(maps) // 2
        box, cls = maps[:nl], maps[nl:]
        if gl is None:
            return (None,) * (6 + len(maps))
        if gl.dtype != torch.float32 or not gl.is_contiguous():
            gl = gl.to(torch.float32).contiguous()
        dbox = [torch.empty_like(t) for t in box]
        pairs = [padded_grad_like(t, zero=False) for t in cls]
        if len({p[0].shape[1] for p in pairs}) != 1:  # (levels padded differently: the kernel takes one width) zeroed buffers, class channels only
            pairs = [(v, v) for v in (padded_grad_like(t)[1] for t in cls)]
        dcls_k, dcls = [p[0] for p in pairs], [p[1] for p in pairs]
        check(  # the kernel differentiates the unscaled sums: the forward's scale rides along as grad_scale; padding channels are zeroed there
            L().ymi_detect_loss_bwd(nl, _map_array(box), _map_array(cls), ctx.strides, ptr(state), state.numel(), ptr(gl), ptr(scale6), _map_array(dbox),
                                    _map_array(dcls_k), stream_ptr()),
            "detect_loss_bwd",
        )
        return (None, None, None, None, None, None, *dbox, *dcls)


def detect_loss(box_maps, cls_maps, strides, targets, scale6, topk=10, alpha=0.5, beta=6.0):
    """-> (loss [3] * scale6[:3] (differentiable), items [3] = loss * scale6[3:] (detached))."""
    return _DetectLoss.apply(targets, tuple(strides), topk, alpha, beta, scale6, *box_maps, *cls_maps)


def detect_decode(box_maps, cls_maps, strides):
    """Detect._inference of reference head.py:103-142 on the per-level maps -> [B, 4+nc, A] float32 (no gradient)."""
    nl = len(box_maps)
    b = box_maps[0].shape[0]
    nc = cls_maps[0].shape[1]
    anchors = sum(int(t.shape[2] * t.shape[3]) for t in box_maps)
    y = torch.empty((b, 4 + nc, anchors), dtype=torch.float32, device=box_maps[0].device)
    st = (ctypes.c_float * nl)(*[float(s) for s in strides])
    check(L().ymi_detect_decode(nl, _map_array([t.detach() for t in box_maps]), _map_array([t.detach() for t in cls_maps]), st, ptr(y), stream_ptr()), "detect_decode")
    return y
