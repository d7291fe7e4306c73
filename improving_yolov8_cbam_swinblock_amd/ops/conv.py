"""Convolution blocks as autograd Functions over the C ABI: data-gradient helpers (joined sums, multi-problem launches), Conv + BatchNorm + SiLU,
the direct first layer, Detect's sibling pairs and its lockstep training path, biased convolutions / linears (reference: nn/modules/conv.py:37-91,
nn/modules/head.py:36-76)."""
import ctypes
import os

import torch

from .. import _lib
from .._lib import ACT_GELU, ACT_NONE, ACT_SILU, ConvProblem, DgradProblem, as_ymi, check, chunk_elems, empty_nhwc, is_nhwc, ptr, stream_ptr, workspace, ymi_dtype
from .base import (  # noqa: F401
    HOOKS, L, _GradBuffer, _GradSlot, _accumulate, _as4d, _byref, _conv_out_hw, _deferred_twice, _dense_ok, _in_backward, _join_plain, _note_use,
    _prep_adds, _stat_acc, compute_dtype, grad_nhwc, join_of, mark_join, round_up,
)
from .weights import (  # noqa: F401
    _adoptable, _deferred, _flush_wgrads, _new_dw, _wgrad_maybe_async, pack_conv_dgrad, pack_conv_dgrad_pair, pack_conv_fwd, pack_conv_fwd_pair,
)

def _dgrad_prepare(dy, weight4, k, stride, in_shape, dtype, adds=None, out=None, packed=None):
    """the arguments of one data-gradient GEMM (see _dgrad) -> job dict; _dgrad_finish completes it after the launch."""
    n, cp, h, w = in_shape
    ty = as_ymi(dy)
    if packed is not None:
        wd, cin = packed
    else:
        cin = weight4.shape[1]
        wd = pack_conv_dgrad(weight4, ty.c, stride, dtype)
    dx = out if (out is not None and tuple(out.shape) == (n, cp, h, w) and cp == cin) else empty_nhwc(n, cp, h, w, dtype, dy.device)
    dxv = dx
    if cp != cin:
        dx.zero_()
        dxv = dx[:, :cin]
    adds = _prep_adds(adds, dtype, True)
    sparse = k == 1 and stride > 1  # pixels between the strides receive no gradient: the kernel leaves them as prepared here (zeros)
    if sparse and cp == cin:
        dx.zero_()  # (_dgrad_joined never passes an in-place `out` in this case)
    fused = adds[:2] if (cp == cin and not sparse) else []
    return {"dy": dy, "ty": ty, "wd": wd, "cin": cin, "k": k, "stride": stride, "fused": fused, "rest": adds[len(fused):], "dx": dx, "dxv": dxv}


def _dgrad_finish(job):
    if job["rest"]:
        _accumulate(job["dxv"], job["rest"])
    return job["dx"]


def _dgrad(dy, weight4, k, stride, in_shape, dtype, adds=None, out=None, packed=None):
    """dx [N, C_in(padded), H, W] (NHWC) from dy and the OIHW weight (+ up to two addends summed in the GEMM's
    epilogue, further ones by accumulate launches); zero-padded input channels get zero.  out: write the result into this
    NHWC view (it may be one of the addends: the epilogue reads an addend before it stores the sum).
    packed: (operand, cin) of an already packed data-gradient operand (weight4 is then unused)."""
    return _dgrad_launch(_dgrad_prepare(dy, weight4, k, stride, in_shape, dtype, adds, out, packed))


def _dgrad_launch(j):
    fused = j["fused"]
    a1 = _byref(as_ymi(fused[0])) if len(fused) > 0 else None
    a2 = _byref(as_ymi(fused[1])) if len(fused) > 1 else None
    check(L().ymi_conv2d_bwd_data_add(_byref(j["ty"]), ptr(j["wd"]), j["cin"], j["k"], j["k"], j["stride"], a1, a2, _byref(as_ymi(j["dxv"])), stream_ptr()),
          "conv2d_bwd_data")
    return _dgrad_finish(j)


def _width_class(c, dtype):
    """problems of one multi-problem GEMM launch must agree on this (the K-step form of the kernel: csrc/igemm.hip launch_igemm_n)."""
    cpt = int(c) // chunk_elems(dtype)
    return (cpt % 4 == 0, cpt % 8 == 0)


def _dgrad_multi(jobs, dtype):
    """the stride-1 data-gradient GEMMs of several INDEPENDENT convolutions (jobs of _dgrad_prepare), one launch per width class."""
    groups = {}
    for j in jobs:
        if j["stride"] != 1:
            raise RuntimeError("_dgrad_multi: stride 1 only")
        groups.setdefault(_width_class(j["ty"].c, dtype), []).append(j)
    for g in groups.values():
        for s in range(0, len(g), 8):
            chunk = g[s : s + 8]
            arr = (DgradProblem * len(chunk))()
            keep = []
            for q, j in zip(arr, chunk):
                ts = [j["ty"], as_ymi(j["dxv"])] + [as_ymi(a) for a in j["fused"]]
                keep.append(ts)
                q.dy, q.dx = ctypes.pointer(ts[0]), ctypes.pointer(ts[1])
                q.w_dgrad_packed, q.cin, q.k = j["wd"].data_ptr(), j["cin"], j["k"]
                if len(ts) > 2:
                    q.add1 = ctypes.pointer(ts[2])
                if len(ts) > 3:
                    q.add2 = ctypes.pointer(ts[3])
            check(L().ymi_conv2d_bwd_data_multi(arr, len(chunk), stream_ptr()), "conv2d_bwd_data_multi")
    return [_dgrad_finish(j) for j in jobs]


def _conv_fwd_multi(problems, dtype):
    """several INDEPENDENT stride-1 convolutions, one launch per width class.  problems: dicts x, wp, cout, k, y and optionally bias, or
    part / pstride / poff (statistics rows, see ymi_conv_problem); 'blocks' (statistics rows written) is filled in."""
    groups = {}
    for p in problems:
        groups.setdefault((_width_class(p["x"].shape[1], dtype), p.get("part") is not None), []).append(p)
    for g in groups.values():
        for s in range(0, len(g), 8):
            chunk = g[s : s + 8]
            arr = (ConvProblem * len(chunk))()
            keep = []
            for q, p in zip(arr, chunk):
                tx, ty = as_ymi(p["x"]), as_ymi(p["y"])
                keep.append((tx, ty))
                q.x, q.y = ctypes.pointer(tx), ctypes.pointer(ty)
                q.w_packed, q.cout, q.kh, q.kw, q.stride, q.act = p["wp"].data_ptr(), p["cout"], p["k"], p["k"], 1, ACT_NONE
                if p.get("bias") is not None:
                    q.bias = p["bias"].data_ptr()
                if p.get("part") is not None:
                    q.stat_partials, q.stat_stride, q.stat_offset = p["part"].data_ptr(), p.get("pstride", 0), p.get("poff", 0)
            check(L().ymi_conv2d_fwd_multi(arr, len(chunk), stream_ptr()), "conv2d_fwd_multi")
            for q, p in zip(arr, chunk):
                p["blocks"] = int(q.stat_blocks)


def _dgrad_joined_prepare(join, dy, weight4, k, stride, in_shape, dtype, packed=None):
    """-> (job, deposit): the data-gradient GEMM of a consumer of a (possibly joined) tensor, with the join's earlier contributions as
    addends when this is the last consumer; deposit: the result is a contribution to hand to the join (_dgrad_joined_finish)."""
    adds = join.arrive() if join is not None else []
    out = None
    if adds is not None and join is not None and join.dst is not None:
        # the tensor's gradient has a prepared place (a channel slice of its producer's gradient buffer: _ChanSplit2): the total goes there
        cand = join.dst.view(dtype)
        join.dst = None
        if cand is not None and tuple(cand.shape) == tuple(in_shape) and _dense_ok(cand, dtype) and not (k == 1 and stride > 1):
            out = cand
    elif adds is not None and join is not None and join.out is not None:
        out, join.out = join.out, None
        if len(adds) < 2 and _dense_ok(out, dtype) and tuple(out.shape) == tuple(in_shape) and not (k == 1 and stride > 1):
            adds = list(adds) + [out]  # the contribution already in the buffer rides as an addend; the total replaces it
        else:
            out = None  # (left for _C2fSplit's own add)
    return _dgrad_prepare(dy, weight4, k, stride, in_shape, dtype, adds, out, packed), adds is None


def _dgrad_joined_finish(join, dx, deposit):
    if deposit:
        join.deposit(dx)
        return None
    return dx


def _dgrad_joined(join, dy, weight4, k, stride, in_shape, dtype, packed=None):
    """data gradient of a consumer of a (possibly joined) tensor: deposits (and returns None) unless it is the last consumer."""
    job, deposit = _dgrad_joined_prepare(join, dy, weight4, k, stride, in_shape, dtype, packed)
    return _dgrad_joined_finish(join, _dgrad_launch(job), deposit)


class _ConvBnAct(torch.autograd.Function):
    """act(BatchNorm_train(conv(x))) (+ residual).  Reference: Conv.forward, nn/modules/conv.py:69-79
    (+ Bottleneck add, nn/modules/block.py:488)."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, running_mean, running_var, stride, eps, momentum, act, residual, slot=None, join=None, res_join=None):
        dtype = x.dtype
        o, i, k, _ = weight.shape
        n, cp, h, w = x.shape
        ho, wo = _conv_out_hw(h, w, k, stride)
        dev = x.device
        _note_use(weight)
        wp = pack_conv_fwd(weight, cp, dtype)
        raw = empty_nhwc(n, o, ho, wo, dtype, dev)
        out = slot.view(n, o, ho, wo, dtype) if slot is not None else empty_nhwc(n, o, ho, wo, dtype, dev)
        stats = torch.empty((2, o), dtype=torch.float32, device=dev)
        m = n * ho * wo
        need = (L().ymi_conv2d_stat_blocks(m, o) * 2 * o + 2 * o) * 4
        ws = workspace(need, dev, "conv")
        res = residual
        tres = _byref(as_ymi(res)) if res is not None else None
        if (HOOKS["stat_atomics"] and o % 4 == 0 and L().ymi_conv2d_bn_silu_fwd_acc_ok(_byref(as_ymi(raw)), _byref(as_ymi(out)), tres)
                and all(t is None or t.data_ptr() % 16 == 0 for t in (gamma, beta, running_mean, running_var))):
            # statistics as fixed-point atomic sums, finalized in the affine pass's prologue: two launches instead of three (or four)
            acc = _stat_acc(o, dev)
            check(
                L().ymi_conv2d_bn_silu_fwd_acc(
                    _byref(as_ymi(x)), ptr(wp), o, k, k, stride, ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
                    momentum, eps, act, tres, _byref(as_ymi(raw)), _byref(as_ymi(out)), ptr(stats[0]), ptr(stats[1]), ptr(acc), stream_ptr(),
                ),
                "conv2d_bn_silu_fwd_acc",
            )
        else:
            check(
                L().ymi_conv2d_bn_silu_fwd(
                    _byref(as_ymi(x)), ptr(wp), o, k, k, stride, ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
                    momentum, eps, act, tres, _byref(as_ymi(raw)), _byref(as_ymi(out)), ptr(stats[0]), ptr(stats[1]), ptr(ws), ws.numel(), stream_ptr(),
                ),
                "conv2d_bn_silu_fwd",
            )
        ctx.save_for_backward(x, weight, gamma, beta, raw, stats)
        ctx.cfg = (stride, act, i, residual is not None)
        ctx.joins = (join, res_join)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight, gamma, beta, raw, stats = ctx.saved_tensors
        stride, act, cin, has_res = ctx.cfg
        dtype = x.dtype
        o, _, k, _ = weight.shape
        dev = x.device
        dout = grad_nhwc(dout, dtype)
        draw = empty_nhwc(*raw.shape, dtype, dev)
        # two separate tensors: AccumulateGrad adopts them as .grad without a clone (views would be copied)
        dgamma = torch.empty(o, dtype=torch.float32, device=dev)
        dbeta = torch.empty(o, dtype=torch.float32, device=dev)
        ws = workspace(2048 * 2 * o * 4 + 256, dev, "bnbwd")
        check(
            L().ymi_bn_act_bwd(_byref(as_ymi(dout)), _byref(as_ymi(raw)), ptr(gamma), ptr(stats[0]), ptr(stats[1]), ptr(beta), act,
                               _byref(as_ymi(draw)), ptr(dgamma), ptr(dbeta), ptr(ws), ws.numel(), stream_ptr()),
            "bn_act_bwd",
        )
        join, res_join = ctx.joins
        # the residual hand-through first: when x is both the input and the residual (Bottleneck shortcut), the data
        # gradient below is then the join's last consumer and adds `dout` in its epilogue
        dres = _join_plain(res_join, dout) if (has_res and ctx.needs_input_grad[10]) else None
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _dgrad_joined(join, draw, weight, k, stride, x.shape, dtype)
        dw = None
        if ctx.needs_input_grad[1]:  # (a frozen conv weight: no GEMM, and nothing for a deferred slab sum to write into)
            dw, _ = _wgrad_maybe_async(x, draw, o, cin, k, stride, False, (weight,))
        return dx, dw, dgamma, dbeta, None, None, None, None, None, None, dres, None, None, None


def conv_bn_act(x, weight, bn, stride, act=ACT_SILU, residual=None, slot=None):
    """train-mode Conv block on an internal (NHWC) tensor; updates bn.running_* in place.  slot: optional OutSlot.
    Tensors marked with mark_join() (several consumers) have their gradient sums formed in the data-gradient epilogue."""
    if bn.momentum is None:
        raise RuntimeError("BatchNorm with cumulative moving average (momentum=None) is not supported")
    if torch.is_grad_enabled() and weight.shape[0] % chunk_elems(x.dtype) != 0 and (x.requires_grad or weight.requires_grad):
        # the backward kernels (BatchNorm backward, data / weight gradient GEMMs) read the output gradient in 16-byte chunks
        raise NotImplementedError(f"training a Conv with {weight.shape[0]} output channels in {x.dtype}: the backward kernels need channel counts in "
                                  f"whole 16-byte chunks (multiples of {chunk_elems(x.dtype)}); every standard YOLOv8 width is - use float32 for this width")
    out = _ConvBnAct.apply(x, weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, int(stride), float(bn.eps), float(bn.momentum), int(act), residual, slot,
                           join_of(x), join_of(residual) if residual is not None else None)
    if bn.num_batches_tracked is not None:
        if _deferred_counters is not None:
            _deferred_counters.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked.add_(1)
    return out


class _FirstConvBnAct(torch.autograd.Function):
    """The model's first Conv block on the caller's float32 NCHW image (reference yolov8.yaml:738 through conv.py:50-79) as direct
    kernels that never store the raw convolution output (csrc/first_conv.hip): forward = statistics pass, finalize, apply pass;
    backward = reduce pass, final sums, fused BatchNorm-apply + weight-gradient pass - each recomputes the 27-tap convolution from the
    NHWC bfloat16 4-channel copy of the image that the statistics pass leaves behind (26 MB saved for backward instead of a 52 MB
    padded copy + the 210 MB raw output).  The image receives no gradient."""

    @staticmethod
    def forward(ctx, img, weight, gamma, beta, running_mean, running_var, eps, momentum, act):
        n, c, h, w = img.shape
        o = weight.shape[0]
        dt = torch.bfloat16
        dev = img.device
        ho, wo = _conv_out_hw(h, w, 3, 2)
        _note_use(weight)
        x4 = torch.empty((n, h, w, 4), dtype=dt, device=dev).permute(0, 3, 1, 2)
        out = empty_nhwc(n, o, ho, wo, dt, dev)
        stats = torch.empty((2, o), dtype=torch.float32, device=dev)
        need = (2 * o + (L().ymi_first_conv_stat_blocks(n, h, w) + 64) * 2 * o) * 4
        ws = workspace(need, dev, "conv")
        check(
            L().ymi_first_conv_bn_act_fwd(ptr(img), n, c, h, w, ptr(weight.detach()), o, ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), momentum, eps,
                                          act, _byref(as_ymi(x4)), _byref(as_ymi(out)), ptr(stats[0]), ptr(stats[1]), ptr(ws), ws.numel(), stream_ptr()),
            "first_conv_bn_act_fwd",
        )
        ctx.save_for_backward(x4, weight, gamma, beta, stats)
        ctx.act = act
        return out

    @staticmethod
    def backward(ctx, dout):
        x4, weight, gamma, beta, stats = ctx.saved_tensors
        o, cin, k, _ = weight.shape
        dev = x4.device
        dout = grad_nhwc(dout, torch.bfloat16)
        dgamma = torch.empty(o, dtype=torch.float32, device=dev)
        dbeta = torch.empty(o, dtype=torch.float32, device=dev)
        n, _, h, w = x4.shape
        need = int(L().ymi_first_conv_bwd_workspace(n, h, w, o))
        defer = (ctx.needs_input_grad[1] and _deferred["on"] and _in_backward() and _adoptable((weight,)) and not _deferred_twice((weight,)))
        dw = _new_dw(o, cin, k, dev, (weight,), None)
        # (the slabs live in the workspace: a deferred sum needs it alive until the end-of-pass flush)
        ws = torch.empty(need, dtype=torch.uint8, device=dev) if defer else workspace(need, dev, "firstconv")
        rec = _lib.WgradPending() if defer else None
        check(
            L().ymi_first_conv_bn_act_bwd(_byref(as_ymi(x4)), ptr(weight.detach()), cin, o, ptr(gamma), ptr(beta), ptr(stats[0]), ptr(stats[1]), ctx.act,
                                          _byref(as_ymi(dout)), ptr(dgamma), ptr(dbeta), ptr(dw), ptr(ws), ws.numel(), _byref(rec) if defer else None, stream_ptr()),
            "first_conv_bn_act_bwd",
        )
        if defer:
            task = torch._C._current_graph_task_id()
            if _deferred["task"] != task:
                _deferred["records"], _deferred["keep"], _deferred["owners"], _deferred["bias"] = [], [], [], []
                torch.autograd.Variable._execution_engine.queue_callback(_flush_wgrads)
                _deferred["task"] = task
            _deferred["records"].append(rec)
            _deferred["keep"].append((ws, x4, dout))
            _deferred["owners"].append(weight)
            _deferred["bias"].append(None)
        return None, (dw if ctx.needs_input_grad[1] else None), dgamma, dbeta, None, None, None, None, None


def first_conv_ok(x, conv, residual, slot):
    """the direct first-layer kernel applies: a float32 NCHW image that needs no gradient, bfloat16 compute, 3x3 stride 2, <= 4 input channels."""
    return (residual is None and slot is None and torch.is_tensor(x) and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.is_contiguous()
            and x.shape[1] <= 4 and not x.requires_grad and conv.kernel_size == (3, 3) and conv.stride == (2, 2) and conv.in_channels == x.shape[1]
            and conv.out_channels in (16, 32, 48, 64) and x.shape[2] % 2 == 0 and x.shape[3] % 4 == 0 and x.data_ptr() % 16 == 0 and compute_dtype(x) == torch.bfloat16
            and HOOKS["first_conv"])


def first_conv_bn_act(img, weight, bn, act=ACT_SILU):
    if bn.momentum is None:
        raise RuntimeError("BatchNorm with cumulative moving average (momentum=None) is not supported")
    out = _FirstConvBnAct.apply(img, weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, float(bn.eps), float(bn.momentum), int(act))
    if bn.num_batches_tracked is not None:
        if _deferred_counters is not None:
            _deferred_counters.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked.add_(1)
    return out


class _ChanSplit2(torch.autograd.Function):
    """(t[:, :c], t[:, c:]) of an NHWC tensor as two views whose gradients are formed IN PLACE in one buffer: each view carries a GradJoin
    with a prepared destination (GradJoin.dst), its consumer's data-gradient GEMM writes its slice of the buffer, and backward hands the
    buffer on without a copy (autograd's own slices would zero-fill two full tensors and add them).  Falls back to copies when a gradient
    arrives somewhere else."""

    @staticmethod
    def forward(ctx, t, c):
        ctx.c, ctx.shape = c, tuple(t.shape)
        ctx.gb = _ChanSplit2.last = _GradBuffer(t.shape, t.device)  # (chan_split2 picks it up right after apply)
        return t[:, :c], t[:, c:]

    @staticmethod
    def backward(ctx, ga, gb):
        c = ctx.c
        buf = ctx.gb.buf
        ctx.gb.buf = None
        n, ctot, h, w = ctx.shape
        g0 = ga if ga is not None else gb
        if buf is None or buf.dtype != g0.dtype:
            buf = empty_nhwc(n, ctot, h, w, g0.dtype, g0.device)
        for g, dst in ((ga, buf[:, :c]), (gb, buf[:, c:])):
            if g is None:
                dst.zero_()
            elif not (g.data_ptr() == dst.data_ptr() and g.stride() == dst.stride() and g.dtype == dst.dtype):
                check(L().ymi_copy(_byref(as_ymi(grad_nhwc(g, buf.dtype))), _byref(as_ymi(dst)), stream_ptr()), "copy")
        return buf, None


def chan_split2(t, c):
    """-> (t[:, :c], t[:, c:]); in training each half is marked with a join whose total lands in the matching slice of ONE gradient buffer."""
    a, b = _ChanSplit2.apply(t, int(c))
    if torch.is_grad_enabled() and t.requires_grad:
        gb, _ChanSplit2.last = getattr(_ChanSplit2, "last", None), None
        if gb is not None:
            for v, lo, hi in ((a, 0, int(c)), (b, int(c), t.shape[1])):
                mark_join(v, 1, force=True)
                j = join_of(v)
                if j is not None:
                    j.dst = _GradSlot(gb, lo, hi)
    return a, b


class _ConvBnActPair(torch.autograd.Function):
    """two Conv blocks of the SAME input as one convolution with oA + oB output channels: Detect's sibling branches cv2[i][0] / cv3[i][0]
    (reference head.py:44-59,71-72).  BatchNorm is per channel, so the result is exactly the two separate blocks; parameters stay the
    reference's separate tensors (their packed operands lie side by side in the weight arena).  The data gradient is ONE GEMM with
    K = taps * (oA + oB) (the GradJoin sum of the two branches disappears into the accumulator), the weight gradient one GEMM whose
    [oA + oB, cin, k, k] result is handed out as two views."""

    @staticmethod
    def forward(ctx, x, wa, ga, ba, rma, rva, wb, gb, bb, rmb, rvb, stride, eps, momentum, act, join):
        dtype = x.dtype
        oa, i, k, _ = wa.shape
        ob = wb.shape[0]
        o = oa + ob
        n, cp, h, w = x.shape
        ho, wo = _conv_out_hw(h, w, k, stride)
        dev = x.device
        _note_use(wa, wb)
        wp = pack_conv_fwd_pair(wa, wb, cp, dtype)
        raw = empty_nhwc(n, o, ho, wo, dtype, dev)
        out = empty_nhwc(n, o, ho, wo, dtype, dev)
        stats = torch.empty((2, o), dtype=torch.float32, device=dev)
        m = n * ho * wo
        need = (L().ymi_conv2d_stat_blocks(m, o) * 2 * o + 2 * o) * 4
        ws = workspace(need, dev, "conv")
        check(
            L().ymi_conv2d_bn_silu_fwd_pair(
                _byref(as_ymi(x)), ptr(wp), o, oa, k, k, stride, ptr(ga), ptr(ba), ptr(rma), ptr(rva), ptr(gb), ptr(bb), ptr(rmb), ptr(rvb),
                momentum, eps, act, _byref(as_ymi(raw)), _byref(as_ymi(out)), ptr(stats[0]), ptr(stats[1]), ptr(ws), ws.numel(), stream_ptr(),
            ),
            "conv2d_bn_silu_fwd_pair",
        )
        ctx.save_for_backward(x, wa, wb, ga, ba, gb, bb, raw, stats)
        ctx.cfg = (stride, act, i)
        ctx.join = join
        return out

    @staticmethod
    def backward(ctx, dout):
        x, wa, wb, ga, ba, gb, bb, raw, stats = ctx.saved_tensors
        stride, act, cin = ctx.cfg
        dtype = x.dtype
        oa, _, k, _ = wa.shape
        ob = wb.shape[0]
        o = oa + ob
        dev = x.device
        dout = grad_nhwc(dout, dtype)
        draw = empty_nhwc(*raw.shape, dtype, dev)
        dgamma = torch.empty(o, dtype=torch.float32, device=dev)
        dbeta = torch.empty(o, dtype=torch.float32, device=dev)
        ws = workspace(2048 * 2 * o * 4 + 256, dev, "bnbwd")
        check(
            L().ymi_bn_act_bwd_pair(_byref(as_ymi(dout)), _byref(as_ymi(raw)), ptr(ga), ptr(ba), ptr(gb), ptr(bb), oa, ptr(stats[0]), ptr(stats[1]), act,
                                    _byref(as_ymi(draw)), ptr(dgamma), ptr(dbeta), ptr(ws), ws.numel(), stream_ptr()),
            "bn_act_bwd_pair",
        )
        dx = None
        if ctx.needs_input_grad[0]:
            wd = pack_conv_dgrad_pair(wa, wb, stride, dtype)
            dx = _dgrad_joined(ctx.join, draw, None, k, stride, x.shape, dtype, packed=(wd, cin))
        nig = ctx.needs_input_grad
        dwa = dwb = None
        if nig[1] or nig[6]:
            # one GEMM for both weights; the two gradients are the halves of its [oA + oB, cin, k, k] result.  Deferred only when BOTH
            # parameters adopt their half (a frozen one would leave its half's memory to the allocator before the batched slab sum runs)
            dw, _ = _wgrad_maybe_async(x, draw, o, cin, k, stride, False, (wa, wb), pair_rows=oa)
            dwa, dwb = (dw[:oa] if nig[1] else None), (dw[oa:] if nig[6] else None)
        return (dx, dwa, dgamma[:oa], dbeta[:oa], None, None, dwb, dgamma[oa:], dbeta[oa:], None, None, None, None, None, None, None)


def conv_bn_act_pair(x, conv_a, bn_a, conv_b, bn_b, act=ACT_SILU):
    """train-mode act(BN_a(conv_a(x))) and act(BN_b(conv_b(x))) from ONE convolution -> [N, oA + oB, H', W'] (the first block's channels first).
    conv_*: nn.Conv2d parameter containers of equal kernel / stride / input width; widths in whole 16-byte chunks."""
    wa, wb = conv_a.weight, conv_b.weight
    if wa.shape[1:] != wb.shape[1:] or conv_a.stride != conv_b.stride:
        raise RuntimeError("conv_bn_act_pair: the two convolutions must share kernel size, stride and input width")
    if bn_a.momentum is None or bn_b.momentum is None or bn_a.eps != bn_b.eps or bn_a.momentum != bn_b.momentum:
        raise RuntimeError("conv_bn_act_pair: the two BatchNorms must share eps and momentum (initialize_weights sets them alike)")
    ch = chunk_elems(x.dtype)
    if wa.shape[0] % ch or wb.shape[0] % ch:
        raise NotImplementedError(f"conv_bn_act_pair: output widths {wa.shape[0]} / {wb.shape[0]} must be multiples of {ch} in {x.dtype}")
    out = _ConvBnActPair.apply(x, wa, bn_a.weight, bn_a.bias, bn_a.running_mean, bn_a.running_var, wb, bn_b.weight, bn_b.bias, bn_b.running_mean,
                               bn_b.running_var, int(conv_a.stride[0]), float(bn_a.eps), float(bn_a.momentum), int(act), join_of(x))
    for bn in (bn_a, bn_b):
        if bn.num_batches_tracked is not None:
            if _deferred_counters is not None:
                _deferred_counters.append(bn.num_batches_tracked)
            else:
                bn.num_batches_tracked.add_(1)
    return out


_DT = 25  # tensors per level of _DetectTrain: x, 4 x (weight, gamma, beta, running_mean, running_var), 2 x (weight, bias)


class _DetectTrain(torch.autograd.Function):
    """the train-mode Detect head of ALL levels (reference head.py:66-74 loops over the levels; per level cv2[i] / cv3[i] are
    Conv -> Conv -> biased 1x1, head.py:44-59), its stages run in lockstep across the levels: every GEMM stage is ONE launch over the
    problems of all levels (ymi_conv2d_fwd_multi / ymi_conv2d_bwd_data_multi) - the 40 x 40 and 20 x 20 levels fill a fraction of the
    chip on their own.  Per level the arithmetic is that of _ConvBnActPair (first convolutions), two _ConvBnAct side by side in one
    buffer (second convolutions, one BatchNorm pass over both) and two _ConvAffineAct.  Outputs per level: box map [N, 64, H, W] and
    the class map in a buffer padded to whole 16-byte rows."""

    @staticmethod
    def forward(ctx, meta, *t):
        nl, eps, momentum, joins, ncpad = meta
        lv = [t[l * _DT : (l + 1) * _DT] for l in range(nl)]
        dtype, dev = lv[0][0].dtype, lv[0][0].device
        lib = L()
        geo, bufs = [], []
        need = 0
        for v in lv:
            n, cp, h, w = v[0].shape
            c2, c3 = v[1].shape[0], v[6].shape[0]
            o, m = c2 + c3, n * h * w
            rows = lib.ymi_conv2d_stat_blocks(m, o)
            geo.append((n, cp, h, w, c2, c3, o, m, need, rows))
            need += 2 * (rows * 2 * o + 2 * o) * 4  # two BatchNorm stages: scale, shift, statistics rows
            _note_use(v[1], v[6], v[11], v[16], v[21], v[23])
        ws = workspace(need, dev, "detect").view(torch.float32)

        def region(g, stage):
            n, cp, h, w, c2, c3, o, m, off, rows = g
            base = off // 4 + stage * (rows * 2 * o + 2 * o)
            return ws[base : base + o], ws[base + o : base + 2 * o], ws[base + 2 * o : base + 2 * o + rows * 2 * o]

        def bn_stage(probs_of_level, stage, raws, params):
            """the multi-problem GEMM of a stage, then per level: statistics -> scale / shift, affine + SiLU."""
            _conv_fwd_multi([p for ps in probs_of_level for p in ps], dtype)
            outs, stats = [], []
            for g, ps, raw, (ga, ba, rma, rva, gb, bb, rmb, rvb) in zip(geo, probs_of_level, raws, params):
                n, cp, h, w, c2, c3, o, m, off, rows = g
                blocks = ps[0]["blocks"]
                if any(p["blocks"] != blocks for p in ps):
                    raise RuntimeError("_DetectTrain: the convolutions of one BatchNorm group ran with different row tiles")
                scale, shift, part = region(g, stage)
                st = torch.empty((2, o), dtype=torch.float32, device=dev)
                out = empty_nhwc(n, o, h, w, dtype, dev)
                check(lib.ymi_bn_finalize_pair(ptr(part), blocks, m, o, c2, ptr(ga), ptr(ba), ptr(rma), ptr(rva), ptr(gb), ptr(bb), ptr(rmb), ptr(rvb),
                                               momentum, eps, ptr(scale), ptr(shift), ptr(st[0]), ptr(st[1]), stream_ptr()), "bn_finalize_pair")
                check(lib.ymi_scale_shift_act(_byref(as_ymi(raw)), ptr(scale), ptr(shift), ACT_SILU, None, _byref(as_ymi(out)), stream_ptr()), "scale_shift_act")
                outs.append(out)
                stats.append(st)
            return outs, stats

        # stage 1: the two first convolutions of a level as ONE (they read the same input)
        raw1 = [empty_nhwc(g[0], g[6], g[2], g[3], dtype, dev) for g in geo]
        probs = [[{"x": v[0], "wp": pack_conv_fwd_pair(v[1], v[6], g[1], dtype), "cout": g[6], "k": v[1].shape[2], "y": r, "part": region(g, 0)[2],
                   "pstride": g[6], "poff": 0}] for v, g, r in zip(lv, geo, raw1)]
        h1, st1 = bn_stage(probs, 0, raw1, [(v[2], v[3], v[4], v[5], v[7], v[8], v[9], v[10]) for v in lv])
        # stage 2: the second convolutions read their half of h1 and write their half of one buffer; one BatchNorm pass over both
        raw2 = [empty_nhwc(g[0], g[6], g[2], g[3], dtype, dev) for g in geo]
        probs = []
        for v, g, hh, r in zip(lv, geo, h1, raw2):
            c2, c3, o = g[4], g[5], g[6]
            part = region(g, 1)[2]
            probs.append([{"x": hh[:, :c2], "wp": pack_conv_fwd(v[11], c2, dtype), "cout": c2, "k": v[11].shape[2], "y": r[:, :c2], "part": part, "pstride": o, "poff": 0},
                          {"x": hh[:, c2:], "wp": pack_conv_fwd(v[16], c3, dtype), "cout": c3, "k": v[16].shape[2], "y": r[:, c2:], "part": part, "pstride": o, "poff": c2}])
        h2, st2 = bn_stage(probs, 1, raw2, [(v[12], v[13], v[14], v[15], v[17], v[18], v[19], v[20]) for v in lv])
        # stage 3: the biased 1x1 outputs
        outs, probs = [], []
        for v, g, hh in zip(lv, geo, h2):
            n, cp, h, w, c2, c3 = g[:6]
            ob, nc = v[21].shape[0], v[23].shape[0]
            box = empty_nhwc(n, ob, h, w, dtype, dev)
            cls = empty_nhwc(n, ncpad, h, w, dtype, dev)  # (padded channels are never read: see _ConvAffineAct)
            probs += [{"x": hh[:, :c2], "wp": pack_conv_fwd(v[21], c2, dtype), "cout": ob, "k": 1, "bias": v[22], "y": box},
                      {"x": hh[:, c2:], "wp": pack_conv_fwd(v[23], c3, dtype), "cout": nc, "k": 1, "bias": v[24], "y": cls[:, :nc] if ncpad != nc else cls}]
            outs += [box, cls]
        _conv_fwd_multi(probs, dtype)
        saved = []
        for v, r1, s1, a1, r2, s2, a2 in zip(lv, raw1, st1, h1, raw2, st2, h2):
            saved += [v[0], v[1], v[6], v[2], v[3], v[7], v[8], r1, s1, a1, v[11], v[16], v[12], v[13], v[17], v[18], r2, s2, a2, v[21], v[23]]
        ctx.save_for_backward(*saved)
        ctx.bias_params = [(v[22] if v[22].requires_grad else None, v[24] if v[24].requires_grad else None) for v in lv]  # (leaf parameters: no cycle)
        ctx.meta = (nl, joins, ncpad, [g[:7] for g in geo])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gout):
        nl, joins, ncpad, geo = ctx.meta
        S = 21
        sv = [ctx.saved_tensors[l * S : (l + 1) * S] for l in range(nl)]
        dtype, dev = sv[0][0].dtype, sv[0][0].device
        lib = L()
        nig = ctx.needs_input_grad
        grads = [None] * (1 + nl * _DT)

        def need(l, i):
            return nig[1 + l * _DT + i]

        def put(l, i, g):
            grads[1 + l * _DT + i] = g

        def bn_bwd(dout, raw, ga, ba, gb, bb, c2, st, o):
            draw = empty_nhwc(*raw.shape, dtype, dev)
            dgamma = torch.empty(o, dtype=torch.float32, device=dev)
            dbeta = torch.empty(o, dtype=torch.float32, device=dev)
            ws = workspace(2048 * 2 * o * 4 + 256, dev, "bnbwd")
            check(lib.ymi_bn_act_bwd_pair(_byref(as_ymi(dout)), _byref(as_ymi(raw)), ptr(ga), ptr(ba), ptr(gb), ptr(bb), c2, ptr(st[0]), ptr(st[1]), ACT_SILU,
                                          _byref(as_ymi(draw)), ptr(dgamma), ptr(dbeta), ptr(ws), ws.numel(), stream_ptr()), "bn_act_bwd_pair")
            return draw, dgamma, dbeta

        # stage 3: data gradients of the 1x1 outputs into the two halves of dh2, weight / bias gradients per convolution
        jobs, dh2, dys = [], [], []
        for l in range(nl):
            n, cp, h, w, c2, c3, o = geo[l]
            wa2, wb2 = sv[l][19], sv[l][20]
            dbox = gout[2 * l]
            dcls = gout[2 * l + 1]
            dbox = grad_nhwc(dbox, dtype) if dbox is not None else torch.zeros((n, h, w, wa2.shape[0]), dtype=dtype, device=dev).permute(0, 3, 1, 2)
            dcls = grad_nhwc(dcls, dtype) if dcls is not None else torch.zeros((n, h, w, ncpad), dtype=dtype, device=dev).permute(0, 3, 1, 2)
            buf = empty_nhwc(n, o, h, w, dtype, dev)
            jobs.append(_dgrad_prepare(dbox, None, 1, 1, (n, c2, h, w), dtype, [], buf[:, :c2], (pack_conv_dgrad(wa2, dbox.shape[1], 1, dtype), c2)))
            jobs.append(_dgrad_prepare(dcls, None, 1, 1, (n, c3, h, w), dtype, [], buf[:, c2:], (pack_conv_dgrad(wb2, dcls.shape[1], 1, dtype), c3)))
            dh2.append(buf)
            dys.append((dbox, dcls))
        _dgrad_multi(jobs, dtype)
        for l in range(nl):
            n, cp, h, w, c2, c3, o = geo[l]
            h2 = sv[l][18]
            for which, (wt, xin, cin, dy) in enumerate(((sv[l][19], h2[:, :c2], c2, dys[l][0]), (sv[l][20], h2[:, c2:], c3, dys[l][1]))):
                iw = 21 + 2 * which
                if need(l, iw) or need(l, iw + 1):
                    dw, db = _wgrad_maybe_async(xin, dy, wt.shape[0], cin, 1, 1, True, (wt, ctx.bias_params[l][which]))
                    put(l, iw, dw.view(wt.shape) if need(l, iw) else None)
                    put(l, iw + 1, db if need(l, iw + 1) else None)
        # stage 2
        jobs, dh1, draws = [], [], []
        for l in range(nl):
            n, cp, h, w, c2, c3, o = geo[l]
            wa1, wb1, ga, ba, gb, bb, raw2, st2 = sv[l][10:18]
            draw, dgamma, dbeta = bn_bwd(dh2[l], raw2, ga, ba, gb, bb, c2, st2, o)
            put(l, 12, dgamma[:c2]); put(l, 13, dbeta[:c2]); put(l, 17, dgamma[c2:]); put(l, 18, dbeta[c2:])
            buf = empty_nhwc(n, o, h, w, dtype, dev)
            k = wa1.shape[2]
            jobs.append(_dgrad_prepare(draw[:, :c2], None, k, 1, (n, c2, h, w), dtype, [], buf[:, :c2], (pack_conv_dgrad(wa1, c2, 1, dtype), c2)))
            jobs.append(_dgrad_prepare(draw[:, c2:], None, k, 1, (n, c3, h, w), dtype, [], buf[:, c2:], (pack_conv_dgrad(wb1, c3, 1, dtype), c3)))
            dh1.append(buf)
            draws.append(draw)
        dh2 = None
        _dgrad_multi(jobs, dtype)
        for l in range(nl):
            n, cp, h, w, c2, c3, o = geo[l]
            h1 = sv[l][9]
            for iw, wt, lo, hi in ((11, sv[l][10], 0, c2), (16, sv[l][11], c2, o)):
                if need(l, iw):
                    dw, _ = _wgrad_maybe_async(h1[:, lo:hi], draws[l][:, lo:hi], hi - lo, hi - lo, wt.shape[2], 1, False, (wt,))
                    put(l, iw, dw)
        # stage 1
        jobs, draws = [], []
        for l in range(nl):
            n, cp, h, w, c2, c3, o = geo[l]
            x, wa, wb, ga, ba, gb, bb, raw1, st1 = sv[l][:9]
            draw, dgamma, dbeta = bn_bwd(dh1[l], raw1, ga, ba, gb, bb, c2, st1, o)
            put(l, 2, dgamma[:c2]); put(l, 3, dbeta[:c2]); put(l, 7, dgamma[c2:]); put(l, 8, dbeta[c2:])
            draws.append(draw)
            if need(l, 0):
                jobs.append((l,) + _dgrad_joined_prepare(joins[l], draw, None, wa.shape[2], 1, x.shape, dtype, (pack_conv_dgrad_pair(wa, wb, 1, dtype), wa.shape[1])))
        dh1 = None
        if jobs:
            dxs = _dgrad_multi([j[1] for j in jobs], dtype)
            for (l, _, deposit), dx in zip(jobs, dxs):
                put(l, 0, _dgrad_joined_finish(joins[l], dx, deposit))
        for l in range(nl):
            n, cp, h, w, c2, c3, o = geo[l]
            x, wa, wb = sv[l][:3]
            if need(l, 1) or need(l, 6):
                dw, _ = _wgrad_maybe_async(x, draws[l], o, wa.shape[1], wa.shape[2], 1, False, (wa, wb), pair_rows=c2)
                put(l, 1, dw[:c2] if need(l, 1) else None)
                put(l, 6, dw[c2:] if need(l, 6) else None)
        return tuple(grads)


def detect_train_ok(levels, dtype):
    """the lockstep form needs: at most 4 levels (8 problems a launch), 3x3 / 3x3 / 1x1 stride-1 branches with SiLU Conv blocks, branch
    widths in whole 16-byte chunks and of ONE width class (both halves of a stage ride in one launch, which fixes the row tile the
    shared statistics rows are counted in), BatchNorms with one eps / momentum."""
    if not HOOKS["detect_multi"] or not (1 <= len(levels) <= 4):
        return False
    ch = chunk_elems(dtype)
    eps = mom = None
    for a0, b0, a1, b1, oa, ob in levels:
        c2, c3 = a0.conv.out_channels, b0.conv.out_channels
        if c2 % ch or c3 % ch or _width_class(c2, dtype) != _width_class(c3, dtype) or oa.out_channels % ch:
            return False
        for m, k in ((a0, 3), (b0, 3), (a1, 3), (b1, 3)):
            cv, bn = m.conv, getattr(m, "bn", None)
            if (bn is None or not isinstance(m.act, torch.nn.SiLU) or cv.kernel_size != (k, k) or cv.stride != (1, 1) or cv.groups != 1 or cv.dilation != (1, 1)
                    or cv.bias is not None or bn.momentum is None):
                return False
            if eps is None:
                eps, mom = bn.eps, bn.momentum
            if bn.eps != eps or bn.momentum != mom:
                return False
        if a1.conv.in_channels != c2 or b1.conv.in_channels != c3 or a0.conv.in_channels != b0.conv.in_channels:
            return False
        for m, cin in ((oa, c2), (ob, c3)):
            if m.kernel_size != (1, 1) or m.stride != (1, 1) or m.in_channels != cin or m.bias is None:
                return False
    return True


def detect_train(xs, levels):
    """xs: the internal input tensor of each level; levels: per level (cv2[i][0], cv3[i][0], cv2[i][1], cv3[i][1], cv2[i][2], cv3[i][2]) ->
    ([box map], [class map]) exactly as the per-level modules would give them."""
    dtype = xs[0].dtype
    t, joins = [], []
    for x, (a0, b0, a1, b1, oa, ob) in zip(xs, levels):
        t.append(x)
        for m in (a0, b0, a1, b1):
            t += [m.conv.weight, m.bn.weight, m.bn.bias, m.bn.running_mean, m.bn.running_var]
        t += [oa.weight, oa.bias, ob.weight, ob.bias]
        joins.append(join_of(x))
    bn0 = levels[0][0].bn
    nc = levels[0][5].out_channels
    ncpad = round_up(nc, chunk_elems(dtype))
    outs = _DetectTrain.apply((len(levels), float(bn0.eps), float(bn0.momentum), joins, ncpad), *t)
    for lv in levels:
        for m in lv[:4]:
            if m.bn.num_batches_tracked is not None:
                if _deferred_counters is not None:
                    _deferred_counters.append(m.bn.num_batches_tracked)
                else:
                    m.bn.num_batches_tracked.add_(1)
    box = list(outs[0::2])
    cls = [(_ChanSlice.apply(c, nc) if ncpad != nc else c) for c in outs[1::2]]
    return box, cls


_deferred_counters = None


class deferred_bn_counters:
    """inside this context the `num_batches_tracked += 1` of every Conv is collected and applied as ONE
    multi-tensor add on exit (59 tiny launches -> 1 per forward)."""

    def __enter__(self):
        global _deferred_counters
        self.prev = _deferred_counters
        _deferred_counters = []
        return self

    def __exit__(self, *exc):
        global _deferred_counters
        pending, _deferred_counters = _deferred_counters, self.prev
        if pending:
            torch._foreach_add_(pending, 1)
        return False


# ------------------------------------------------------------------- conv / linear with affine epilogue
class _ConvAffineAct(torch.autograd.Function):
    """y = act(scale*conv(x) + bias) (+ residual), one kernel.  Used for eval-mode Conv (BN folded into
    scale/bias: conv.py:81-91 and utils/torch_utils.py:240-271), Detect's biased 1x1 outputs
    (head.py:45-59) and every nn.Linear of SwinBlock (swin_block.py:29-35) with k = 1."""

    @staticmethod
    def forward(ctx, x, weight, scale, bias, stride, act, residual, cout_pad, join=None, res_join=None):
        dtype = x.dtype
        w4 = _as4d(weight)
        o, i, k, _ = w4.shape
        dev = x.device
        _note_use(weight)
        wp = pack_conv_fwd(weight, x.shape[1], dtype)
        if x.dim() == 4:
            n, cp, h, w = x.shape
            ho, wo = _conv_out_hw(h, w, k, stride)
            # (padded channels - an output whose width is not a whole 16-byte chunk, e.g. Detect's class map at nc = 1 - are never
            # read: every consumer sees the [:, :o] view, and the GRADIENT's padded channels are zeros written by its producer)
            y = empty_nhwc(n, cout_pad, ho, wo, dtype, dev)
            yv = y[:, :o] if cout_pad != o else y
        else:
            y = torch.empty((x.shape[0], cout_pad), dtype=dtype, device=dev)
            yv = y[:, :o] if cout_pad != o else y
        check(
            L().ymi_conv2d_fwd(_byref(as_ymi(x)), ptr(wp), o, k, k, stride, ptr(scale), ptr(bias), act,
                               _byref(as_ymi(residual)) if residual is not None else None, _byref(as_ymi(yv)), None, None, stream_ptr()),
            "conv2d_fwd",
        )
        if act != ACT_NONE or scale is not None:
            ctx.unsupported = "backward through a fused activation / BN-folded conv is not implemented (use train mode or act=none)"
        else:
            ctx.unsupported = None
        ctx.save_for_backward(x, weight)
        ctx.bias_param = bias if (bias is not None and bias.requires_grad) else None  # (a leaf parameter: no cycle)
        ctx.cfg = (stride, i, bias is not None, residual is not None, cout_pad)
        ctx.joins = (join, res_join)
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.unsupported:
            raise NotImplementedError(ctx.unsupported)
        x, weight = ctx.saved_tensors
        stride, cin, has_bias, has_res, cout_pad = ctx.cfg
        dtype = x.dtype
        w4 = _as4d(weight)
        o, _, k, _ = w4.shape
        dy = grad_nhwc(dy, dtype)
        join, res_join = ctx.joins
        dres = None
        if has_res and ctx.needs_input_grad[6]:
            dres = _join_plain(res_join, dy[:, :o] if cout_pad != o else dy)
        dx = None
        if ctx.needs_input_grad[0]:
            if x.dim() == 4:
                dx = _dgrad_joined(join, dy, w4, k, stride, x.shape, dtype)
            else:
                adds = join.arrive() if join is not None else []
                ty = as_ymi(dy)
                wd = pack_conv_dgrad(weight, ty.c, 1, dtype)
                dx = torch.empty((x.shape[0], x.shape[1]), dtype=dtype, device=x.device)
                fa = _prep_adds(adds, dtype, False)
                a1 = _byref(as_ymi(fa[0])) if len(fa) > 0 else None
                a2 = _byref(as_ymi(fa[1])) if len(fa) > 1 else None
                check(L().ymi_conv2d_bwd_data_add(_byref(ty), ptr(wd), x.shape[1], 1, 1, 1, a1, a2, _byref(as_ymi(dx)), stream_ptr()), "conv2d_bwd_data")
                if len(fa) > 2:
                    _accumulate(dx, fa[2:])
                if adds is None:
                    join.deposit(dx)
                    dx = None
        dw = db = None
        need_w, need_b = ctx.needs_input_grad[1], has_bias and ctx.needs_input_grad[3]
        if need_w or need_b:
            # (the bias gradient comes out of the same launch; a frozen weight with a trainable bias gets its complete, un-deferred result)
            dw, db = _wgrad_maybe_async(x, dy, o, cin, k, stride, has_bias, (weight, ctx.bias_param))
            dw = dw.view(weight.shape) if need_w else None
            db = db if need_b else None
        return dx, dw, None, db, None, None, dres, None, None, None


def conv_affine_act(x, weight, scale=None, bias=None, stride=1, act=ACT_NONE, residual=None, pad_out=False):
    w4 = _as4d(weight)
    o = w4.shape[0]
    cout_pad = round_up(o, chunk_elems(x.dtype)) if pad_out else o
    y = _ConvAffineAct.apply(x, weight, scale, bias, int(stride), int(act), residual, cout_pad, join_of(x), join_of(residual) if residual is not None else None)
    if cout_pad != o:
        y = _ChanSlice.apply(y, o)
    return y


class _ChanSlice(torch.autograd.Function):
    """y[:, :o] of a channel-padded tensor.  Backward: the gradient of the padded tensor with zeros in the padded channels.  When
    the incoming gradient already IS the [:, :o] view of such a padded buffer (the detection loss writes its class-map gradients
    that way, pads zeroed) the buffer is handed through; otherwise zeros + copy, as autograd's own slice would do."""

    @staticmethod
    def forward(ctx, y, o):
        ctx.shape = tuple(y.shape)
        return y[:, :o]

    @staticmethod
    def backward(ctx, g):
        base = _zero_padded.pop(g.data_ptr(), None)  # (python attributes do not survive the trip through the engine: keyed on the address)
        if (base is not None and tuple(base.shape) == ctx.shape and base.dtype == g.dtype and base.stride() == g.stride()
                and tuple(g.shape) == (ctx.shape[0], g.shape[1]) + ctx.shape[2:]):
            return base, None
        full = torch.zeros(ctx.shape, dtype=g.dtype, device=g.device).contiguous(memory_format=torch.channels_last) if len(ctx.shape) == 4 \
            else torch.zeros(ctx.shape, dtype=g.dtype, device=g.device)
        full[:, : g.shape[1]].copy_(g)
        return full, None


_zero_padded = {}  # data_ptr -> padded buffer whose [:, :c] view padded_grad_like handed out (dropped when _ChanSlice.backward takes it)


def padded_grad_like(t, zero=True):
    """a gradient buffer for tensor t: if t is the [:, :c] view of a channel-padded NHWC tensor (pixel stride ld > c), a buffer of the
    PADDED shape (see _ChanSlice.backward, which hands it through); else an empty tensor like t.  -> (tensor for the kernel, [:, :c] view
    to return to autograd).  zero=False: the kernel that fills it writes zeros into the padding channels itself."""
    if t.dim() == 4 and is_nhwc(t):
        ld = as_ymi(t).ld
        n, c, h, w = t.shape
        if ld != c and ld % chunk_elems(t.dtype) == 0 and ld - c < chunk_elems(t.dtype):
            base = empty_nhwc(n, ld, h, w, t.dtype, t.device)
            if zero:
                base.zero_()
            if len(_zero_padded) > 64:  # (gradients that never reached a _ChanSlice: do not keep their buffers alive)
                _zero_padded.clear()
            _zero_padded[base.data_ptr()] = base
            return base, base[:, :c]
    e = torch.empty_like(t)
    return e, e


def linear(x, weight, bias=None, residual=None):
    """token GEMM: x [T, Cin] @ weight[Cout, Cin]^T + bias (+ residual)."""
    return _ConvAffineAct.apply(x, weight, None, bias, 1, ACT_NONE, residual, weight.shape[0], join_of(x), join_of(residual) if residual is not None else None)
