"""Foundations of the op layer: library handle, test hooks, layout conversion at the model edges, forward epochs (shared-weight detection,
BatchNorm statistics accumulators), gradient joins (sums of multi-consumer tensors formed inside kernels), output slots of concat buffers."""
import ctypes
import os

import torch

from .. import _lib
from .._lib import ACT_GELU, ACT_NONE, ACT_SILU, ConvProblem, DgradProblem, as_ymi, check, chunk_elems, empty_nhwc, is_nhwc, ptr, stream_ptr, workspace, ymi_dtype

_byref = ctypes.byref


def L():
    return _lib.lib()


# Test / A-B hooks (python side; the library's own are behind _lib.set_option): each is the "before" arm of a measured change.  Production code
# never touches them; tests flip them in place, bench.py --hook name=0 sets them for a same-box A/B (tools/r5_ab.sh).
HOOKS = {
    "fused_swin_mlp": True,  # SwinBlock's second half as the fused kernels of csrc/swin_mlp.hip (False: LayerNorm + two token GEMMs)
    "detect_pair": True,     # Detect's sibling first convolutions as one (False: separately)
    "detect_multi": True,    # Detect's levels in lockstep, one multi-problem launch per stage (False: level by level)
    "first_conv": True,      # layer 0 through the direct kernels of csrc/first_conv.hip (False: the generic path)
    "stat_atomics": True,    # BatchNorm statistics as fixed-point atomic sums, finalized inside the affine pass (False: per-block rows + a finalize launch)
    "wgrad_rider": True,     # BatchNorm-backward final passes ride in the previous layer's weight-gradient launch (False: their own launches)
}


def compute_dtype(x):
    if torch.is_autocast_enabled("cuda") if hasattr(torch, "is_autocast_enabled") else False:
        dt = torch.get_autocast_dtype("cuda")
        if dt != torch.bfloat16:
            raise RuntimeError(f"libyolo_mi355 supports bfloat16 autocast only (got {dt}); fp16 has no kernels here")
        return dt
    return x.dtype if x.dtype in (torch.float32, torch.bfloat16) else torch.float32


def round_up(v, m):
    return (v + m - 1) // m * m


def _dense_ok(t, dtype):
    ch = chunk_elems(dtype)
    if t.dtype != dtype or not is_nhwc(t):
        return False
    n, c, h, w = t.shape
    ld = as_ymi(t).ld
    return c % ch == 0 and ld % ch == 0 and t.data_ptr() % 16 == 0


class _ToInternal(torch.autograd.Function):
    """NCHW float32 (the reference's API format) -> NHWC compute dtype, channels zero-padded to a
    16-byte multiple.  API edge of the model (first Conv input)."""

    @staticmethod
    def forward(ctx, x, dtype):
        n, c, h, w = x.shape
        cp = round_up(c, chunk_elems(dtype))
        src = x.detach()
        if src.dtype != torch.float32 or not src.is_contiguous():
            src = src.float().contiguous()
        out = empty_nhwc(n, cp, h, w, dtype, x.device)
        check(L().ymi_nchw_to_nhwc(ptr(src), n, c, h, w, _byref(as_ymi(out)), stream_ptr()), "nchw_to_nhwc")
        ctx.c = c
        ctx.in_dtype = x.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        g = grad_nhwc(g, g.dtype if g.dtype in (torch.float32, torch.bfloat16) else torch.float32)
        n, cp, h, w = g.shape
        full = torch.empty((n, cp, h, w), dtype=torch.float32, device=g.device)
        check(L().ymi_nhwc_to_nchw(_byref(as_ymi(g)), ptr(full), stream_ptr()), "nhwc_to_nchw")
        return full[:, : ctx.c].to(ctx.in_dtype), None


def to_internal(x, dtype=None):
    dtype = dtype or compute_dtype(x)
    if not x.is_cuda:
        raise RuntimeError("improving_yolov8_cbam_swinblock_amd runs on the MI355X only: move the input to 'cuda' (no CPU path)")
    if _dense_ok(x, dtype):
        return x
    return _ToInternal.apply(x, dtype)


def grad_nhwc(g, dtype):
    """normalise an incoming gradient to dense NHWC memory of `dtype` (torch-side plumbing)."""
    if g.dtype != dtype:
        g = g.to(dtype)
    if g.dim() == 4:
        if not _dense_ok(g, dtype):
            n, c, h, w = g.shape
            buf = empty_nhwc(n, c, h, w, dtype, g.device)
            buf.copy_(g)
            g = buf
    elif g.dim() == 2:
        if g.stride(1) != 1 or g.stride(0) % 4 != 0 or g.data_ptr() % 16 != 0:
            g = g.contiguous()
    return g


def to_nchw_float(x):
    """NHWC compute-dtype tensor -> contiguous NCHW float32 (for callers that need the reference format)."""
    n, c, h, w = x.shape
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
    check(L().ymi_nhwc_to_nchw(_byref(as_ymi(x.detach())), ptr(out), stream_ptr()), "nhwc_to_nchw")
    return out


def _as4d(w):
    return w if w.dim() == 4 else w.view(w.shape[0], w.shape[1], 1, 1)


def _conv_out_hw(h, w, k, s):
    p = k // 2
    return (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1


# A weight used more than once in one forward (shared weights): autograd's input buffer sums the gradients of the uses BEFORE AccumulateGrad
# sees them, i.e. it READS them during the pass - none of them may be deferred (round-3 ADVICE).  Which weights are shared is only
# known once the forward is over, so every weight-consuming Function counts its uses per forward "epoch" (a model forward starts a new
# one: BaseModel._predict_once); modules called on their own never start an epoch, so a second call already counts as sharing - the
# safe side: their gradients are complete when the Function returns.
_use_epoch = [0]

# BatchNorm statistics accumulators (ymi_conv2d_bn_silu_fwd_acc): every Conv block of a forward takes a [4][2][cout] int64 block that must be
# ZERO when its GEMM starts.  A model forward starts an epoch: ONE fill zeroes the whole arena (sized by the previous epoch's demand) and the
# blocks are handed out in call order; a block asked for outside an epoch - modules called on their own - or beyond the arena is a fresh
# zeroed tensor.
_stat_arena = {"buf": None, "cursor": 0, "need": 0}


def new_forward_epoch(device=None):
    _use_epoch[0] += 1
    a = _stat_arena
    if device is not None and HOOKS["stat_atomics"]:
        if a["buf"] is None or a["buf"].device != device or a["need"] > a["buf"].numel():
            a["buf"] = torch.zeros(max(a["need"] * 2, 1 << 16), dtype=torch.int64, device=device)
        else:
            a["buf"].zero_()
    else:
        a["buf"] = None
    a["cursor"], a["need"] = 0, 0


def _stat_acc(cout, device):
    a = _stat_arena
    n = 8 * cout
    a["need"] += n
    buf = a["buf"]
    if buf is not None and buf.device == device and a["cursor"] + n <= buf.numel():
        out = buf[a["cursor"]: a["cursor"] + n]
        a["cursor"] += n
        return out
    return torch.zeros(n, dtype=torch.int64, device=device)


def _note_use(*params):
    for w in params:
        if w is None:
            continue
        st = getattr(w, "_ymi_use", None)
        if st is None or st[0] != _use_epoch[0]:
            w._ymi_use = [_use_epoch[0], 1]
        else:
            st[1] += 1


def _deferred_twice(params):
    """True when one of these parameters was used more than once in the forward this backward belongs to."""
    for p in params:
        st = getattr(p, "_ymi_use", None) if p is not None else None
        if st is not None and st[1] > 1:
            return True
    return False


def _in_backward():
    """True inside an autograd backward pass (where the engine accepts end-of-pass callbacks)."""
    try:
        return torch._C._current_graph_task_id() != -1
    except AttributeError:  # very old torch: no way to tell, never defer
        return False


class GradJoin:
    """Gradient sum of a tensor with several consumers, formed inside kernels instead of by autograd `add`s.

    The module that creates the tensor (and knows every consumer is one of this package's ops) hands the same GradJoin to
    all `n` consumers.  In backward every consumer but the last to run DEPOSITS its gradient contribution here and returns
    None to autograd; the last one fetches the deposits and adds them in the epilogue of its own kernel (data-gradient
    GEMM, LayerNorm backward, upsample adjoint) - or, if it has no such kernel, with explicit accumulate launches.
    Which consumer is last is decided at run time, so the result does not depend on autograd's node order.
    Reference sites: Bottleneck shortcut (block.py:488), the two Detect branches (head.py:72), neck skip connections
    (yolov8.yaml:760-773), SwinBlock residuals (swin_block.py:52-53), C2f chunk / concat (block.py:302-304)."""

    __slots__ = ("n", "seen", "pending", "out", "dst")

    def __init__(self, n):
        self.n, self.seen, self.pending = int(n), 0, []
        # optional: where the total should be WRITTEN (a _GradSlot: a channel slice of a wider gradient buffer; see _ChanSplit2)
        self.dst = None
        # optional: a buffer that already holds one more contribution (set during backward by the producer of that contribution);
        # the last-arriving data gradient then adds it as an addend AND writes the total there (C2f's chunk: see _C2fSplit)
        self.out = None

    def arrive(self):
        """-> the deposits if the caller is the last consumer (it must return the total), else None (it must deposit)."""
        self.seen += 1
        if self.seen < self.n:
            return None
        out, self.pending, self.seen = self.pending, [], 0
        return out

    def deposit(self, g):
        self.pending.append(g)


def mark_join(t, consumers, force=False):
    """attach a GradJoin for `consumers` join-aware consumers to tensor t (training, grad enabled, > 1 consumer - or `force`:
    a single consumer whose data gradient should pick up GradJoin.out)."""
    if (consumers > 1 or force) and torch.is_grad_enabled() and t.requires_grad:
        t._ymi_join = GradJoin(consumers)
    return t


def join_of(t):
    return getattr(t, "_ymi_join", None) if torch.is_grad_enabled() else None


def _accumulate(total, adds):
    """total += each addend (explicit launches: the fall-back of consumers without a fusing kernel); total is private."""
    for a in adds:
        a = grad_nhwc(a, total.dtype) if a.dim() == 4 else a.to(total.dtype)
        check(L().ymi_add_inplace(_byref(as_ymi(a)), _byref(as_ymi(total)), stream_ptr()), "add_inplace")
    return total


def _join_plain(join, g):
    """consumer without a fusing kernel (residual hand-through, concat slice): deposit, or return the total if last."""
    if join is None:
        return g
    adds = join.arrive()
    if adds is None:
        join.deposit(g)
        return None
    if not adds:
        return g
    total = empty_nhwc(*g.shape, g.dtype, g.device) if g.dim() == 4 else torch.empty_like(g)
    total.copy_(g)
    return _accumulate(total, adds)


def _prep_adds(adds, dtype, like4d):
    """addends as tensors the epilogue can read: same dtype, NHWC memory (4-D) / unit channel stride (2-D)."""
    out = []
    for a in adds or ():
        if a.dtype != dtype:
            a = a.to(dtype)
        a = grad_nhwc(a, dtype)
        out.append(a)
    return out


# ------------------------------------------------------------------- Conv + BN(train) + act
class OutSlot:
    """where a producer writes its output: channels [off, off + c) of a pre-allocated NHWC concat buffer, so the
    concat itself (block.py:304 `torch.cat`) needs no copy.  A plain Python object: the buffer is storage only and
    never takes part in autograd; the producer's output is a view created inside its Function.forward."""

    def __init__(self, buf, off, lazy=None):
        self.buf, self.off, self.lazy = buf, int(off), lazy

    def view(self, n, c, h, w, dtype):
        if self.buf is None and self.lazy is not None:  # concat buffer of the model graph: created by its first producer
            self.buf = self.lazy.get(n, h, w, dtype)
        b = self.buf
        if b.dtype != dtype or b.shape[0] != n or b.shape[2] != h or b.shape[3] != w or self.off + c > b.shape[1]:
            raise RuntimeError(f"output slot [{self.off}:{self.off + c}] of {tuple(b.shape)} {b.dtype} does not fit a {(n, c, h, w)} {dtype} result")
        return b[:, self.off : self.off + c]


class LazyConcatBuffer:
    """the NHWC buffer of one Concat layer of the model graph: every producer writes its slice (OutSlot), the Concat
    itself copies nothing (nn/modules/conv.py:683 `torch.cat` of the reference).  Allocated when the first producer runs."""

    def __init__(self, channels, device):
        self.channels, self.device, self.buf = int(channels), device, None

    def get(self, n, h, w, dtype):
        if self.buf is None:
            self.buf = empty_nhwc(n, self.channels, h, w, dtype, self.device)
        return self.buf


class _GradBuffer:
    """the gradient buffer of a tensor whose channel slices are consumed separately (_ChanSplit2): allocated when the first consumer's
    data gradient needs its slice, so that every slice's gradient is WRITTEN where the whole tensor's gradient will be read."""

    def __init__(self, shape, device):
        self.shape, self.device, self.buf = tuple(shape), device, None

    def get(self, dtype):
        if self.buf is None:
            n, c, h, w = self.shape
            self.buf = empty_nhwc(n, c, h, w, dtype, self.device)
        return self.buf if self.buf.dtype == dtype else None


class _GradSlot:
    __slots__ = ("gb", "lo", "hi")

    def __init__(self, gb, lo, hi):
        self.gb, self.lo, self.hi = gb, lo, hi

    def view(self, dtype):
        b = self.gb.get(dtype)
        return None if b is None else b[:, self.lo : self.hi]
