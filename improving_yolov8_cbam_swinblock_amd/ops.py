"""torch.autograd.Function wrappers over the C ABI of libyolo_mi355.so.

PyTorch is plumbing here: it owns device memory, streams and the autograd tape; every tensor
operation on the hot path is a hand-written gfx950 kernel reached through `_lib` (ctypes).  There is
no fallback: without the library, or with CPU tensors, these functions raise.

Tensor convention between ops: logical [N, C, H, W] tensors whose MEMORY is NHWC (channels_last),
possibly a channel slice of a wider buffer (pixel stride ld > C); token matrices are plain [T, C].
Compute dtype is bfloat16 under `torch.autocast("cuda", dtype=torch.bfloat16)` and the input's
dtype (float32 = parity mode) otherwise; parameters stay float32 and are packed per call.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import ACT_GELU, ACT_NONE, ACT_SILU, ConvProblem, DgradProblem, as_ymi, check, chunk_elems, empty_nhwc, is_nhwc, ptr, stream_ptr, workspace, ymi_dtype

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


# ------------------------------------------------------------------------------------ weights
class _PackDesc(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("dst_fwd", ctypes.c_void_p), ("dst_dgrad", ctypes.c_void_p), ("o", ctypes.c_int32),
                ("i", ctypes.c_int32), ("kh", ctypes.c_int32), ("kw", ctypes.c_int32), ("ipad", ctypes.c_int32), ("opad", ctypes.c_int32),
                ("stride", ctypes.c_int32), ("ostride", ctypes.c_int32), ("o_off", ctypes.c_int32), ("_pad", ctypes.c_int32)]


class WeightArena:
    """All conv / linear weights of a model packed into kernel operand layouts by ONE launch per step.

    Life cycle: while `recording`, the per-call packers below note (weight, padding, stride) of every use during one
    full forward+backward; `build()` then allocates one arena and a device descriptor table; afterwards `pack()`
    (called at the start of each training forward) refreshes every operand with a single kernel and the per-call
    packers return views of the arena.  A use that was not recorded simply falls back to its own pack launch.

    A spec holds ONE weight, or a PAIR of weights of the same input that run as one convolution (Detect's sibling branches,
    reference head.py:71-72): the pair's forward operands lie back to back ([oA + oB][tap][ipad]) and its data-gradient operand has
    the two column ranges side by side ([i][tap][oA + oB], ymi_pack_desc.ostride / o_off)."""

    def __init__(self):
        self.specs = {}      # id(weight) | (id(wA), id(wB)) -> dict(weights, o (tuple), i, k, ipad, opad, stride)
        self.dtype = None
        self.built = False
        self.fresh = False   # operands correspond to the current weight values
        self.views = {}

    def note(self, weight, dtype, ipad=None, opad=None, stride=1, pair=None):
        """pair: the second weight when (weight, pair) run as one convolution (opad is then the two real widths' sum)."""
        if self.built:
            return
        ws = (weight,) if pair is None else (weight, pair)
        w4 = [_as4d(w) for w in ws]
        i, k = w4[0].shape[1], w4[0].shape[2]
        key = id(weight) if pair is None else (id(weight), id(pair))
        sp = self.specs.setdefault(key, dict(weights=ws, o=tuple(w.shape[0] for w in w4), i=i, k=k, ipad=None, opad=None, stride=1))
        if ipad is not None:
            sp["ipad"] = ipad
        if opad is not None:
            sp["opad"], sp["stride"] = opad, stride
        self.dtype = dtype

    def build(self):
        if not self.specs:
            return
        dev = next(iter(self.specs.values()))["weights"][0].device
        es = 2 if self.dtype == torch.bfloat16 else 4
        total, plan = 0, []
        for key, sp in self.specs.items():
            if sp["k"] > 3:
                raise RuntimeError("weight arena: kernels larger than 3x3 are not packed in one launch (the pack kernel's LDS tile holds nine taps)")
            osum = sum(sp["o"])
            nf = osum * sp["k"] ** 2 * sp["ipad"] if sp["ipad"] else 0
            nd = sp["i"] * sp["k"] ** 2 * sp["opad"] if sp["opad"] else 0
            offf, total = total, total + round_up(nf, 8)
            offd, total = total, total + round_up(nd, 8)
            plan.append((key, sp, nf, nd, offf, offd))
        self.arena = torch.empty(total, dtype=self.dtype, device=dev)
        ndesc = sum(len(sp["weights"]) for _, sp, *_ in plan)
        descs = (_PackDesc * ndesc)()
        starts = [0]
        base = self.arena.data_ptr()
        self.params = []
        n = 0
        for key, sp, nf, nd, offf, offd in plan:
            pair = len(sp["weights"]) == 2
            k2 = sp["k"] ** 2
            o_off = 0
            for w, o in zip(sp["weights"], sp["o"]):
                # (a pair: forward rows of the second weight follow the first's; in the data-gradient operand each weight owns the
                # columns [o_off, o_off + o) of rows that are opad = oA + oB long - the widths are whole 16-byte chunks, no padding between)
                opad_w = (o if pair else sp["opad"]) if nd else 0
                descs[n] = _PackDesc(w.data_ptr(), base + (offf + o_off * k2 * (sp["ipad"] or 0)) * es if nf else None, base + offd * es if nd else None,
                                     o, sp["i"], sp["k"], sp["k"], sp["ipad"] or 0, opad_w, sp["stride"], (sp["opad"] if (pair and nd) else 0), o_off if pair else 0, 0)
                # workgroups: one per 32 x 32 tile of (output, input) channels, padded extents included (include/ymi.h)
                wgs = ((max(o, opad_w) + 31) // 32) * ((max(sp["i"], sp["ipad"] if nf else 0) + 31) // 32)
                starts.append(starts[-1] + wgs)
                self.params.append(w)
                o_off += o
                n += 1
            self.views[key] = (self.arena[offf : offf + nf] if nf else None, self.arena[offd : offd + nd] if nd else None, sp["ipad"], sp["opad"], sp["stride"])
        raw = bytes(descs)
        self.descs = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        self.starts = torch.tensor(starts, dtype=torch.int32).to(dev)
        self.count, self.blocks = ndesc, starts[-1]
        self.ptrs = [w.data_ptr() for w in self.params]  # the descriptor table holds these raw addresses
        self.built = True

    def stale(self):
        """a parameter's storage moved since build() (model.to() / .float() / any re-allocating Module._apply swaps
        param.data under the same Parameter object): the descriptor table then points at freed memory."""
        return any(w.data_ptr() != p or w.device != self.arena.device for w, p in zip(self.params, self.ptrs))

    def pack(self):
        check(L().ymi_pack_conv_weights_batch(ptr(self.descs), ptr(self.starts), self.count, self.blocks, ymi_dtype(self.dtype), stream_ptr()), "pack_conv_weights_batch")
        # an operand is valid for exactly the weight VALUES it was packed from: Tensor._version counts in-place
        # updates (optimizer steps, load_state_dict), so a module called on its own after an update never sees the
        # operands of the previous step
        self.versions = {id(w): w._version for w in self.params}
        self.fresh = True

    def _view(self, weight, dtype, pair=None):
        if not (self.built and self.fresh and dtype == self.dtype):
            return None
        key = id(weight) if pair is None else (id(weight), id(pair))
        v = self.views.get(key)
        if v is None or any(self.versions.get(id(w)) != w._version for w in ((weight,) if pair is None else (weight, pair))):
            return None
        return v

    def lookup_fwd(self, weight, ipad, dtype, pair=None):
        v = self._view(weight, dtype, pair)
        return v[0] if v is not None and v[0] is not None and v[2] == ipad else None

    def lookup_dgrad(self, weight, opad, stride, dtype, pair=None):
        v = self._view(weight, dtype, pair)
        return v[1] if v is not None and v[1] is not None and v[3] == opad and v[4] == stride else None


_arena = None  # the WeightArena of the model whose forward/backward is running (set by DetectionModel)


def set_weight_arena(arena):
    global _arena
    _arena = arena


def pack_conv_fwd(weight, cin_pad, dtype):
    if _arena is not None:
        hit = _arena.lookup_fwd(weight, cin_pad, dtype)
        if hit is not None:
            return hit
        _arena.note(weight, dtype, ipad=cin_pad)
    weight = _as4d(weight)
    o, i, kh, kw = weight.shape
    buf = torch.empty(o * kh * kw * cin_pad, dtype=dtype, device=weight.device)
    check(L().ymi_pack_conv_weight_fwd(ptr(weight.detach()), o, i, kh, kw, cin_pad, ymi_dtype(dtype), ptr(buf), stream_ptr()), "pack_conv_weight_fwd")
    return buf


def pack_conv_dgrad(weight, cout_pad, stride, dtype):
    if _arena is not None:
        hit = _arena.lookup_dgrad(weight, cout_pad, stride, dtype)
        if hit is not None:
            return hit
        _arena.note(weight, dtype, opad=cout_pad, stride=stride)
    weight = _as4d(weight)
    o, i, kh, kw = weight.shape
    buf = torch.empty(cout_pad * i * kh * kw, dtype=dtype, device=weight.device)
    check(L().ymi_pack_conv_weight_dgrad_ex(ptr(weight.detach()), o, cout_pad, i, kh, kw, stride, ymi_dtype(dtype), ptr(buf), stream_ptr()), "pack_conv_weight_dgrad")
    return buf


def pack_conv_fwd_pair(wa, wb, cin_pad, dtype):
    """forward operand of two convolutions of one input run as ONE: [oA + oB][kh][kw][cin_pad] (a view of the arena once it is built)."""
    if _arena is not None:
        hit = _arena.lookup_fwd(wa, cin_pad, dtype, pair=wb)
        if hit is not None:
            return hit
        _arena.note(wa, dtype, ipad=cin_pad, pair=wb)
    w = torch.cat([wa.detach(), wb.detach()], 0)  # (only until the arena exists - the warm-up steps - and in stand-alone use: never inside a captured graph)
    o, i, kh, kw = w.shape
    buf = torch.empty(o * kh * kw * cin_pad, dtype=dtype, device=w.device)
    check(L().ymi_pack_conv_weight_fwd(ptr(w), o, i, kh, kw, cin_pad, ymi_dtype(dtype), ptr(buf), stream_ptr()), "pack_conv_weight_fwd")
    return buf


def pack_conv_dgrad_pair(wa, wb, stride, dtype):
    """data-gradient operand of the pair: [i][tap][oA + oB] per stride-parity class."""
    otot = wa.shape[0] + wb.shape[0]
    if _arena is not None:
        hit = _arena.lookup_dgrad(wa, otot, stride, dtype, pair=wb)
        if hit is not None:
            return hit
        _arena.note(wa, dtype, opad=otot, stride=stride, pair=wb)
    w = torch.cat([wa.detach(), wb.detach()], 0)
    o, i, kh, kw = w.shape
    buf = torch.empty(o * i * kh * kw, dtype=dtype, device=w.device)
    check(L().ymi_pack_conv_weight_dgrad_ex(ptr(w), o, o, i, kh, kw, stride, ymi_dtype(dtype), ptr(buf), stream_ptr()), "pack_conv_weight_dgrad")
    return buf


def _as4d(w):
    return w if w.dim() == 4 else w.view(w.shape[0], w.shape[1], 1, 1)


def _conv_out_hw(h, w, k, s):
    p = k // 2
    return (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1


# Where weight gradients should be WRITTEN: {id(parameter): float32 tensor of the parameter's shape}.  engine.trainer's several-rank
# schedule registers the slices of its flat all-reduce buckets here, so that a weight gradient is born inside its bucket and no pack
# copy of the 53.6 MB of gradients is needed before the exchange (None: fresh tensors).
_grad_arena = None


class grad_arena:
    def __init__(self, views):
        self.views = views

    def __enter__(self):
        global _grad_arena
        self.prev, _grad_arena = _grad_arena, self.views
        return self

    def __exit__(self, *exc):
        global _grad_arena
        _grad_arena = self.prev
        return False


def _new_dw(cout, cin, k, dev, params, pair_rows):
    if _grad_arena is not None and not pair_rows and params and params[0] is not None:
        v = _grad_arena.get(id(params[0]))
        if v is not None and v.numel() == cout * cin * k * k and v.dtype == torch.float32 and v.is_contiguous() and v.device == dev:
            return v.view(cout, cin, k, k)
    return torch.empty((cout, cin, k, k), dtype=torch.float32, device=dev)


def _wgrad(x, dy, cout, cin, k, stride, want_bias, params=(), pair_rows=None):
    """-> (dw [cout, cin, k, k] f32, dbias [cout] f32 | None).  params: the parameters these gradients belong to.
    pair_rows: params are TWO weights whose gradients are the row ranges [0, pair_rows) and [pair_rows, cout) of dw (_ConvBnActPair)."""
    if _deferred["on"] and _in_backward() and _adoptable(params) and not _deferred_twice(params):
        owner = (params[0], params[1], int(pair_rows)) if pair_rows else (params[0] if params else None)
        bias_owner = params[1] if (want_bias and not pair_rows and len(params) > 1) else None
        return _wgrad_deferred(x, dy, cout, cin, k, stride, want_bias, owner, _new_dw(cout, cin, k, x.device, params, pair_rows), bias_owner)
    dev = x.device
    dw = _new_dw(cout, cin, k, dev, params, pair_rows)
    ty, tx = as_ymi(dy), as_ymi(x)
    db = torch.empty(ty.c, dtype=torch.float32, device=dev) if want_bias else None  # (column sums of every channel of dy, padded ones included)
    need = L().ymi_conv2d_bwd_weight_workspace(ty.n * ty.h * ty.w, ty.c, tx.c, k, k)
    ws = workspace(need, dev, "wgrad")
    check(L().ymi_conv2d_bwd_weight(_byref(tx), _byref(ty), cout, cin, k, k, stride, ptr(dw), ptr(db), ptr(ws), ws.numel(), stream_ptr()), "conv2d_bwd_weight")
    return dw, (db[:cout] if want_bias else None)


def _adoptable(params):
    """True when AccumulateGrad will adopt freshly returned gradient tensors of these parameters as `.grad` WITHOUT reading
    them during the pass: no gradient accumulated yet, and no hooks that run when the gradient arrives."""
    for p in params:
        if p is None:
            continue
        if not p.requires_grad:
            # autograd drops the returned tensor at once: a deferred slab sum would later write into memory the allocator has
            # already handed to another tensor of the pass (round-3 ADVICE)
            return False
        if p.grad is not None or getattr(p, "_backward_hooks", None) or getattr(p, "_post_accumulate_grad_hooks", None):
            return False
    return True


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


# ---- weight gradients on a second stream ------------------------------------------------------------------------
# Inside `async_wgrad()` (the training step's backward) every weight-gradient GEMM (+ its slab reduce) is enqueued on a
# side stream that forks from the current one: it depends only on (x, dz), and nothing downstream needs dW before the
# optimizer, while the data-gradient / BatchNorm chain of the next layers continues on the current stream.  The small
# 20x20 / 40x40 layers leave CUs idle in their tails; two independent kernel chains fill them.  Captured in a HIP graph
# this becomes a parallel branch.  The operands are kept alive until the join so the allocator cannot hand their memory
# to the main stream while the side stream still reads them.
_side_streams = {}
_async = {"on": False, "pending": False, "keep": []}


def _side_stream(dev):
    s = _side_streams.get(dev.index)
    if s is None:
        s = _side_streams[dev.index] = torch.cuda.Stream(device=dev)
    return s


class async_wgrad:
    def __init__(self, enabled=True):
        self.enabled = enabled

    def __enter__(self):
        self.prev = _async["on"]
        _async["on"] = bool(self.enabled)
        return self

    def __exit__(self, *exc):
        join_side_stream()
        _async["on"] = self.prev
        return False


def join_side_stream():
    """make the current stream wait for the weight-gradient stream (no-op when nothing is pending)."""
    if _async["pending"]:
        cur = torch.cuda.current_stream()
        cur.wait_stream(_side_stream(cur.device))
        _async["pending"] = False
        _async["keep"].clear()


def _wgrad_maybe_async(x, dy, cout, cin, k, stride, want_bias, params=(), pair_rows=None):
    if not _async["on"]:
        return _wgrad(x, dy, cout, cin, k, stride, want_bias, params, pair_rows)
    cur = torch.cuda.current_stream()
    side = _side_stream(x.device)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        out = _wgrad(x, dy, cout, cin, k, stride, want_bias, params, pair_rows)
    _async["keep"].append((x, dy))
    _async["pending"] = True
    return out


# ---- slab sums of ALL weight gradients of a backward pass in one launch ---------------------------------------------
# Inside `deferred_wgrad()` every weight-gradient GEMM of a backward pass leaves its split-K slabs un-summed and registers a
# record; a callback the autograd engine runs when the pass ends (before backward() returns) sums them all with ONE launch
# (ymi_wgrad_reduce_batch).  Until then the returned dW tensors hold no data.  That is only safe when NOTHING reads a weight
# gradient before the pass is over:
#   * AccumulateGrad must adopt the tensor as `.grad` (p.grad is None when the pass starts).  With gradient accumulation
#     - a second backward() before zero_grad, as the reference trainer does for nbs / batch > 1 (trainer.py:305,397) -
#     AccumulateGrad runs `p.grad += dw` DURING the pass and would read the unfilled tensor;
#   * no post-accumulate-grad hooks (the overlapped DDP schedule of engine/ddp.py) and no tensor hooks on parameters.
# So the deferral is OPT-IN: engine.trainer.TrainStep, which zeroes gradients with set_to_none=True after every step and
# knows its DDP schedule, enables it around its backward.  Everywhere else (plain autograd use of the modules, gradient
# accumulation, hooks) each weight gradient is complete when its Function returns.
_deferred = {"on": False, "records": [], "keep": [], "owners": [], "bias": [], "task": None, "table": None}


class deferred_wgrad:
    """context manager: batch the split-K slab sums of every weight gradient of the backward passes run inside it.  The
    caller guarantees the conditions above; parameters that already hold a gradient are detected by _wgrad and not deferred."""

    def __init__(self, enabled=True):
        self.enabled = bool(enabled)

    def __enter__(self):
        self.prev = _deferred["on"]
        _deferred["on"] = self.enabled
        return self

    def __exit__(self, *exc):
        _deferred["on"] = self.prev
        return False


class wgrad_riders:
    """context manager around a backward pass with deferred weight gradients on ONE stream (engine.trainer.TrainStep's captured steps): the
    library holds each deferred weight-gradient launch back until the next BatchNorm backward, whose final pass then rides in it
    (include/ymi.h: ymi_wgrad_hold).  Leaving the context issues a launch still held."""

    def __init__(self, enabled=True):
        self.enabled = bool(enabled) and HOOKS["wgrad_rider"]

    def __enter__(self):
        if self.enabled:
            check(L().ymi_wgrad_hold(1), "wgrad_hold")
        return self

    def __exit__(self, *exc):
        if self.enabled:
            check(L().ymi_wgrad_hold(2 if exc[0] is not None else 0), "wgrad_hold")
            if exc[0] is not None:
                L().ymi_wgrad_hold(0)
        return False


def set_wgrad_deferred(flag):
    """process-wide switch (tests / tools); prefer the `deferred_wgrad` context manager."""
    _deferred["on"] = bool(flag)


def _flush_wgrads():
    recs, keep, owners, biases = _deferred["records"], _deferred["keep"], _deferred["owners"], _deferred["bias"]
    _deferred["records"], _deferred["keep"], _deferred["owners"], _deferred["bias"], _deferred["task"] = [], [], [], [], None
    if not recs:
        return
    # Every node of the pass has run: a parameter's AccumulateGrad has either ADOPTED the returned tensor (p.grad is that
    # memory - the usual case, and the reason the tensor must not be referenced from here: a second reference makes
    # AccumulateGrad clone it) or, if the gradient layout contract failed, stored a clone of the still unfilled tensor and
    # dropped the original.  In the second case the sum is written straight into p.grad instead of into freed memory.
    for rec, owner in zip(recs, owners):
        if isinstance(owner, tuple):
            # two parameters share one [oA + oB, ...] result as its two row ranges (_ConvBnActPair): both must have ADOPTED their view -
            # a cloned half cannot be redirected (the sum is one write of the whole tensor), so that case is refused loudly
            pa, pb, rows = owner
            per_row = rec.cin_real * rec.ntaps * 4
            if pa.grad is None and pb.grad is None:
                continue  # (torch.autograd.grad: the two views are handed to the caller as they are, nothing was accumulated)
            ok = (pa.grad is not None and pb.grad is not None and pa.grad.data_ptr() == rec.dw and pb.grad.data_ptr() == rec.dw + rows * per_row)
            if not ok:
                raise RuntimeError("deferred weight gradient of a convolution pair: AccumulateGrad did not adopt both halves of the result")
            continue
        g = owner.grad if owner is not None else None
        if g is not None and g.data_ptr() != rec.dw and g.dtype == torch.float32 and g.is_contiguous():
            rec.dw = g.data_ptr()
    dev = keep[0][0].device
    n = len(recs)
    tab = _deferred["table"]
    if tab is None or tab.device != dev or tab.numel() < n * ctypes.sizeof(_lib.WgradPending):
        tab = _deferred["table"] = torch.empty(max(n, 128) * ctypes.sizeof(_lib.WgradPending), dtype=torch.uint8, device=dev)
    arr = (_lib.WgradPending * n)(*recs)
    def late_bias():
        # bias gradients are summed by the batched launch too (into the buffer their Function returned a view of).  Usually AccumulateGrad
        # adopted that view; where it stored a clone instead, the clone gets the finished sum here
        for b in biases:
            if b is not None and b[0] is not None:
                g = b[0].grad
                if g is not None and g.data_ptr() != b[1].data_ptr():
                    g.copy_(b[1][: b[2]])

    if _async["on"]:  # the GEMMs ran on the side stream: the sum follows them there (joined by async_wgrad's exit)
        # (slabs produced on the CURRENT stream - the first layer's fused backward - must be complete too)
        _side_stream(dev).wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(_side_stream(dev)):
            check(L().ymi_wgrad_reduce_batch(arr, n, ptr(tab), stream_ptr()), "wgrad_reduce_batch")
            late_bias()
        _async["pending"] = True
    else:
        check(L().ymi_wgrad_reduce_batch(arr, n, ptr(tab), stream_ptr()), "wgrad_reduce_batch")
        late_bias()
    del keep


def _wgrad_deferred(x, dy, cout, cin, k, stride, want_bias, owner=None, dw=None, bias_owner=None):
    """as _wgrad, with the slab sum left to the end of the backward pass.  Slabs and operands stay alive in _deferred['keep']
    until the flush has been enqueued; the gradient tensor itself is owned by autograd (see _flush_wgrads).  owner: the weight."""
    task = torch._C._current_graph_task_id()
    if _deferred["task"] != task:
        # first deferred gradient of this pass.  Records of an earlier pass whose end-of-pass callback never ran (the engine
        # drops callbacks when a backward raises) are stale: their gradient tensors are gone - discard them.
        if _deferred["records"]:
            L().ymi_wgrad_hold(2)  # (a launch the library still holds back for a rider belongs to that pass too: its operands are gone)
        _deferred["records"], _deferred["keep"], _deferred["owners"], _deferred["bias"] = [], [], [], []
        torch.autograd.Variable._execution_engine.queue_callback(_flush_wgrads)
        _deferred["task"] = task
    dev = x.device
    if dw is None:
        dw = torch.empty((cout, cin, k, k), dtype=torch.float32, device=dev)
    ty, tx = as_ymi(dy), as_ymi(x)
    db = torch.empty(ty.c, dtype=torch.float32, device=dev) if want_bias else None
    need = L().ymi_conv2d_bwd_weight_workspace(ty.n * ty.h * ty.w, ty.c, tx.c, k, k)
    ws = torch.empty(int(need), dtype=torch.uint8, device=dev)
    rec = _lib.WgradPending()
    check(L().ymi_conv2d_bwd_weight_deferred(_byref(tx), _byref(ty), cout, cin, k, k, stride, ptr(dw), ptr(db), ptr(ws), ws.numel(), _byref(rec), stream_ptr()),
          "conv2d_bwd_weight")
    _deferred["records"].append(rec)
    # (db, the BASE of the returned bias-gradient view, stays referenced until the flush: the batched sum also writes the bias gradient
    # - its per-split partials come out of the GEMM - so that memory must not return to the allocator first; holding the base does not
    # keep AccumulateGrad from adopting the view)
    _deferred["keep"].append((ws, x, dy, db))
    _deferred["owners"].append(owner)
    _deferred["bias"].append((bias_owner, db, cout) if want_bias else None)
    return dw, (db[:cout] if want_bias else None)


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
