"""Weight operands and weight gradients: the per-step weight arena and its pack launches, gradient buckets as GEMM outputs (grad_arena), the
split-K weight-gradient GEMM with its deferred (batched) slab sum, side-stream and rider schedules (reference: the backward of
Conv / nn.Linear weights, nn/modules/conv.py:50-91)."""
import ctypes
import os

import torch

from .. import _lib
from .._lib import ACT_GELU, ACT_NONE, ACT_SILU, ConvProblem, DgradProblem, as_ymi, check, chunk_elems, empty_nhwc, is_nhwc, ptr, stream_ptr, workspace, ymi_dtype
from .base import (  # noqa: F401
    HOOKS, L, _as4d, _byref, _deferred_twice, _in_backward, round_up,
)

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
