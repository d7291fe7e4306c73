"""Training step of the hot path (reference: ultralytics/engine/trainer.py:383-399,614-622,788-849 and
models/yolo/detect/train.py:90-115), reduced to what the benchmark step needs: bf16 autocast forward,
v8 detection loss, backward (+ RCCL gradient mean), gradient clip 10.0, SGD-nesterov step, EMA update."""
import os

import torch
import torch.nn as nn

from .. import ops
from .ddp import GradientBuckets
from .optim import FusedAdamax, FusedAdamW, FusedNAdam, FusedRAdam, FusedRMSprop, FusedSGD, ModelEMA

# weight-gradient GEMMs on a second stream during backward (ops.async_wgrad) in EAGER steps; YMI_WGRAD_STREAM=0 keeps
# one stream.  Graph-replayed steps stay single-stream: measured no wall-time gain there, and concurrent kernels stretch
# each other's durations, which would blur the per-kernel roofline measurement.
ASYNC_WGRAD = os.environ.get("YMI_WGRAD_STREAM", "1") != "0"


def build_optimizer(model, name="SGD", lr=0.01, momentum=0.937, decay=5e-4, ema=None, iterations=1e5, nc=None):
    """reference build_optimizer (trainer.py:788-849) + optimizer_step's clip (:617): three parameter groups (biases, decayed
    weights, norm weights) and
      * 'SGD'  (:832-833)  nesterov momentum, as the fused HIP step FusedSGD;
      * 'AdamW' / 'Adam' (:829-830)  betas = (momentum, 0.999), as FusedAdamW;
      * 'Adamax' / 'NAdam' / 'RAdam' (:829-830) and 'RMSProp' (:831-832) with torch's default hyper-parameters, as FusedAdamax / FusedNAdam /
        FusedRAdam / FusedRMSprop (the same three launches, another compiled rule);
      * 'auto' (:804-813)  SGD(lr 0.01, momentum 0.9) for more than 10000 iterations, else AdamW(lr = round(0.002 * 5 / (4 + nc), 6),
        beta1 0.9) - the caller's lr / momentum are ignored, as in the reference.
    The gradients the step is handed are already the mean over ranks (GradientBuckets.finish divides once), so the step itself
    never scales by the world size."""
    if name == "auto":
        if nc is None:
            nc = getattr(model, "nc", None) or getattr(model.model[-1], "nc", 10)
        lr_fit = round(0.002 * 5 / (4 + nc), 6)
        name, lr, momentum = ("SGD", 0.01, 0.9) if iterations > 10000 else ("AdamW", lr_fit, 0.9)
    known = {x.lower(): x for x in ("Adam", "Adamax", "AdamW", "NAdam", "RAdam", "RMSProp", "SGD")}
    name = known.get(str(name).lower())
    if name == "SGD":
        return FusedSGD(model, lr=lr, momentum=momentum, decay=decay, nesterov=True, max_norm=10.0, ema=ema)
    if name in ("AdamW", "Adam"):
        return FusedAdamW(model, lr=lr, betas=(momentum, 0.999), decay=decay, max_norm=10.0, ema=ema, decoupled=name == "AdamW")
    if name in ("Adamax", "NAdam", "RAdam"):  # trainer.py:829-830: getattr(optim, name)(g[2], lr=lr, betas=(momentum, 0.999), weight_decay=0.0)
        cls = {"Adamax": FusedAdamax, "NAdam": FusedNAdam, "RAdam": FusedRAdam}[name]
        return cls(model, lr=lr, betas=(momentum, 0.999), decay=decay, max_norm=10.0, ema=ema)
    if name == "RMSProp":  # trainer.py:831-832: optim.RMSprop(g[2], lr=lr, momentum=momentum)
        return FusedRMSprop(model, lr=lr, momentum=momentum, decay=decay, max_norm=10.0, ema=ema)
    raise NotImplementedError(f"optimizer {name!r} is not one of the reference's (trainer.py:827-840)")


def synthetic_batch(batch, imgsz, device, seed, boxes_per_image=4):
    """the synthetic batch of SURVEY.md section 8(d) config 3: U[0,1) images, 4 boxes per image, class 0."""
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(batch, 3, imgsz, imgsz, generator=g)
    n = batch * boxes_per_image
    ctr = torch.rand(n, 2, generator=g) * 0.6 + 0.2
    wh = torch.rand(n, 2, generator=g) * 0.3 + 0.05
    return {
        "img": img.to(device),
        "batch_idx": torch.arange(batch).repeat_interleave(boxes_per_image).float().to(device),
        "cls": torch.zeros(n, 1, device=device),
        "bboxes": torch.cat((ctr, wh), 1).to(device),
        "max_boxes": boxes_per_image,
    }


class TrainStep:
    """one optimisation step: forward under autocast, loss.sum() * world (reference trainer.py:386-388), backward
    with bucketed RCCL mean, then the reference's optimizer_step (trainer.py:614-622) as the fused HIP step: global-norm
    clip 10.0 (bf16 needs no GradScaler), SGD-nesterov over the three parameter groups, EMA update; zero_grad.

    graph=True replays the step as a HIP graph (every kernel of libyolo_mi355 only enqueues on the stream it is
    given, so the capture is legal; needs static shapes: batch["max_boxes"] must be set):
      * one rank: the whole step (forward, loss, backward, clip, update, EMA) is one graph;
      * several ranks (or graph="split": the same schedule on one rank, for tests): THREE graphs with the RCCL calls - which cannot
        be captured - between them, so that the exchange overlaps compute as the reference's DDP reducer does (trainer.py:278):
          G1  forward + loss + the HEAD's backward (down to the backbone / head boundary of the YAML), head gradients copied
              into bucket 0's flat buffer                     -> all-reduce of bucket 0 starts (RCCL's own stream)
          G2  the BACKBONE's backward, its gradients copied into bucket 1's flat buffer (runs beside bucket 0's all-reduce)
                                                              -> all-reduce of bucket 1 starts
          G3  clip + update + EMA reading the gradient SUMS from the flat buffers (the step scales by 1 / world itself)
        The backward is split with torch.autograd.grad at the boundary tensors (BaseModel.boundary_layers); gradient joins of
        boundary tensors (ops.GradJoin) carry the head's contribution into the backbone's pass.
      * graph="tail": the round-3 multi-rank form (forward + backward as one graph, gradient mean and update eager behind it).
    Learning rates / momentum changed through `opt.param_groups` reach a replayed graph: they are read from a device
    array (`FusedSGD.sync_hyper`).

    What must NOT be inside a captured region (found in round 2, tools/graph_cat_probe.py): torch.cat / torch.stack -
    and therefore torch.nn.utils.clip_grad_norm_.  On this ROCm build ATen's cat stages its tensor metadata through a
    host buffer that a memcpy NODE copies to the device; a replay copies whatever that host buffer holds by then (any
    later eager cat/stack rewrites it), so the replayed cat reads wrong pointers: silently wrong values, or a memory
    fault.  The captured step contains only kernels of this library and elementwise ATen ops."""

    def __init__(self, model, world_size=1, lr=0.01, dtype=torch.bfloat16, bucket_bytes=32 << 20, graph=False, ema=True, optimizer="SGD",
                 momentum=0.937, decay=5e-4):
        self.model = model
        self.world = world_size
        self.dtype = dtype
        self.ema = ModelEMA(model) if ema is True else (ema or None)
        self.opt = build_optimizer(model, name=optimizer, lr=lr, momentum=momentum, decay=decay, ema=self.ema)
        self.use_graph = bool(graph)
        self.full_graph = self.use_graph and world_size == 1 and graph not in ("split", "tail")
        self.overlap_graphs = self.use_graph and not self.full_graph and graph != "tail"
        self.params = [p for p in model.parameters() if p.requires_grad]
        groups = None
        if self.overlap_graphs:
            # bucket 0 = the head's parameters (their gradients are complete when G1 ends), bucket 1 = the backbone's
            nb = len(model.yaml["backbone"])
            head_ids = {id(p) for m in model.model if m.i >= nb for p in m.parameters()}
            self._head_params = [p for p in reversed(self.params) if id(p) in head_ids]
            self._back_params = [p for p in reversed(self.params) if id(p) not in head_ids]
            self._boundary = model.boundary_layers()
            groups = [self._head_params, self._back_params]
        self.buckets = GradientBuckets(model, world_size, bucket_bytes, overlap=not self.use_graph, groups=groups)
        # the buckets hold gradient SUMS over ranks; the fused step applies 1 / world (hyper[11]) - no divide launches
        self.opt.world = world_size
        self._graph = None
        self._static = None
        self._static_items = None
        self._graph_grads = None
        self._seed = None
        self._comm_events = None  # time_exposed_communication(): [(event after the backward's last graph, event after the wait for the buckets)]

    def __call__(self, batch):
        if not self.use_graph:
            return self.eager_step(batch)
        if self._graph is None:
            if batch.get("max_boxes") is None:
                raise ValueError("graph=True needs batch['max_boxes'] (static target shape)")
            # The first batch's tensors become the graph's static inputs (later batches are copied into them).
            self._static = dict(batch)
            for _ in range(3):  # warm-up: allocator, lazy state (weight arena, optimizer tables), workspaces
                self.eager_step(self._static)
            torch.cuda.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            # several ranks: the process group's watchdog thread polls its events while this thread captures; only
            # this thread's calls may invalidate the capture
            mode = "global" if self.world == 1 else "thread_local"
            if self.overlap_graphs:
                self._capture_overlap(mode)
            else:
                with torch.cuda.graph(self._graph, capture_error_mode=mode):
                    if self.full_graph:
                        self._static_items = self.eager_step(self._static)
                    else:
                        self._static_items = self._forward_backward(self._static)
                if not self.full_graph:  # the gradients the replays rewrite in place
                    self._graph_grads = {p: p.grad for p in self.params if p.grad is not None}
                elif self.ema is not None:
                    self.opt.count_updates(-1)  # the capture recorded the update without running it
        else:
            for k, v in batch.items():
                if torch.is_tensor(v) and v is not self._static[k]:
                    if v.shape != self._static[k].shape:  # copy_ would broadcast silently (e.g. a shorter label tensor)
                        raise ValueError(f"graph=True replays static shapes: batch['{k}'] is {tuple(v.shape)}, captured {tuple(self._static[k].shape)}")
                    self._static[k].copy_(v)
                elif not torch.is_tensor(v) and v != self._static[k]:
                    raise ValueError(f"graph=True: batch['{k}'] = {v!r} differs from the captured value {self._static[k]!r}")
        if self.full_graph or self.overlap_graphs:
            self.opt.sync_hyper()  # scheduler changes reach the captured update through the device array
        self._graph.replay()
        if self.full_graph:
            if self.ema is not None:
                self.opt.count_updates(+1)  # the captured step advanced the device counter
        elif self.overlap_graphs:
            self.buckets.start(0)       # the head's gradient sums travel ...
            self._graph2.replay()       # ... while the backbone's backward runs
            self.buckets.start(1)
            ev = self._comm_event_pair()
            if ev:
                ev[0].record()          # (completes when the backbone's backward does)
            self.buckets.wait_all(divide=False)  # the current stream waits for both; .grad = the flat buffers' slices
            if ev:
                ev[1].record()          # (completes once the compute stream may go on: the gap is communication nothing hid)
            self._graph3.replay()
            if self.ema is not None:
                self.opt.count_updates(+1)
            self.opt.zero_grad(set_to_none=True)
        else:
            self._reduce_and_update(self._graph_grads)
            self.opt.zero_grad(set_to_none=True)  # drops references only: the graph owns its gradient buffers
        return self._static_items

    # ---- the three-graph schedule -----------------------------------------------------------------------------------------------
    def _head_pass(self, batch):
        """forward + loss + the backward of everything behind the backbone / head boundary (the head reads detached leaves of the
        boundary tensors: BaseModel._predict_once) -> (loss items, head gradients aligned with self._head_params,
        [(boundary tensor, gradient the head formed for it)])."""
        model = self.model
        model.train()
        model._taps = dict(self._boundary)
        try:
            with torch.autocast("cuda", dtype=self.dtype, enabled=self.dtype != torch.float32):
                loss, items = model(batch)
            taps = model._taps
        finally:
            model._taps = None
        if self._seed is None or self._seed.device != loss.device:
            self._seed = torch.full((3,), float(self.world), dtype=torch.float32, device=loss.device)
        pairs = [v for v in taps.values() if isinstance(v, tuple)]
        leaves = [leaf for _, leaf in pairs]
        # torch.autograd.grad, not backward(): the leaves are channel slices of concat buffers (not dense), and AccumulateGrad would
        # re-lay every gradient it stores for them out as NCHW-contiguous copies; captured gradients are handed over as they are
        with ops.deferred_wgrad(True), ops.wgrad_riders(self.use_graph):
            grads = torch.autograd.grad([loss], leaves + self._head_params, [self._seed], allow_unused=True)
        return items, list(grads[len(leaves):]), [(orig, g) for (orig, _), g in zip(pairs, grads[: len(leaves)])]

    def _backbone_pass(self, pairs):
        """the backbone's backward, from the boundary tensors with the gradients the head left in their leaves.  A boundary tensor that
        also has backbone consumers carries a gradient join: the head's gradient is deposited there and the backbone consumer that
        arrives last adds it in its data-gradient epilogue; the others are roots of the pass."""
        roots, grads = [], []
        for orig, g in pairs:
            if g is None:
                continue
            j = ops.join_of(orig)
            if j is not None:
                adds = j.arrive()
                if adds is None:
                    j.deposit(g)
                    continue
                g = ops._accumulate(g, adds) if adds else g  # (no backbone consumer left to arrive: the head's gradient is the total)
            roots.append(orig)
            grads.append(g)
        with ops.deferred_wgrad(True), ops.wgrad_riders(self.use_graph):
            torch.autograd.backward(roots, grads)

    def _pack(self, bi, params, grads):
        """gradients -> the slices of bucket bi's flat buffer (one multi-tensor copy; parameters without a gradient keep zeros there)."""
        views = self.buckets.flat_views(bi)
        where = {id(p): v for p, v in zip(self.buckets.buckets[bi], views)}
        dst, src = [], []
        for p, g in zip(params, grads):
            if g is not None and not (g.data_ptr() == where[id(p)].data_ptr() and g.dtype == where[id(p)].dtype):  # (weight gradients are born there: ops.grad_arena)
                dst.append(where[id(p)])
                src.append(g if g.dtype == where[id(p)].dtype else g.to(where[id(p)].dtype))
        if dst:
            torch._foreach_copy_(dst, src)

    def _capture_overlap(self, mode):
        b = self._static
        # conv / linear weight gradients are written straight into the flat buckets (ops.grad_arena): only the small vectors (BatchNorm
        # and LayerNorm parameters, biases, the paired Detect weights) are copied there
        arena = {id(p): v for bi in range(len(self.buckets.buckets)) for p, v in zip(self.buckets.buckets[bi], self.buckets.flat_views(bi)) if p.dim() >= 2}
        with torch.cuda.graph(self._graph, capture_error_mode=mode), ops.grad_arena(arena):
            self._static_items, hg, pairs = self._head_pass(b)
            self._pack(0, self._head_params, hg)
        del hg
        self._graph2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph2, pool=self._graph.pool(), capture_error_mode=mode), ops.grad_arena(arena):
            self._backbone_pass(pairs)
            self._pack(1, self._back_params, [p.grad for p in self._back_params])
        del pairs
        self.buckets.wait_all(divide=False)  # (nothing in flight: points .grad at the flat slices the update graph will read)
        self._graph3 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph3, pool=self._graph.pool(), capture_error_mode=mode):
            self.opt.step(None)
        if self.ema is not None:
            self.opt.count_updates(-1)  # captured, not executed
        self.opt.zero_grad(set_to_none=True)

    def _forward_backward(self, batch):
        self.model.train()
        with torch.autocast("cuda", dtype=self.dtype, enabled=self.dtype != torch.float32):
            loss, items = self.model(batch)
        # this step owns its gradients: zero_grad(set_to_none=True) after every update, so AccumulateGrad adopts the tensors
        # the weight-gradient Functions return and nothing reads them before backward() is over - the condition under which
        # their slab sums may be batched into one launch at the end of the pass (ops.deferred_wgrad; parameters that do hold a
        # gradient or a hook - the overlapped DDP schedule - are detected there and reduced at once)
        # backward of loss.sum() * world (reference trainer.py:386-388, 394) seeded directly with d(total)/d(loss) = world: the
        # sum, the multiplication and their backward nodes would be five one-element launches
        if self._seed is None or self._seed.device != loss.device:
            self._seed = torch.full((3,), float(self.world), dtype=torch.float32, device=loss.device)
        # (captured steps run on one stream: there the BatchNorm-backward final passes ride in the weight-gradient launches, ops.wgrad_riders)
        with ops.deferred_wgrad(True), ops.async_wgrad(ASYNC_WGRAD and not self.use_graph), ops.wgrad_riders(self.use_graph):  # joins the side stream on exit
            torch.autograd.backward([loss], [self._seed])
        return items

    def _reduce_and_update(self, grads_of=None):
        ev = self._comm_event_pair()
        if ev:
            ev[0].record()
        self.buckets.finish(grads_of, divide=False)  # world > 1: leaves the SUM in .grad (views of the flat buckets); the step scales by 1 / world
        if ev:
            ev[1].record()
        self.opt.step(grads_of if self.world == 1 else None)

    # ---- exposed-communication probe (bench.py's several-rank line) ---------------------------------------------------------------
    def time_exposed_communication(self, on=True):
        """from now on every step brackets its wait for the gradient exchange with two events on the compute stream: what elapses between them
        is exchange time the backward did not hide (three-graph schedule: behind the backbone's backward; tail / eager: the whole exchange)."""
        self._comm_events = [] if on else None

    def _comm_event_pair(self):
        if self._comm_events is None or self.world == 1:
            return None
        pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        self._comm_events.append(pair)
        return pair

    def exposed_communication_ms(self):
        """mean milliseconds per step between the two events (call after a synchronize); None when nothing was timed"""
        ev = self._comm_events or []
        return sum(a.elapsed_time(b) for a, b in ev) / len(ev) if ev else None

    def eager_step(self, batch):
        items = self._forward_backward(batch)
        self._reduce_and_update()
        self.opt.zero_grad(set_to_none=True)
        return items
