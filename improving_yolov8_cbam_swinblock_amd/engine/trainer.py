"""Training step of the hot path (reference: ultralytics/engine/trainer.py:383-399,614-622,788-849 and
models/yolo/detect/train.py:90-115), reduced to what the benchmark step needs: bf16 autocast forward,
v8 detection loss, backward (+ RCCL gradient mean), gradient clip 10.0, SGD-nesterov step."""
import os

import torch
import torch.nn as nn

from .. import ops
from .ddp import GradientBuckets

# weight-gradient GEMMs on a second stream during backward (ops.async_wgrad) in EAGER steps; YMI_WGRAD_STREAM=0 keeps
# one stream.  Graph-replayed steps stay single-stream: measured no wall-time gain there, and concurrent kernels stretch
# each other's durations, which would blur the per-kernel roofline measurement.
ASYNC_WGRAD = os.environ.get("YMI_WGRAD_STREAM", "1") != "0"


def build_optimizer(model, lr=0.01, momentum=0.937, decay=5e-4):
    """parameter groups of reference build_optimizer (trainer.py:788-849): weights with decay, BN/LN weights and
    all biases without; SGD with nesterov momentum (the 'SGD' branch, trainer.py:832-833)."""
    g_w, g_n, g_b = [], [], []
    norm = tuple(v for k, v in nn.__dict__.items() if "Norm" in k)
    for mod_name, mod in model.named_modules():
        for pn, p in mod.named_parameters(recurse=False):
            if not p.requires_grad:
                continue
            if "bias" in pn:
                g_b.append(p)
            elif isinstance(mod, norm):
                g_n.append(p)
            else:
                g_w.append(p)
    # fused multi-tensor update on the GPU (one kernel per parameter group instead of a chain of foreach passes)
    fused = bool(g_b) and g_b[0].is_cuda and os.environ.get("YMI_FUSED_SGD", "1") != "0"
    opt = torch.optim.SGD(g_b, lr=lr, momentum=momentum, nesterov=True, fused=True if fused else None)  # None: foreach
    opt.add_param_group({"params": g_w, "weight_decay": decay})
    opt.add_param_group({"params": g_n, "weight_decay": 0.0})
    return opt


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
    with bucketed RCCL mean, unscale-free clip (bf16 needs no GradScaler), optimizer step, zero_grad.

    graph=True replays the step as a HIP graph (every kernel of libyolo_mi355 only enqueues on the stream it is
    given, so the capture is legal; needs static shapes: batch["max_boxes"] must be set):
      * one rank: the whole step (forward, loss, backward, clip, update) is one graph;
      * several ranks: forward + loss + backward are one graph; the RCCL gradient mean, the clip and the update run
        eagerly after the replay (a dozen launches), because RCCL calls are not captured."""

    def __init__(self, model, world_size=1, lr=0.01, dtype=torch.bfloat16, bucket_bytes=32 << 20, graph=False):
        self.model = model
        self.world = world_size
        self.dtype = dtype
        self.opt = build_optimizer(model, lr=lr)
        self.use_graph = bool(graph)
        self.full_graph = self.use_graph and world_size == 1 and graph != "split"  # graph="split": the multi-rank form on one rank
        self.buckets = GradientBuckets(model, world_size, bucket_bytes, overlap=not self.use_graph)
        self.params = [p for p in model.parameters() if p.requires_grad]
        self._graph = None
        self._static = None
        self._static_items = None
        self._graph_grads = None

    def __call__(self, batch):
        if not self.use_graph:
            return self.eager_step(batch)
        if self._graph is None:
            if batch.get("max_boxes") is None:
                raise ValueError("graph=True needs batch['max_boxes'] (static target shape)")
            # The first batch's tensors become the graph's static inputs (later batches are copied into them).
            self._static = dict(batch)
            # Warm-up (allocator, lazy state, workspaces) on the CURRENT stream.  Measured on ROCm 7.0 / torch 2.10:
            # warming up on a side stream, as the CUDA recipe suggests, left the caching allocator handing the
            # graph's private-pool blocks to later eager allocations (corrupted replays at bs >= 16); with the
            # warm-up on the current stream replays stay isolated (tools/graph_probe.py).
            for _ in range(3):
                self.eager_step(self._static)
            torch.cuda.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            # several ranks: the process group's watchdog thread polls its events while this thread captures; only
            # this thread's calls may invalidate the capture
            mode = "global" if self.world == 1 else "thread_local"
            with torch.cuda.graph(self._graph, capture_error_mode=mode):
                if self.full_graph:
                    self._static_items = self.eager_step(self._static)
                else:
                    self._static_items = self._forward_backward(self._static)
            if not self.full_graph:  # the gradients the replays rewrite in place
                self._graph_grads = {p: p.grad for p in self.params if p.grad is not None}
        else:
            for k, v in batch.items():
                if torch.is_tensor(v) and v is not self._static[k]:
                    if v.shape != self._static[k].shape:  # copy_ would broadcast silently (e.g. a shorter label tensor)
                        raise ValueError(f"graph=True replays static shapes: batch['{k}'] is {tuple(v.shape)}, captured {tuple(self._static[k].shape)}")
                    self._static[k].copy_(v)
                elif not torch.is_tensor(v) and v != self._static[k]:
                    raise ValueError(f"graph=True: batch['{k}'] = {v!r} differs from the captured value {self._static[k]!r}")
        self._graph.replay()
        if not self.full_graph:
            self._reduce_and_update(self._graph_grads)
            self.opt.zero_grad(set_to_none=True)  # drops references only: the graph owns its gradient buffers
        return self._static_items

    def _forward_backward(self, batch):
        self.model.train()
        with torch.autocast("cuda", dtype=self.dtype, enabled=self.dtype != torch.float32):
            loss, items = self.model(batch)
            total = loss.sum() * self.world
        with ops.async_wgrad(ASYNC_WGRAD and not self.use_graph):  # joins the side stream on exit
            total.backward()
        return items

    def _reduce_and_update(self, grads_of=None):
        if grads_of is not None and self.world == 1:  # (single-rank use of the split path: tests)
            for p, g in grads_of.items():
                p.grad = g
        self.buckets.finish(grads_of)
        torch.nn.utils.clip_grad_norm_(self.params, max_norm=10.0)
        self.opt.step()

    def eager_step(self, batch):
        items = self._forward_backward(batch)
        self._reduce_and_update()
        self.opt.zero_grad(set_to_none=True)
        return items
