"""Data-parallel strategy of the reference (ultralytics/engine/trainer.py:221-232,278,387-388), MI355X-first.

One process per GPU; `torch.distributed` backend "nccl" is RCCL on ROCm (gloo on CPU for tests).  The
reference wraps the model in DistributedDataParallel; here the only exchange step of the path - the
gradient mean over ranks - is done explicitly: parameters are grouped into a few large flat buckets
(xGMI is point-to-point, 7 links per GPU: few large messages beat many small ones); a bucket is packed
with one multi-tensor copy, all-reduced asynchronously (from a post-accumulate-grad hook as soon as its
last gradient is ready when backward runs eagerly, right after the replay when backward is a HIP graph),
and `finish()` waits, divides the flat buffer once and points `.grad` at its slices (no copy back).
BatchNorm statistics stay per-rank (the reference has no SyncBatchNorm).
"""
import os

import torch
import torch.distributed as dist


def shard_seed(base, rank):
    """per-rank data seed (reference: DistributedSampler shards, data/build.py:166; synthetic data here)."""
    return int(base) + int(rank)


def setup(backend=None, device_index=None):
    """init the process group from torchrun's environment (reference _setup_ddp, trainer.py:221-232: NCCL if available, else
    Gloo).  device_index: the GPU this rank uses (default LOCAL_RANK; a gloo rehearsal may place several ranks on one device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if torch.cuda.is_available():
            torch.cuda.set_device(local if device_index is None else device_index)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL buffer sharing)
        dist.init_process_group(backend or ("nccl" if torch.cuda.is_available() else "gloo"), rank=rank, world_size=world)
    return rank, local, world


def broadcast_parameters(module, src=0, bucket_bytes=64 << 20):
    """same initial weights and buffers on every rank (what DDP's constructor does, trainer.py:278): the tensors are
    packed into a few flat buffers per dtype (one multi-tensor copy in, one out), so the exchange is a handful of large
    broadcasts - xGMI is point-to-point, 141 small messages would each pay the link latency."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers()]
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    for dt, group in by_dtype.items():
        cur, size = [], 0
        chunks = []
        for t in group:
            cur.append(t)
            size += t.numel() * t.element_size()
            if size >= bucket_bytes:
                chunks.append(cur)
                cur, size = [], 0
        if cur:
            chunks.append(cur)
        for chunk in chunks:
            flat = torch.empty(sum(t.numel() for t in chunk), dtype=dt, device=chunk[0].device)
            views, off = [], 0
            for t in chunk:
                views.append(flat[off : off + t.numel()].view_as(t))
                off += t.numel()
            torch._foreach_copy_(views, chunk)
            dist.broadcast(flat, src)
            torch._foreach_copy_(chunk, views)


class GradientBuckets:
    """bucketed gradient mean.  overlap=True: each bucket's all-reduce starts from a post-accumulate-grad hook as
    soon as its last gradient is ready (eager backward); overlap=False: `finish()` reduces everything after backward
    (used when backward is a replayed HIP graph, which cannot call into RCCL).
    Usage: gb = GradientBuckets(model, world); ... loss.backward(); gb.finish()."""

    def __init__(self, module, world_size, bucket_bytes=32 << 20, comm_dtype=None, overlap=True, groups=None):
        """groups: explicit buckets (lists of parameters, in the order their gradients become ready) instead of size-based ones - the
        split-graph schedule of engine.trainer.TrainStep uses [head parameters], [backbone parameters]."""
        self.world = world_size
        self.comm_dtype = comm_dtype
        params = [p for p in module.parameters() if p.requires_grad]
        params.reverse()  # gradients become ready roughly in reverse registration order
        self.buckets = []
        if groups is not None:
            self.buckets = [list(g) for g in groups if len(g)]
            if sorted(id(p) for b in self.buckets for p in b) != sorted(id(p) for p in params):
                raise ValueError("GradientBuckets: the groups must partition the trainable parameters")
        else:
            cur, size = [], 0
            for p in params:
                cur.append(p)
                size += p.numel() * 4
                if size >= bucket_bytes:
                    self.buckets.append(cur)
                    cur, size = [], 0
            if cur:
                self.buckets.append(cur)
        self._where = {}
        self._flat = []
        self._views = []  # per bucket: slices of the flat buffer shaped like the parameters
        for bi, b in enumerate(self.buckets):
            for p in b:
                self._where[p] = bi
            n = sum(p.numel() for p in b)
            flat = torch.zeros(n, dtype=comm_dtype or b[0].dtype, device=b[0].device)
            views, off = [], 0
            for p in b:
                views.append(flat[off : off + p.numel()].view_as(p))
                off += p.numel()
            self._flat.append(flat)
            self._views.append(views)
        self._pending = [len(b) for b in self.buckets]
        self._work = [None] * len(self.buckets)
        # the hooks read gradients DURING backward: every weight gradient must be complete when its hook fires.  ops._wgrad
        # sees the hooks on the parameters and keeps those gradients out of the end-of-pass batched slab sum.
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in params] if (world_size > 1 and overlap) else []

    def _launch(self, bi, grads=None):
        """pack the bucket's gradients into its flat buffer (one multi-tensor copy) and start the all-reduce."""
        if grads is None:
            grads = [p.grad for p in self.buckets[bi]]
        if grads and grads[0].is_cuda:
            from .. import ops

            ops.join_side_stream()  # weight gradients may still be in flight on the side stream (ops.async_wgrad)
        torch._foreach_copy_(self._views[bi], grads)
        self._work[bi] = dist.all_reduce(self._flat[bi], op=dist.ReduceOp.SUM, async_op=True)

    def _on_grad(self, p):
        bi = self._where[p]
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._launch(bi)

    def flat_views(self, bi):
        """the slices of bucket bi's flat buffer, shaped like its parameters (static storage: a captured graph may write them)."""
        return self._views[bi]

    def start(self, bi, packed=True):
        """start bucket bi's all-reduce; packed: its gradients already lie in the flat buffer (written there by a captured multi-tensor
        copy) - else they are packed from .grad first.  The collective is issued on the communicator's stream behind whatever the
        current stream has enqueued so far, so work enqueued afterwards (the backbone's backward graph) overlaps it."""
        if self.world == 1 and not (dist.is_available() and dist.is_initialized()):
            return  # one rank, no process group (the schedule's one-rank test form): nothing to exchange
        if packed:
            self._work[bi] = dist.all_reduce(self._flat[bi], op=dist.ReduceOp.SUM, async_op=True)
        else:
            self._launch(bi)

    def wait_all(self, divide=True):
        """wait for the started buckets (the current stream waits; the host does not block on NCCL) and point .grad at the flat slices.
        divide=False leaves the SUM there: the fused optimizer step scales by 1 / world itself (FusedSGD.world)."""
        for bi, b in enumerate(self.buckets):
            if self._work[bi] is not None:
                self._work[bi].wait()
                self._work[bi] = None
            if divide:
                self._flat[bi].div_(self.world)
            for p, v in zip(b, self._views[bi]):
                p.grad = v if v.dtype == p.dtype else v.to(p.dtype)
            self._pending[bi] = len(b)

    def finish(self, grads_of=None, force=False, divide=True):
        """wait for every bucket and leave grad = sum / world in .grad (as views of the flat buffers: no copy back);
        re-arm for the next step.  grads_of: optional {param: gradient tensor} to reduce instead of .grad.
        force: run the collectives even on one rank (the RCCL smoke test of a one-GPU box).
        divide=False: leave the SUM (engine.trainer: the fused optimizer step scales by 1 / world itself)."""
        if self.world == 1 and not force:
            return
        for bi, b in enumerate(self.buckets):
            if self._work[bi] is None:  # no hook fired (overlap off, or parameters without a gradient this step)
                if grads_of is not None:
                    src = [grads_of[p] if p in grads_of else torch.zeros_like(p) for p in b]
                else:
                    for p in b:
                        if p.grad is None:
                            p.grad = torch.zeros_like(p)
                    src = None
                self._launch(bi, src)
        for bi, b in enumerate(self.buckets):
            self._work[bi].wait()
            if divide:
                self._flat[bi].div_(self.world)
            for p, v in zip(b, self._views[bi]):
                p.grad = v if v.dtype == p.dtype else v.to(p.dtype)
            self._work[bi] = None
            self._pending[bi] = len(b)


def allreduce_mean_gradients(module, world_size, bucket_bytes=32 << 20):
    """non-overlapped form (tests, and callers that already ran backward)."""
    if world_size == 1:
        return
    gb = GradientBuckets(module, world_size, bucket_bytes, overlap=False)
    gb.finish()
