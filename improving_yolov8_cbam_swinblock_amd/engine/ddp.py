"""Data-parallel strategy of the reference (ultralytics/engine/trainer.py:221-232,278,387-388), MI355X-first.

One process per GPU; `torch.distributed` backend "nccl" is RCCL on ROCm (gloo on CPU for tests).  The
reference wraps the model in DistributedDataParallel; here the only exchange step of the path - the
gradient mean over ranks - is done explicitly: parameters are grouped into a few large flat buckets
(xGMI is point-to-point, 7 links per GPU: few large messages beat many small ones), each bucket's
all-reduce is launched asynchronously from a post-accumulate-grad hook as soon as its last gradient
is ready, so the collectives overlap the rest of backward, and `finish()` waits and scatters the
averaged values back.  BatchNorm statistics stay per-rank (the reference has no SyncBatchNorm).
"""
import os

import torch
import torch.distributed as dist


def shard_seed(base, rank):
    """per-rank data seed (reference: DistributedSampler shards, data/build.py:166; synthetic data here)."""
    return int(base) + int(rank)


def setup(backend=None):
    """init the process group from torchrun's environment (reference _setup_ddp, trainer.py:221-232)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend or ("nccl" if torch.cuda.is_available() else "gloo"), rank=rank, world_size=world)
    return rank, local, world


def broadcast_parameters(module, src=0):
    """same initial weights and buffers on every rank (what DDP's constructor does, trainer.py:278)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers() if b.dtype.is_floating_point]
    for t in tensors:
        dist.broadcast(t, src)


class GradientBuckets:
    """bucketed, backward-overlapped gradient mean.  Usage: gb = GradientBuckets(model, world); ...
    loss.backward(); gb.finish()."""

    def __init__(self, module, world_size, bucket_bytes=32 << 20, comm_dtype=None):
        self.world = world_size
        self.comm_dtype = comm_dtype
        params = [p for p in module.parameters() if p.requires_grad]
        params.reverse()  # gradients become ready roughly in reverse registration order
        self.buckets = []
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
        for bi, b in enumerate(self.buckets):
            for p in b:
                self._where[p] = bi
            n = sum(p.numel() for p in b)
            self._flat.append(torch.zeros(n, dtype=comm_dtype or b[0].dtype, device=b[0].device))
        self._pending = [len(b) for b in self.buckets]
        self._work = [None] * len(self.buckets)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in params] if world_size > 1 else []

    def _launch(self, bi):
        flat = self._flat[bi]
        off = 0
        for p in self.buckets[bi]:
            n = p.numel()
            flat[off : off + n].copy_(p.grad.reshape(-1))
            off += n
        self._work[bi] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)

    def _on_grad(self, p):
        bi = self._where[p]
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._launch(bi)

    def finish(self):
        """wait for every bucket, write grad = sum / world back into .grad, re-arm for the next step."""
        if self.world == 1:
            return
        for bi, b in enumerate(self.buckets):
            if self._work[bi] is None:  # parameters without a gradient this step (unused): reduce what exists
                for p in b:
                    if p.grad is None:
                        p.grad = torch.zeros_like(p)
                self._launch(bi)
            self._work[bi].wait()
            flat = self._flat[bi]
            off = 0
            for p in b:
                n = p.numel()
                p.grad.copy_(flat[off : off + n].view_as(p.grad)).div_(self.world)
                off += n
            self._work[bi] = None
            self._pending[bi] = len(b)


def allreduce_mean_gradients(module, world_size, bucket_bytes=32 << 20):
    """non-overlapped form (tests, and callers that already ran backward)."""
    if world_size == 1:
        return
    gb = GradientBuckets(module, 1, bucket_bytes)  # no hooks
    gb.world = world_size
    for bi, b in enumerate(gb.buckets):
        for p in b:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        gb._launch(bi)
    gb.finish()
