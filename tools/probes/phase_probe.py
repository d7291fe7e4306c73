"""Development probe: HIP-graph replay time of cumulative pieces of the training step (bs=32 640x640 bf16):
net forward | + loss | + backward | + clip + optimizer.   usage: phase_probe.py [BS] [SZ]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from improving_yolov8_cbam_swinblock_amd.engine.trainer import build_optimizer, synthetic_batch  # noqa: E402
from improving_yolov8_cbam_swinblock_amd.nn.tasks import DetectionModel  # noqa: E402

dev = torch.device("cuda:0")
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
sz = int(sys.argv[2]) if len(sys.argv) > 2 else 640
torch.manual_seed(0)
model = DetectionModel("yolov8s.yaml", ch=3, nc=1).to(dev).train()
batch = synthetic_batch(bs, sz, dev, 1)
opt = build_optimizer(model)
params = [p for p in model.parameters() if p.requires_grad]


def body(mode):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        if mode == "net":
            preds = model(batch["img"])
            return preds[0] if isinstance(preds, (list, tuple)) else preds
        loss, items = model(batch)
    if mode == "net+loss":
        return items
    loss.sum().backward()
    if mode == "full":
        torch.nn.utils.clip_grad_norm_(params, max_norm=10.0)
        opt.step()
    model.zero_grad(set_to_none=True)
    return items


prev = 0.0
for mode in ("net", "net+loss", "net+loss+bwd", "full"):
    for _ in range(3):
        body(mode)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = body(mode)
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{mode:14s} {ms:8.3f} ms   (+{ms - prev:.3f})", flush=True)
    prev = ms
    del g
