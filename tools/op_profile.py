"""Development tool: torch.profiler view of one eager training step: which ATen ops still run on the path, and the
Python call sites of the copies / adds among them."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from improving_yolov8_cbam_swinblock_amd.engine.trainer import TrainStep, synthetic_batch  # noqa: E402
from improving_yolov8_cbam_swinblock_amd.nn.tasks import DetectionModel  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = DetectionModel("yolov8s.yaml", ch=3, nc=1).to(dev)
step = TrainStep(model, graph=False)
batch = synthetic_batch(32, 640, dev, 1)
for _ in range(3):
    step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(batch)
    torch.cuda.synchronize()
ka = prof.key_averages()
print("aten ops by count")
for e in sorted(ka, key=lambda e: -e.count):
    if e.key.startswith("aten::") and e.count >= 4:
        print(f"{e.count:5d}  cpu {e.cpu_time_total / 1e3:8.2f} ms  cuda {getattr(e, 'device_time_total', 0) / 1e3:8.2f} ms  {e.key[:70]}")
print("\ncall sites of copy / clone / add / fill ops (count, op, innermost repo frames)")
ks = prof.key_averages(group_by_stack_n=12)
for e in sorted(ks, key=lambda e: -e.count):
    if e.key in ("aten::copy_", "aten::clone", "aten::add", "aten::add_", "aten::contiguous", "aten::zero_", "aten::fill_", "aten::zeros", "aten::to") and e.count >= 2:
        frames = [s for s in e.stack if "improving_yolov8" in s or "torch/optim" in s or "clip_grad" in s][:3]
        print(f"{e.count:4d} {e.key:16s} cuda {getattr(e, 'device_time_total', 0) / 1e3:6.3f} ms  {frames}")
