"""Development tool: torch.profiler op counts for one eager training step (which ATen ops still run on the path)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from torch.profiler import profile, ProfilerActivity
from improving_yolov8_cbam_swinblock_amd.engine.trainer import TrainStep, synthetic_batch
from improving_yolov8_cbam_swinblock_amd.nn.tasks import DetectionModel
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = DetectionModel("yolov8s.yaml", ch=3, nc=1).to(dev)
step = TrainStep(model, graph=False)
batch = synthetic_batch(32, 640, dev, 1)
for _ in range(3): step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(batch); torch.cuda.synchronize()
ka = prof.key_averages()
rows = sorted(ka, key=lambda e: -e.count)
print("top ops by count")
for e in rows[:45]:
    print(f"{e.count:5d}  cpu {e.cpu_time_total/1e3:8.2f} ms  cuda {getattr(e,'device_time_total',0)/1e3:8.2f} ms  {e.key[:70]}")
# memcpy sources
print("\nmemcpy DtoD stacks")
ks = prof.key_averages(group_by_stack_n=6)
for e in sorted(ks, key=lambda e: -e.count):
    if "copy_" in e.key and e.count >= 5:
        print(e.count, e.key, [s for s in e.stack[:6]])
