"""Development probe: which host-side tensor copies / fills does one eager training step issue?  (captured into the HIP graph they
become blit kernels).  usage: python tools/probes/copy_probe.py"""
import collections, os, sys, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from improving_yolov8_cbam_swinblock_amd.engine.trainer import TrainStep, synthetic_batch
from improving_yolov8_cbam_swinblock_amd.nn.tasks import DetectionModel

dev = torch.device("cuda:0")
model = DetectionModel("yolov8s.yaml", ch=3, nc=1).to(dev).train()
batch = synthetic_batch(32, 640, dev, 1)
step = TrainStep(model, graph=False)
for _ in range(2):
    step(batch)
torch.cuda.synchronize()
log = collections.Counter()

def wrap(name):
    orig = getattr(torch.Tensor, name)
    def f(self, *a, **k):
        if self.is_cuda:
            fr = [x for x in traceback.extract_stack()[:-1] if "improving_yolov8" in x.filename or "bench" in x.filename]
            where = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno}" if fr else "?"
            log[(name, where, tuple(self.shape), str(self.dtype))] += 1
        return orig(self, *a, **k)
    setattr(torch.Tensor, name, f)

for n in ("copy_", "clone", "zero_", "fill_", "contiguous", "to", "float", "add_", "mul_"):
    wrap(n)
step(batch)
torch.cuda.synchronize()
for k, v in sorted(log.items(), key=lambda kv: -kv[1]):
    print(v, k)
