"""Development probe: 300 graph-replayed training steps on four rotating synthetic batches (bs 32, 640x640, bf16): the
three loss terms must fall and every parameter / buffer must stay finite.  YMI_PROBE_OPT=AdamW (lr 0.002) runs the AdamW branch."""
import torch, sys, os
sys.path.insert(0, os.getcwd())
from improving_yolov8_cbam_swinblock_amd.engine.trainer import TrainStep, synthetic_batch
from improving_yolov8_cbam_swinblock_amd.nn.tasks import DetectionModel
dev = torch.device("cuda:0"); torch.manual_seed(0)
model = DetectionModel("yolov8s.yaml", ch=3, nc=1).to(dev)
opt = os.environ.get("YMI_PROBE_OPT", "SGD")
step = TrainStep(model, graph=True, lr=0.01 if opt == "SGD" else 0.002, optimizer=opt, momentum=0.937 if opt == "SGD" else 0.9)
batches = [synthetic_batch(32, 640, dev, s) for s in range(4)]
hist = []
for i in range(300):
    it = step(batches[i % 4])
    if i % 25 == 0 or i == 299:
        v = it.float().cpu(); hist.append(v.sum().item()); print(i, [round(x, 4) for x in v.tolist()], flush=True)
assert all(map(lambda x: x == x, hist)), "NaN"
assert hist[-1] < hist[0]
bad = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
bs = [n for n, b in model.named_buffers() if b.dtype.is_floating_point and not torch.isfinite(b).all()]
print("non-finite params:", bad, "buffers:", bs)
