"""diagnostic: where does a bf16 C2f leave the storage-matched oracle?  prints per-slice errors of the concat input of cv2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oracle.modules as OM
from oracle import quant
import improving_yolov8_cbam_swinblock_amd.nn.modules as PM

def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / b.norm().clamp(min=1e-12))

torch.manual_seed(0)
args = (128, 128, 2, True)
o = OM.C2f(*args)
for b in o.modules():
    if isinstance(b, torch.nn.BatchNorm2d):
        b.eps, b.momentum = 1e-3, 0.03
        b.weight.data.uniform_(0.5, 1.5); b.bias.data.normal_(0, 0.3)
quant.round_weights_(o)
m = PM.C2f(*args)
for b in m.modules():
    if isinstance(b, torch.nn.BatchNorm2d):
        b.eps, b.momentum = 1e-3, 0.03
m.load_state_dict(o.state_dict())
m = m.cuda().train(); o.train()
x = torch.randn(4, 128, 40, 40).bfloat16().float()
cap = {}
o.cv2.register_forward_pre_hook(lambda mod, inp: cap.__setitem__("o", inp[0].detach().clone()))
m.cv2.register_forward_pre_hook(lambda mod, inp: cap.__setitem__("p", inp[0].detach().float().cpu().clone()))
for name, mod in o.named_modules():
    if isinstance(mod, OM.Conv):
        mod.register_forward_hook(lambda md, i, out, name=name: cap.__setitem__("o." + name, out.detach().clone()))
for name, mod in m.named_modules():
    if isinstance(mod, PM.Conv):
        mod.register_forward_hook(lambda md, i, out, name=name: cap.__setitem__("p." + name, out.detach().float().cpu().clone()))
with quant.storage(torch.bfloat16):
    yo = o(x)
with torch.autocast("cuda", dtype=torch.bfloat16):
    yg = m(x.cuda())
print("out", rel(yg, yo))
c = 64
for j in range(4):
    print("concat slice", j, rel(cap["p"][:, j * c:(j + 1) * c], cap["o"][:, j * c:(j + 1) * c]))
for k in sorted(cap):
    if k.startswith("o.") and ("p." + k[2:]) in cap:
        print(k[2:], rel(cap["p." + k[2:]], cap[k]))
