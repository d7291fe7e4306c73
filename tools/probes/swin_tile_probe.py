"""Development probe: one SwinBlock(256) forward + backward at the benchmark's shape (bs 32, 40x40 map) under the tile the
environment forces (YMI_IGEMM_TILE=bm,bn; unset: the default rule).  Prints ms per forward+backward (HIP events, 20 iterations)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from improving_yolov8_cbam_swinblock_amd.nn.modules import SwinBlock
from improving_yolov8_cbam_swinblock_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = SwinBlock(256, 8, 7).to(dev).train()
x = torch.randn(32, 256, 40, 40, device=dev)
xi = ops.to_internal(x, torch.bfloat16).requires_grad_(True)
g = torch.randn(32, 256, 40, 40, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
def it():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = m(xi)
    y.backward(g)
    for p in m.parameters():
        p.grad = None
    xi.grad = None
for _ in range(5):
    it()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(20):
    it()
e1.record()
torch.cuda.synchronize()
print(os.environ.get("YMI_IGEMM_TILE", "default"), round(e0.elapsed_time(e1) / 20, 3), "ms per SwinBlock fwd+bwd")
