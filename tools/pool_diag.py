"""Development probe: SPPF pool cascade forward/backward vs chained torch max_pool2d on the GPU (f32/bf16, k=5/7)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from improving_yolov8_cbam_swinblock_amd import ops
dev = torch.device("cuda:0")
F = torch.nn.functional
for dt in (torch.float32, torch.bfloat16):
    for (n, c, h, w) in ((4, 256, 20, 20), (2, 64, 9, 11), (4, 256, 40, 40), (32, 256, 20, 20), (1, 288, 40, 40)):
        for k in (5, 7):
            for dist in ("randn", "silu"):
                torch.manual_seed(1)
                x = torch.randn(n, c, h, w, device=dev)
                if dist == "silu":
                    x = F.silu(x * 2)
                x = x.to(dt)
                xi = ops.to_internal(x.float(), dt).detach().requires_grad_(True)
                cat = ops.sppf_pool_cat(xi, k)
                g = torch.randn(n, 4 * c, h, w, device=dev).to(dt)
                gi = ops.to_internal(g.float(), dt)
                (dx,) = torch.autograd.grad(cat, xi, gi)
                xr = x.float().requires_grad_(True)
                y1 = F.max_pool2d(xr, k, 1, k // 2); y2 = F.max_pool2d(y1, k, 1, k // 2); y3 = F.max_pool2d(y2, k, 1, k // 2)
                ref = torch.cat((xr, y1, y2, y3), 1)
                (dr,) = torch.autograd.grad(ref, xr, g.float())
                fe = float((cat.float() - ref).abs().max())
                be = float((dx.float() - dr).norm() / dr.norm())
                nbad = int(((dx.float() - dr).abs() > 1e-2 * dr.abs().max()).sum())
                print(f"{str(dt):15s} {(n,c,h,w)} k={k} {dist:5s} fwd max-abs {fe:.2e}  bwd rel {be:.3e}  bad elems {nbad}", flush=True)
