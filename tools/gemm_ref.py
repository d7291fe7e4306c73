"""Development probe: what the vendor GEMM (torch.matmul -> hipBLASLt/rocBLAS) reaches on the GEMM shapes of this
model's layers (plain dense operands, no im2col): an upper reference for the implicit-GEMM kernel at the same M, N, K."""
import torch

dev = torch.device("cuda:0")
SHAPES = [("det.cv3[0] 3x3 128->128 @80", 204800, 128, 1152), ("c2f4.m 3x3 64->64 @80", 204800, 64, 576), ("L1 3x3 32->64 s2", 819200, 64, 288),
          ("L5 3x3 128->256 s2", 51200, 256, 1152), ("c2f6.m 3x3 128->128 @40", 51200, 128, 1152), ("c2f9.m 3x3 256->256 @20", 12800, 256, 2304),
          ("c2f2.cv1 1x1 64->64 @160", 819200, 64, 64), ("c2f4.cv2 1x1 256->128 @80", 204800, 128, 256), ("swin.fc1 256->1024", 56448, 1024, 256),
          ("swin.fc2 1024->256", 56448, 256, 1024), ("square 8192", 8192, 8192, 8192)]
for name, m, n, k in SHAPES:
    a = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    b = torch.randn(n, k, device=dev, dtype=torch.bfloat16)
    f = lambda: torch.matmul(a, b.t())
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"{name:32s} M={m:7d} N={n:5d} K={k:5d}  {us:8.1f} us  {2.0 * m * n * k / us * 1e-6:7.1f} TF")

print("weight-gradient shapes of 1x1 convs / token GEMMs: dW[N_out, C_in] = dY^T [N_out, P] @ X [P, C_in]")
for name, p_, co, ci in [("c2f4.cv2 256->128 @80", 204800, 128, 256), ("c2f19.cv1 384->128 @80", 204800, 128, 384), ("c2f2.cv2 96->64 @160", 819200, 64, 96),
                         ("swin.fc1 256->1024", 56448, 1024, 256), ("sppf.cv2 1024->512 @20", 12800, 512, 1024)]:
    dy = torch.randn(p_, co, device=dev, dtype=torch.bfloat16)
    x = torch.randn(p_, ci, device=dev, dtype=torch.bfloat16)
    f = lambda: torch.matmul(dy.t(), x)
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"{name:32s} P={p_:7d} Cout={co:5d} Cin={ci:5d}  {us:8.1f} us  {2.0 * p_ * co * ci / us * 1e-6:7.1f} TF")
