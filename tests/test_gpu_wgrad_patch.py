"""The weight gradient's two pixel orders (csrc/wgrad.hip): PATCH order with scalar addressing and out-of-range lanes as padding (round 5)
against the raster walk and against a float64 sum of the same bfloat16 operands - every map geometry the choice depends on: patch widths
32 / 16 / 8, maps that do not tile (20 x 20: raster fallback), non-square maps, stride 2, 1 x 1 and 3 x 3, channel counts below a tile.
Reference arithmetic: torch.nn.functional.conv2d's weight gradient (the reference's Conv.forward is nn.Conv2d, nn/modules/conv.py:69-79)."""
import pytest
import torch

from improving_yolov8_cbam_swinblock_amd import _lib, ops
from improving_yolov8_cbam_swinblock_amd.ops import weights as W

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp(min=1e-12))


CASES = [
    # n, cin, cout, k, stride, H, W (input)
    (4, 64, 64, 3, 1, 160, 160),   # 1 x 32 patches
    (8, 128, 128, 3, 1, 80, 80),   # 2 x 16
    (8, 128, 256, 3, 2, 80, 80),   # stride 2 -> 40 x 40 outputs: 4 x 8
    (16, 256, 128, 3, 1, 20, 20),  # does not tile: raster walk
    (4, 96, 64, 1, 1, 80, 48),     # non-square, 1 x 1, 48 = 3 x 16
    (2, 32, 32, 3, 1, 64, 96),     # 32-row tile, 96 = 3 x 32
    (3, 64, 128, 3, 2, 96, 64),    # stride 2 -> 48 x 32 outputs
    (2, 16, 24, 3, 1, 40, 40),     # channel counts below a tile
]


@pytest.mark.parametrize("n,cin,cout,k,s,h,w", CASES)
def test_patch_order_vs_raster_and_float64(n, cin, cout, k, s, h, w):
    torch.manual_seed(n * 1000 + cin + cout + h)
    x = torch.randn(n, cin, h, w, device=dev()).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
    dy = torch.randn(n, cout, ho, wo, device=dev()).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    ref = torch.nn.grad.conv2d_weight(x.double(), (cout, cin, k, k), dy.double(), stride=s, padding=k // 2)
    out = {}
    try:
        for mode in (1, 0):
            _lib.set_option("wgrad_patch", mode)
            dw, _ = W._wgrad(x, dy, cout, cin, k, s, False)
            torch.cuda.synchronize()
            out[mode] = dw.clone()
    finally:
        _lib.set_option("wgrad_patch", 1)
    e_patch, e_raster, e_pair = rel(out[1], ref), rel(out[0], ref), rel(out[1], out[0])
    print(f"\n[wgrad {cin}->{cout} k{k} s{s} {h}x{w} n{n}] patch {e_patch:.2e} raster {e_raster:.2e} patch vs raster {e_pair:.2e}")
    # bfloat16 first-level slabs put ~1.7e-3 on dW whatever the order (tests/test_gpu_bf16_matched.py); float32 slabs (few splits) ~1e-6
    assert e_patch <= 2.5e-3 and e_raster <= 2.5e-3 and e_pair <= 3.5e-3, (e_patch, e_raster, e_pair)


def test_patch_order_is_deterministic():
    torch.manual_seed(5)
    x = torch.randn(8, 128, 80, 80, device=dev()).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(8, 128, 80, 80, device=dev()).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    a, _ = W._wgrad(x, dy, 128, 128, 3, 1, False)
    a = a.clone()
    for _ in range(3):
        b, _ = W._wgrad(x, dy, 128, 128, 3, 1, False)
        assert torch.equal(a, b)
