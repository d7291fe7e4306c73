"""Per-shape timing of the HBM-bound BatchNorm kernels (scale_shift_act; bn_act_bwd = reduce + final + apply) on the
conv-output shapes of YOLOv8s-CBAM-Swin at bs=32 640x640, against a plain device copy of the same bytes.
Development tool.   python tools/bn_bench.py [--iters 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from improving_yolov8_cbam_swinblock_amd import _lib, ops  # noqa: E402
from improving_yolov8_cbam_swinblock_amd.ops import _byref, as_ymi, check, ptr, stream_ptr, workspace  # noqa: E402

# (C, H, count) : conv outputs of the model by shape
SHAPES = [(32, 320, 1), (64, 160, 3), (32, 160, 2), (128, 80, 8), (64, 80, 7), (256, 40, 8), (128, 40, 7), (64, 40, 2), (512, 20, 5), (256, 20, 6),
          (128, 20, 2), (64, 20, 2)]


def timeit(fn, iters):
    for _ in range(3):
        fn()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    L = _lib.lib()
    dt = torch.bfloat16
    tot = {"ssa": 0.0, "bwd": 0.0, "copy": 0.0}
    print(f"{'C':>4s} {'H':>4s} {'MB':>7s} | {'ssa us':>8s} {'TB/s':>5s} | {'bn_bwd us':>9s} {'TB/s':>5s} | {'copy us':>8s} {'TB/s':>5s}")
    for c, h, count in SHAPES:
        raw = ops.empty_nhwc(args.batch, c, h, h, dt, dev)
        raw.copy_(torch.randn(args.batch, c, h, h, device=dev))
        out = torch.empty_like(raw)
        dout = torch.empty_like(raw)
        dout.copy_(torch.randn(args.batch, c, h, h, device=dev))
        draw = torch.empty_like(raw)
        scale = torch.rand(c, device=dev) + 0.5
        shift = torch.randn(c, device=dev)
        mean = torch.zeros(c, device=dev)
        inv = torch.ones(c, device=dev)
        dg = torch.empty(c, device=dev)
        db = torch.empty(c, device=dev)
        ws = workspace(2048 * 2 * c * 4 + 256, dev, "bnbwd")
        mb = raw.numel() * 2 / 1e6

        def ssa():
            check(L.ymi_scale_shift_act(_byref(as_ymi(raw)), ptr(scale), ptr(shift), 1, None, _byref(as_ymi(out)), stream_ptr()), "ssa")

        def bwd():
            check(L.ymi_bn_act_bwd(_byref(as_ymi(dout)), _byref(as_ymi(raw)), ptr(scale), ptr(mean), ptr(inv), ptr(shift), 1, _byref(as_ymi(draw)),
                                   ptr(dg), ptr(db), ptr(ws), ws.numel(), stream_ptr()), "bn_act_bwd")

        def cp():
            out.copy_(raw)

        t_ssa, t_bwd, t_cp = timeit(ssa, args.iters), timeit(bwd, args.iters), timeit(cp, args.iters)
        tot["ssa"] += t_ssa * count
        tot["bwd"] += t_bwd * count
        tot["copy"] += t_cp * count
        print(f"{c:4d} {h:4d} {mb:7.1f} | {t_ssa:8.1f} {2 * mb / t_ssa:5.2f} | {t_bwd:9.1f} {5 * mb / t_bwd:5.2f} | {t_cp:8.1f} {2 * mb / t_cp:5.2f}")
    print("model totals (us per step):", {k: round(v) for k, v in tot.items()})


if __name__ == "__main__":
    main()
