"""Per-layer timing of the three MFMA GEMM kernels on the conv shapes of YOLOv8s-CBAM-Swin at bs=32, 640x640
(and the Swin token GEMMs).  Development tool: prints one line per shape with TFLOP/s for forward (raw+stats),
data gradient and weight gradient.   python tools/conv_bench.py [--dtype bf16|f32] [--iters 20]"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from improving_yolov8_cbam_swinblock_amd import _lib, ops  # noqa: E402

# (name, cin, cout, k, stride, H_in) at bs=32 ; count = occurrences in the model
SHAPES = [
    ("L0 3->32 s2", 8, 32, 3, 2, 640, 1),
    ("L1 32->64 s2", 32, 64, 3, 2, 320, 1),
    ("c2f2.cv1 64->64 1x1", 64, 64, 1, 1, 160, 1),
    ("c2f2.m 32->32 3x3", 32, 32, 3, 1, 160, 2),
    ("c2f2.cv2 96->64 1x1", 96, 64, 1, 1, 160, 1),
    ("L3 64->128 s2", 64, 128, 3, 2, 160, 1),
    ("c2f4.cv1 128->128 1x1", 128, 128, 1, 1, 80, 1),
    ("c2f4.m 64->64 3x3", 64, 64, 3, 1, 80, 4),
    ("c2f4.cv2 256->128 1x1", 256, 128, 1, 1, 80, 1),
    ("L5 128->256 s2", 128, 256, 3, 2, 80, 1),
    ("c2f6.cv1 256->256 1x1", 256, 256, 1, 1, 40, 1),
    ("c2f6.m 128->128 3x3", 128, 128, 3, 1, 40, 4),
    ("c2f6.cv2 512->256 1x1", 512, 256, 1, 1, 40, 1),
    ("L8 256->512 s2", 256, 512, 3, 2, 40, 1),
    ("c2f9.cv1 512->512 1x1", 512, 512, 1, 1, 20, 1),
    ("c2f9.m 256->256 3x3", 256, 256, 3, 1, 20, 2),
    ("c2f9.cv2 768->512 1x1", 768, 512, 1, 1, 20, 1),
    ("sppf.cv1 512->256 1x1", 512, 256, 1, 1, 20, 2),
    ("sppf.cv2 1024->512 1x1", 1024, 512, 1, 1, 20, 2),
    ("c2f15.cv1 768->256 1x1", 768, 256, 1, 1, 40, 1),
    ("c2f19.cv1 384->128 1x1", 384, 128, 1, 1, 80, 1),
    ("c2f19.cv2 192->128 1x1", 192, 128, 1, 1, 80, 1),
    ("L20 128->128 s2", 128, 128, 3, 2, 80, 1),
    ("L23 256->256 s2", 256, 256, 3, 2, 40, 1),
    ("det.cv2[0] 128->64 3x3", 128, 64, 3, 1, 80, 1),
    ("det.cv3[0] 128->128 3x3", 128, 128, 3, 1, 80, 2),
    ("det.cv2[1] 256->64 3x3", 256, 64, 3, 1, 40, 1),
    ("det.cv3[1] 256->128 3x3", 256, 128, 3, 1, 40, 1),
    ("det.cv2[2] 512->64 3x3", 512, 64, 3, 1, 20, 1),
    ("det.cv3[2] 512->128 3x3", 512, 128, 3, 1, 20, 1),
    ("swin.qkv 256->768 tok", 256, 768, 1, 1, -56448, 2),
    ("swin.fc1 256->1024 tok", 256, 1024, 1, 1, -56448, 2),
    ("swin.fc2 1024->256 tok", 1024, 256, 1, 1, -56448, 2),
]


def stamp_dump(L, fn, title, wgrad=False):
    """one more launch with the stamp buffer armed; prints, per wave, cycles between the 4 marks of K steps 2..9."""
    buf = torch.zeros(8 * 64 + 16 + 64, dtype=torch.int64, device="cuda:0")
    setter = L.ymi_debug_stamp_buffer_wgrad if wgrad else L.ymi_debug_stamp_buffer
    setter.argtypes = [ctypes.c_void_p]
    setter(ctypes.c_void_p(buf.data_ptr()))
    fn()
    torch.cuda.synchronize()
    setter(ctypes.c_void_p(0))
    host = buf.cpu()
    st = host[: 8 * 64].view(8, 64)
    cal = host[8 * 64: 8 * 64 + 16].view(8, 2)
    marks = host[8 * 64 + 16:].view(8, 8)
    for w in range(8):
        if int(marks[w, 2]) > 0:
            if wgrad:
                print(f"wave {w}: {int(marks[w, 6])} K steps; cycles from kernel entry: prologue done {int(marks[w, 0])}, K loop done {int(marks[w, 1])}, stores issued {int(marks[w, 5])}, stores retired {int(marks[w, 2])}")
                continue
            print(f"wave {w}: cycles from kernel entry: prologue done {int(marks[w, 0])}, K loop done {int(marks[w, 1])}, tile in LDS {int(marks[w, 3])}, past barrier {int(marks[w, 4])}, stores issued {int(marks[w, 5])}, stores retired {int(marks[w, 2])}")
    for w in range(8):
        if int(cal[w, 1]) > 0:
            print(f"wave {w}: K loop {int(cal[w, 0])} shader cycles in {int(cal[w, 1]) * 10} ns -> in-kernel clock {int(cal[w, 0]) / (int(cal[w, 1]) * 10.0):.2f} GHz")
    print(f"--- stamps: {title}  (s_memtime ticks; marks 0->1->2->3->next 0)")
    t0 = min(int(v) for v in st[:, 0] if int(v) > 0) if (st[:, 0] > 0).any() else 0
    for w in range(8):
        row = [int(v) for v in st[w] if int(v) > 0]
        if not row:
            continue
        d = [row[i + 1] - row[i] for i in range(len(row) - 1)]
        steps = [d[i:i + 4] for i in range(0, len(d) - 3, 4)]
        print(f"wave {w}: first mark at +{row[0] - t0}; per step [0->1, 1->2, 2->3, 3->next]:", " ".join(str(x) for x in steps[:8]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--only", default="", help="substring filter on the shape name")
    ap.add_argument("--ops", default="fwd,dgrad,wgrad")
    ap.add_argument("--shape", action="append", default=[], help="extra shape 'name,cin,cout,k,stride,H' (square map; replaces the model's list)")
    ap.add_argument("--cold", action="store_true", help="evict the caches (1 GiB fill) before every timed launch: per-launch events, as a layer meets its operands inside the step")
    ap.add_argument("--producer", action="store_true", help="forward only: the BatchNorm affine + SiLU pass that WRITES the GEMM's input runs before every timed launch, as in "
                    "the step (chains of pass -> GEMM; columns: the pass alone, the pair, the GEMM's share); with YMI_XCD_SHIFT=k the pass works on another XCD's eighth")
    ap.add_argument("--stamps", action="store_true", help="diagnostic build (-DYMI_STAMPS) only: print the s_memtime stamps of one workgroup's K steps")
    args = ap.parse_args()
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    dev = torch.device("cuda:0")
    L = _lib.lib()
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    print(f"{'shape':28s} {'M':>9s} {'GF':>7s} | {'fwd us':>8s} {'TF':>6s} | {'dgrad us':>8s} {'TF':>6s} | {'wgrad us':>8s} {'TF':>6s}")
    shapes = SHAPES
    if args.shape:
        shapes = []
        for spec in args.shape:
            nm, cin_, cout_, k_, s_, h_ = spec.split(",")
            shapes.append((nm, int(cin_), int(cout_), int(k_), int(s_), int(h_), 1))
    for name, cin, cout, k, s, h, count in shapes:
        if args.only and args.only not in name:
            continue
        w = torch.randn(cout, cin, k, k, device=dev) * 0.05
        if h < 0:
            x = torch.randn(-h, cin, device=dev).to(dt)
            M = -h
            y = torch.empty(M, cout, dtype=dt, device=dev)
        else:
            x = ops.empty_nhwc(args.batch, cin, h, h, dt, dev)
            x.copy_(torch.randn(args.batch, cin, h, h, device=dev))
            ho = (h + 2 * (k // 2) - k) // s + 1
            M = args.batch * ho * ho
            y = ops.empty_nhwc(args.batch, cout, ho, ho, dt, dev)
        wp = ops.pack_conv_fwd(w, cin, dt)
        wd = ops.pack_conv_dgrad(w, cout, s, dt)
        part = torch.empty(int(L.ymi_conv2d_stat_blocks(M, cout)) * 2 * cout, dtype=torch.float32, device=dev)
        dy = torch.randn_like(y.float()).to(dt) if h < 0 else ops.empty_nhwc(*y.shape, dt, dev).copy_(torch.randn(y.shape, device=dev))
        dx = torch.empty_like(x) if h < 0 else ops.empty_nhwc(*x.shape, dt, dev)
        dw = torch.empty(cout, cin, k, k, device=dev)
        ws = torch.empty(int(L.ymi_conv2d_bwd_weight_workspace(M, cout, cin, k, k)), dtype=torch.uint8, device=dev)
        nb = ctypes.c_int64(0)
        tx, ty, tdy, tdx = _lib.as_ymi(x), _lib.as_ymi(y), _lib.as_ymi(dy), _lib.as_ymi(dx)
        sp = _lib.stream_ptr()

        def fwd():
            _lib.check(L.ymi_conv2d_fwd(ctypes.byref(tx), _lib.ptr(wp), cout, k, k, s, None, None, 0, None, ctypes.byref(ty), _lib.ptr(part), ctypes.byref(nb), sp))

        def dgrad():
            _lib.check(L.ymi_conv2d_bwd_data(ctypes.byref(tdy), _lib.ptr(wd), cin, k, k, s, ctypes.byref(tdx), sp))

        def wgrad():
            _lib.check(L.ymi_conv2d_bwd_weight(ctypes.byref(tx), ctypes.byref(tdy), cout, cin, k, k, s, _lib.ptr(dw), None, _lib.ptr(ws), ws.numel(), sp))

        gf = 2.0 * M * cout * cin * k * k / 1e9
        line = f"{name:28s} {M:9d} {gf:7.1f} |"
        if args.producer:
            if h < 0:
                continue
            src = ops.empty_nhwc(*x.shape, dt, dev).copy_(torch.randn(x.shape, device=dev))
            sc = torch.ones(x.shape[1], device=dev)
            sh = torch.zeros(x.shape[1], device=dev)
            tsrc = _lib.as_ymi(src)

            def produce():
                _lib.check(L.ymi_scale_shift_act(ctypes.byref(tsrc), _lib.ptr(sc), _lib.ptr(sh), 1, None, ctypes.byref(tx), sp))

            def chain(fns):
                for f in fns:
                    f()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    for f in fns:
                        f()
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) * 1e3 / args.iters

            t_p, t_g, t_pg = chain([produce]), chain([fwd]), chain([produce, fwd])
            print(f"{name:28s} {M:9d} | pass {t_p:7.1f}  gemm warm {t_g:7.1f}  pair {t_pg:7.1f}  gemm in chain {t_pg - t_p:7.1f} us", flush=True)
            tot["fwd"] += (t_pg - t_p) * count
            tot["dgrad"] += t_g * count
            continue
        for nm, fn in (("fwd", fwd), ("dgrad", dgrad), ("wgrad", wgrad)):
            if nm not in args.ops.split(","):
                continue
            for _ in range(3):
                fn()
            if args.cold:
                if "evict" not in globals():
                    globals()["evict"] = torch.empty(1 << 28, dtype=torch.float32, device=dev)  # 1 GiB > L2 + Infinity Cache
                tot_ms = 0.0
                for _ in range(args.iters):
                    globals()["evict"].fill_(1.0)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    fn()
                    e1.record()
                    torch.cuda.synchronize()
                    tot_ms += e0.elapsed_time(e1)
                us = tot_ms * 1e3 / args.iters
            else:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / args.iters
            if args.stamps:
                stamp_dump(L, fn, f"{name} {nm} ({us:.1f} us)", nm == "wgrad")
            tot[nm] += us * count
            line += f" {us:8.1f} {gf / us * 1e-3 * 1e3 / 1e3 * 1e3:6.0f} |" if False else f" {us:8.1f} {gf / (us * 1e-6) / 1e3:6.0f} |"
        print(line, flush=True)
    print("model totals (us, weighted by occurrence):", {k: round(v) for k, v in tot.items()})


if __name__ == "__main__":
    main()
