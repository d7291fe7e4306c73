"""Benchmark of the hot path: images/sec, forward+backward, YOLOv8s-CBAM-Swin, bs=32 per GPU, 640x640.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = forward (bf16 autocast) + v8 detection loss + backward (+ RCCL gradient mean over ranks) +
grad-clip + SGD-nesterov update on one synthetic batch already resident in HBM.  Rank 0 prints ONE JSON
line.  `roofline` is measured live: HIP events on the launch stream around every launch of the MFMA GEMM
kernels during the timed steps (algorithmic FLOP / elapsed).  `cpu_baseline` times the CPU oracle (a port
of the reference's PyTorch-CPU path, oracle/) on a bounded sample, rank 0, N=1 only.
"""
import argparse
import ctypes
import json
import os
import sys
import time

# multi-process GPU work on this pool needs dmabuf IPC (set before the first HIP call)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GFLOP_PER_IMG_FWD_BWD = 103.87  # SURVEY.md section 8(d), measured on the reference with FlopCounterMode
PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)


def cpu_baseline(batch=4, iters=8):
    """reference's CPU path as restated by the oracle: fwd + v8 loss + bwd, fp32, all host cores."""
    from oracle.loss import v8DetectionLoss
    from oracle.tasks import DetectionModel

    # the GPU box gives one GPU's share of the host (16 cores); os.cpu_count() reports the whole machine
    threads = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    model = DetectionModel("yolov8s.yaml", ch=3, nc=1)
    model.train()
    crit = v8DetectionLoss(model)
    g = torch.Generator().manual_seed(1)
    img = torch.rand(batch, 3, 640, 640, generator=g)
    n = batch * 4
    tb = {
        "batch_idx": torch.arange(batch).repeat_interleave(4).float(),
        "cls": torch.zeros(n, 1),
        "bboxes": torch.cat((torch.rand(n, 2, generator=g) * 0.6 + 0.2, torch.rand(n, 2, generator=g) * 0.3 + 0.05), 1),
    }

    def step():
        loss, _ = crit(model(img), tb)
        loss.sum().backward()
        model.zero_grad(set_to_none=True)

    step()  # warm-up
    t0 = time.perf_counter()
    done = 0
    for _ in range(iters):
        step()
        done += 1
        if time.perf_counter() - t0 > 30.0:  # bounded sample
            break
    dt = time.perf_counter() - t0
    iters = done
    return {
        "value": round(batch * iters / dt, 3),
        "unit": "images/sec",
        "cores": threads,
        "kind": "port",
        "sample": f"oracle (CPU restatement of the reference path) fwd+loss+bwd fp32, bs={batch} 640x640, 1 warm-up + {iters} timed iterations",
    }


def traffic_bytes(family):
    """HBM bytes per launch of the kernel family from the committed PMC passes (tools/pmc_traffic.py); None if absent."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as fh:
            return json.load(fh)["families"][family]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--imgsz", type=int, default=640)
    ap.add_argument("--model", default="yolov8s.yaml")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--graph", type=int, default=1, help="1: replay the step as a HIP graph (several ranks: forward+backward graph, RCCL mean + update eager), 0: eager")
    args = ap.parse_args()

    from improving_yolov8_cbam_swinblock_amd import _lib
    from improving_yolov8_cbam_swinblock_amd.engine import ddp
    from improving_yolov8_cbam_swinblock_amd.engine.trainer import TrainStep, synthetic_batch
    from improving_yolov8_cbam_swinblock_amd.nn.tasks import DetectionModel

    rank, local, world = ddp.setup()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    torch.manual_seed(0)
    model = DetectionModel(args.model, ch=3, nc=1).to(dev)
    ddp.broadcast_parameters(model)
    use_graph = bool(args.graph)
    step = TrainStep(model, world_size=world, graph=use_graph)
    batch = synthetic_batch(args.batch, args.imgsz, dev, ddp.shard_seed(1, rank))

    for _ in range(args.warmup):
        step(batch)
    lib = _lib.lib()
    timing = not args.no_kernel_timing
    instrument_inline = timing and not use_graph  # HIP events cannot be recorded inside a replayed graph
    if instrument_inline:
        _lib.check(lib.ymi_profile_begin(args.steps * 1024), "profile_begin")
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        items = step(batch)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    roof = None
    if timing:
        prof_steps = args.steps
        if not instrument_inline:
            # graph mode: the same kernels, launched eagerly with HIP events around each MFMA-GEMM launch,
            # measured right after the timed region (the graph replays them without host-visible boundaries)
            prof_steps = min(args.steps, 5)
            _lib.check(lib.ymi_profile_begin(prof_steps * 1024), "profile_begin")
            for _ in range(prof_steps):
                step.eager_step(batch)
        ms = (ctypes.c_double * 2)()
        fl = (ctypes.c_double * 2)()
        cnt = (ctypes.c_int64 * 2)()
        byt = (ctypes.c_double * 2)()
        bnd = (ctypes.c_double * 2)()
        _lib.check(lib.ymi_profile_end_ex(ms, fl, cnt, byt, bnd), "profile_end_ex")
        fam = 0 if ms[0] >= ms[1] else 1
        names = ["igemm_kernel (implicit-GEMM conv fwd / dgrad / token GEMM)", "wgrad_kernel (weight-gradient split-K GEMM)"]
        ach = fl[fam] / (ms[fam] * 1e-3) / 1e12 if ms[fam] > 0 else 0.0
        roof = {
            "bound": "mfma",
            "kernel": names[fam],
            "achieved": round(ach, 2),
            "peak": PEAK_BF16_TFLOPS,
            "unit": "TFLOP/s",
            "frac": round(ach / PEAK_BF16_TFLOPS, 4),
            "traffic": traffic_bytes(["igemm", "wgrad"][fam]),
            "traffic_unit": "HBM bytes per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, profiles/r01_pmc_traffic.json)",
            "algorithmic_bytes_per_launch": round(byt[fam] / max(cnt[fam], 1)),
            # every launch against ITS OWN roofline, max(flop / 2.5 PF, algorithmic bytes / 8 TB/s): the 1x1 and
            # narrow convs of this model are HBM-bound, so the family's MFMA fraction alone understates them
            "frac_of_per_launch_bounds": round(bnd[fam] / max(ms[fam], 1e-9), 4),
            "launches_per_step": cnt[fam] // max(prof_steps, 1),
            "avg_launch_us": round(ms[fam] * 1e3 / max(cnt[fam], 1), 2),
            "measured_over": "the timed steps" if instrument_inline else f"{prof_steps} eager steps right after the timed (graph-replayed) steps",
            "families": {
                "igemm": {"ms_per_step": round(ms[0] / prof_steps, 3), "tflops": round(fl[0] / max(ms[0], 1e-9) / 1e9, 2),
                          "frac_of_per_launch_bounds": round(bnd[0] / max(ms[0], 1e-9), 4)},
                "wgrad": {"ms_per_step": round(ms[1] / prof_steps, 3), "tflops": round(fl[1] / max(ms[1], 1e-9) / 1e9, 2),
                          "frac_of_per_launch_bounds": round(bnd[1] / max(ms[1], 1e-9), 4)},
            },
        }
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    value = args.gpus * args.batch * args.steps / dt
    if rank == 0:
        out = {
            # BASELINE.json's metric on its configuration; other --model / --batch / --imgsz values are named as they are
            "metric": (f"images/sec fwd+bwd YOLOv8s-CBAM-Swin bs={args.batch} {args.imgsz}x{args.imgsz}" if args.model == "yolov8s.yaml"
                       else f"images/sec fwd+bwd {args.model} bs={args.batch} {args.imgsz}x{args.imgsz}"),
            "value": round(value, 2),
            "unit": "images/sec",
            "n_gpus": args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "config": {
                "workload": f"{args.model} (CBAM + 2x SwinBlock + SPPF5 + SPPF7, nc=1) forward + v8 loss + backward + SGD step, bs={args.batch}/GPU {args.imgsz}x{args.imgsz}",
                "global_batch": args.batch * args.gpus,
                "imgsz": args.imgsz,
                "parallelism": f"dp{args.gpus}",
            },
            "hip_graph": bool(use_graph),
            "model_tflops": round(value * GFLOP_PER_IMG_FWD_BWD / 1e3, 2),
            "model_mfma_frac": round(value * GFLOP_PER_IMG_FWD_BWD / 1e3 / (PEAK_BF16_TFLOPS * args.gpus), 4),
            "loss_items": [round(float(v), 4) for v in items],
        }
        if roof:
            out["roofline"] = roof
        if args.gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
