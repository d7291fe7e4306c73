"""Benchmark of the hot path: images/sec, forward+backward, YOLOv8s-CBAM-Swin, bs=32 per GPU, 640x640.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N ...            # N > 1 without torchrun's environment: launches itself (see self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = forward (bf16 autocast) + v8 detection loss + backward (+ RCCL gradient mean over ranks) + grad-clip +
SGD-nesterov update + EMA on one synthetic batch already resident in HBM.  Rank 0 prints ONE JSON line.
  roofline      measured live: HIP events on the launch stream around every launch of the MFMA GEMM kernels
                (algorithmic FLOP / elapsed);
  forward       north_star's target number: the train-mode forward alone (graph-replayed), ms and MFMA fraction;
  cpu_baseline  the CPU oracle (a port of the reference's PyTorch-CPU path, oracle/) timed on the host cores on a
                bounded sample, rank 0, N=1 only (BASELINE.md section 3 protocol).
"""
import argparse
import ctypes
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

# multi-process GPU work on this pool needs dmabuf IPC (set before the first HIP call)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# GFLOP per image (2 x MAC over conv / linear / bmm), measured on the reference with FlopCounterMode: SURVEY.md section 8(d) /
# BASELINE.md section 2.  (model yaml) -> (image size it was measured at, forward, forward + backward); convolutions and
# window attention both scale with the pixel count, so other sizes scale by (imgsz / size)^2.
GFLOP_TABLE = {
    "yolov8s.yaml": (640, 34.68, 103.87),                   # configs 3 / 4
    "yolov8m-cbam-swin384.yaml": (1280, 368.38, 1104.08),   # config 5
    "yolov8n-cbam.yaml": (640, 8.08, 24.16),                # config 2
    "yolov8n-stock.yaml": (640, 8.74, 26.14),               # config 1
}


def gflop_per_img(model, imgsz):
    """-> (forward, forward + backward) GFLOP per image, or (None, None) for a model without a measured figure."""
    if model not in GFLOP_TABLE:
        return None, None
    size, fwd, both = GFLOP_TABLE[model]
    k = (imgsz / size) ** 2
    return fwd * k, both * k


PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
HOST_SHARE = 16  # host cores that belong to one GPU of the box (the pool's rule for worker sizing)


def self_launch(args):
    """`bench.py --gpus N` started by hand (no torchrun environment): start N ranks as CHILD processes through
    torch.distributed.run - before this process has made any GPU call, never by re-executing it - relay their output
    (rank 0 prints the JSON line) and return their exit code.  Reference launcher: ultralytics/utils/dist.py:78-98."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "4"))
    return subprocess.run(cmd, env=env).returncode


def host_info():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return model, avail


def cpu_baseline():
    """reference's CPU path as restated by the oracle, BASELINE.md section 3 protocol: forward-only (train-mode BN,
    no_grad) at bs=32 and forward + v8 loss + backward at bs=8, fp32 NCHW, 1 warm-up + up to 3 timed iterations each
    (each leg stops early after 20 s of timed work), same synthetic batch recipe as the GPU run."""
    from improving_yolov8_cbam_swinblock_amd.engine.trainer import synthetic_batch
    from oracle.loss import v8DetectionLoss
    from oracle.tasks import DetectionModel

    cpu_model, avail = host_info()
    threads = min(HOST_SHARE, avail)  # one GPU's share of the host; the machine's other cores belong to other boxes
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    model = DetectionModel("yolov8s.yaml", ch=3, nc=1)
    model.train()
    crit = v8DetectionLoss(model)

    def timed(fn, iters=3, budget=20.0):
        fn()  # warm-up
        t0 = time.perf_counter()
        done = 0
        for _ in range(iters):
            fn()
            done += 1
            if time.perf_counter() - t0 > budget:
                break
        return done, time.perf_counter() - t0

    b32 = synthetic_batch(32, 640, torch.device("cpu"), 1)
    b8 = synthetic_batch(8, 640, torch.device("cpu"), 1)

    def fwd():
        with torch.no_grad():
            model(b32["img"])

    def fwd_bwd():
        loss, _ = crit(model(b8["img"]), b8)
        loss.sum().backward()
        model.zero_grad(set_to_none=True)

    nf, tf = timed(fwd)
    nb, tb = timed(fwd_bwd)
    return {
        "value": round(8 * nb / tb, 3),
        "unit": "images/sec",
        "cores": threads,
        "kind": "port",
        "cpu_model": cpu_model,
        "logical_cpus_visible": avail,
        "forward_only": {"value": round(32 * nf / tf, 3), "unit": "images/sec", "batch": 32, "iterations": nf},
        "sample": (f"oracle (CPU restatement of the reference path, fp32 NCHW) on {threads} threads of '{cpu_model}': value = forward + v8 loss + "
                   f"backward at bs=8 640x640 (1 warm-up + {nb} timed); forward_only = train-mode forward, no_grad, bs=32 (1 warm-up + {nf} timed)"),
    }


def kernel_rev():
    """hash of the kernel sources: PMC traffic measured on other kernels is not reported for these."""
    h = hashlib.sha1()
    d = os.path.join(ROOT, "improving_yolov8_cbam_swinblock_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(d, name), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:12]


def traffic_record(family, workload):
    """HBM bytes per launch from the latest committed PMC passes (tools/pmc_traffic.py) IF they were taken on the
    kernels this run executes (same source hash) AND on this workload (model, batch, image size; files without a
    "workload" entry are the default configuration's); otherwise None: hardware counters cannot be read from inside the run."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for name in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if name.endswith("pmc_traffic.json"):
            try:
                with open(os.path.join(pdir, name)) as fh:
                    d = json.load(fh)
                if d.get("kernel_rev") == kernel_rev() and d.get("workload", "yolov8s.yaml bs32 640") == workload:
                    best = (d["families"][family]["hbm_bytes_per_launch"], name)
            except (OSError, KeyError, ValueError):
                continue
    return best


def forward_record(model, batch, steps, gf_fwd):
    """north_star's target metric: the train-mode forward of the step alone (bf16 autocast, batch statistics, tensors
    saved for backward, weight pack included), replayed as a HIP graph; 1.110 TFLOP per batch of 32."""
    model.train()

    def fwd():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return model(batch["img"])

    for _ in range(2):
        fwd()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fwd()
    del out
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    del g
    n = batch["img"].shape[0]
    tf = n * gf_fwd / 1e3 / (ms * 1e-3)
    return {"ms": round(ms, 3), "images_per_sec": round(n / (ms * 1e-3), 1), "tflops": round(tf, 1), "mfma_frac": round(tf / PEAK_BF16_TFLOPS, 4),
            "target_mfma_frac": 0.40, "what": f"train-mode forward (Detect maps), bs={n}, bf16, HIP-graph replay, {steps} timed replays"}


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
    ap.add_argument("--no-forward", action="store_true", help="skip the forward-only sub-record")
    ap.add_argument("--graph", type=int, default=1, help="1: replay the step as a HIP graph (several ranks: forward+backward graph, RCCL mean + update eager), 0: eager")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend of the gradient mean: nccl (= RCCL over xGMI, the measured configuration) or gloo "
                         "(rehearsal of the multi-rank path on fewer GPUs than ranks: ranks then share devices round-robin)")
    ap.add_argument("--schedule", choices=["auto", "split", "tail"], default="auto",
                    help="graph schedule: auto = one graph on one rank, three graphs with the bucket exchanges between them on several; split = "
                         "the several-rank schedule also on one rank (what it costs without the collectives); tail = round 3's several-rank form")
    ap.add_argument("--sustained", type=int, default=850,
                    help="graph replays of the sustained-throughput sub-record (0: skip); the default keeps the device busy for >= 10 s, long enough "
                         "for an outside 5-second GPU-busy sampler to see it")
    ap.add_argument("--forward-only", action="store_true",
                    help="profiling aid: run ONLY the train-mode forward (north_star's target metric) - warm-up + `steps` graph replays - and print "
                         "its record; under rocprofv3 --kernel-trace --stats this gives the forward's own kernel table")
    ap.add_argument("--hook", action="append", default=[], metavar="NAME=VALUE",
                    help="development aid for same-box A/B runs (tools/r5_ab.sh): set a python hook (ops.HOOKS: fused_swin_mlp, detect_pair, detect_multi, "
                         "first_conv) or a library option (ymi_set_option: bn_tail, ew_cap, ...) before the model is built; the default command sets none")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))  # no GPU call has happened in this process

    from improving_yolov8_cbam_swinblock_amd import _lib
    from improving_yolov8_cbam_swinblock_amd.engine import ddp
    from improving_yolov8_cbam_swinblock_amd.engine.trainer import TrainStep, synthetic_batch
    from improving_yolov8_cbam_swinblock_amd.nn.tasks import DetectionModel

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    for kv in args.hook:
        from improving_yolov8_cbam_swinblock_amd import ops

        name, _, val = kv.partition("=")
        if name in ops.HOOKS:
            ops.HOOKS[name] = bool(int(val))
        else:
            _lib.set_option(name, int(val))
    ndev = torch.cuda.device_count()
    local_env = int(os.environ.get("LOCAL_RANK", "0"))
    if args.backend == "nccl" and local_env >= ndev:
        raise SystemExit(f"--backend nccl needs one GPU per rank (LOCAL_RANK {local_env}, {ndev} visible): RCCL refuses ranks that share a device; "
                         "use --backend gloo to rehearse the multi-rank path on fewer GPUs")
    device_index = local_env % ndev  # gloo rehearsal: ranks share the visible devices round-robin
    torch.cuda.set_device(device_index)
    rank, local, world = ddp.setup(args.backend, device_index=device_index)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    dev = torch.device("cuda", device_index)

    torch.manual_seed(0)
    model = DetectionModel(args.model, ch=3, nc=1).to(dev)
    ddp.broadcast_parameters(model)
    use_graph = bool(args.graph)
    batch = synthetic_batch(args.batch, args.imgsz, dev, ddp.shard_seed(1, rank))
    if args.forward_only:
        gf_fwd, _ = gflop_per_img(args.model, args.imgsz)
        rec = forward_record(model, batch, args.steps, gf_fwd or 0.0)
        rec["model"] = args.model
        print(json.dumps({"forward": rec}), flush=True)
        return
    step = TrainStep(model, world_size=world, graph=(args.schedule if (use_graph and args.schedule != "auto") else use_graph))

    for _ in range(args.warmup):
        step(batch)
    comm = None
    if world > 1:
        # what the exchange layer itself reports, so that a several-rank line explains itself: the world size as the process group sees it, an
        # all-reduce of ones (= the number of ranks that took part), the flat buckets' sizes; the exposed exchange time is added after the timed steps
        ones = torch.ones(1, dtype=torch.float32, device=dev)
        torch.distributed.all_reduce(ones)
        comm = {"backend": torch.distributed.get_backend(), "world_size": torch.distributed.get_world_size(), "allreduce_of_ones": float(ones.item()),
                "bucket_bytes": [int(sum(p.numel() for p in b) * 4) for b in step.buckets.buckets]}
        step.time_exposed_communication(True)
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
    if comm is not None:
        comm["exposed_ms_per_step"] = round(step.exposed_communication_ms() or 0.0, 4)
        comm["exposed_is"] = ("compute-stream time between the end of the backbone's backward graph and the end of the wait for both buckets" if step.overlap_graphs
                              else "the whole gradient exchange (nothing overlaps it in this schedule)")
        step.time_exposed_communication(False)
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
        tr = traffic_record(["igemm", "wgrad"][fam], f"{args.model} bs{args.batch} {args.imgsz}")
        roof = {
            "bound": "mfma",
            "kernel": names[fam],
            "achieved": round(ach, 2),
            "peak": PEAK_BF16_TFLOPS,
            "unit": "TFLOP/s",
            "frac": round(ach / PEAK_BF16_TFLOPS, 4),
            # hardware counters cannot be sampled from inside this process: the figure comes from separate rocprofv3 --pmc
            # passes of this same command (tools/pmc_traffic.py) and is reported only when those passes ran the kernels of
            # this source revision
            "traffic": tr[0] if tr else None,
            "traffic_source": (f"profiles/{tr[1]} (HBM bytes per launch, 2*FETCH_SIZE + WRITE_SIZE KiB, kernel_rev {kernel_rev()})" if tr
                               else f"no PMC pass committed for kernel_rev {kernel_rev()} on this workload"),
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
    sustained = None
    if use_graph and args.sustained > 0:
        # the same step, replayed back to back for seconds instead of the K timed steps (0.3 s at the defaults): long enough for an
        # external sampler (rocm-smi, the driver's GPU-busy probe) to see the device busy and for the clock to settle; same formula
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for _ in range(args.sustained):
            step(batch)
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        dts = time.perf_counter() - ts
        if world > 1:
            tm = torch.tensor([dts], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(tm, op=torch.distributed.ReduceOp.MAX)
            dts = float(tm.item())
        sustained = {"steps": args.sustained, "seconds": round(dts, 3), "ms_per_step": round(dts / args.sustained * 1e3, 3),
                     "images_per_sec": round(args.gpus * args.batch * args.sustained / dts, 2)}
    fwd = None
    gf_fwd, gf_both = gflop_per_img(args.model, args.imgsz)
    if rank == 0 and not args.no_forward and gf_fwd is not None:
        fwd = forward_record(model, batch, max(args.steps, 10), gf_fwd)
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
                "workload": f"{args.model} (nc=1) forward + v8 loss + backward + clip + SGD-nesterov + EMA step, bs={args.batch}/GPU {args.imgsz}x{args.imgsz}"
                            + (" (CBAM + 2x SwinBlock + SPPF5 + SPPF7)" if args.model in ("yolov8s.yaml", "yolov8m-cbam-swin384.yaml") else ""),
                "global_batch": args.batch * args.gpus,
                "imgsz": args.imgsz,
                "parallelism": f"dp{args.gpus}",
                "backend": ("rccl" if args.backend == "nccl" else "gloo") if args.gpus > 1 else None,
            },
            "hip_graph": bool(use_graph),
            "schedule": ("one graph" if step.full_graph else "three graphs, bucket exchanges between them" if step.overlap_graphs else
                         "forward+backward graph, eager reduction and update" if use_graph else "eager"),
            "gflop_per_image": round(gf_both, 2) if gf_both is not None else None,
            "model_tflops": round(value * gf_both / 1e3, 2) if gf_both is not None else None,
            "model_mfma_frac": round(value * gf_both / 1e3 / (PEAK_BF16_TFLOPS * args.gpus), 4) if gf_both is not None else None,
            "loss_items": [round(float(v), 4) for v in items],
        }
        if comm:
            out["communication"] = comm
        if sustained:
            out["sustained"] = sustained
        if roof:
            out["roofline"] = roof
        if fwd:
            out["forward"] = fwd
        if args.gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
