"""GPU: full-size (BASELINE config 3 shapes) checks through size-independent properties, plus oracle
comparisons at the largest sizes the CPU oracle finishes in seconds."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    return float((a.float().cpu() - b.float().cpu()).norm() / b.float().cpu().norm().clamp(min=1e-9))


@pytest.mark.parametrize("cin,cout,k,s,hw", [(32, 64, 3, 2, 64), (64, 64, 3, 1, 40), (96, 64, 1, 1, 40), (256, 128, 1, 1, 20), (128, 256, 3, 2, 40),
                                             (64, 64, 3, 2, 21), (64, 128, 3, 2, 37),  # odd sizes: the stride-2 parity classes differ in size
                                             # >= 300 tiles of 256x128: the ping-pong form of igemm_kernel (bf16), forward and data gradient,
                                             # with a ragged last tile; 1x1; and the four parity classes of a stride-2 data gradient in one launch
                                             (128, 128, 3, 1, 141), (256, 256, 1, 1, 100), (128, 256, 3, 2, 200)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_conv_block_vs_oracle_midsize(cin, cout, k, s, hw, dtype):
    """multi-tile shapes (several M/N blocks, K loops over all taps) against the CPU oracle, fwd + all grads."""
    import oracle.modules as OM
    from improving_yolov8_cbam_swinblock_amd.nn.modules import Conv

    torch.manual_seed(cin + cout)
    o = OM.Conv(cin, cout, k, s)
    for b in o.modules():
        if isinstance(b, torch.nn.BatchNorm2d):
            b.eps, b.momentum = 1e-3, 0.03
            b.weight.data.uniform_(0.5, 1.5)
            b.bias.data.normal_(0, 0.3)
    m = Conv(cin, cout, k, s)
    for b in m.modules():
        if isinstance(b, torch.nn.BatchNorm2d):
            b.eps, b.momentum = 1e-3, 0.03
    m.load_state_dict(o.state_dict())
    m = m.to(dev()).train()
    o.train()
    x = torch.randn(4, cin, hw, hw)
    ho = (hw + 2 * (k // 2) - k) // s + 1
    gy = torch.randn(4, cout, ho, ho)
    xo = x.clone().requires_grad_(True)
    yo = o(xo)
    go = torch.autograd.grad(yo, [xo] + list(o.parameters()), gy)
    xg = x.to(dev()).requires_grad_(True)
    if dtype == torch.bfloat16:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            yg = m(xg)
    else:
        yg = m(xg)
    gg = torch.autograd.grad(yg, [xg] + list(m.parameters()), gy.to(dev()).to(yg.dtype))
    tol = 2e-4 if dtype == torch.float32 else 2e-2
    assert rel(yg, yo) < tol, ("fwd", rel(yg, yo))
    for a, b, n in zip(gg, go, ["x", "w", "gamma", "beta"]):
        assert rel(a, b) < tol * 3, (n, rel(a, b))
    if dtype == torch.float32:
        assert float((yg.float().cpu() - yo).abs().max()) < 1e-3 * max(1.0, float(yo.abs().max()))


def test_fullsize_forward_properties_bs32_640():
    """config 3 shapes: finite outputs, BatchNorm statistics consistent with the stored raw tensors, max-pool
    cascade idempotence/ordering, window partition round trip - properties that do not need a CPU reference."""
    from improving_yolov8_cbam_swinblock_amd import ops
    from improving_yolov8_cbam_swinblock_amd.nn.modules import SPPF, Conv, SwinBlock
    from improving_yolov8_cbam_swinblock_amd.nn.tasks import DetectionModel

    torch.manual_seed(0)
    model = DetectionModel("yolov8s.yaml", ch=3, nc=1).to(dev()).train()
    img = torch.rand(32, 3, 640, 640, device=dev())
    with torch.autocast("cuda", dtype=torch.bfloat16):
        preds = model(img)
    assert [tuple(p.shape) for p in preds] == [(32, 65, 80, 80), (32, 65, 40, 40), (32, 65, 20, 20)]
    assert all(torch.isfinite(p.float()).all() for p in preds)
    # BN train-mode output of a Conv has per-channel mean/var of the pre-activation equal to beta / gamma^2:
    conv = Conv(64, 128, 3, 2).to(dev()).train()
    conv.act = torch.nn.Identity()
    x = torch.randn(32, 64, 160, 160, device=dev())
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = conv(x).float()
    mean = y.mean(dim=(0, 2, 3))
    var = y.var(dim=(0, 2, 3), unbiased=False)
    assert float(mean.abs().max()) < 2e-2 and float((var - 1).abs().max()) < 3e-2  # gamma=1, beta=0 at init
    # SPPF pools: y1 <= y2 <= y3 elementwise, and pooling a constant-per-channel map is the identity
    from improving_yolov8_cbam_swinblock_amd.ops import sppf_pool_cat, to_internal

    y0 = to_internal(torch.randn(32, 256, 20, 20, device=dev()), torch.bfloat16)
    cat = sppf_pool_cat(y0, 5)
    a, b, c, d = cat[:, :256].float(), cat[:, 256:512].float(), cat[:, 512:768].float(), cat[:, 768:].float()
    assert torch.equal(a, y0.float()) and bool((b >= a).all()) and bool((c >= b).all()) and bool((d >= c).all())
    ref = torch.nn.functional.max_pool2d(torch.nn.functional.max_pool2d(torch.nn.functional.max_pool2d(y0.float(), 5, 1, 2), 5, 1, 2), 5, 1, 2)
    assert torch.equal(d, ref)  # max is exact: bit-identical to the chained reference pools
    # window partition / reverse round trip at the model's Swin shape [32,256,40,40] (pad to 42)
    xs = to_internal(torch.randn(32, 256, 40, 40, device=dev()), torch.bfloat16)
    tok = ops.window_partition(xs, 7)
    assert tok.shape == (32 * 36 * 49, 256)
    back = ops.window_reverse(tok, 32, 40, 40, 7)
    assert torch.equal(back, xs)


def test_fullsize_train_step_decreases_loss():
    from improving_yolov8_cbam_swinblock_amd.engine.trainer import TrainStep, synthetic_batch
    from improving_yolov8_cbam_swinblock_amd.nn.tasks import DetectionModel

    torch.manual_seed(0)
    model = DetectionModel("yolov8s.yaml", ch=3, nc=1).to(dev())
    step = TrainStep(model, world_size=1, lr=0.01)
    batch = synthetic_batch(8, 640, dev(), 1)
    first = step(batch)
    for _ in range(12):
        last = step(batch)
    assert torch.isfinite(last).all()
    assert float(last.sum()) < float(first.sum()), (first, last)


@pytest.mark.parametrize("bs,sz", [(4, 320), (16, 640)])
def test_hip_graph_step_matches_eager_and_is_isolated(bs, sz):
    """TrainStep(graph=True) replays exactly the kernels of the eager step: the same losses step by step AND the same
    parameters afterwards, with eager allocations, pinned-memory traffic and a second stream's work made between the
    replays (they must not disturb the graph's memory).  Graph mode runs 3 eager warm-up steps before its first replay,
    so replay i is eager step i + 3: 7 replays are compared with eager steps 3..9."""
    from improving_yolov8_cbam_swinblock_amd.engine.trainer import TrainStep, synthetic_batch
    from improving_yolov8_cbam_swinblock_amd.nn.tasks import DetectionModel

    losses, params = {}, {}
    probe = ("model.0.conv.weight", "model.7.attn.in_proj_weight", "model.10.sa.conv.weight", "model.22.cv2.bn.weight", "model.26.cv3.0.2.bias")
    # "split": the multi-rank schedule on one rank - three graphs (forward + head backward | backbone backward | update) with the
    # bucket exchanges between them; "tail": round 3's form (forward + backward graph, eager reduction / update behind it)
    for mode in ("eager", "graph", "split", "tail"):
        torch.manual_seed(0)
        model = DetectionModel("yolov8s.yaml", ch=3, nc=1).to(dev())
        step = TrainStep(model, world_size=1, lr=0.01, graph={"eager": False, "graph": True, "split": "split", "tail": "tail"}[mode])
        batch = synthetic_batch(bs, sz, dev(), 1)
        out = []
        for i in range(10 if mode == "eager" else 7):
            if mode != "eager" and i >= 1:  # eager work between replays: device allocations of many sizes, pinned staging, a side stream
                junk = [torch.full((n,), 7.0, device=dev()) for n in (1, 3, 17, 1000, 100000, 5000000)]
                junk.append(torch.randn(1000, 1000, device=dev()).sum())
                _ = float(junk[-1]) + float(junk[3].sum())
                pinned = torch.arange(1000.0).pin_memory().to(dev(), non_blocking=True)
                side = torch.cuda.Stream()
                with torch.cuda.stream(side):
                    junk.append(torch.ones(1 << 20, device=dev()).cumsum(0))
                torch.cuda.synchronize()
                del junk, pinned
            out.append(step(batch).float().cpu().clone())
        losses[mode] = torch.stack(out)
        sd = model.state_dict()
        params[mode] = {k: sd[k].detach().float().cpu().clone() for k in probe}
        del step, model
        torch.cuda.empty_cache()
    assert torch.isfinite(losses["graph"]).all() and torch.isfinite(losses["split"]).all()
    assert float(losses["eager"][-1].sum()) < float(losses["eager"][0].sum())
    # identical kernels on identical weights, deterministic: equal to float noise (2e-2 covers bf16 re-association only)
    torch.testing.assert_close(losses["graph"], losses["eager"][3:10], rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(losses["split"], losses["eager"][3:10], rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(losses["tail"], losses["eager"][3:10], rtol=2e-2, atol=2e-2)
    for mode in ("graph", "split", "tail"):  # after 10 updates each: the replicas of the four schedules still agree
        for k in probe:
            a, b = params[mode][k], params["eager"][k]
            err = float((a - b).norm() / b.norm().clamp(min=1e-9))
            assert err < 5e-3, (mode, k, err)


def _copy_state(dst, src):
    missing = dst.load_state_dict(src.state_dict(), strict=True)
    return missing


def test_config2_yolov8n_cbam_fp32_forward_parity_bs16_640():
    """BASELINE config 2: YOLOv8n + CBAM only, bs=16, 640x640, fp32 forward on the GPU within 1e-3 of the CPU oracle
    (train-mode BatchNorm, i.e. batch statistics, on both sides)."""
    from improving_yolov8_cbam_swinblock_amd.nn.tasks import DetectionModel
    from oracle.tasks import DetectionModel as OracleModel

    torch.manual_seed(0)
    oracle = OracleModel("yolov8n-cbam.yaml", ch=3, nc=1).train()
    model = DetectionModel("yolov8n-cbam.yaml", ch=3, nc=1)
    _copy_state(model, oracle)
    model = model.to(dev()).train()
    g = torch.Generator().manual_seed(1)
    img = torch.rand(16, 3, 640, 640, generator=g)
    with torch.no_grad():
        ref = oracle(img)
        got = model(img.to(dev()))
    for i, (a, b) in enumerate(zip(got, ref)):
        err = float((a.float().cpu() - b).abs().max())
        assert err <= 1e-3 * max(1.0, float(b.abs().max())), (i, err, float(b.abs().max()))


def test_config1_stock_yolov8n_eval_predict_bs4_640():
    """BASELINE config 1: stock YOLOv8n (nc=80), random init, 4x3x640x640, eval-mode predict: decoded [4, 84, 8400]
    and the three raw maps against the CPU oracle (fp32)."""
    from improving_yolov8_cbam_swinblock_amd.nn.tasks import DetectionModel
    from oracle.tasks import DetectionModel as OracleModel

    torch.manual_seed(0)
    oracle = OracleModel("yolov8n-stock.yaml", ch=3).eval()
    for m in oracle.modules():  # non-trivial running statistics
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    model = DetectionModel("yolov8n-stock.yaml", ch=3)
    _copy_state(model, oracle)
    model = model.to(dev()).eval()
    g = torch.Generator().manual_seed(1)
    img = torch.rand(4, 3, 640, 640, generator=g)
    with torch.no_grad():
        y_ref, maps_ref = oracle(img)
        y, maps = model(img.to(dev()))
    assert tuple(y.shape) == (4, 84, 8400)
    for a, b in zip(maps, maps_ref):
        assert float((a.float().cpu() - b).abs().max()) <= 1e-3 * max(1.0, float(b.abs().max()))
    # boxes are in pixels (<= 640): 1e-3 relative to the magnitude
    assert float((y.float().cpu() - y_ref).abs().max()) <= 1e-3 * max(1.0, float(y_ref.abs().max()))
    # fused (Conv+BN folded) inference gives the same predictions
    model.fuse()
    with torch.no_grad():
        y2, _ = model(img.to(dev()))
    assert float((y2.float().cpu() - y_ref).abs().max()) <= 2e-3 * max(1.0, float(y_ref.abs().max()))


def test_config5_yolov8m_swin384_bs16_1280_train_step_properties():
    """BASELINE config 5 at its stated per-GPU size: yolov8m-cbam-swin384 (SwinBlock(384): head_dim 192, 144 windows per
    image, 2304 windows per block), bs=16, 1280x1280, bf16.  No CPU reference finishes at this size, so properties: finite loss
    items and gradients for every trainable parameter, per-channel BatchNorm moments of a P3 layer, a loss that decreases over a
    few optimizer steps, and the peak device memory (printed; 288 GB HBM per GPU)."""
    from improving_yolov8_cbam_swinblock_amd.engine.trainer import TrainStep, synthetic_batch
    from improving_yolov8_cbam_swinblock_amd.nn.tasks import DetectionModel

    torch.manual_seed(0)
    torch.cuda.reset_peak_memory_stats()
    model = DetectionModel("yolov8m-cbam-swin384.yaml", ch=3, nc=1).to(dev()).train()
    batch = synthetic_batch(16, 1280, dev(), 1)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss, items = model(batch)
    loss.sum().backward()
    torch.cuda.synchronize()
    assert torch.isfinite(items).all()
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    missing = [n for n, p in named if p.grad is None]
    assert missing == [], missing
    gn = torch.stack([p.grad.float().norm() for _, p in named])
    assert len(named) > 250 and bool(torch.isfinite(gn).all()) and float(gn.max()) > 0
    peak = torch.cuda.max_memory_allocated() / 2**30
    print(f"\n[cfg5 bs16 1280 bf16] loss items {[round(float(v), 4) for v in items]}, {len(named)} gradients finite, peak device memory {peak:.1f} GiB")
    assert peak < 200
    model.zero_grad(set_to_none=True)
    step = TrainStep(model, world_size=1, lr=0.01)
    first = step(batch)
    for _ in range(5):
        last = step(batch)
    assert torch.isfinite(last).all() and float(last.sum()) < float(first.sum()), (first, last)


def test_hip_loss_fullsize_anchors_vs_oracle():
    """8400 anchors per image (640x640), bf16 maps as in training, 6 images with 0..12 labels: HIP loss and its
    gradients against the CPU oracle evaluated on the same (bf16-rounded) values."""
    from improving_yolov8_cbam_swinblock_amd.nn.tasks import DetectionModel
    from oracle.loss import v8DetectionLoss as OracleLoss
    from oracle.tasks import DetectionModel as OracleModel

    torch.manual_seed(0)
    crit = DetectionModel("yolov8n-cbam.yaml", ch=3, nc=1).to(dev()).init_criterion()
    ocrit = OracleLoss(OracleModel("yolov8n-cbam.yaml", ch=3, nc=1))
    g = torch.Generator().manual_seed(5)
    B = 6
    preds = [(torch.randn(B, 65, s, s, generator=g) * 2).bfloat16() for s in (80, 40, 20)]
    counts = [0, 1, 3, 12, 5, 2]
    bi = torch.cat([torch.full((c,), float(i)) for i, c in enumerate(counts)])
    n = int(bi.numel())
    boxes = torch.cat((torch.rand(n, 2, generator=g) * 0.7 + 0.15, torch.rand(n, 2, generator=g) * 0.5 + 0.03), 1)
    batch = {"batch_idx": bi, "cls": torch.zeros(n, 1), "bboxes": boxes}
    pg = [p.clone().to(dev()).requires_grad_(True) for p in preds]
    po = [p.float().requires_grad_(True) for p in preds]
    a, _ = crit(pg, {k: v.to(dev()) for k, v in batch.items()})
    b, _ = ocrit(po, batch)
    torch.testing.assert_close(a.cpu(), b, rtol=5e-4, atol=1e-5)
    a.sum().backward()
    b.sum().backward()
    for x, y in zip(pg, po):
        assert rel(x.grad.detach(), y.grad) < 5e-3


def _ddp_worker(rank, world, port, graph, q):
    import os

    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    # both ranks share the box's single GPU, which RCCL refuses: gloo moves the CUDA buckets instead; everything
    # above the collective (bucket packing, graph replay + eager tail, update) is the code the multi-GPU run uses
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from improving_yolov8_cbam_swinblock_amd.engine import ddp
    from improving_yolov8_cbam_swinblock_amd.engine.trainer import TrainStep, synthetic_batch
    from improving_yolov8_cbam_swinblock_amd.nn.tasks import DetectionModel

    d = torch.device("cuda", 0)
    torch.manual_seed(100 + rank)  # different initial weights: the broadcast must equalise them
    model = DetectionModel("yolov8n-cbam.yaml", ch=3, nc=1).to(d)
    ddp.broadcast_parameters(model)
    step = TrainStep(model, world_size=world, graph=graph)
    batch = synthetic_batch(2, 320, d, ddp.shard_seed(1, rank))
    items = [step(batch).float().cpu() for _ in range(5)]
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().float().reshape(-1).cpu() for p in model.parameters()])
    q.put((rank, torch.stack(items).numpy(), flat.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("graph", [False, True, "tail"], ids=["eager_overlapped", "three_graphs_overlapped", "graph_plus_eager_tail"])
def test_two_rank_training_keeps_replicas_identical(graph):
    """2 ranks (gloo over CUDA tensors, one GPU): after 5 steps on different shards the replicas hold identical
    parameters (gradient mean applied on both) and the losses differ per shard."""
    import socket

    import numpy as np
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, graph, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(120)
    assert np.isfinite(res[0][1]).all() and np.isfinite(res[1][1]).all()
    assert not np.allclose(res[0][1], res[1][1])  # different shards
    np.testing.assert_allclose(res[0][2], res[1][2], rtol=0, atol=0)


def _rccl_worker(port, q):
    import os

    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)  # "nccl" IS RCCL on ROCm
    from improving_yolov8_cbam_swinblock_amd.engine import ddp
    from improving_yolov8_cbam_swinblock_amd.engine.trainer import TrainStep, synthetic_batch
    from improving_yolov8_cbam_swinblock_amd.nn.tasks import DetectionModel

    d = torch.device("cuda", 0)
    torch.manual_seed(3)
    model = DetectionModel("yolov8n-cbam.yaml", ch=3, nc=1).to(d)
    # the flat-bucket broadcast and the bucketed gradient all-reduce through RCCL (one rank: identity, but every call,
    # buffer and stream hand-off of the multi-GPU schedule runs)
    flat = torch.arange(1 << 20, dtype=torch.float32, device=d)
    dist.broadcast(flat, 0)
    step = TrainStep(model, world_size=1, graph=False)
    batch = synthetic_batch(2, 320, d, 1)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss, _ = model(batch)
    loss.sum().backward()
    ref = {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None}
    gb = ddp.GradientBuckets(model, 1, bucket_bytes=4 << 20, overlap=False)
    gb.finish(force=True)
    torch.cuda.synchronize()
    ok = all(torch.equal(p.grad.float(), ref[n]) for n, p in model.named_parameters() if n in ref)
    views = all(p.grad.data_ptr() != 0 for p in model.parameters() if p.grad is not None)
    q.put((bool(ok), bool(views), len(gb.buckets), float(flat.sum().item())))
    dist.barrier()
    dist.destroy_process_group()
    del step


def test_rccl_single_rank_bucketed_allreduce():
    """the RCCL backend itself (not gloo) under the bucketed gradient mean, on the one GPU of the box: world_size 1, so
    the all-reduce is the identity - what is checked is that RCCL initialises with the dmabuf IPC setting, accepts the flat
    f32 buckets as they are built, completes asynchronously and leaves .grad pointing at the reduced buffers."""
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    proc = ctx.Process(target=_rccl_worker, args=(port, q))
    proc.start()
    ok, views, nb, total = q.get(timeout=300)
    proc.join(120)
    assert ok and views and nb >= 2
    assert total == float(sum(range(1 << 20)))


def _rccl_split_worker(port, q):
    import os

    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    from improving_yolov8_cbam_swinblock_amd.engine.trainer import TrainStep, synthetic_batch
    from improving_yolov8_cbam_swinblock_amd.nn.tasks import DetectionModel

    d = torch.device("cuda", 0)
    probe = ("model.0.conv.weight", "model.7.attn.in_proj_weight", "model.22.cv2.bn.weight", "model.26.cv3.0.2.bias")
    res = {}
    for mode in ("eager", "split"):
        torch.manual_seed(0)
        model = DetectionModel("yolov8s.yaml", ch=3, nc=1).to(d)
        step = TrainStep(model, world_size=1, lr=0.01, graph=False if mode == "eager" else "split")
        batch = synthetic_batch(4, 320, d, 1)
        out = [step(batch).float().cpu().clone() for _ in range(9 if mode == "eager" else 6)]
        torch.cuda.synchronize()
        sd = model.state_dict()
        res[mode] = (torch.stack(out), {k: sd[k].detach().float().cpu().clone() for k in probe})
        issued = sum(1 for _ in step.buckets.buckets)
        del step, model
    q.put((res["eager"][0][3:9].tolist(), res["split"][0].tolist(),
           {k: float((res["split"][1][k] - res["eager"][1][k]).norm() / res["eager"][1][k].norm().clamp(min=1e-9)) for k in probe}, issued))
    dist.barrier()
    dist.destroy_process_group()


def test_three_graph_schedule_with_rccl_between_the_graphs():
    """the several-rank schedule (G1 -> all_reduce(bucket 0) -> G2 -> all_reduce(bucket 1) -> wait -> G3) with REAL RCCL calls between the
    replayed graphs: a world-size-1 nccl process group on the box's one GPU, so every collective is the identity, but the captures run
    beside RCCL's watchdog thread, the async work objects are waited on by the update graph's stream, and the flat buckets are the
    buffers RCCL reads and writes.  Losses and parameters must follow the eager step's (replay i = eager step i + 3)."""
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    proc = ctx.Process(target=_rccl_split_worker, args=(port, q))
    proc.start()
    eager, split, errs, nb = q.get(timeout=600)
    proc.join(120)
    assert nb == 2
    torch.testing.assert_close(torch.tensor(split), torch.tensor(eager), rtol=2e-2, atol=2e-2)
    assert all(e < 5e-3 for e in errs.values()), errs


def test_profile_table_lists_every_layer():
    """utils/profile.py: the per-module forward / backward table (reference torch_utils.py:792-870 reporting format)."""
    from improving_yolov8_cbam_swinblock_amd.nn.tasks import DetectionModel
    from improving_yolov8_cbam_swinblock_amd.utils.profile import profile_model

    torch.manual_seed(0)
    model = DetectionModel("yolov8s.yaml", ch=3, nc=1).to(dev())
    rows = profile_model(model, torch.rand(2, 3, 320, 320, device=dev()), n=2, verbose=False)
    assert [r[0] for r in rows] == list(range(27))
    assert [r[1] for r in rows][:3] == ["Conv", "Conv", "C2f"] and rows[-1][1] == "Detect"
    # .np is what parse_model recorded (the reference records it before CBAM's lazy MLP exists: 32,768 parameters fewer)
    assert all(r[4] > 0 for r in rows) and sum(r[2] for r in rows) == 13405269 - 32768
    assert all(r[5] > 0 for r in rows if r[1] in ("Conv", "C2f", "SPPF", "CBAM", "SwinBlock", "Detect"))
