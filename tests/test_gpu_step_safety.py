"""GPU: the training step's plumbing under the usage patterns the reference trainer has and TrainStep's own loop does not
exercise: gradient accumulation, a backward pass that raises, EMA counters set after the optimizer tables exist,
C2f widths that cannot use the concat buffer - and the multi-rank entry of bench.py as a child process.
"""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parents[1]


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / b.norm().clamp(min=1e-12))


def _small_model(seed=0):
    from improving_yolov8_cbam_swinblock_amd.engine.trainer import synthetic_batch
    from improving_yolov8_cbam_swinblock_amd.nn.tasks import DetectionModel

    torch.manual_seed(seed)
    model = DetectionModel("yolov8n-cbam.yaml", ch=3, nc=1).to(dev()).train()
    return model, synthetic_batch(2, 320, dev(), 1)


def _backward(model, batch):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss, _ = model(batch)
    loss.sum().backward()


def _grads(model):
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


def test_gradient_accumulation_is_exact_with_batched_slab_sums_enabled():
    """two backward passes without zero_grad (the reference accumulates nbs / batch iterations, trainer.py:305,397) inside
    ops.deferred_wgrad: the second pass finds .grad set, so AccumulateGrad would read each returned dW DURING the pass - those
    gradients must not be deferred.  The kernels are deterministic, so the accumulated gradient is exactly 2 x one pass."""
    from improving_yolov8_cbam_swinblock_amd import ops

    model, batch = _small_model()
    stats = {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}
    with ops.deferred_wgrad(True):
        _backward(model, batch)
        g1 = _grads(model)
        model.load_state_dict(stats, strict=False)
        _backward(model, batch)  # accumulates onto the first pass
        g2 = _grads(model)
    assert set(g1) == set(g2) and len(g1) > 100
    # (the batched slab sum of pass 1 and the per-layer sum of pass 2 add the same slabs in different groupings: float32 rounding)
    bad = [(n, rel(g2[n], 2 * g1[n])) for n in g1 if rel(g2[n], 2 * g1[n]) > 1e-5]
    assert not bad, bad[:5]
    # and the deferred path itself equals the immediate one
    model.zero_grad(set_to_none=True)
    model.load_state_dict(stats, strict=False)
    _backward(model, batch)
    g0 = _grads(model)
    assert all(rel(g0[n], g1[n]) <= 1e-5 for n in g1)


def test_backward_that_raises_does_not_poison_the_next_pass():
    """the autograd engine drops end-of-pass callbacks when a backward raises: the records of that pass must not leak into
    the next one, and the next pass must queue its own flush (else every weight gradient stays uninitialised memory)."""
    from improving_yolov8_cbam_swinblock_amd import ops

    model, batch = _small_model()
    stats = {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}
    _backward(model, batch)
    ref = _grads(model)
    model.zero_grad(set_to_none=True)

    class Boom(RuntimeError):
        pass

    def raise_late(mod, inp, out):  # layer 1's output gradient arrives after ~50 weight gradients have been deferred
        out.register_hook(lambda g: (_ for _ in ()).throw(Boom("injected")))

    with ops.deferred_wgrad(True):
        h = model.model[1].register_forward_hook(raise_late)
        model.load_state_dict(stats, strict=False)
        with pytest.raises(Boom):
            _backward(model, batch)
        h.remove()
        assert ops._deferred["records"], "the failed pass should have left deferred records behind (else this test checks nothing)"
        model.zero_grad(set_to_none=True)
        model.load_state_dict(stats, strict=False)
        _backward(model, batch)
        got = _grads(model)
    assert not ops._deferred["records"] and ops._deferred["task"] is None
    assert set(got) == set(ref)
    bad = [(n, rel(got[n], ref[n])) for n in ref if not rel(got[n], ref[n]) <= 1e-5]   # (uninitialised memory would be O(1) or NaN)
    assert not bad, bad[:5]


def test_ema_update_count_set_after_the_tables_exist_reaches_the_device():
    """reference idiom `ema.updates = ckpt["updates"]` (trainer.py:771) AFTER the optimizer's device tables were built (a warm-up
    step, state_dict(), load_state_dict()): the kernel computes the EMA decay from a device-side counter, which must follow."""
    import math

    from improving_yolov8_cbam_swinblock_amd.engine.trainer import TrainStep

    model, batch = _small_model()
    step = TrainStep(model, world_size=1, lr=0.01)
    step(batch)                      # builds the tables; device counter = 1
    step.opt.state_dict()
    n = 20000
    step.ema.updates = n             # as resume does
    key = "model.0.conv.weight"
    step.ema.ema.state_dict()[key].add_(0.5)   # an EMA far from the raw weights: the decay decides where it lands
    e0 = step.ema.ema.state_dict()[key].clone()
    step(batch)
    torch.cuda.synchronize()
    assert step.ema.updates == n + 1
    p1 = model.state_dict()[key]
    e1 = step.ema.ema.state_dict()[key]
    d = 0.9999 * (1 - math.exp(-(n + 1) / 2000))
    want = d * e0 + (1 - d) * p1
    assert float((e1 - want).abs().max()) <= 1e-6 * max(1.0, float(want.abs().max()))
    assert float((e1 - p1).abs().min()) > 0.4   # decay(2) ~ 1e-3 would have collapsed the EMA onto the raw weights
    # the graph-replayed step keeps host count, device counter and mirror together
    model2, batch2 = _small_model(1)
    gs = TrainStep(model2, world_size=1, lr=0.01, graph=True)
    for _ in range(3):
        gs(batch2)
    gs.ema.updates = n
    gs.ema.ema.state_dict()[key].add_(0.5)
    e0 = gs.ema.ema.state_dict()[key].clone()
    gs(batch2)
    torch.cuda.synchronize()
    assert gs.ema.updates == n + 1
    e1, p1 = gs.ema.ema.state_dict()[key], model2.state_dict()[key]
    want = d * e0 + (1 - d) * p1
    assert float((e1 - want).abs().max()) <= 1e-6 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_c2f_width_that_cannot_use_the_concat_buffer(dtype):
    """C2f with c = 12 (not a multiple of the 16-byte chunk): the Bottlenecks write no concat slots and ops.concat copies;
    every tensor with several consumers must still receive ALL its gradient contributions (join counts, nn/modules/block.py)."""
    import oracle.modules as OM
    from oracle import quant
    from improving_yolov8_cbam_swinblock_amd.nn.modules import C2f

    torch.manual_seed(5)
    o = OM.C2f(24, 24, 2, True)
    for b in o.modules():
        if isinstance(b, torch.nn.BatchNorm2d):
            b.eps, b.momentum = 1e-3, 0.03
            b.weight.data.uniform_(0.5, 1.5)
            b.bias.data.normal_(0, 0.3)
    bf = dtype == torch.bfloat16
    if bf:
        quant.round_weights_(o)
    m = C2f(24, 24, 2, True)
    for b in m.modules():
        if isinstance(b, torch.nn.BatchNorm2d):
            b.eps, b.momentum = 1e-3, 0.03
    m.load_state_dict(o.state_dict())
    m = m.to(dev()).train()
    o.train()
    x, gy = torch.randn(2, 24, 16, 16), torch.randn(2, 24, 16, 16)
    if bf:
        x, gy = x.bfloat16().float(), gy.bfloat16().float()
    xo = x.clone().requires_grad_(True)
    import contextlib

    with (quant.storage(torch.bfloat16) if bf else contextlib.nullcontext()):
        yo = o(xo)
        go = torch.autograd.grad(yo, [xo] + list(o.parameters()), gy)
    xg = x.to(dev()).requires_grad_(True)
    if bf:
        # 12 channels are 24 bytes: the bf16 backward kernels work in 16-byte chunks, and say so at the forward call
        with pytest.raises(NotImplementedError, match="16-byte chunks"):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                m(xg)
        return
    yg = m(xg)
    gg = torch.autograd.grad(yg, [xg] + list(m.parameters()), gy.to(dev()).to(yg.dtype))
    # f32: float32 noise; bf16 against the storage-matched oracle: a dropped gradient contribution is an O(1) error.  The slices
    # of this width are not 16-byte aligned, so they pass through float32 NCHW copies: two more roundings than the matched oracle has
    tol_f, tol_g = (1e-4, 2e-4) if not bf else (8e-3, 2e-2)
    assert rel(yg, yo) <= tol_f, rel(yg, yo)
    names = ["x"] + [n for n, _ in m.named_parameters()]
    errs = {n: rel(a, b) for n, a, b in zip(names, gg, go)}
    assert all(e <= tol_g for e in errs.values()), errs


def test_bench_two_ranks_gloo_as_a_child_process():
    """`python bench.py --gpus 2 --backend gloo ...` started by hand: self_launch -> torch.distributed.run -> two ranks that share
    the box's one GPU; exercises the launcher, the rendezvous, the barriers, the MAX-over-ranks timing and the rank-0 JSON line
    (RCCL itself refuses two ranks on one device: the gradient mean runs over gloo here, over RCCL on the 8-GPU node)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1", "--batch", "2", "--imgsz", "320",
           "--no-cpu-baseline", "--sustained", "0"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 4 and d["config"]["backend"] == "gloo"
    assert d["steps"] == 3 and d["value"] > 0 and d["scaling"] == "weak" and all(v == v for v in d["loss_items"])
    # the several-rank line explains itself: what the process group reports, an all-reduce of ones, the buckets, the exchange time nothing hid
    c = d["communication"]
    assert c["world_size"] == 2 and c["allreduce_of_ones"] == 2.0 and c["backend"] == "gloo"
    assert len(c["bucket_bytes"]) == 2 and sum(c["bucket_bytes"]) > 1 << 20 and c["exposed_ms_per_step"] >= 0.0


def test_frozen_mid_network_conv_gets_no_deferred_weight_gradient():
    """a Conv weight with requires_grad=False whose input still needs a gradient (round-3 ADVICE): autograd drops the tensor a
    weight-gradient Function returns for it at once, so a DEFERRED slab sum would later write into memory the allocator has handed to
    another gradient of the same pass.  Frozen weights get no GEMM and no record; every other gradient equals the non-deferred pass."""
    from improving_yolov8_cbam_swinblock_amd import ops

    model, batch = _small_model()
    frozen = [model.model[4].cv1.conv.weight, model.model[6].m[0].cv2.conv.weight, model.model[-1].cv3[1][2].weight]
    for p in frozen:
        p.requires_grad_(False)
    stats = {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}
    _backward(model, batch)  # immediate, complete gradients
    ref = _grads(model)
    model.zero_grad(set_to_none=True)
    model.load_state_dict(stats, strict=False)
    with ops.deferred_wgrad(True):
        _backward(model, batch)
    got = _grads(model)
    assert all(p.grad is None for p in frozen)
    assert set(got) == set(ref) and len(ref) > 100
    bad = [(n, rel(got[n], ref[n])) for n in ref if not rel(got[n], ref[n]) <= 1e-5]
    assert not bad, bad[:5]
    # the frozen Detect output conv keeps its trainable bias: that gradient is still produced
    assert model.model[-1].cv3[1][2].bias.grad is not None


def test_weight_used_twice_in_one_pass_accumulates_both_gradients():
    """one Conv applied twice in a graph (round-3 ADVICE, low): with deferral on, the second use must not be deferred as well - both
    slab sums would overwrite the same .grad.  The total equals the non-deferred result."""
    from improving_yolov8_cbam_swinblock_amd import ops
    from improving_yolov8_cbam_swinblock_amd.nn.modules import Conv

    torch.manual_seed(3)
    conv = Conv(32, 32, 3, 1).to(dev()).train()
    x = torch.randn(2, 32, 20, 20, device=dev())

    def run(deferred):
        conv.zero_grad(set_to_none=True)
        with ops.deferred_wgrad(deferred):
            y = conv(conv(x))
            y.float().square().mean().backward()
        torch.cuda.synchronize()
        return conv.conv.weight.grad.detach().clone()

    g0, g1 = run(False), run(True)
    assert rel(g1, g0) <= 1e-5, rel(g1, g0)


def test_bn_backward_final_passes_riding_in_weight_gradient_launches():
    """ops.wgrad_riders: every deferred weight-gradient launch is held back until the next BatchNorm backward, whose final pass (dgamma / dbeta /
    the apply pass's coefficients) then rides in it as extra workgroups.  The rider repeats chan_reduce_final_kernel's arithmetic bit for bit, so the
    gradients are IDENTICAL to a pass with the final passes as their own launches; a raised pass leaves no launch behind."""
    from improving_yolov8_cbam_swinblock_amd import _lib as L, ops

    model, batch = _small_model()
    stats = {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}
    with ops.deferred_wgrad(True):
        _backward(model, batch)
    ref = _grads(model)

    def ride():
        model.zero_grad(set_to_none=True)
        model.load_state_dict(stats, strict=False)
        with ops.deferred_wgrad(True), ops.wgrad_riders(True):
            _backward(model, batch)
        return _grads(model)

    got = ride()
    assert set(got) == set(ref) and len(ref) > 100
    bad = [n for n in ref if not torch.equal(got[n], ref[n])]
    assert not bad, bad[:5]
    again = ride()
    assert all(torch.equal(again[n], got[n]) for n in got)
    # a pass that raises inside the riders context: the held launch is forgotten, the next pass is unaffected
    model.zero_grad(set_to_none=True)
    model.load_state_dict(stats, strict=False)
    with pytest.raises(RuntimeError, match="boom"):
        with ops.deferred_wgrad(True), ops.wgrad_riders(True):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss, _ = model(batch)
            h = loss.register_hook(lambda g: (_ for _ in ()).throw(RuntimeError("boom")))
            loss.sum().backward()
    third = ride()
    assert all(torch.equal(third[n], got[n]) for n in got)
