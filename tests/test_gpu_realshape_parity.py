"""GPU parity at the REAL shapes of BASELINE configs 3 and 5 against the CPU oracle (same weights, same batch).

The oracle (oracle/, pinned to the reference by tests/golden) runs the fp32 NCHW reference arithmetic on the host
cores; the product runs through libyolo_mi355.so.  Two kinds of comparison, tolerances written at the asserts:

* end to end: Detect maps, the three loss terms and EVERY parameter's gradient (relative L2 per tensor);
* layer by layer ("teacher forced"): every top-level layer of the graph gets the ORACLE's input activation and the
  ORACLE's output gradient, so its forward, its input gradient and its parameter gradients are compared at the real
  shape without the chaotic part of the chain.

Why both: the chain contains discrete decisions - max-pool arg-max routing in the two SPPF blocks and in CBAM's
channel/spatial max, top-k anchor assignment in the loss.  A forward difference of 5e-6 (float32 re-association over 10
layers) flips 2 of 409,600 first-pool arg-max positions of layer 11 at bs=4, and each flip re-routes ~1e-2 of the gradient
that flows to layers 0-10; the oracle ITSELF moves by that amount when it is fed the product's layer-10 output
(tools/grad_diag.py prints both numbers).  End-to-end gradient bounds are therefore those of the discontinuity, and
the tight bounds sit on the teacher-forced comparison, where both sides take the same decisions.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def rel_l2(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / b.norm().clamp(min=1e-12))


def _threads():
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))


def _pair(cfg, seed=0, round_weights=False):
    from improving_yolov8_cbam_swinblock_amd.nn.tasks import DetectionModel
    from oracle import quant
    from oracle.tasks import DetectionModel as OracleModel

    torch.manual_seed(seed)
    oracle = OracleModel(cfg, ch=3, nc=1).train()
    # non-trivial affine parameters everywhere a fresh model has ones / zeros, so every gradient term is exercised
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for m in oracle.modules():
            if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.LayerNorm)):
                m.weight.copy_(1.0 + 0.2 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
    if round_weights:  # bf16-representable conv / linear weights on both sides (the product packs them to bf16)
        quant.round_weights_(oracle)
    model = DetectionModel(cfg, ch=3, nc=1)
    model.load_state_dict(oracle.state_dict(), strict=True)
    return oracle, model.to(dev()).train()


def _oracle_run(oracle, batch, storage=None):
    """oracle forward + loss + backward; records every top-level layer's input(s), output(s) and output gradient(s).
    storage: run the oracle in storage-precision mode (oracle/quant.py) - the quantisation-matched reference of the bf16 path."""
    import contextlib

    from oracle import quant
    from oracle.loss import v8DetectionLoss as OracleLoss

    _threads()
    with (quant.storage(storage) if storage is not None else contextlib.nullcontext()):
        return _oracle_run_inner(oracle, batch, OracleLoss)


def _oracle_run_inner(oracle, batch, OracleLoss):
    rec = {}

    def hook(mod, inp, out):
        x = inp[0]
        outs = list(out) if isinstance(out, (list, tuple)) else [out]
        r = rec[mod.i] = {"x": [t.detach() for t in x] if isinstance(x, (list, tuple)) else x.detach(), "y": [t.detach() for t in outs], "gy": [None] * len(outs)}
        for k, t in enumerate(outs):
            if t.requires_grad:
                t.register_hook(lambda g, r=r, k=k: r["gy"].__setitem__(k, g.detach().clone()))

    handles = [m.register_forward_hook(hook) for m in oracle.model]
    preds = oracle(batch["img"])
    loss, _ = OracleLoss(oracle)(preds, batch)
    loss.sum().backward()
    for h in handles:
        h.remove()
    grads = {n: p.grad.clone() for n, p in oracle.named_parameters() if p.grad is not None}
    return [p.detach() for p in preds], loss.detach(), grads, rec


def _product_step(model, batch, dtype):
    """the training path exactly as engine.trainer.TrainStep drives it: model(batch) -> (loss * B, items), backward
    (so the graph-level concat slots and gradient joins of nn/tasks.py are what is tested); the Detect maps for the
    forward comparison come from a second, gradient-free forward of the same weights."""
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
        with torch.no_grad():
            stats = {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}
            preds = model(batch["img"])
            model.load_state_dict(stats, strict=False)  # the extra forward must not move the BatchNorm running statistics
        loss, _ = model(batch)
    loss.sum().backward()
    torch.cuda.synchronize()
    return preds, loss.detach(), {n: p.grad for n, p in model.named_parameters() if p.grad is not None}


def _noise_floor(name, ref_grads):
    """`mlp.2.bias` of a SwinBlock adds a per-channel constant right in front of a Conv + train-mode BatchNorm, which
    removes it: its true gradient is zero (1e-16 in float64) and what either side computes is rounding noise."""
    return name.endswith("mlp.2.bias")


def _teacher_forced(oracle, model, rec, dtype, tol_fwd, tol_grad, loose=None, storage=None):
    """every top-level layer alone, on the oracle's input and output gradient.  loose: {layer type: bound} overrides."""
    import contextlib

    from oracle import quant

    with (quant.storage(storage) if storage is not None else contextlib.nullcontext()):
        return _teacher_forced_inner(oracle, model, rec, dtype, tol_fwd, tol_grad, loose)


def _teacher_forced_inner(oracle, model, rec, dtype, tol_fwd, tol_grad, loose):
    worst = []
    for om, gm in zip(oracle.model, model.model):
        r = rec[om.i]
        if all(g is None for g in r["gy"]):
            continue
        for p in list(om.parameters()) + list(gm.parameters()):
            p.grad = None
        multi = isinstance(r["x"], list)
        xo = [t.clone().requires_grad_(True) for t in (r["x"] if multi else [r["x"]])]
        yo = om(xo if multi else xo[0])
        yo = list(yo) if isinstance(yo, (list, tuple)) else [yo]
        torch.autograd.backward(yo, [g if g is not None else torch.zeros_like(y) for y, g in zip(yo, r["gy"])])
        xg = [t.to(dev()).clone().requires_grad_(True) for t in (r["x"] if multi else [r["x"]])]
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
            yg = gm(xg if multi else xg[0])
        yg = list(yg) if isinstance(yg, (list, tuple)) else [yg]
        torch.autograd.backward(yg, [(g if g is not None else torch.zeros_like(y)).to(dev()).to(y.dtype) for y, g in zip(yg, r["gy"])])
        kind = type(gm).__name__
        tf, tg = (loose or {}).get(kind, (tol_fwd, tol_grad))
        for k, (a, b) in enumerate(zip(yg, yo)):
            e = rel_l2(a, b)
            worst.append((f"layer {om.i} {kind} out{k}", e, tf))
        for k, (a, b) in enumerate(zip(xg, xo)):
            if b.grad is not None and float(b.grad.norm()) > 0:
                worst.append((f"layer {om.i} {kind} dx{k}", rel_l2(a.grad, b.grad), tg))
        for (n, a), (_, b) in zip(gm.named_parameters(), om.named_parameters()):
            if b.grad is not None and not _noise_floor(n, None):
                worst.append((f"layer {om.i} {kind} {n}", rel_l2(a.grad, b.grad), tg))
    torch.cuda.synchronize()
    return worst


def _print_worst(tag, rows, n=10):
    rows = sorted(rows, key=lambda t: -t[1] / t[2])[:n]
    print(f"\n[{tag}] closest to their bounds:", "; ".join(f"{w} {e:.2e}/{t:.0e}" for w, e, t in rows))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_config3_yolov8s_bs4_640_vs_oracle(dtype):
    """BASELINE config 3's graph (yolov8s.yaml: CBAM + 2 SwinBlock(256) + SPPF5 + SPPF7) at 640x640, bs=4."""
    from improving_yolov8_cbam_swinblock_amd.engine.trainer import synthetic_batch

    oracle, model = _pair("yolov8s.yaml")
    cpu = synthetic_batch(4, 640, torch.device("cpu"), 1)
    ref_preds, ref_loss, ref_grads, rec = _oracle_run(oracle, cpu)
    batch = {k: (v.to(dev()) if torch.is_tensor(v) else v) for k, v in cpu.items()}
    preds, loss, got = _product_step(model, batch, dtype)
    assert set(got) == set(ref_grads), set(got) ^ set(ref_grads)
    errs = [(n, rel_l2(got[n], ref_grads[n])) for n in ref_grads if not _noise_floor(n, ref_grads)]
    lrel = float(((loss.float().cpu() - ref_loss).abs() / ref_loss.abs()).max())
    perr = [float((a.float().cpu() - b).abs().max()) / max(1.0, float(b.abs().max())) for a, b in zip(preds, ref_preds)]
    print(f"\n[cfg3 {dtype}] pred max-abs (scaled) {[f'{e:.1e}' for e in perr]}, loss rel {lrel:.2e}; end-to-end gradient rel-L2 max {max(e for _, e in errs):.2e} "
          f"median {sorted(e for _, e in errs)[len(errs) // 2]:.2e}")
    if dtype == torch.float32:
        assert max(perr) <= 1e-3, perr                       # north_star: within 1e-3 of the fp32 CPU reference
        assert lrel <= 1e-3, (loss, ref_loss)
        # layers behind the last arg-max decision of the backward chain (SPPF 12 -> head): plain float32 agreement
        tail = [(n, e) for n, e in errs if int(n.split(".")[1]) >= 12]
        assert all(e <= 1e-3 for _, e in tail), [t for t in tail if t[1] > 1e-3]
        # layers 0-11: bounded by the routing discontinuity described in the module docstring (observed 8e-3: two flipped
        # near-ties of 409,600 first-pool windows; tools/grad_diag.py)
        assert all(e <= 2e-2 for _, e in errs), [t for t in errs if t[1] > 2e-2]
        rows = _teacher_forced(oracle, model, rec, dtype, 1e-4, 1e-4)   # same decisions on both sides: float32 noise only
    else:
        assert lrel <= 2e-2, (loss, ref_loss)                # bf16: loss within 2e-2 relative
        # bf16 maps change which anchors the top-k assigner picks, so the end-to-end gradient is a different (equally
        # valid) sparse pattern: the per-parameter bound of 5e-2 is asserted layer by layer, where targets are shared.
        # CBAM / SPPF route gradients through arg-max over bf16-rounded values (ties and near-ties decided differently
        # from the float32 oracle): their bound is the routing's, not the arithmetic's.
        rows = _teacher_forced(oracle, model, rec, dtype, 2e-2, 5e-2, loose={"SPPF": (2e-2, 0.35), "CBAM": (2e-2, 0.1)})
    _print_worst(f"cfg3 {dtype} teacher-forced", rows)
    bad = [(w, e, t) for w, e, t in rows if not e <= t]
    assert not bad, bad


def test_config5_yolov8m_swin384_bs1_1280_forward_vs_oracle():
    """BASELINE config 5's graph (m scale, SwinBlock(384): head_dim 192, 144 windows per image) at 1280x1280, bs=1,
    float32 forward against the oracle: Detect maps within 1e-3 (scaled max-abs)."""
    oracle, model = _pair("yolov8m-cbam-swin384.yaml")
    _threads()
    g = torch.Generator().manual_seed(1)
    img = torch.rand(1, 3, 1280, 1280, generator=g)
    with torch.no_grad():
        ref = oracle(img)
        got = model(img.to(dev()))
    perr = [float((a.float().cpu() - b).abs().max()) / max(1.0, float(b.abs().max())) for a, b in zip(got, ref)]
    print(f"\n[cfg5 f32 fwd 1280] pred max-abs (scaled) {perr}")
    assert max(perr) <= 1e-3, perr


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_config5_yolov8m_swin384_bs2_640_vs_oracle(dtype):
    """config 5's graph with every gradient against the oracle at a size the CPU finishes in seconds (bs=2, 640x640:
    SwinBlock(384), head_dim 192, attention forward and backward; widths 192/384/576)."""
    from improving_yolov8_cbam_swinblock_amd.engine.trainer import synthetic_batch

    oracle, model = _pair("yolov8m-cbam-swin384.yaml")
    cpu = synthetic_batch(2, 640, torch.device("cpu"), 3)
    ref_preds, ref_loss, ref_grads, rec = _oracle_run(oracle, cpu)
    batch = {k: (v.to(dev()) if torch.is_tensor(v) else v) for k, v in cpu.items()}
    preds, loss, got = _product_step(model, batch, dtype)
    assert set(got) == set(ref_grads)
    lrel = float(((loss.float().cpu() - ref_loss).abs() / ref_loss.abs()).max())
    errs = [(n, rel_l2(got[n], ref_grads[n])) for n in ref_grads if not _noise_floor(n, ref_grads)]
    print(f"\n[cfg5 {dtype}] loss rel {lrel:.2e}; end-to-end gradient rel-L2 max {max(e for _, e in errs):.2e} median {sorted(e for _, e in errs)[len(errs) // 2]:.2e}")
    if dtype == torch.float32:
        assert lrel <= 1e-3
        assert all(e <= 5e-2 for _, e in errs), [t for t in errs if t[1] > 5e-2]   # routing discontinuity bound (docstring)
        rows = _teacher_forced(oracle, model, rec, dtype, 1e-4, 1e-4)
    else:
        assert lrel <= 2e-2
        rows = _teacher_forced(oracle, model, rec, dtype, 2e-2, 5e-2, loose={"SPPF": (2e-2, 0.35), "CBAM": (2e-2, 0.1)})
    _print_worst(f"cfg5 {dtype} teacher-forced", rows)
    bad = [(w, e, t) for w, e, t in rows if not e <= t]
    assert not bad, bad


@pytest.mark.parametrize("cfg,bs,seed", [("yolov8s.yaml", 4, 1), ("yolov8m-cbam-swin384.yaml", 2, 3)], ids=["cfg3", "cfg5"])
def test_bf16_layers_vs_quantisation_matched_oracle(cfg, bs, seed):
    """the bf16 step's kernels at the real shapes of configs 3 and 5 (640x640), layer by layer, against the oracle in
    storage-precision mode on identical bf16-valued weights, inputs and output gradients: what remains is
    accumulation order and values that round the other way, so the bounds are 4e-3 (outputs) / 1e-2 (gradients) relative L2
    instead of the 2e-2 .. 0.35 a float32 oracle needs - a mis-routed window or a wrong row of a ragged tile does not fit.
    Measured (MI355X): Conv 7e-5, Detect 3e-4, SwinBlock 4e-4, CBAM 2e-5 on outputs.  Two layer kinds get their own bounds:
    * C2f (8e-3 / 2e-2): a value that rounds the other way is a full-ulp error for the next convolution, so along a chain
      of stored tensors the difference grows towards the unmatched rounding level - tools/c2f_matched_diag.py prints
      1e-5, 1.5e-4, 4e-4, 1.4e-3, 1.7e-3, 2.6e-3 along the six blocks of a C2f with two Bottlenecks (5.6e-3 with four);
    * SPPF gradients (3e-2): such a value in front of the pool cascade re-routes the gradient of every window whose maximum
      it was (arg-max is discontinuous).  With identical values both sides take the same decisions, ties included
      (tests/test_gpu_bf16_matched.py checks that rule bit for bit)."""
    from improving_yolov8_cbam_swinblock_amd.engine.trainer import synthetic_batch

    oracle, model = _pair(cfg, round_weights=True)
    cpu = synthetic_batch(bs, 640, torch.device("cpu"), seed)
    cpu["img"] = cpu["img"].bfloat16().float()
    _, _, _, rec = _oracle_run(oracle, cpu, storage=torch.bfloat16)
    rows = _teacher_forced(oracle, model, rec, torch.bfloat16, 4e-3, 1e-2, loose={"C2f": (8e-3, 2e-2), "SPPF": (4e-3, 3e-2)}, storage=torch.bfloat16)
    _print_worst(f"{cfg} bf16 vs matched oracle", rows, n=14)
    bad = [(w, e, tol) for w, e, tol in rows if not e <= tol]
    assert not bad, bad
