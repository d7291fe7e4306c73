"""Generate the golden fixtures under tests/golden/ by running the REAL reference on CPU.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

The reference never travels to the GPU box; what is committed is data: inputs, weights and the
outputs the reference produced for them (float32 .npz), plus a JSON table of the model graphs.
The leaf modules (cbam.py, swin_block.py) are loaded straight from their files; the full model /
loss path is imported as the `ultralytics` package with the two import-time dependencies that are
absent from this image (cv2, torchvision) stubbed, exactly as recorded in SURVEY.md section 8(c).
"""
import importlib.metadata as md
import importlib.util
import json
import os
import sys
from pathlib import Path
from types import SimpleNamespace
from unittest.mock import MagicMock

import numpy as np
import torch

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent
REPO = OUT.parents[1]
CFG = REPO / "improving_yolov8_cbam_swinblock_amd" / "cfg" / "models" / "v8"


def load_leaf(name, rel):
    spec = importlib.util.spec_from_file_location(name, REF / rel)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def import_reference_package():
    os.environ.setdefault("YOLO_OFFLINE", "true")
    os.environ.setdefault("YOLO_CONFIG_DIR", "/tmp/ulcfg")
    os.environ.setdefault("YOLO_VERBOSE", "false")
    sys.modules.setdefault("cv2", MagicMock(__version__="4.10.0"))
    orig = md.version
    md.version = lambda n: "0.25.0" if n == "torchvision" else orig(n)
    sys.path.insert(0, str(REF))
    import ultralytics.nn.tasks as tasks  # noqa

    return tasks


def randomize(module, gen):
    """Non-trivial weights everywhere (LN/BN affine, running stats) so every term is pinned."""
    with torch.no_grad():
        for n, p in module.named_parameters():
            if p.ndim <= 1:
                p.copy_(torch.randn(p.shape, generator=gen) * 0.3 + (1.0 if n.endswith("weight") else 0.0))
            else:
                fan = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=gen) / fan**0.5)
        for n, b in module.named_buffers():
            if n.endswith("running_mean"):
                b.copy_(torch.randn(b.shape, generator=gen) * 0.2)
            elif n.endswith("running_var"):
                b.copy_(torch.rand(b.shape, generator=gen) + 0.5)


def sd_np(module, prefix="w."):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in module.state_dict().items()}


def save(name, **arrays):
    path = OUT / f"{name}.npz"
    np.savez_compressed(path, **arrays)
    print(f"{name}.npz  {path.stat().st_size / 1024:.1f} kB")


def main():
    torch.set_num_threads(4)
    g = torch.Generator().manual_seed(1234)
    cbam = load_leaf("ref_cbam", "ultralytics/nn/modules/cbam.py")
    swin = load_leaf("ref_swin", "ultralytics/nn/modules/swin_block.py")

    # ---- CBAM: lazy (ratio 16) and explicit (ratio 8 for C < 128), train-free module ----------
    for name, ctor_c, C, H, W in [("cbam_lazy_c32", None, 32, 12, 12), ("cbam_c64", 64, 64, 9, 7), ("cbam_lazy_c512", None, 512, 5, 5)]:
        m = cbam.CBAM(ctor_c) if ctor_c else cbam.CBAM()
        x = torch.randn(2, C, H, W, generator=g)
        m(x)  # creates the lazy MLP
        randomize(m, g)
        x.requires_grad_(True)
        y = m(x)
        gy = torch.randn(y.shape, generator=g)
        grads = torch.autograd.grad(y, [x] + list(m.parameters()), gy)
        names = ["x"] + [n for n, _ in m.named_parameters()]
        save(
            name,
            x=x.detach().numpy(),
            y=y.detach().numpy(),
            ca=m.ca(x).detach().numpy(),
            gy=gy.numpy(),
            **{"g." + n: t.numpy() for n, t in zip(names, grads)},
            **sd_np(m),
        )

    # ---- window partition / reverse: integer index maps (bit-exact) --------------------------
    for H, W, ws in [(42, 42, 7), (84, 84, 7), (14, 21, 7)]:
        B = 2
        idx = torch.arange(B * H * W, dtype=torch.int64).view(B, H, W, 1)
        part = swin.window_partition(idx, ws)  # [B*nW, ws*ws, 1]
        back = swin.window_reverse(part, ws, H, W)
        assert torch.equal(back, idx)
        save(f"window_index_{H}x{W}_ws{ws}", partition=part.squeeze(-1).numpy().astype(np.int32), shape=np.array([B, H, W, ws]))

    # ---- SwinBlock: pad 0 and pad > 0, non-square, non-zero LN bias ---------------------------
    for name, dim, heads, H, W in [("swin_d32_7x7", 32, 2, 7, 7), ("swin_d32_10x10", 32, 2, 10, 10), ("swin_d64_14x21", 64, 2, 14, 21), ("swin_d64_h4_20x20", 64, 4, 20, 20)]:
        m = swin.SwinBlock(dim, heads)
        randomize(m, g)
        x = torch.randn(2, dim, H, W, generator=g, requires_grad=True)
        y = m(x)
        gy = torch.randn(y.shape, generator=g)
        params = list(m.parameters())
        grads = torch.autograd.grad(y, [x] + params, gy)
        names = ["x"] + [n for n, _ in m.named_parameters()]
        save(name, x=x.detach().numpy(), y=y.detach().numpy(), gy=gy.numpy(), heads=np.array(heads),
             **{"g." + n: t.numpy() for n, t in zip(names, grads)}, **sd_np(m))

    # ---- package-level modules ---------------------------------------------------------------
    tasks = import_reference_package()
    from ultralytics.nn.modules.block import SPPF, C2f, Bottleneck
    from ultralytics.nn.modules.conv import Conv
    from ultralytics.nn.modules.head import Detect
    from ultralytics.utils.torch_utils import initialize_weights
    from ultralytics.utils.loss import v8DetectionLoss

    def fwd_bwd_fixture(name, m, x, extra=None):
        """train-mode forward + backward, then eval-mode forward with the updated running stats."""
        initialize_weights(m)
        randomize(m, g)
        before = sd_np(m)
        m.train()
        x = x.clone().requires_grad_(True)
        y = m(x)
        gy = torch.randn(y.shape, generator=g)
        params = [p for p in m.parameters() if p.requires_grad]
        pnames = [n for n, p in m.named_parameters() if p.requires_grad]
        grads = torch.autograd.grad(y, [x] + params, gy)
        after = sd_np(m, "after.")
        m.eval()
        with torch.no_grad():
            y_eval = m(x)
        save(name, x=x.detach().numpy(), y_train=y.detach().numpy(), y_eval=y_eval.numpy(), gy=gy.numpy(),
             **{"g." + n: t.numpy() for n, t in zip(["x"] + pnames, grads)}, **before,
             **{k: v for k, v in after.items() if "running" in k or "num_batches" in k}, **(extra or {}))

    fwd_bwd_fixture("conv_3x3_s1", Conv(8, 16, 3, 1), torch.randn(2, 8, 9, 11, generator=g))
    fwd_bwd_fixture("conv_3x3_s2", Conv(8, 16, 3, 2), torch.randn(2, 8, 10, 12, generator=g))
    fwd_bwd_fixture("conv_3x3_s2_c3", Conv(3, 16, 3, 2), torch.randn(2, 3, 16, 16, generator=g))
    fwd_bwd_fixture("conv_1x1", Conv(24, 16, 1, 1), torch.randn(2, 24, 7, 5, generator=g))
    fwd_bwd_fixture("bottleneck_add", Bottleneck(16, 16, True, 1, k=((3, 3), (3, 3)), e=1.0), torch.randn(2, 16, 8, 8, generator=g))
    fwd_bwd_fixture("c2f_n2_shortcut", C2f(16, 32, 2, True), torch.randn(2, 16, 8, 8, generator=g))
    fwd_bwd_fixture("c2f_n1_noshortcut", C2f(24, 16, 1, False), torch.randn(2, 24, 6, 10, generator=g))
    fwd_bwd_fixture("sppf_k5", SPPF(16, 16, 5), torch.randn(2, 16, 9, 11, generator=g))
    fwd_bwd_fixture("sppf_k7", SPPF(16, 24, 7), torch.randn(2, 16, 10, 10, generator=g))

    # fused-eval Conv (BaseModel.fuse path)
    from ultralytics.utils.torch_utils import fuse_conv_and_bn
    m = Conv(8, 16, 3, 1)
    initialize_weights(m)
    randomize(m, g)
    m.eval()
    x = torch.randn(2, 8, 6, 6, generator=g)
    w = sd_np(m)
    m.conv = fuse_conv_and_bn(m.conv, m.bn)
    delattr(m, "bn")
    with torch.no_grad():
        y = m.forward_fuse(x)
    save("conv_fused_eval", x=x.numpy(), y=y.numpy(), fused_weight=m.conv.weight.detach().numpy(), fused_bias=m.conv.bias.detach().numpy(), **w)

    # Detect: train maps and eval decode
    Detect.legacy = True
    det = Detect(3, [16, 32, 64])
    det.stride = torch.tensor([8.0, 16.0, 32.0])
    initialize_weights(det)
    randomize(det, g)
    det.bias_init()
    det.dfl.conv.weight.data[:] = torch.arange(16, dtype=torch.float).view(1, 16, 1, 1)
    xs = [torch.randn(2, c, s, s, generator=g) for c, s in [(16, 8), (32, 4), (64, 2)]]
    w = sd_np(det)
    det.train()
    yt = det([t.clone() for t in xs])
    det.eval()
    with torch.no_grad():
        ye, _ = det([t.clone() for t in xs])
    save("detect_nc3", **{f"x{i}": t.numpy() for i, t in enumerate(xs)}, **{f"y_train{i}": t.detach().numpy() for i, t in enumerate(yt)},
         y_eval=ye.numpy(), **w, **{"after." + k: v.numpy() for k, v in det.state_dict().items() if "running" in k})

    # ---- parse_model tables ------------------------------------------------------------------
    import yaml

    table = {}
    for fname, scale in [("yolov8-stock.yaml", "n"), ("yolov8-cbam.yaml", "n"), ("yolov8.yaml", "s"), ("yolov8-cbam-swin384.yaml", "m")]:
        d = yaml.safe_load((CFG / fname).read_text())
        d["scale"] = scale
        torch.manual_seed(0)
        model = tasks.DetectionModel(d, ch=3, verbose=False)
        table[f"{fname}:{scale}"] = {
            "layers": [{"i": m.i, "f": m.f, "type": m.type.split(".")[-1], "np": int(m.np)} for m in model.model],
            "save": list(model.save),
            "stride": [float(s) for s in model.stride],
            "params": int(sum(p.numel() for p in model.parameters())),
            "keys": {k: list(v.shape) for k, v in model.state_dict().items()},
        }
    (OUT / "parse_model_tables.json").write_text(json.dumps(table, indent=0))
    print("parse_model_tables.json", (OUT / "parse_model_tables.json").stat().st_size // 1024, "kB")

    # ---- end-to-end: width-reduced active graph, seed-constructed weights --------------------
    d = yaml.safe_load((CFG / "yolov8.yaml").read_text())
    d["scales"] = {"t": [0.33, 0.125, 1024]}
    d["scale"] = "t"
    for row in d["backbone"] + d["head"]:
        if row[2] == "SwinBlock":
            row[3] = [64]
    torch.manual_seed(7)
    model = tasks.DetectionModel(d, ch=3, nc=1, verbose=False)
    model.args = SimpleNamespace(box=7.5, cls=0.5, dfl=1.5)
    state0 = {k: v.clone() for k, v in model.state_dict().items()}
    B, S = 2, 64
    img = torch.rand(B, 3, S, S, generator=g)
    nb = 3
    ctr = torch.rand(B * nb, 2, generator=g) * 0.6 + 0.2
    wh = torch.rand(B * nb, 2, generator=g) * 0.3 + 0.05
    batch = {"img": img, "batch_idx": torch.arange(B).repeat_interleave(nb).float(), "cls": torch.zeros(B * nb, 1), "bboxes": torch.cat((ctr, wh), 1)}
    model.train()
    preds = model(img)
    crit = v8DetectionLoss(model)
    loss, items = crit(preds, batch)
    loss.sum().backward()
    gnorm = {n: float(p.grad.norm()) for n, p in model.named_parameters() if p.grad is not None}
    model.eval()
    with torch.no_grad():
        ye, _ = model(img)
    save("e2e_tiny_seed7", img=img.numpy(), batch_idx=batch["batch_idx"].numpy(), cls=batch["cls"].numpy(), bboxes=batch["bboxes"].numpy(),
         **{f"pred{i}": p.detach().numpy() for i, p in enumerate(preds)}, loss=loss.detach().numpy(), loss_items=items.numpy(), y_eval=ye.numpy(),
         **{"w." + k: v.numpy() for k, v in state0.items()})
    (OUT / "e2e_tiny_seed7_gradnorms.json").write_text(json.dumps(gnorm, indent=0))
    (OUT / "e2e_tiny_seed7_yaml.json").write_text(json.dumps(d))

    # loss-only fixture at nc=3 with crowded boxes (exercises multi-gt anchors) on fixed preds
    det = model.model[-1]
    preds2 = [torch.randn(2, 65, s, s, generator=g) for s in (8, 4, 2)]
    ctr = torch.rand(10, 2, generator=g) * 0.3 + 0.35
    wh = torch.rand(10, 2, generator=g) * 0.4 + 0.2
    batch2 = {"batch_idx": torch.tensor([0.0] * 6 + [1.0] * 4), "cls": torch.zeros(10, 1), "bboxes": torch.cat((ctr, wh), 1)}
    pl = [p.clone().requires_grad_(True) for p in preds2]
    loss2, items2 = crit(pl, batch2)
    loss2.sum().backward()
    save("loss_crowded", **{f"pred{i}": p.numpy() for i, p in enumerate(preds2)}, **{f"g.pred{i}": p.grad.numpy() for i, p in enumerate(pl)},
         batch_idx=batch2["batch_idx"].numpy(), cls=batch2["cls"].numpy(), bboxes=batch2["bboxes"].numpy(), loss=loss2.detach().numpy(), loss_items=items2.numpy(),
         stride=det.stride.numpy())


# SwinBlock cases beyond the small ones above: head_dim 192 (config 5's SwinBlock(384, 2)) and windows of
# more than 64 tokens (ws = 14 -> 196).  Own RNG streams (numpy RandomState), so adding cases never shifts main()'s.
LARGE_SWIN = [
    # name, dim, heads, ws, B, H, W, seed, seeded weights (True: not stored, rebuilt from the seed by tests/golden_weights.py)
    ("swin_d384_14x14", 384, 2, 7, 1, 14, 14, 11, True),
    ("swin_d384_20x20", 384, 2, 7, 1, 20, 20, 12, True),
    ("swin_d64_ws14_20x20", 64, 2, 14, 2, 20, 20, 13, False),
    ("swin_d64_ws14_28x14", 64, 4, 14, 1, 28, 14, 14, False),
    ("swin_d384_ws14_28x28", 384, 2, 14, 1, 28, 28, 15, True),
]


def swin_large_fixtures():
    sys.path.insert(0, str(REPO / "tests"))
    from golden_weights import grad_record, seeded_inputs, seeded_state

    torch.set_num_threads(4)
    swin = load_leaf("ref_swin", "ultralytics/nn/modules/swin_block.py")
    for name, dim, heads, ws, B, H, W, seed, seeded in LARGE_SWIN:
        m = swin.SwinBlock(dim, heads, ws)
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        m.load_state_dict(seeded_state(shapes, seed), strict=True)
        x, gy = seeded_inputs(seed, (B, dim, H, W), (B, dim, H, W))
        x.requires_grad_(True)
        y = m(x)
        params = list(m.parameters())
        names = [n for n, _ in m.named_parameters()]
        grads = torch.autograd.grad(y, [x] + params, gy)
        arrays = dict(meta=np.array([dim, heads, ws, B, H, W, seed, int(seeded)]), y=y.detach().numpy())
        arrays["g.x"] = grads[0].numpy()
        for n, t in zip(names, grads[1:]):
            if seeded:
                for kind, v in grad_record(t).items():
                    arrays[f"g{kind}.{n}"] = v
            else:
                arrays["g." + n] = t.numpy()
        if not seeded:
            arrays.update(sd_np(m))
            arrays["x"] = x.detach().numpy()
            arrays["gy"] = gy.numpy()
        save(name, **arrays)


def opt_step_fixture(opt_name="SGD", lr=0.01, momentum=0.937, out_name="opt_step_tiny"):
    """two optimizer steps of the REAL reference trainer code on the width-reduced model of e2e_tiny: BaseTrainer.build_optimizer
    ('SGD' branch, trainer.py:788-849), optimizer_step's clip + step + zero_grad (trainer.py:614-622, GradScaler disabled as
    for fp32 / bf16) and ModelEMA.update (torch_utils.py:657-673).  Stored: group membership, gradient norms, and strided
    samples + norms of every parameter / EMA entry after each step (the model's initial weights are e2e_tiny's)."""
    sys.path.insert(0, str(REPO / "tests"))
    from golden_weights import grad_record

    tasks = import_reference_package()
    from ultralytics.engine.trainer import BaseTrainer
    from ultralytics.utils.loss import v8DetectionLoss
    from ultralytics.utils.torch_utils import ModelEMA

    torch.set_num_threads(4)
    d = json.loads((OUT / "e2e_tiny_seed7_yaml.json").read_text())
    z = np.load(OUT / "e2e_tiny_seed7.npz")
    torch.manual_seed(7)
    model = tasks.DetectionModel(d, ch=3, nc=1, verbose=False)
    model.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w.")}, strict=True)
    model.args = SimpleNamespace(box=7.5, cls=0.5, dfl=1.5)
    for k, v in model.named_parameters():  # trainer.py:244-256: '.dfl' is always frozen
        if ".dfl" in k:
            v.requires_grad = False
    fake = SimpleNamespace(args=SimpleNamespace(lr0=0.01, momentum=0.937, warmup_bias_lr=0.1), data={"nc": 1})
    opt = BaseTrainer.build_optimizer(fake, model, name=opt_name, lr=lr, momentum=momentum, decay=5e-4)
    ema = ModelEMA(model)
    batch = {"img": torch.from_numpy(z["img"]), "batch_idx": torch.from_numpy(z["batch_idx"]), "cls": torch.from_numpy(z["cls"]),
             "bboxes": torch.from_numpy(z["bboxes"])}
    names = {id(p): n for n, p in model.named_parameters()}
    arrays = {}
    meta = {"groups": [[names[id(p)] for p in g["params"]] for g in opt.param_groups],
            "group_hyper": [{k: (list(g[k]) if isinstance(g[k], tuple) else g[k]) for k in ("lr", "momentum", "weight_decay", "nesterov", "betas", "eps") if k in g}
                            for g in opt.param_groups], "norms": [], "loss": [], "optimizer": type(opt).__name__}
    model.train()
    crit = v8DetectionLoss(model)
    for step in range(2):
        loss, items = crit(model(batch["img"]), batch)
        loss.sum().backward()
        norm = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=10.0)  # trainer.py:617
        opt.step()
        opt.zero_grad()
        ema.update(model)
        meta["norms"].append(float(norm))
        meta["loss"].append([float(v) for v in loss])
        for n, p in model.state_dict().items():
            if p.dtype.is_floating_point:
                for kind, v in grad_record(p).items():
                    arrays[f"s{step}.p{kind}.{n}"] = v
        for n, p in ema.ema.state_dict().items():
            if p.dtype.is_floating_point:
                for kind, v in grad_record(p).items():
                    arrays[f"s{step}.e{kind}.{n}"] = v
    meta["ema_updates"] = ema.updates
    save(out_name, **arrays)
    (OUT / f"{out_name}.json").write_text(json.dumps(meta))


def tie_fixtures():
    """arg-max routing on TIED values, produced by the reference's own modules: inputs quantised to 16 levels,
    so nearly every max has several maximal elements.  nn.MaxPool2d of the reference's SPPF (block.py:220,225), and the
    reference CBAM (AdaptiveMaxPool2d(1) cbam.py:9, torch.max(dim=1) cbam.py:50) once with random weights (ties in the
    per-channel spatial max) and once with a zero second MLP layer (ca = 0.5 for every channel, so the per-pixel channel
    max of x * ca is tied too).  Output gradients are small dyadic numbers: every gradient sum of the pool fixtures is
    exact in float32, so they are compared bit for bit.  Own RNG stream."""
    torch.set_num_threads(4)
    g = torch.Generator().manual_seed(4321)
    cbam = load_leaf("ref_cbam", "ultralytics/nn/modules/cbam.py")
    import_reference_package()
    from ultralytics.nn.modules.block import SPPF

    def levels(shape, n=16, step=0.25):
        return (torch.randint(0, n, shape, generator=g).float() - n // 2) * step

    for k, (H, W) in ((5, (11, 13)), (7, (10, 9))):
        pool = SPPF(16, 16, k).m  # the reference's nn.MaxPool2d(k, 1, k // 2)
        x = levels((2, 8, H, W)).requires_grad_(True)
        ys = [x]
        ys.extend(pool(ys[-1]) for _ in range(3))  # block.py:224-225
        cat = torch.cat(ys, 1)
        gy = levels(cat.shape, 16, 0.125)
        (gx,) = torch.autograd.grad(cat, x, gy)
        save(f"pool_ties_k{k}", x=x.detach().numpy(), y=cat.detach().numpy(), gy=gy.numpy(), **{"g.x": gx.numpy()}, k=np.array(k))

    for name, flat in (("cbam_ties_c32", False), ("cbam_ties_flatca_c32", True)):
        m = cbam.CBAM()
        x = levels((2, 32, 12, 12))
        m(x)  # creates the lazy MLP
        randomize(m, g)
        if flat:
            with torch.no_grad():
                m.ca.shared_MLP[2].weight.zero_()
        x.requires_grad_(True)
        y = m(x)
        gy = levels(y.shape, 16, 0.125)
        grads = torch.autograd.grad(y, [x] + list(m.parameters()), gy)
        names = ["x"] + [n for n, _ in m.named_parameters()]
        save(name, x=x.detach().numpy(), y=y.detach().numpy(), gy=gy.numpy(), **{"g." + n: t.numpy() for n, t in zip(names, grads)}, **sd_np(m))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "ties":
        tie_fixtures()
    elif len(sys.argv) > 1 and sys.argv[1] == "swin_large":
        swin_large_fixtures()
    elif len(sys.argv) > 1 and sys.argv[1] == "opt_step":
        opt_step_fixture()
    elif len(sys.argv) > 1 and sys.argv[1] == "opt_step_adamw":  # the 'AdamW' branch (what optimizer='auto' picks for short runs)
        opt_step_fixture("AdamW", lr=0.002, momentum=0.9, out_name="opt_step_tiny_adamw")
    else:
        main()
        swin_large_fixtures()
        opt_step_fixture()
        opt_step_fixture("AdamW", lr=0.002, momentum=0.9, out_name="opt_step_tiny_adamw")
        tie_fixtures()
