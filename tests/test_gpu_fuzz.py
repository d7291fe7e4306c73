"""GPU: randomised shapes for the Conv block (conv -> train-mode BatchNorm -> SiLU, forward and all four gradients) in both
dtypes against the oracle: non-square maps, odd sizes, channel counts that are not tile multiples, batch 1..3, 1x1 / 3x3,
stride 1 / 2.  The fixed-shape tests pin the shapes of the benchmark's layers; this one looks for what they do not cover (ragged
last tiles in both GEMM directions, parity classes of stride-2 data gradients on odd maps, split-K ranges that end mid-image).
Default: 24 cases per dtype (a few seconds); YMI_FUZZ_CASES=N runs N.  Soak at the end of round 3 (YMI_FUZZ_CASES=3000: 3000 Conv cases per
dtype, 1500 module cases, 44 s): float32 worst 2.0e-6; bf16 against the matched oracle worst 6.8e-4 (4.7e-4 forward), weight gradient 3.8e-3;
modules worst 6.4e-6.  The soak's first run found the one unsupported combination (1x1 stride-2 data gradient), now built."""
import os
import random

import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = int(os.environ.get("YMI_FUZZ_CASES", "24"))


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / b.norm().clamp(min=1e-12))


def q(x):
    return x.bfloat16().float()


def _cases(seed, n, chunk):
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        cin = rng.choice([3, 8, 16, 24, 32, 40, 64, 72, 96, 128, 136, 192, 256])
        cout = rng.choice([8, 16, 24, 32, 48, 64, 72, 96, 128, 160, 256, 320])
        if chunk == 4:
            cout = rng.choice([cout, cout + 4])  # float32: any multiple of 4
        out.append((cin, cout, rng.choice([1, 3]), rng.choice([1, 2]), rng.randint(1, 3), rng.randint(5, 47), rng.randint(5, 47)))
    return out


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_conv_block_random_shapes_vs_oracle(dtype):
    import oracle.modules as OM
    from oracle import quant
    from improving_yolov8_cbam_swinblock_amd.nn.modules import Conv

    bf = dtype == torch.bfloat16
    worst = {}
    for idx, (cin, cout, k, s, n, h, w) in enumerate(_cases(20260 + int(bf), CASES, 8 if bf else 4)):
        torch.manual_seed(idx)
        o = OM.Conv(cin, cout, k, s)
        for b in o.modules():
            if isinstance(b, torch.nn.BatchNorm2d):
                b.eps, b.momentum = 1e-3, 0.03
                b.weight.data.uniform_(0.5, 1.5)
                b.bias.data.normal_(0, 0.3)
        if bf:
            quant.round_weights_(o)
        m = Conv(cin, cout, k, s)
        for b in m.modules():
            if isinstance(b, torch.nn.BatchNorm2d):
                b.eps, b.momentum = 1e-3, 0.03
        m.load_state_dict(o.state_dict())
        m = m.to(dev()).train()
        o.train()
        x = torch.randn(n, cin, h, w)
        ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
        gy = torch.randn(n, cout, ho, wo)
        if bf:
            x, gy = q(x), q(gy)
        xo = x.clone().requires_grad_(True)
        if bf:
            with quant.storage(torch.bfloat16):
                yo = o(xo)
                go = torch.autograd.grad(yo, [xo] + list(o.parameters()), gy)
        else:
            yo = o(xo)
            go = torch.autograd.grad(yo, [xo] + list(o.parameters()), gy)
        xg = x.to(dev()).requires_grad_(True)
        if bf:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                yg = m(xg)
        else:
            yg = m(xg)
        gg = torch.autograd.grad(yg, [xg] + list(m.parameters()), gy.to(dev()).to(yg.dtype))
        case = f"{cin}->{cout} k{k} s{s} n{n} {h}x{w}"
        errs = {"fwd": rel(yg, yo)}
        for a, b, name in zip(gg, go, ["x", "w", "gamma", "beta"]):
            errs[name] = rel(a, q(b) if (bf and name == "x") else b)
        # float32: north_star's 1e-3 (measured <= 2e-6).  bf16 against the matched oracle: the fixed-shape tests' 1e-3, and 5e-3 for the weight
        # gradient (bf16 split-K slabs; 4e-3 there: a 1x1 layer on 3 input channels has very few terms per weight to average the rounding over)
        for name, e in errs.items():
            bound = 1e-3 if not bf else (5e-3 if name == "w" else 1e-3)
            assert e <= bound, (case, name, e, errs)
            if e > worst.get(name, (0.0, ""))[0]:
                worst[name] = (e, case)
    print("\n[fuzz conv %s, %d cases] worst: " % ("bf16" if bf else "f32", CASES) + "; ".join(f"{k} {v[0]:.2e} ({v[1]})" for k, v in worst.items()))


def _module_cases(seed, n):
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        kind = rng.choice(["C2f", "C2f", "SPPF", "CBAM", "SwinBlock", "Bottleneck"])
        b, h, w = rng.randint(1, 3), rng.randint(6, 33), rng.randint(6, 33)
        if kind == "C2f":
            c1, c2 = rng.choice([16, 32, 48, 64, 96]), rng.choice([16, 32, 64, 128])  # hidden width c2/2 stays a multiple of 8
            args = (c1, c2, rng.randint(1, 3), rng.choice([True, False]))
        elif kind == "Bottleneck":
            c = rng.choice([16, 32, 64])
            c1, args = c, (c, c, rng.choice([True, False]), 1, ((3, 3), (3, 3)), rng.choice([0.5, 1.0]))
        elif kind == "SPPF":
            c1 = rng.choice([16, 32, 64, 128])
            args = (c1, rng.choice([16, 32, 64]), rng.choice([3, 5, 7]))
        elif kind == "CBAM":
            c1 = rng.choice([16, 32, 64, 128, 256])
            args = (c1,)
        else:
            heads = rng.choice([1, 2, 4])
            c1 = heads * rng.choice([8, 16, 32])
            args = (c1, heads, rng.choice([3, 4, 7]))
        out.append((kind, args, (b, c1, h, w)))
    return out


def test_composite_modules_random_shapes_vs_oracle():
    """float32: C2f (1-3 Bottlenecks, with and without shortcut), Bottleneck, SPPF (k 3/5/7), CBAM, SwinBlock (1/2/4 heads, windows 3/4/7,
    maps that need padding) on random non-square maps against the oracle: forward, input gradient and every parameter gradient at
    north_star's 1e-3 (measured 1e-6 .. 1e-5; arg-max routed modules can, rarely, break a near-tie the other way - such a case would show as
    an O(1e-2) input-gradient error on SPPF / CBAM and is re-run on a perturbed input before it counts)."""
    import oracle.modules as OM
    from improving_yolov8_cbam_swinblock_amd.nn import modules as PM

    worst = (0.0, "")
    for idx, (kind, args, shape) in enumerate(_module_cases(77, max(CASES // 2, 8))):
        for attempt in range(2):
            torch.manual_seed(1000 * attempt + idx)
            o = getattr(OM, kind)(*args)
            if kind == "CBAM" and o.ca.shared_MLP is None:
                o.ca.create_mlp(shape[1])
            for bn in o.modules():
                if isinstance(bn, torch.nn.BatchNorm2d):
                    bn.eps, bn.momentum = 1e-3, 0.03
                    bn.weight.data.uniform_(0.5, 1.5)
                    bn.bias.data.normal_(0, 0.3)
            m = getattr(PM, kind)(*args)
            if kind == "CBAM" and m.ca.shared_MLP is None:
                m.ca.create_mlp(shape[1])
            for bn in m.modules():
                if isinstance(bn, torch.nn.BatchNorm2d):
                    bn.eps, bn.momentum = 1e-3, 0.03
            m.load_state_dict(o.state_dict())
            m = m.to(dev()).train()
            o.train()
            x = torch.randn(*shape)
            xo = x.clone().requires_grad_(True)
            yo = o(xo)
            gy = torch.randn_like(yo)
            po = [p for p in o.parameters() if p.requires_grad]
            go = torch.autograd.grad(yo, [xo] + po, gy)
            xg = x.to(dev()).requires_grad_(True)
            yg = m(xg)
            pg = [p for p in m.parameters() if p.requires_grad]
            gg = torch.autograd.grad(yg, [xg] + pg, gy.to(dev()))
            errs = [rel(yg, yo)] + [rel(a, b) for a, b in zip(gg, go)]
            case = f"{kind}{args} {shape}"
            if max(errs) <= 1e-3 or attempt == 1 or kind not in ("SPPF", "CBAM"):
                break
        assert max(errs) <= 1e-3, (case, errs)
        if max(errs) > worst[0]:
            worst = (max(errs), case)
    print(f"\n[fuzz modules f32, {max(CASES // 2, 8)} cases] worst {worst[0]:.2e} ({worst[1]})")


# (forward, gradient) relative-L2 bounds per module kind.  Soak of 1000 cases (round 3): worst C2f 5.0e-3 / 2.4e-2 (up to three Bottlenecks: a
# chain of eight stored tensors, roundings that fall differently cascade - tools/c2f_matched_diag.py), Bottleneck 6.9e-4 / 2.2e-3, SwinBlock
# 3.2e-4 / 2.6e-3, CBAM 1.1e-4 / 3.1e-3, SPPF 7.2e-4 / 3.0e-2 (the gradient is routed by arg-max over bf16-rounded values).  A wrong row,
# channel or tap is O(1).
BF16_MODULE_BOUNDS = {"C2f": (8e-3, 4e-2), "Bottleneck": (2e-3, 6e-3), "SPPF": (2e-3, 5e-2), "CBAM": (1e-3, 1e-2), "SwinBlock": (1e-3, 8e-3)}


def test_composite_modules_random_shapes_bf16_vs_matched_oracle():
    """the same modules in bf16 against the quantisation-matched oracle, per-kind bounds (BF16_MODULE_BOUNDS above)."""
    import oracle.modules as OM
    from oracle import quant
    from improving_yolov8_cbam_swinblock_amd.nn import modules as PM

    worst = {}
    for idx, (kind, args, shape) in enumerate(_module_cases(78, max(CASES // 2, 8))):
        torch.manual_seed(idx)
        o = getattr(OM, kind)(*args)
        if kind == "CBAM" and o.ca.shared_MLP is None:
            o.ca.create_mlp(shape[1])
        for bn in o.modules():
            if isinstance(bn, torch.nn.BatchNorm2d):
                bn.eps, bn.momentum = 1e-3, 0.03
                bn.weight.data.uniform_(0.5, 1.5)
                bn.bias.data.normal_(0, 0.3)
        quant.round_weights_(o)
        m = getattr(PM, kind)(*args)
        if kind == "CBAM" and m.ca.shared_MLP is None:
            m.ca.create_mlp(shape[1])
        for bn in m.modules():
            if isinstance(bn, torch.nn.BatchNorm2d):
                bn.eps, bn.momentum = 1e-3, 0.03
        m.load_state_dict(o.state_dict())
        m = m.to(dev()).train()
        o.train()
        x = q(torch.randn(*shape))
        xo = x.clone().requires_grad_(True)
        with quant.storage(torch.bfloat16):
            yo = o(xo)
            gy = q(torch.randn_like(yo))
            po = [p for p in o.parameters() if p.requires_grad]
            go = torch.autograd.grad(yo, [xo] + po, gy)
        xg = x.to(dev()).requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            yg = m(xg)
        pg = [p for p in m.parameters() if p.requires_grad]
        gg = torch.autograd.grad(yg, [xg] + pg, gy.to(dev()).to(yg.dtype))
        case = f"{kind}{args} {shape}"
        ef = rel(yg, yo)
        eg = max(rel(a, q(b) if i == 0 else b) for i, (a, b) in enumerate(zip(gg, go)))
        fb, gb = BF16_MODULE_BOUNDS[kind]
        assert ef <= fb and eg <= gb, (case, ef, eg)
        w = worst.setdefault(kind, [0.0, 0.0])
        w[0], w[1] = max(w[0], ef), max(w[1], eg)
    print(f"\n[fuzz modules bf16, {max(CASES // 2, 8)} cases] worst (forward, gradient): " + "; ".join(f"{k} {v[0]:.1e} {v[1]:.1e}" for k, v in worst.items()))


def test_whole_models_random_scales_vs_oracle():
    """float32: the four YAMLs of the path (stock, CBAM, CBAM + SwinBlock, SwinBlock(384) ws 14) at random scale letters, class counts, image
    sizes (multiples of 32) and batch sizes: train-mode predictions, the three loss terms, and the gradients of the Detect head's parameters
    against the oracle graph with the same weights (1e-3 / 2e-3 / 2e-2 by norm).  Gradients further upstream pass the SPPF / CBAM arg-max,
    whose near-ties two float32 implementations may break differently: they are covered teacher-forced elsewhere."""
    from oracle.loss import v8DetectionLoss as OracleLoss
    from oracle.tasks import DetectionModel as OracleModel
    from improving_yolov8_cbam_swinblock_amd.nn.tasks import DetectionModel

    rng = random.Random(5)
    n_cases = max(CASES // 8, 3)
    worst = [0.0, 0.0, 0.0]
    for idx in range(n_cases):
        # (the SwinBlock(256) / SwinBlock(384) rows of the last two YAMLs fit the 's' / 'm' widths only, as in the reference)
        name = rng.choice(["yolov8{}-stock.yaml", "yolov8{}-cbam.yaml", "yolov8s.yaml", "yolov8s.yaml", "yolov8m-cbam-swin384.yaml"]).format(rng.choice(["n", "n", "s"]))
        nc, imgsz, bs = rng.choice([1, 2, 5, 80]), 32 * rng.randint(2, 6), rng.randint(1, 2)
        torch.manual_seed(idx)
        oracle = OracleModel(name, ch=3, nc=nc)
        model = DetectionModel(name, ch=3, nc=nc)
        model.load_state_dict(oracle.state_dict(), strict=True)
        model = model.to(dev()).train()
        oracle.train()
        g = torch.Generator().manual_seed(idx)
        img = torch.rand(bs, 3, imgsz, imgsz, generator=g)
        nb = rng.randint(0, 6)
        batch = {"batch_idx": torch.randint(0, bs, (nb,), generator=g).float(), "cls": torch.randint(0, nc, (nb, 1), generator=g).float(),
                 "bboxes": torch.cat((torch.rand(nb, 2, generator=g) * 0.6 + 0.2, torch.rand(nb, 2, generator=g) * 0.3 + 0.05), 1)}
        po = oracle(img)
        lo, _ = OracleLoss(oracle)(po, batch)
        lo.sum().backward()
        pg = model(img.to(dev()))
        lg, _ = model.init_criterion()(pg, {k: v.to(dev()) for k, v in batch.items()})
        lg.sum().backward()
        case = f"{name} nc{nc} {imgsz} bs{bs} boxes{nb}"
        ep = max(rel(a, b) for a, b in zip(pg, po))
        el = float((lg.detach().cpu() - lo.detach()).abs().max() / lo.detach().abs().max().clamp(min=1e-6))
        last = f"model.{len(model.model) - 1}."
        og = {n: p.grad for n, p in oracle.named_parameters() if n.startswith(last) and p.grad is not None}
        eg = max(rel(p.grad, og[n]) for n, p in model.named_parameters() if n in og and float(og[n].norm()) > 1e-6)
        assert ep <= 1e-3 and el <= 2e-3 and eg <= 2e-2, (case, ep, el, eg)
        worst = [max(worst[0], ep), max(worst[1], el), max(worst[2], eg)]
    print(f"\n[fuzz models f32, {n_cases} cases] worst: predictions {worst[0]:.1e}, loss {worst[1]:.1e}, Detect gradients {worst[2]:.1e}")
