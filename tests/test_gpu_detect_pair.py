"""GPU: Detect's sibling first convolutions (cv2[i][0], cv3[i][0] read the same x[i]: reference head.py:71-72) run as ONE convolution
(ops.conv_bn_act_pair) - outputs, every gradient, running statistics and the deferred weight-gradient path against the two separate
Conv blocks, in both dtypes; and the merged operands of the weight arena against the per-call packers."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / b.norm().clamp(min=1e-12))


def _detect(seed, nc=3, ch=(32, 64, 128)):
    from improving_yolov8_cbam_swinblock_amd.nn.modules import Detect

    torch.manual_seed(seed)
    Detect.legacy = True
    m = Detect(nc, ch).to(dev()).train()
    m.stride = torch.tensor([8.0, 16.0, 32.0])
    m.bias_init()
    for p in m.parameters():
        if p.dim() == 1 and p.requires_grad:
            p.data.uniform_(0.5, 1.5)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.eps, mod.momentum = 1e-3, 0.03
    return m


def _run(m, xs, pair, dtype, deferred=False):
    from improving_yolov8_cbam_swinblock_amd import ops
    from improving_yolov8_cbam_swinblock_amd.nn.modules import Detect

    m.zero_grad(set_to_none=True)
    saved = Detect.pair_ok
    Detect.pair_ok = property(lambda self: pair)
    try:
        xs = [x.clone().requires_grad_(True) for x in xs]
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
            box, cls = m.forward_split(xs)
        loss = sum((b.float() * torch.linspace(-1, 1, b.shape[1], device=b.device).view(1, -1, 1, 1)).sum() for b in box) * 1e-3
        loss = loss + sum(c.float().square().mean() for c in cls)
        with ops.deferred_wgrad(deferred):
            loss.backward()
    finally:
        Detect.pair_ok = saved
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    bufs = {n: b.detach().clone() for n, b in m.named_buffers() if "running" in n}
    return [b.detach().float() for b in box], [c.detach().float() for c in cls], [x.grad.detach().clone() for x in xs], grads, bufs


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_detect_pair_matches_the_separate_branches(dtype):
    m = _detect(0)
    state = {k: v.clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    xs = [torch.randn(2, c, s, s, generator=g).to(dev()) for c, s in ((32, 24), (64, 12), (128, 6))]
    ref = _run(m, xs, False, dtype)
    m.load_state_dict(state)
    got = _run(m, xs, True, dtype)
    # float32: the same sums in another order; bfloat16: the merged data gradient keeps the two branches' sum in float32 where the
    # separate form rounds one branch's result to bfloat16 before the other adds it, and the weight-gradient splits differ
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    for a, b in zip(got[0] + got[1], ref[0] + ref[1]):
        assert rel(a, b) <= (1e-6 if dtype == torch.float32 else 4e-3), rel(a, b)
    for a, b in zip(got[2], ref[2]):
        assert rel(a, b) <= tol, rel(a, b)
    assert set(got[3]) == set(ref[3])
    bad = [(n, rel(got[3][n], ref[3][n])) for n in ref[3] if not rel(got[3][n], ref[3][n]) <= tol]
    assert not bad, bad[:6]
    bad = [(n, rel(got[4][n], ref[4][n])) for n in ref[4] if not rel(got[4][n], ref[4][n]) <= 1e-5]
    assert not bad, bad[:6]


def test_detect_pair_with_deferred_slab_sums_and_a_frozen_half():
    """the pair's weight gradient is one [c2 + c3, cin, 3, 3] result handed out as two views: with batched slab sums both parameters
    must adopt their view; with one of them frozen nothing is deferred and the other half is still right."""
    m = _detect(2)
    state = {k: v.clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(3)
    xs = [torch.randn(2, c, s, s, generator=g).to(dev()) for c, s in ((32, 24), (64, 12), (128, 6))]
    ref = _run(m, xs, True, torch.bfloat16, deferred=False)
    m.load_state_dict(state)
    got = _run(m, xs, True, torch.bfloat16, deferred=True)
    bad = [(n, rel(got[3][n], ref[3][n])) for n in ref[3] if not rel(got[3][n], ref[3][n]) <= 1e-5]
    assert not bad, bad[:6]
    m.load_state_dict(state)
    m.cv3[1][0].conv.weight.requires_grad_(False)
    frz = _run(m, xs, True, torch.bfloat16, deferred=True)
    assert "cv3.1.0.conv.weight" not in frz[3]
    bad = [(n, rel(frz[3][n], ref[3][n])) for n in frz[3] if not rel(frz[3][n], ref[3][n]) <= 1e-5]
    assert not bad, bad[:6]


def test_arena_pair_operands_equal_the_per_call_packers():
    """the merged forward / data-gradient operands written by the one-launch weight pack (ymi_pack_desc.ostride / o_off) against
    packing the concatenated weight directly, stride 1 and 2."""
    from improving_yolov8_cbam_swinblock_amd import ops

    torch.manual_seed(5)
    for stride in (1, 2):
        wa = torch.randn(64, 40, 3, 3, device=dev()).requires_grad_(True)
        wb = torch.randn(24, 40, 3, 3, device=dev()).requires_grad_(True)
        for dt in (torch.bfloat16, torch.float32):
            arena = ops.WeightArena()
            arena.note(wa, dt, ipad=40, pair=wb)
            arena.note(wa, dt, opad=88, stride=stride, pair=wb)
            arena.build()
            arena.pack()
            ops.set_weight_arena(None)
            f_ref = ops.pack_conv_fwd(torch.cat([wa, wb], 0).detach(), 40, dt)
            d_ref = ops.pack_conv_dgrad(torch.cat([wa, wb], 0).detach(), 88, stride, dt)
            torch.cuda.synchronize()
            assert torch.equal(arena.lookup_fwd(wa, 40, dt, pair=wb), f_ref)
            assert torch.equal(arena.lookup_dgrad(wa, 88, stride, dt, pair=wb), d_ref)
