"""In-launch hand-offs between workgroups (csrc/common.h: write-through stores, one ticket per workgroup, the last arriver combines).

The failure these tests look for is a STALE read: a last-arriving workgroup that sums another workgroup's partial row from a cache that
still holds an earlier launch's bytes.  So every case reuses the same workspace addresses launch after launch with DIFFERENT data,
checks every output word against an independent float64 reference, and runs beside an unrelated stream of copies on a second stream
(uneven load: workgroups finish in a different order every launch).  Reference behaviour: BatchNorm2d's backward inside Conv
(nn/modules/conv.py:66-67,79) - per-channel sums over all pixels, whatever the order the hardware produced the partials in.
"""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _bn_bwd_reference(dout, raw, gamma, beta, mean, inv):
    """float64: dz = dout * silu'(z), z = (raw - mean) * inv * gamma + beta; dbeta = sum dz, dgamma = sum dz * xhat, draw as BatchNorm's backward"""
    d = dout.double()
    x = raw.double()
    g, b, mu, iv = (t.double()[None, :, None, None] for t in (gamma, beta, mean, inv))
    xh = (x - mu) * iv
    z = xh * g + b
    s = torch.sigmoid(z)
    dz = d * (s * (1 + z * (1 - s)))
    dbeta = dz.sum((0, 2, 3))
    dgamma = (dz * xh).sum((0, 2, 3))
    n = d.shape[0] * d.shape[2] * d.shape[3]
    draw = g * iv * (dz - dbeta[None, :, None, None] / n - xh * dgamma[None, :, None, None] / n)
    # the sums of magnitudes: what a float32 accumulation error (or a stale partial row) is measured against
    return draw, dgamma, dbeta, (dz * xh).abs().sum((0, 2, 3)), dz.abs().sum((0, 2, 3))


def _call_bn_bwd(dout, raw, gamma, beta, mean, inv, ws):
    from improving_yolov8_cbam_swinblock_amd import _lib as L

    c = raw.shape[1]
    draw = L.empty_nhwc(*raw.shape, raw.dtype, raw.device)
    dgamma = torch.empty(c, dtype=torch.float32, device=raw.device)
    dbeta = torch.empty(c, dtype=torch.float32, device=raw.device)
    L.check(
        L.lib().ymi_bn_act_bwd(ctypes.byref(L.as_ymi(dout)), ctypes.byref(L.as_ymi(raw)), L.ptr(gamma), L.ptr(mean), L.ptr(inv), L.ptr(beta), L.ACT_SILU,
                               ctypes.byref(L.as_ymi(draw)), L.ptr(dgamma), L.ptr(dbeta), L.ptr(ws), ws.numel(), L.stream_ptr()),
        "bn_act_bwd",
    )
    return draw, dgamma, dbeta


SHAPES = [(2, 512, 20, 20), (4, 256, 40, 40), (4, 64, 80, 80), (2, 32, 160, 160), (3, 192, 17, 23), (1, 8, 5, 7)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", SHAPES)
def test_bn_backward_tail_under_load_every_word(shape, dtype):
    """the last-arriver final pass against float64, 40 launches on the same workspace with 4 different inputs in rotation, a copy stream beside"""
    from improving_yolov8_cbam_swinblock_amd import _lib as L

    dev = torch.device("cuda:0")
    n, c, h, w = shape
    L.set_option("bn_tail", 1)  # (not the default: profiles/r05_bn_tail_ab.txt)
    gen = torch.Generator(device="cpu").manual_seed(n * 1000 + c)
    sets = []
    for k in range(4):
        raw = L.empty_nhwc(n, c, h, w, dtype, dev)
        dout = L.empty_nhwc(n, c, h, w, dtype, dev)
        raw.copy_(torch.randn(n, c, h, w, generator=gen) * (1.0 + k))
        dout.copy_(torch.randn(n, c, h, w, generator=gen) * (0.1 * (k + 1)))
        gamma = (torch.rand(c, generator=gen) + 0.5).to(dev)
        beta = (torch.randn(c, generator=gen) * 0.2).to(dev)
        mean = raw.float().mean((0, 2, 3))
        inv = torch.rsqrt(raw.float().var((0, 2, 3), unbiased=False) + 1e-3)
        ref = _bn_bwd_reference(dout, raw, gamma, beta, mean, inv)
        sets.append((dout, raw, gamma, beta, mean, inv, ref))
    ws = torch.empty(2048 * 2 * c * 4 + 256, dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream()
    junk_a = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    junk_b = torch.empty_like(junk_a)
    outs = []
    try:
        for it in range(40):
            if it % 3 != 2:  # uneven: two launches in three share the chip with a 64 MB copy
                with torch.cuda.stream(side):
                    junk_b.copy_(junk_a)
            k = (it * 7 + it // 5) % 4
            outs.append((k, _call_bn_bwd(*sets[k][:6], ws)))
        torch.cuda.synchronize()
    finally:
        L.set_option("bn_tail", 0)
    # f32 accumulation of <= 2000-term chains: <= 1e-5 of the sum of magnitudes; one stale partial row of 512 would be ~2e-3 of it
    for it, (k, (draw, dgamma, dbeta)) in enumerate(outs):
        rdraw, rdgamma, rdbeta, mag_g, mag_b = sets[k][6]
        assert ((dgamma.double() - rdgamma).abs() <= 1e-5 * mag_g + 1e-30).all(), f"dgamma, launch {it}"
        assert ((dbeta.double() - rdbeta).abs() <= 1e-5 * mag_b + 1e-30).all(), f"dbeta, launch {it}"
        dtol = 1e-4 if dtype == torch.float32 else 2e-2  # (bf16: the stored rounding)
        assert (draw.double() - rdraw).abs().max().item() <= dtol * (rdraw.abs().max().item() + 1e-6), f"draw, launch {it}"


@pytest.mark.parametrize("shape", SHAPES[:4])
def test_bn_backward_tail_is_deterministic_and_matches_the_final_launch(shape):
    """same inputs, 30 launches under load: bit-identical every time (fixed rows in fixed order whoever arrives last); and against the
    separate final launch (option bn_tail = 0) within float32 rounding of the sums"""
    from improving_yolov8_cbam_swinblock_amd import _lib as L

    dev = torch.device("cuda:0")
    n, c, h, w = shape
    dtype = torch.bfloat16
    raw = L.empty_nhwc(n, c, h, w, dtype, dev)
    dout = L.empty_nhwc(n, c, h, w, dtype, dev)
    raw.copy_(torch.randn(n, c, h, w))
    dout.copy_(torch.randn(n, c, h, w) * 0.3)
    gamma = torch.rand(c, device=dev) + 0.5
    beta = torch.randn(c, device=dev) * 0.2
    mean = raw.float().mean((0, 2, 3))
    inv = torch.rsqrt(raw.float().var((0, 2, 3), unbiased=False) + 1e-3)
    ws = torch.empty(2048 * 2 * c * 4 + 256, dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream()
    junk_a = torch.empty(32 << 20, dtype=torch.uint8, device=dev)
    junk_b = torch.empty_like(junk_a)
    first = None
    L.set_option("bn_tail", 1)
    try:
        for it in range(30):
            if it % 2:
                with torch.cuda.stream(side):
                    junk_b.copy_(junk_a)
            ws.random_(0, 255)  # poison: nothing may depend on what an earlier launch left behind
            got = _call_bn_bwd(dout, raw, gamma, beta, mean, inv, ws)
            if first is None:
                first = got
            else:
                for a, b in zip(first, got):
                    assert torch.equal(a, b), f"launch {it} differs from launch 0"
    finally:
        L.set_option("bn_tail", 0)
    sep = _call_bn_bwd(dout, raw, gamma, beta, mean, inv, ws)
    torch.cuda.synchronize()
    for name, a, b in zip(("draw", "dgamma", "dbeta"), first, sep):
        err = (a.double() - b.double()).abs().max().item()
        bound = (1e-2 if name == "draw" else 1e-5) * (b.double().abs().max().item() + 1e-6)  # draw: one bf16 step where a sum's last bit differs
        assert err <= bound, f"{name}: tail vs final launch {err} > {bound}"
