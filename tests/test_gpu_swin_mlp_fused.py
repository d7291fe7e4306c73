"""SwinBlock's fused second half (csrc/swin_mlp.hip: LayerNorm-2 + fc1 + GELU + fc2 + skip in one kernel per direction) against the oracle's
SwinBlock arithmetic (reference nn/modules/swin_block.py:31-35,53) and against the unfused kernels it replaces."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _decode_pre(pre_priv, tokens, hidden):
    """private register-order layout [tile of 256 tokens][chunk][wave][g][half][lane32][8]: hidden unit 32 chunk + 16 half + 8 g + e of token
    256 tile + 32 wave + lane32 -> [tokens, hidden] (csrc/swin_mlp.hip)"""
    tiles = (tokens + 255) // 256
    v = pre_priv.view(tiles, hidden // 32, 8, 2, 2, 32, 8).permute(0, 2, 5, 1, 4, 3, 6).reshape(tiles * 256, hidden)
    return v[:tokens]


def _reference(x, gamma, beta, eps, w1, b1, w2, b2):
    """float64 on the bf16-rounded operands, with the product's storage roundings (u, pre, post in bf16)"""
    xd = x.double()
    mu = xd.mean(1, keepdim=True)
    var = ((xd - mu) ** 2).mean(1, keepdim=True)
    rs = 1.0 / torch.sqrt(var + eps)
    u = ((xd - mu) * rs * gamma.double() + beta.double()).to(torch.bfloat16)
    w1b, w2b = w1.to(torch.bfloat16).double(), w2.to(torch.bfloat16).double()
    pre = (u.double() @ w1b.t() + b1.double()).to(torch.bfloat16)
    post = torch.nn.functional.gelu(pre.double()).to(torch.bfloat16)
    out = post.double() @ w2b.t() + b2.double() + xd
    return u, mu[:, 0], rs[:, 0], pre, out


def _fused_fwd(x, gamma, beta, eps, w1, b1, w2, b2, train=True):
    from improving_yolov8_cbam_swinblock_amd import _lib as L

    lib = L.lib()
    t, c = x.shape
    hidden = w1.shape[0]
    dev = x.device
    assert lib.ymi_swin_ln_mlp_supported(c, hidden, L.YMI_BF16)
    packed = torch.empty(lib.ymi_swin_ln_mlp_pack_elems(c, hidden), dtype=torch.bfloat16, device=dev)
    L.check(lib.ymi_swin_ln_mlp_pack(L.ptr(w1), L.ptr(w2), c, hidden, L.ptr(packed), L.stream_ptr()), "pack")
    out = torch.empty_like(x)
    u = torch.empty_like(x) if train else None
    stats = torch.empty((2, t), dtype=torch.float32, device=dev) if train else None
    pre = torch.empty(lib.ymi_swin_ln_mlp_pre_elems(t, hidden), dtype=torch.bfloat16, device=dev) if train else None
    L.check(
        lib.ymi_swin_ln_mlp_fwd(ctypes.byref(L.as_ymi(x)), L.ptr(gamma), L.ptr(beta), eps, L.ptr(packed), L.ptr(b1), L.ptr(b2), hidden,
                                ctypes.byref(L.as_ymi(u)) if train else None, L.ptr(stats[0]) if train else None, L.ptr(stats[1]) if train else None,
                                L.ptr(pre), ctypes.byref(L.as_ymi(out)), L.stream_ptr()),
        "swin_ln_mlp_fwd",
    )
    return out, u, stats, pre, packed


def _inputs(t, hidden, seed):
    g = torch.Generator().manual_seed(seed)
    dev = torch.device("cuda:0")
    c = 256
    x = (torch.randn(t, c, generator=g) * 1.5 + 0.3).to(torch.bfloat16).to(dev)
    gamma = (torch.rand(c, generator=g) + 0.5).to(dev)
    beta = (torch.randn(c, generator=g) * 0.3).to(dev)
    w1 = (torch.randn(hidden, c, generator=g) / 16).to(dev)
    b1 = (torch.randn(hidden, generator=g) * 0.2).to(dev)
    w2 = (torch.randn(c, hidden, generator=g) / (hidden ** 0.5)).to(dev)
    b2 = (torch.randn(c, generator=g) * 0.2).to(dev)
    return x, gamma, beta, 1e-5, w1, b1, w2, b2


@pytest.mark.parametrize("t,hidden", [(128, 32), (256, 64), (421, 1024), (640, 64), (56448 // 8, 1024)])
def test_fused_forward_against_float64(t, hidden):
    args = _inputs(t, hidden, t + hidden)
    out, u, stats, pre, _ = _fused_fwd(*args)
    torch.cuda.synchronize()
    ru, rmu, rrs, rpre, rout = _reference(*args)
    assert (stats[0].double() - rmu).abs().max().item() < 1e-5
    assert ((stats[1].double() - rrs) / rrs).abs().max().item() < 1e-5
    # u: the same value up to ONE bf16 step where the f32 LayerNorm arithmetic rounds the other way
    du = (u.double() - ru.double()).abs()
    assert (du <= 2.0 ** -7 * ru.double().abs() + 1e-6).all() and (du > 0).double().mean().item() < 0.02
    # pre: K = 256 f32 accumulation of bf16 products against float64 + the bf16 rounding; a wrong lane / permutation would be O(1)
    dp = (_decode_pre(pre, t, hidden).double() - rpre.double()).abs()
    assert dp.max().item() <= 2.0 ** -6 * rpre.double().abs().max().item()
    assert (dp > 2.0 ** -8 * (rpre.double().abs() + 0.05)).double().mean().item() < 0.02
    err = (out.double() - rout).abs().max().item()
    assert err <= 2e-2 * rout.abs().max().item(), f"out: {err}"
    rel = ((out.double() - rout).norm() / rout.norm()).item()
    assert rel < 4e-3, f"out relative L2 {rel}"  # (the SwinBlock matched-oracle forward bound)


def test_fused_forward_matches_the_unfused_kernels():
    """the kernels it replaces (ymi_layernorm_fwd + ymi_swin_mlp_fwd) on the same operands: same rounding points, different accumulation order"""
    from improving_yolov8_cbam_swinblock_amd import _lib as L

    t, hidden = 1152, 1024
    x, gamma, beta, eps, w1, b1, w2, b2 = _inputs(t, hidden, 7)
    out, u, stats, pre, _ = _fused_fwd(x, gamma, beta, eps, w1, b1, w2, b2)
    out_eval, _, _, _, _ = _fused_fwd(x, gamma, beta, eps, w1, b1, w2, b2, train=False)
    lib = L.lib()
    u2 = torch.empty_like(x)
    st2 = torch.empty((2, t), dtype=torch.float32, device=x.device)
    L.check(lib.ymi_layernorm_fwd(ctypes.byref(L.as_ymi(x)), 0, L.ptr(gamma), L.ptr(beta), eps, ctypes.byref(L.as_ymi(u2)), L.ptr(st2[0]), L.ptr(st2[1]), L.stream_ptr()), "ln")
    w1p = w1.to(torch.bfloat16).contiguous()
    w2p = w2.to(torch.bfloat16).contiguous()
    pre2 = torch.empty((t, hidden), dtype=torch.bfloat16, device=x.device)
    post2 = torch.empty_like(pre2)
    out2 = torch.empty_like(x)
    L.check(lib.ymi_swin_mlp_fwd(ctypes.byref(L.as_ymi(u2)), L.ptr(w1p), L.ptr(b1), hidden, L.ptr(w2p), L.ptr(b2), ctypes.byref(L.as_ymi(x)),
                                 ctypes.byref(L.as_ymi(pre2)), ctypes.byref(L.as_ymi(post2)), ctypes.byref(L.as_ymi(out2)), L.stream_ptr()), "mlp")
    torch.cuda.synchronize()
    assert torch.equal(out, out_eval)  # the training variant only stores more
    assert (u.float() - u2.float()).abs().max().item() <= 2.0 ** -7 * u2.float().abs().max().item()
    assert ((out.float() - out2.float()).norm() / out2.float().norm()).item() < 3e-3
    assert ((_decode_pre(pre, t, hidden).float() - pre2.float()).norm() / pre2.float().norm()).item() < 3e-3


@pytest.mark.parametrize("t,hidden", [(128, 32), (421, 1024), (56448 // 8, 1024)])
def test_fused_backward_data_path_against_float64(t, hidden):
    """post = gelu(pre) (bit for bit the forward's fc2 operand), d_pre = bf16(d_out W2) * gelu'(pre), d_u = d_pre W1"""
    from improving_yolov8_cbam_swinblock_amd import _lib as L

    x, gamma, beta, eps, w1, b1, w2, b2 = _inputs(t, hidden, 3 * t + hidden)
    out, u, stats, pre, packed = _fused_fwd(x, gamma, beta, eps, w1, b1, w2, b2)
    g = torch.Generator().manual_seed(t)
    dout = (torch.randn(t, 256, generator=g) * 0.5).to(torch.bfloat16).to(x.device)
    # post / dpre: the first t rows of buffers padded to whole 256-token tiles (the kernel stores every row of a tile)
    cap = L.lib().ymi_swin_ln_mlp_pre_elems(t, hidden)
    post = torch.full((cap,), float("nan"), dtype=torch.bfloat16, device=x.device).view(-1, hidden)[:t]
    dpre = torch.full((cap,), float("nan"), dtype=torch.bfloat16, device=x.device).view(-1, hidden)[:t]
    du = torch.empty_like(x)
    L.check(L.lib().ymi_swin_ln_mlp_bwd_data(ctypes.byref(L.as_ymi(dout)), L.ptr(packed), L.ptr(pre), hidden, ctypes.byref(L.as_ymi(post)), ctypes.byref(L.as_ymi(dpre)),
                                             ctypes.byref(L.as_ymi(du)), L.stream_ptr()), "bwd_data")
    torch.cuda.synchronize()
    prem = _decode_pre(pre, t, hidden)
    pd = prem.double()
    rpost = torch.nn.functional.gelu(pd)
    w1b, w2b = w1.to(torch.bfloat16).double(), w2.to(torch.bfloat16).double()
    dpost = (dout.double() @ w2b).to(torch.bfloat16).double()
    grad = 0.5 * (1 + torch.erf(pd / 2 ** 0.5)) + pd * torch.exp(-pd * pd / 2) / (2 * torch.pi) ** 0.5
    rdpre = dpost * grad
    rdu = rdpre.to(torch.bfloat16).double() @ w1b
    # post: one bf16 step at most (the erf approximation's 1.5e-7 moves a rounding now and then)
    dpo = (post.double() - rpost).abs()
    assert (dpo <= 2.0 ** -7 * rpost.abs() + 1e-6).all()
    e_dpre = ((dpre.double() - rdpre).norm() / rdpre.norm()).item()
    assert e_dpre < 4e-3, f"dpre {e_dpre}"
    assert (dpre.double() - rdpre).abs().max().item() <= 3e-2 * rdpre.abs().max().item()
    e_du = ((du.double() - rdu).norm() / rdu.norm()).item()
    assert e_du < 6e-3, f"du {e_du}"
    assert (du.double() - rdu).abs().max().item() <= 3e-2 * rdu.abs().max().item()
