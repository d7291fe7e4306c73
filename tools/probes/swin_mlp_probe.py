"""time SwinBlock's second half at the model's shape (56,448 tokens x 256, hidden 1024): the fused kernel against LayerNorm + the two token GEMMs"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from improving_yolov8_cbam_swinblock_amd import _lib as L
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_swin_mlp_fused import _inputs, _fused_fwd

t, hidden = 56448, 1024
x, gamma, beta, eps, w1, b1, w2, b2 = _inputs(t, hidden, 1)
lib = L.lib()
dev = x.device
packed = torch.empty(lib.ymi_swin_ln_mlp_pack_elems(256, hidden), dtype=torch.bfloat16, device=dev)
L.check(lib.ymi_swin_ln_mlp_pack(L.ptr(w1), L.ptr(w2), 256, hidden, L.ptr(packed), L.stream_ptr()), "pack")
out = torch.empty_like(x); u = torch.empty_like(x); st = torch.empty((2, t), dtype=torch.float32, device=dev)
pre = torch.empty(lib.ymi_swin_ln_mlp_pre_elems(t, hidden), dtype=torch.bfloat16, device=dev)
w1p = w1.to(torch.bfloat16).contiguous(); w2p = w2.to(torch.bfloat16).contiguous()
pre2 = torch.empty((t, hidden), dtype=torch.bfloat16, device=dev); post2 = torch.empty_like(pre2); out2 = torch.empty_like(x); u2 = torch.empty_like(x)
junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev)

def fused(train=True):
    L.check(lib.ymi_swin_ln_mlp_fwd(ctypes.byref(L.as_ymi(x)), L.ptr(gamma), L.ptr(beta), eps, L.ptr(packed), L.ptr(b1), L.ptr(b2), hidden,
                                    ctypes.byref(L.as_ymi(u)) if train else None, L.ptr(st[0]) if train else None, L.ptr(st[1]) if train else None,
                                    L.ptr(pre) if train else None, ctypes.byref(L.as_ymi(out)), L.stream_ptr()), "fwd")
def unfused():
    L.check(lib.ymi_layernorm_fwd(ctypes.byref(L.as_ymi(x)), 0, L.ptr(gamma), L.ptr(beta), eps, ctypes.byref(L.as_ymi(u2)), L.ptr(st[0]), L.ptr(st[1]), L.stream_ptr()), "ln")
    L.check(lib.ymi_swin_mlp_fwd(ctypes.byref(L.as_ymi(u2)), L.ptr(w1p), L.ptr(b1), hidden, L.ptr(w2p), L.ptr(b2), ctypes.byref(L.as_ymi(x)),
                                 ctypes.byref(L.as_ymi(pre2)), ctypes.byref(L.as_ymi(post2)), ctypes.byref(L.as_ymi(out2)), L.stream_ptr()), "mlp")
def timeit(fn, cold, n=20):
    e0 = [torch.cuda.Event(enable_timing=True) for _ in range(n)]; e1 = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    for i in range(n):
        if cold: junk.zero_()
        e0[i].record(); fn(); e1[i].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in zip(e0, e1))
    return ts[len(ts) // 2], ts[0]
dout = (torch.randn(t, 256) * 0.5).to(torch.bfloat16).to(dev)
cap = lib.ymi_swin_ln_mlp_pre_elems(t, hidden)
post = torch.empty(cap, dtype=torch.bfloat16, device=dev).view(-1, hidden)[:t]; dpre = torch.empty(cap, dtype=torch.bfloat16, device=dev).view(-1, hidden)[:t]; du = torch.empty_like(x)
w2d = w2.to(torch.bfloat16).t().contiguous()   # [hidden][C]: the data-gradient operand of fc2 ([I][tap][O] with one tap)
w1d = w1.to(torch.bfloat16).t().contiguous()   # [C][hidden]
def fused_bwd():
    L.check(lib.ymi_swin_ln_mlp_bwd_data(ctypes.byref(L.as_ymi(dout)), L.ptr(packed), L.ptr(pre), hidden, ctypes.byref(L.as_ymi(post)), ctypes.byref(L.as_ymi(dpre)),
                                         ctypes.byref(L.as_ymi(du)), L.stream_ptr()), "bwd")
def unfused_bwd():
    L.check(lib.ymi_swin_mlp_bwd_data(ctypes.byref(L.as_ymi(dout)), L.ptr(w2d), ctypes.byref(L.as_ymi(pre2)), ctypes.byref(L.as_ymi(dpre)), L.ptr(w1d), None, None,
                                      ctypes.byref(L.as_ymi(du)), L.stream_ptr()), "bwd0")
for _ in range(3): fused(); unfused(); fused_bwd(); unfused_bwd()
for cold in (False, True):
    print("cold" if cold else "warm", "fused train (median, min us):", timeit(lambda: fused(True), cold), "fused eval:", timeit(lambda: fused(False), cold), "unfused:", timeit(unfused, cold))
    print("cold" if cold else "warm", "fused bwd data:", timeit(fused_bwd, cold), "unfused bwd data:", timeit(unfused_bwd, cold))

if os.environ.get("YMI_MLP_STAMPS") == "1":  # a -DYMI_MLP_ABL=11 build: print the in-kernel phase stamps (cycles relative to the first)
    stamps = torch.zeros(64, dtype=torch.int64, device=dev)
    for _ in range(3):
        L.check(lib.ymi_swin_ln_mlp_fwd(ctypes.byref(L.as_ymi(x)), L.ptr(gamma), L.ptr(beta), eps, L.ptr(packed), L.ptr(b1), L.ptr(b2), hidden,
                                        None, L.ptr(stamps), None, None, ctypes.byref(L.as_ymi(out)), L.stream_ptr()), "fwd")
    torch.cuda.synchronize()
    v = stamps[:48].view(2, 4, 6).cpu()
    base = int(v.min())
    for hh in range(2):
        for j in range(4):
            print("half", hh, "chunk", 8 + j, "stamps:", [int(t) - base for t in v[hh, j]])
