"""SwinBlock of the fork (reference: ultralytics/nn/modules/swin_block.py)."""
import torch.nn as nn

from ... import ops


def window_partition(x, window_size):
    """[B, H, W, C] -> [B*nW, ws*ws, C] (reference swin_block.py:8-13); H, W multiples of window_size."""
    b, h, w, c = x.shape
    xi = ops.to_internal(x.permute(0, 3, 1, 2))
    return ops.window_partition(xi, window_size).view(-1, window_size * window_size, c)


def window_reverse(windows, window_size, H, W):
    """inverse of window_partition (reference swin_block.py:15-20) -> [B, H, W, C]."""
    c = windows.shape[-1]
    b = int(windows.shape[0] / (H * W / window_size / window_size))
    t = windows.reshape(-1, c)
    return ops.window_reverse(t, b, H, W, window_size).permute(0, 2, 3, 1)


class SwinBlock(nn.Module):
    """LN1 -> windowed MHA -> skip from the NORMALISED tokens -> LN2 -> MLP(GELU) -> skip
    (reference swin_block.py:23-58; no shift, no relative-position bias, no mask: pad tokens are keys).

    Kernel chain: gather+LN1 | QKV GEMM | window attention | out-proj GEMM (+skip) | LN2 |
    fc1 GEMM | GELU | fc2 GEMM (+skip) | scatter+crop.
    """

    def __init__(self, dim, num_heads=2, window_size=7):
        super().__init__()
        self.dim = dim
        self.window_size = window_size
        self.norm1 = nn.LayerNorm(dim)
        self.attn = nn.MultiheadAttention(embed_dim=dim, num_heads=num_heads, batch_first=True)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = nn.Sequential(nn.Linear(dim, dim * 4), nn.GELU(), nn.Linear(dim * 4, dim))

    def forward(self, x, out=None):
        x = ops.to_internal(x)
        n, c, h, w = x.shape
        ws = self.window_size
        if c != self.dim:
            raise RuntimeError(f"Given normalized_shape=[{self.dim}], expected input with {self.dim} channels, got {c}")
        a = self.attn
        join = self.training  # t1 / t2 each feed a GEMM (or LayerNorm) and a residual: gradient sums form in those kernels
        t1 = ops.layernorm(x, self.norm1, ws)                                   # [T, C] window order, pad tokens = norm1.bias
        if join:
            ops.mark_join(t1, 2)
        qkv = ops.linear(t1, a.in_proj_weight, a.in_proj_bias)                  # [T, 3C]
        o = ops.window_attention(qkv, ws * ws, a.num_heads)                     # [T, C]
        t2 = ops.linear(o, a.out_proj.weight, a.out_proj.bias, residual=t1)     # skip from normalised tokens
        if ops.swin_ln_mlp_ok(t2, self.mlp[0]) and self.mlp[0].bias is not None and self.mlp[2].bias is not None:
            # norm2 -> fc1 -> GELU -> fc2 -> + t2 as one kernel per direction: the [T, 4C] activations never leave the registers (csrc/swin_mlp.hip)
            t3 = ops.swin_ln_mlp(t2, self.norm2, self.mlp[0], self.mlp[2])
        else:
            if join:
                ops.mark_join(t2, 2)
            u = ops.layernorm(t2, self.norm2, 0)
            t3 = ops.swin_mlp(u, self.mlp[0], self.mlp[2], residual=t2)  # fc1 -> GELU -> fc2 -> + t2, GELU inside the GEMM epilogues
        return ops.window_reverse(t3, n, h, w, ws, out)
