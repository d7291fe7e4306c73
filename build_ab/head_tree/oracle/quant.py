"""Storage-precision mode of the oracle.  TEST INFRASTRUCTURE ONLY - see oracle/__init__.py.

The reference computes and stores everything in float32.  The product's fast path stores activations and
activation gradients in bfloat16 (arithmetic and accumulation stay float32).  To check that path with a bound a
single wrong row cannot hide in, the oracle can round at the SAME storage points:

    with oracle.quant.storage(torch.bfloat16):
        y = oracle_module(x)          # forward values rounded where the product stores a tensor
        y.backward(gy)                # gradients rounded where the product stores a gradient tensor

`st(x)` marks a stored tensor: identity by default (the float32 reference arithmetic - all golden fixtures are
generated and checked in that mode); inside `storage(dtype)` it rounds the value to `dtype` in forward and the
incoming gradient to `dtype` in backward (straight-through: rounding has derivative 1).  `stf(x)` rounds in forward
only (operands that are rounded for a matrix product but whose gradient is never stored).
"""
import contextlib

import torch

_mode = {"dtype": None}


@contextlib.contextmanager
def storage(dtype):
    prev = _mode["dtype"]
    _mode["dtype"] = dtype
    try:
        yield
    finally:
        _mode["dtype"] = prev


def active():
    return _mode["dtype"]


class _RoundBoth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.dtype = dtype
        return x.to(dtype).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dtype).to(g.dtype), None


class _RoundFwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        return x.to(dtype).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g, None


def st(x):
    """a tensor the product stores (value and gradient)."""
    dt = _mode["dtype"]
    return x if dt is None else _RoundBoth.apply(x, dt)


def stf(x):
    """a value the product rounds on its way into a matrix product (forward only)."""
    dt = _mode["dtype"]
    return x if dt is None else _RoundFwd.apply(x, dt)


def round_weights_(module, dtype=torch.bfloat16):
    """round every matrix-product weight (conv / linear / in_proj, ndim >= 2) of `module` in place to `dtype`-representable
    float32 values: the product packs exactly these tensors to `dtype` before its MFMA products; vectors (BatchNorm /
    LayerNorm affine parameters, biases) stay float32 there and here."""
    with torch.no_grad():
        for p in module.parameters():
            if p.ndim >= 2:
                p.copy_(p.to(dtype).to(p.dtype))
    return module
