"""Seed-constructed tensors shared by tests/golden/make_golden.py and the tests that replay its large fixtures.

A SwinBlock(384) has 1.77 M parameters: committing them (and their gradients) as arrays would be 14 MB per case, so
those fixtures store only what the reference PRODUCED (outputs, input gradient, parameter-gradient samples and norms)
and both sides rebuild the inputs from a seed with numpy's legacy MT19937 `RandomState`, whose stream numpy guarantees
to be frozen across versions and platforms."""
import numpy as np
import torch


def seeded_array(rs, shape, scale=1.0, shift=0.0):
    return (rs.standard_normal(size=tuple(shape)).astype(np.float32) * np.float32(scale) + np.float32(shift)).astype(np.float32)


def seeded_state(shapes, seed):
    """state-dict shaped {name: shape} -> tensors: vectors ~ N(shift, 0.3) with shift 1 for *weight (norm scales),
    0 otherwise; matrices ~ N(0, 1/fan_in).  Same rule as make_golden.randomize, but reproducible from `seed`."""
    rs = np.random.RandomState(seed)
    out = {}
    for name, shape in shapes.items():
        shape = tuple(shape)
        if len(shape) <= 1:
            out[name] = torch.from_numpy(seeded_array(rs, shape, 0.3, 1.0 if name.endswith("weight") else 0.0))
        else:
            fan = int(np.prod(shape[1:]))
            out[name] = torch.from_numpy(seeded_array(rs, shape, 1.0 / np.sqrt(fan)))
    return out


def seeded_inputs(seed, x_shape, y_shape):
    rs = np.random.RandomState(seed + 1000)
    return torch.from_numpy(seeded_array(rs, x_shape)), torch.from_numpy(seeded_array(rs, y_shape))


SAMPLE_STRIDE = 97  # prime: walks every row/column phase of the big gradient matrices
FULL_LIMIT = 4096   # gradients up to this many elements are stored whole


def grad_record(t):
    """what a large fixture keeps of one gradient tensor: all of it when small, else a strided sample + its L2 norm."""
    flat = t.detach().reshape(-1)
    if flat.numel() <= FULL_LIMIT:
        return {"full": flat.numpy().copy()}
    return {"sample": flat[::SAMPLE_STRIDE].numpy().copy(), "norm": np.array(float(flat.double().norm()), dtype=np.float64)}
