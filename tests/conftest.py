import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    """-> dict of numpy arrays; keys 'w.<state-dict key>' are weights, 'g.<name>' gradients."""
    with np.load(GOLDEN / f"{name}.npz") as z:
        return {k: z[k] for k in z.files}


def golden_state(d, prefix="w."):
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in d.items() if k.startswith(prefix)}


@pytest.fixture(autouse=True)
def _seed():
    torch.manual_seed(0)
    np.random.seed(0)
    torch.set_num_threads(min(8, os.cpu_count() or 1))


LARGE_SWIN = ["swin_d384_14x14", "swin_d384_20x20", "swin_d64_ws14_20x20", "swin_d64_ws14_28x14", "swin_d384_ws14_28x28"]


def load_large_swin(name, ctor):
    """a LARGE_SWIN fixture (tests/golden/make_golden.py::swin_large_fixtures) -> (module with the fixture's weights,
    x, gy, expected).  ctor(dim, heads, ws) builds the SwinBlock under test.  Seeded cases rebuild weights and inputs
    from the seed (tests/golden_weights.py); their parameter gradients are strided samples + norms."""
    from golden_weights import seeded_inputs, seeded_state

    d = load_golden(name)
    dim, heads, ws, B, H, W, seed, seeded = (int(v) for v in d["meta"])
    m = ctor(dim, heads, ws)
    if seeded:
        m.load_state_dict(seeded_state({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed), strict=True)
        x, gy = seeded_inputs(seed, (B, dim, H, W), (B, dim, H, W))
    else:
        m.load_state_dict(golden_state(d), strict=True)
        x, gy = torch.from_numpy(d["x"]), torch.from_numpy(d["gy"])
    return m, x, gy, d, bool(seeded)


def large_swin_grad_errors(d, seeded, names, grads):
    """-> [(name, scaled max-abs error, relative-L2 error)] of parameter gradients against a LARGE_SWIN fixture."""
    from golden_weights import SAMPLE_STRIDE

    out = []
    for n, g in zip(names, grads):
        g = g.detach().float().cpu().reshape(-1)
        if not seeded:
            ref = torch.from_numpy(d["g." + n]).reshape(-1)
        elif "gfull." + n in d:
            ref = torch.from_numpy(d["gfull." + n]).reshape(-1)
        else:
            ref = torch.from_numpy(d["gsample." + n]).reshape(-1)
            nerr = abs(float(g.double().norm()) - float(d["gnorm." + n])) / max(float(d["gnorm." + n]), 1e-12)
            out.append((n + " (norm)", nerr, nerr))
            g = g[::SAMPLE_STRIDE]
        scale = max(1.0, float(ref.abs().max()))
        out.append((n, float((g - ref).abs().max()) / scale, float((g - ref).norm() / ref.norm().clamp(min=1e-12))))
    return out


def sample_errors(prefix, d, tensors):
    """compare {name: tensor} with the `grad_record`-style entries `<prefix>full.<name>` / `<prefix>sample.<name>` +
    `<prefix>norm.<name>` of fixture d -> [(name, relative L2 error of the stored elements, relative norm error)]."""
    from golden_weights import SAMPLE_STRIDE

    out = []
    for n, t in tensors.items():
        t = t.detach().float().cpu().reshape(-1)
        if f"{prefix}full.{n}" in d:
            ref = torch.from_numpy(d[f"{prefix}full.{n}"]).reshape(-1)
            nerr = 0.0
        else:
            ref = torch.from_numpy(d[f"{prefix}sample.{n}"]).reshape(-1)
            nerr = abs(float(t.double().norm()) - float(d[f"{prefix}norm.{n}"])) / max(float(d[f"{prefix}norm.{n}"]), 1e-12)
            t = t[::SAMPLE_STRIDE]
        out.append((n, float((t - ref).norm() / ref.norm().clamp(min=1e-12)), nerr))
    return out


def stored_elements(prefix, d, name):
    """the elements a `grad_record` entry of fixture d holds for tensor `name` (all of it, or the strided sample)."""
    k = f"{prefix}full.{name}"
    return torch.from_numpy(d[k] if k in d else d[f"{prefix}sample.{name}"]).reshape(-1)


def same_elements(t):
    """the elements of tensor t that `grad_record` would store."""
    from golden_weights import FULL_LIMIT, SAMPLE_STRIDE

    flat = t.detach().float().cpu().reshape(-1)
    return flat if flat.numel() <= FULL_LIMIT else flat[::SAMPLE_STRIDE]


def check_update_steps(d, init_state, states, ema_states, step_tol=(2e-3, 5e-2)):
    """the UPDATES (after - before, on the stored elements) of two optimizer steps against fixture opt_step_tiny.
    Comparing updates rather than values exposes a wrong rule (no nesterov / clip / decay: >= 0.3) that a comparison of
    the values themselves (update / value ~ 1e-3) would hide.  Bounds: step 0 sees identical weights on both sides, so the
    update agrees to the gradients' float32 agreement; step 1 starts from weights 1e-7 apart and the tiny model's 2-image
    BatchNorm over 2x2 maps amplifies that, hence the looser second bound."""
    worst = []
    for kind, seq in (("p", states), ("e", ema_states)):
        prev_ref = {n: same_elements(v) for n, v in init_state.items() if v.dtype.is_floating_point}
        prev_got = dict(prev_ref)
        for step, st in enumerate(seq):
            tot_r = tot_e = 0.0
            for n, v in st.items():
                if not v.dtype.is_floating_point:
                    continue
                ref = stored_elements(f"s{step}.{kind}", d, n)
                got = same_elements(v)
                dr, dg = ref - prev_ref[n], got - prev_got[n]
                tot_r += float(dr.double().pow(2).sum())
                tot_e += float((dg - dr).double().pow(2).sum())
                if float(dr.norm()) > 1e-7:
                    worst.append((kind, step, n, float((dg - dr).norm() / dr.norm())))
                prev_ref[n], prev_got[n] = ref, got
            err = (tot_e / max(tot_r, 1e-30)) ** 0.5
            print(f"[update steps] {'parameters' if kind == 'p' else 'EMA'} step {step}: relative error of the update {err:.2e} (bound {step_tol[step]:.0e})")
            assert err <= step_tol[step], (kind, step, err, sorted(worst, key=lambda w: -w[3])[:5])
    return worst
