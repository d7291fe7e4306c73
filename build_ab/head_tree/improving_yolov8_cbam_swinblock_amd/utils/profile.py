"""Per-module forward / backward time table (the reporting format of reference utils/torch_utils.py:792-870 `profile`,
and of BaseModel._profile_one_layer, nn/tasks.py:181-208), measured with HIP events on the launch stream.

    from improving_yolov8_cbam_swinblock_amd.utils.profile import profile_model
    rows = profile_model(model, batch["img"], n=10)        # prints the table, returns the rows
"""
import torch


def profile_model(model, x, n=10, dtype=torch.bfloat16, verbose=True):
    """time every top-level layer of a DetectionModel: forward (train mode, autocast `dtype`) and backward (gradient of the
    layer's output sum), `n` timed repetitions after one warm-up, on the layer's real input.  Returns
    [(index, type, params, input shape, forward ms, backward ms)] like the reference's `profile` table
    (Params, GFLOPs is not measured: FLOPs are in SURVEY 8d)."""
    if not x.is_cuda:
        raise RuntimeError("profile_model runs the HIP kernels: put the model and the input on the MI355X (cuda) device")
    model.train()
    rows, y = [], []
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
    cur = x
    with torch.autocast("cuda", dtype=dtype, enabled=dtype != torch.float32):
        for m in model.model:
            if m.f != -1:
                cur = y[m.f] if isinstance(m.f, int) else [cur if j == -1 else y[j] for j in m.f]
            inp = [t.detach().requires_grad_(True) for t in cur] if isinstance(cur, list) else cur.detach().requires_grad_(True)
            tf = tb = 0.0
            for it in range(n + 1):
                e0, e1, e2 = ev(), ev(), ev()
                e0.record()
                out = m(inp)
                e1.record()
                outs = out if isinstance(out, (list, tuple)) else [out]
                loss = sum(o.float().sum() for o in outs)
                has_grad = any(o.requires_grad for o in outs)
                if has_grad:
                    loss.backward()
                e2.record()
                torch.cuda.synchronize()
                if it:  # first repetition is the warm-up
                    tf += e0.elapsed_time(e1)
                    tb += e1.elapsed_time(e2) if has_grad else 0.0
                for p in m.parameters():
                    p.grad = None
            shape = [tuple(t.shape) for t in inp] if isinstance(inp, list) else tuple(inp.shape)
            rows.append((m.i, m.type.split(".")[-1], int(m.np), shape, tf / n, tb / n))
            with torch.no_grad():
                cur = m([t.detach() for t in inp] if isinstance(inp, list) else inp.detach())
            y.append(cur if m.i in model.save else None)
    if verbose:
        print(f"{'layer':>5} {'module':<12}{'params':>10}  {'forward ms':>10} {'backward ms':>11}  input")
        for i, t, np_, shape, f, b in rows:
            print(f"{i:>5} {t:<12}{np_:>10}  {f:>10.3f} {b:>11.3f}  {shape}")
        print(f"{'':>5} {'total':<12}{sum(r[2] for r in rows):>10}  {sum(r[4] for r in rows):>10.3f} {sum(r[5] for r in rows):>11.3f}")
    return rows
