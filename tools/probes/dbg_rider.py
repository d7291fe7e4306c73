import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_step_safety import _small_model, _backward, _grads, rel
from improving_yolov8_cbam_swinblock_amd import ops
model, batch = _small_model()
stats = {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}
with ops.deferred_wgrad(True):
    _backward(model, batch)
ref = _grads(model)
model.zero_grad(set_to_none=True); model.load_state_dict(stats, strict=False)
with ops.deferred_wgrad(True):
    _backward(model, batch)
ref2 = _grads(model)
print("ref vs ref2 max rel:", max(rel(ref2[n], ref[n]) for n in ref))
model.zero_grad(set_to_none=True); model.load_state_dict(stats, strict=False)
with ops.deferred_wgrad(True), ops.wgrad_riders(True):
    _backward(model, batch)
got = _grads(model)
errs = [(n, rel(got[n], ref[n])) for n in ref]
for n, e in errs:
    if e > 1e-5: print(f"{n:50s} {e:.3e}")
print("n bad", sum(e > 1e-5 for _, e in errs), "of", len(errs))
