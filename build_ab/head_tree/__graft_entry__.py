"""Driver entry points.

build(): compile every HIP source of the package for gfx950 into libyolo_mi355.so (hipcc cross-compiles
         without a GPU), load it, check that it exports the whole C ABI, and import the package.  The oracle
         is pure Python/PyTorch (the reference has no native code), so there is nothing else to compile.
smoke(): one small forward + loss + backward of the YOLOv8-CBAM-Swin graph on cuda:0 through the HIP
         kernels, checked against the CPU oracle (oracle/ is used here only as the checker).
"""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build():
    from improving_yolov8_cbam_swinblock_amd import _lib

    _lib.build()
    lib = _lib.lib()  # raises if any symbol declared in include/ymi.h is missing
    assert lib.ymi_version() == 1
    import improving_yolov8_cbam_swinblock_amd.nn.tasks  # noqa: F401
    import improving_yolov8_cbam_swinblock_amd.engine.trainer  # noqa: F401


def smoke():
    import json

    import torch

    from improving_yolov8_cbam_swinblock_amd import _lib
    from improving_yolov8_cbam_swinblock_amd.nn.tasks import DetectionModel
    from oracle.loss import v8DetectionLoss as OracleLoss
    from oracle.tasks import DetectionModel as OracleModel

    assert torch.cuda.is_available(), "smoke() needs the MI355X"
    _lib.lib()
    dev = torch.device("cuda:0")
    cfg = json.load(open(os.path.join(ROOT, "tests", "golden", "e2e_tiny_seed7_yaml.json")))
    torch.manual_seed(0)
    oracle = OracleModel(cfg, ch=3, nc=1)
    model = DetectionModel(cfg, ch=3, nc=1)
    model.load_state_dict(oracle.state_dict(), strict=True)
    model = model.to(dev).train()
    oracle.train()
    g = torch.Generator().manual_seed(1)
    img = torch.rand(2, 3, 64, 64, generator=g)
    batch = {
        "batch_idx": torch.tensor([0.0, 0.0, 1.0]),
        "cls": torch.zeros(3, 1),
        "bboxes": torch.tensor([[0.5, 0.5, 0.3, 0.3], [0.3, 0.6, 0.2, 0.25], [0.6, 0.4, 0.35, 0.3]]),
    }
    ref_preds = oracle(img)
    ref_loss, _ = OracleLoss(oracle)(ref_preds, batch)
    ref_loss.sum().backward()
    gbatch = {k: v.to(dev) for k, v in batch.items()}
    gbatch["img"] = img.to(dev)
    preds = model(gbatch["img"])  # float32 parity mode
    for a, b in zip(preds, ref_preds):
        err = float((a.float().cpu() - b).abs().max())
        assert err < 1e-3 * max(1.0, float(b.abs().max())), f"forward mismatch vs oracle: {err}"
    loss, _ = model.init_criterion()(preds, gbatch)
    assert torch.allclose(loss.cpu(), ref_loss, rtol=2e-3, atol=2e-3), (loss, ref_loss)
    loss.sum().backward()
    gw = model.model[0].conv.weight.grad.cpu()
    rw = oracle.model[0].conv.weight.grad
    assert float((gw - rw).norm() / rw.norm()) < 2e-2, "backward mismatch vs oracle"
    # and one bf16 training step of the fast path
    with torch.autocast("cuda", dtype=torch.bfloat16):
        l2, _ = model(gbatch)
    l2.sum().backward()
    torch.cuda.synchronize()
    assert torch.isfinite(l2).all()
    print("smoke ok: loss", [round(float(v), 4) for v in loss], "bf16", [round(float(v), 4) for v in l2])


if __name__ == "__main__":
    build()
    print("build ok")
