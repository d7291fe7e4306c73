"""MI355X-native YOLOv8-CBAM-Swin hot path: Ultralytics-style operator modules and YAML model builder
whose tensor work runs in hand-written gfx950 HIP kernels (libyolo_mi355.so, C ABI in include/ymi.h).

Drop-in surface (same names, constructor arguments and state-dict keys as the reference's
`ultralytics/nn/modules` and `ultralytics/nn/tasks.py`):

    from improving_yolov8_cbam_swinblock_amd.nn.tasks import DetectionModel
    model = DetectionModel("yolov8s.yaml", ch=3, nc=1).cuda()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss, items = model({"img": img, "batch_idx": bi, "cls": cls, "bboxes": boxes})
"""
__version__ = "0.1.0"
