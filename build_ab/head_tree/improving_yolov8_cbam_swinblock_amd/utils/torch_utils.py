"""step-path utilities with the reference's names (ultralytics/utils/torch_utils.py)."""
import torch
import torch.nn as nn


def initialize_weights(model):
    """BN eps / momentum and in-place activations as the reference sets them (torch_utils.py:462-472)."""
    for m in model.modules():
        t = type(m)
        if t is nn.BatchNorm2d:
            m.eps = 1e-3
            m.momentum = 0.03
        elif t in {nn.Hardswish, nn.LeakyReLU, nn.ReLU, nn.ReLU6, nn.SiLU}:
            m.inplace = True


def fuse_conv_and_bn(conv, bn):
    """fold BN running statistics into a biased conv (reference torch_utils.py:240-271)."""
    fused = (
        nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding, conv.dilation, conv.groups, bias=True)
        .requires_grad_(False)
        .to(conv.weight.device)
    )
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    fused.weight.copy_(conv.weight * scale.view(-1, 1, 1, 1))
    b0 = torch.zeros_like(bn.running_mean) if conv.bias is None else conv.bias
    fused.bias.copy_((b0 - bn.running_mean) * scale + bn.bias)
    return fused


def intersect_dicts(da, db, exclude=()):
    """reference torch_utils.py: keys of da present in db with equal shapes."""
    return {k: v for k, v in da.items() if k in db and all(x not in k for x in exclude) and v.shape == db[k].shape}
