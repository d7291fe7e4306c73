"""ORACLE (test infrastructure): CPU restatement of the reference's optimizer step in plain torch ops.

reference: ultralytics/engine/trainer.py:788-849 (build_optimizer, 'SGD' and 'AdamW' / 'Adam' branches), :614-622 (optimizer_step) and
ultralytics/utils/torch_utils.py:620-673 (ModelEMA).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
may import this package; the product never does.
"""
import math
from copy import deepcopy

import torch
import torch.nn as nn


def build_optimizer(model, lr=0.01, momentum=0.937, decay=5e-4, name="SGD"):
    """trainer.py:814-843: g[2] biases (no decay) first, then add_param_group(g[0] weights, decay), (g[1] norm weights);
    name: 'SGD' (:832-833), 'AdamW' / 'Adam' / 'Adamax' / 'NAdam' / 'RAdam' (:829-830, betas = (momentum, 0.999)) or 'RMSProp' (:831-832)."""
    g = [], [], []  # frozen parameters ('.dfl', trainer.py:244-256) are grouped too: SGD skips tensors without a gradient
    bn = tuple(v for k, v in nn.__dict__.items() if "Norm" in k)
    for module_name, module in model.named_modules():
        for param_name, param in module.named_parameters(recurse=False):
            fullname = f"{module_name}.{param_name}" if module_name else param_name
            if "bias" in fullname:
                g[2].append(param)
            elif isinstance(module, bn) or "logit_scale" in fullname:
                g[1].append(param)
            else:
                g[0].append(param)
    if name in ("AdamW", "Adam", "Adamax", "NAdam", "RAdam"):
        opt = getattr(torch.optim, name)(g[2], lr=lr, betas=(momentum, 0.999), weight_decay=0.0)
    elif name == "RMSProp":
        opt = torch.optim.RMSprop(g[2], lr=lr, momentum=momentum)
    else:
        opt = torch.optim.SGD(g[2], lr=lr, momentum=momentum, nesterov=True)
    opt.add_param_group({"params": g[0], "weight_decay": decay})
    opt.add_param_group({"params": g[1], "weight_decay": 0.0})
    return opt


class ModelEMA:
    """torch_utils.py:636-673."""

    def __init__(self, model, decay=0.9999, tau=2000, updates=0):
        self.ema = deepcopy(model).eval()
        self.updates = updates
        self.decay = lambda x: decay * (1 - math.exp(-x / tau))
        for p in self.ema.parameters():
            p.requires_grad_(False)

    def update(self, model):
        self.updates += 1
        d = self.decay(self.updates)
        msd = model.state_dict()
        for k, v in self.ema.state_dict().items():
            if v.dtype.is_floating_point:
                v *= d
                v += (1 - d) * msd[k].detach()


def optimizer_step(model, optimizer, ema=None, max_norm=10.0):
    """trainer.py:614-622 without the GradScaler (bf16 / fp32 need none): clip -> step -> zero_grad -> EMA.
    Returns the total gradient norm before clipping."""
    norm = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=max_norm)
    optimizer.step()
    optimizer.zero_grad()
    if ema:
        ema.update(model)
    return norm
