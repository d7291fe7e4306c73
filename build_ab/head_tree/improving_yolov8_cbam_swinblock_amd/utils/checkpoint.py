"""Checkpoint compatibility with the reference (state-dict subset only).

reference: ultralytics/engine/trainer.py:531-562 (save_model: the dict a `last.pt` / `best.pt` holds), nn/tasks.py:284-297
(BaseModel.load: intersect a state_dict by key and shape), :1178-1340 (attempt_load_one_weight: `ckpt["ema"] or ckpt["model"]`).

The reference pickles whole nn.Module objects (`"ema": deepcopy(self.ema.ema).half()`), which can only be read back by
importing the reference's classes and executing the pickle.  This package reads and writes the SAME dict layout but with
plain containers in the "ema" / "model" slots - {"state_dict": OrderedDict[str, Tensor], "yaml": dict, "nc": int, "names": dict} -
so everything loads with `torch.load(..., weights_only=True)` (nothing from the file is executed).  A file whose weights
entry is a pickled module is refused with an explanation; a bare state_dict (what `model.state_dict()` of the reference
gives, keys `model.N...`) is accepted as is.
"""
from collections import OrderedDict
from copy import deepcopy
from datetime import datetime

import torch

from .. import __version__

WEIGHT_SLOTS = ("ema", "model")  # order of attempt_load_one_weight (tasks.py:1318: `ckpt.get("ema") or ckpt["model"]`)


def _half_state(module_or_sd):
    sd = module_or_sd.state_dict() if hasattr(module_or_sd, "state_dict") else module_or_sd
    out = OrderedDict()
    for k, v in sd.items():
        v = v.detach().cpu()
        out[k] = v.half() if v.dtype.is_floating_point else v.clone()  # trainer.py:544 stores the EMA as fp16
    return out


def optimizer_state_to_fp16(opt_sd):
    """reference convert_optimizer_state_dict_to_fp16 (utils/torch_utils.py:736-748): float32 state tensors -> fp16 on the CPU."""
    state = {}
    for idx, st in opt_sd["state"].items():
        state[idx] = {k: (v.detach().cpu().half() if torch.is_tensor(v) and v.dtype == torch.float32 else (v.detach().cpu() if torch.is_tensor(v) else v))
                      for k, v in st.items()}
    return {"state": state, "param_groups": deepcopy(opt_sd["param_groups"])}


def _weights_entry(model, source):
    return {"state_dict": _half_state(source), "yaml": deepcopy(getattr(model, "yaml", None)), "nc": int(getattr(model.model[-1], "nc", 0)),
            "names": dict(getattr(model, "names", {}))}


def save_checkpoint(path, model, ema=None, optimizer=None, epoch=-1, best_fitness=None, train_args=None, train_metrics=None):
    """write the reference's checkpoint dict (trainer.py:536-552 keys).  `ema`: engine.optim.ModelEMA or None; when given
    the "model" slot is None exactly as the reference writes it ("resume and final checkpoints derive from EMA")."""
    ckpt = {
        "epoch": int(epoch),
        "best_fitness": best_fitness,
        "model": None if ema is not None else _weights_entry(model, model),
        "ema": _weights_entry(model, ema.ema) if ema is not None else None,
        "updates": int(ema.updates) if ema is not None else 0,
        "optimizer": optimizer_state_to_fp16(optimizer.state_dict()) if optimizer is not None else None,
        "train_args": dict(train_args or {}),
        "train_metrics": dict(train_metrics or {}),
        "train_results": {},
        "date": datetime.now().isoformat(),
        "version": __version__,
        "license": "AGPL-3.0 (https://ultralytics.com/license)",
        "docs": "https://docs.ultralytics.com",
    }
    torch.save(ckpt, path)
    return ckpt


def load_checkpoint(path, map_location="cpu"):
    """read a checkpoint without executing anything from the file -> (state_dict, ckpt dict or None).
    Accepts this package's checkpoints and bare state_dict files; refuses pickled-module checkpoints."""
    try:
        obj = torch.load(path, map_location=map_location, weights_only=True)
    except Exception as e:  # the weights-only unpickler refuses anything but tensors and plain containers
        raise RuntimeError(
            f"{path}: not loadable with weights_only=True ({type(e).__name__}).  Checkpoints written by the reference pickle whole "
            "ultralytics module objects in 'ema' / 'model'; export their state_dict with the reference (torch.save(ckpt['ema'].float()"
            ".state_dict(), f)) and load that file instead"
        ) from e
    return state_dict_of(obj), (obj if isinstance(obj, dict) and "epoch" in obj else None)


def state_dict_of(obj):
    """the weights inside `obj`: a checkpoint dict (ema first, then model), a weights entry, or a bare state_dict."""
    if isinstance(obj, dict) and any(k in obj for k in WEIGHT_SLOTS) and not all(torch.is_tensor(v) for v in obj.values()):
        for slot in WEIGHT_SLOTS:
            w = obj.get(slot)
            if w is not None:
                return state_dict_of(w)
        raise RuntimeError("checkpoint holds neither 'ema' nor 'model' weights")
    if isinstance(obj, dict) and "state_dict" in obj and isinstance(obj["state_dict"], dict):
        return obj["state_dict"]
    if isinstance(obj, dict) and obj and all(torch.is_tensor(v) for v in obj.values()):
        return obj
    if hasattr(obj, "state_dict"):
        return obj.state_dict()
    raise RuntimeError(f"cannot find a state_dict in an object of type {type(obj).__name__}")


def resume(path, model, optimizer=None, ema=None, map_location="cpu"):
    """load weights (+ optimizer momentum, EMA weights and update count) from a checkpoint of this package, as the
    reference's resume_training does (trainer.py:760-786).  Returns the checkpoint dict."""
    sd, ckpt = load_checkpoint(path, map_location)
    model.load(sd)
    if ckpt is not None:
        if ema is not None and ckpt.get("ema") is not None:
            ema.ema.load(state_dict_of(ckpt["ema"]))
            ema.updates = int(ckpt.get("updates", 0))
        if optimizer is not None and ckpt.get("optimizer") is not None:
            optimizer.load_state_dict(ckpt["optimizer"])
    return ckpt
