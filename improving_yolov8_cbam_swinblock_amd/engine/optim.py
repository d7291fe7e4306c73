"""Optimizer step of the reference trainer as multi-tensor HIP launches (csrc/optim.hip).

reference: ultralytics/engine/trainer.py:788-849 (build_optimizer: biases / decayed weights / norm weights, SGD with
nesterov momentum), :614-622 (optimizer_step: clip_grad_norm_(10.0) -> step -> zero_grad -> EMA update) and
utils/torch_utils.py:620-685 (ModelEMA).

`FusedSGD` keeps torch.optim.SGD's surface (`param_groups` with lr / momentum / weight_decay / nesterov that schedulers
may rewrite, `state_dict()` / `load_state_dict()` in torch.optim.SGD's format, `zero_grad`) but `step()` is three
launches over every parameter: global gradient norm, clip coefficient (+ EMA decay, update counter), update (+ EMA).
Hyper-parameters live in a device array that is rewritten only when `param_groups` change, so a HIP graph that captured
`step()` follows the learning-rate schedule.
"""
import ctypes
import math
from copy import deepcopy

import torch
import torch.nn as nn

from .. import _lib
from .._lib import OPT_MAX_GRADS, OptEntry, check, stream_ptr


def param_groups_of(model):
    """(biases, decayed weights, norm weights) by the reference's rule (trainer.py:815-825): 'bias' in the full
    parameter name -> no decay; parameters of normalisation layers -> no decay; everything else decays.  Frozen parameters (the DFL
    projection) are grouped as well, exactly as the reference does: tensors without a gradient are skipped by the step."""
    g_bias, g_w, g_norm = [], [], []
    norm = tuple(v for k, v in nn.__dict__.items() if "Norm" in k)
    for module_name, module in model.named_modules():
        for param_name, p in module.named_parameters(recurse=False):
            fullname = f"{module_name}.{param_name}" if module_name else param_name
            if "bias" in fullname:
                g_bias.append(p)
            elif isinstance(module, norm) or "logit_scale" in fullname:
                g_norm.append(p)
            else:
                g_w.append(p)
    return g_bias, g_w, g_norm


class ModelEMA:
    """exponential moving average of every floating-point entry of the model's state_dict (parameters AND buffers),
    reference utils/torch_utils.py:620-685: decay(x) = decay * (1 - exp(-x / tau)), ema = d*ema + (1-d)*model.
    `update()` alone is one multi-tensor launch; attached to a FusedSGD it rides in the optimizer's update pass."""

    def __init__(self, model, decay=0.9999, tau=2000, updates=0):
        arena = model.__dict__.pop("_arena", None)  # packed-operand cache of the live model: not part of its state
        try:
            self.ema = deepcopy(model).eval()
        finally:
            if arena is not None:
                model.__dict__["_arena"] = arena
        self.updates = updates
        self.decay_max, self.tau = float(decay), float(tau)
        self.decay = lambda x: decay * (1 - math.exp(-x / tau))
        for p in self.ema.parameters():
            p.requires_grad_(False)
        self.enabled = True
        self._own = None  # FusedSGD used for stand-alone updates

    def pairs(self, model):
        """[(model tensor, ema tensor)] over floating-point state_dict entries, in state_dict order."""
        msd, esd = model.state_dict(), self.ema.state_dict()
        return [(msd[k], v) for k, v in esd.items() if v.dtype.is_floating_point]

    def update(self, model):
        if not self.enabled:
            return
        if self._own is None or self._own.model is not model:
            self._own = FusedSGD(model, lr=0.0, ema=self, sgd=False)
        self._own.step()

    def update_attr(self, model, include=(), exclude=("process_group", "reducer")):
        if self.enabled:
            for k, v in model.__dict__.items():
                if (len(include) and k not in include) or k.startswith("_") or k in exclude:
                    continue
                setattr(self.ema, k, v)


class FusedSGD:
    """SGD(momentum, nesterov) over the reference's three parameter groups + gradient clipping (+ EMA), fused."""

    RULE = 0  # csrc/optim.hip hyper[14]: 0 SGD-momentum, 1 AdamW, 2 Adam, 3 Adamax, 4 NAdam, 5 RAdam, 6 RMSprop

    def __init__(self, model, lr=0.01, momentum=0.937, decay=5e-4, nesterov=True, max_norm=10.0, ema=None, sgd=True):
        self.model = model
        g_bias, g_w, g_norm = param_groups_of(model) if sgd else ([], [], [])
        # group order of the reference's optimizer: [biases] + add_param_group(weights, decay) + add_param_group(norm weights)
        self.param_groups = [
            {"params": g_bias, "lr": lr, "initial_lr": lr, "momentum": momentum, "nesterov": nesterov, "weight_decay": 0.0, "dampening": 0},
            {"params": g_w, "lr": lr, "initial_lr": lr, "momentum": momentum, "nesterov": nesterov, "weight_decay": decay, "dampening": 0},
            {"params": g_norm, "lr": lr, "initial_lr": lr, "momentum": momentum, "nesterov": nesterov, "weight_decay": 0.0, "dampening": 0},
        ]
        self.max_norm = float(max_norm) if max_norm else 0.0
        self.world = 1  # gradients are scaled by 1 / world inside the kernels (hyper[11]): TrainStep hands over the SUM over ranks and sets this
        self.ema = ema
        self.sgd = sgd
        self._state = None
        self._table = None
        self._hyper_host = None
        self._keep = []

    # ---- device tables ---------------------------------------------------------------------------------------------
    def _build(self):
        dev = next(self.model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("FusedSGD runs in libyolo_mi355 kernels: the model must be on the MI355X (cuda) device; there is no CPU path")
        ema_of = {}
        if self.ema is not None:
            for mt, et in self.ema.pairs(self.model):
                ema_of[mt.data_ptr()] = et
        old = dict(zip((id(p) for p in getattr(self, "params", [])), getattr(self, "momentum", [])))  # rebuild after model.to(): keep the momentum
        self.params, entries = [], []
        self.momentum = []
        for gi, grp in enumerate(self.param_groups):
            for p in grp["params"]:
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("FusedSGD updates contiguous float32 parameters")
                buf = old[id(p)].to(p.device) if id(p) in old and old[id(p)].shape == p.shape else torch.zeros_like(p)
                self.params.append(p)
                self.momentum.append(buf)
                e = ema_of.pop(p.data_ptr(), None)
                entries.append((p, buf, e, gi))
        old2 = dict(zip(old.keys(), getattr(self, "second", [])))
        self.second = []  # Adam / AdamW: exp_avg_sq per parameter
        if self.RULE:
            for p in self.params:
                self.second.append(old2[id(p)].to(p.device) if id(p) in old2 and old2[id(p)].shape == p.shape else torch.zeros_like(p))
        self.n_sgd = len(entries)
        # EMA-only entries: buffers (BN running statistics) and frozen parameters
        self.ema_only = []
        if self.ema is not None:
            for mt, et in self.ema.pairs(self.model):
                if mt.data_ptr() in ema_of:
                    ema_of.pop(mt.data_ptr())
                    if mt.dtype != torch.float32 or et.dtype != torch.float32:
                        raise RuntimeError("ModelEMA entries must be float32")
                    entries.append((mt, None, et, 0))
                    self.ema_only.append(mt)
        chunk = int(_lib.lib().ymi_opt_chunk_elems())
        tab = (OptEntry * len(entries))()
        cmap, self.ranges = [], []  # ranges: (first tensor, n tensors, first chunk, n chunks, has grads)
        first_chunk_of = []
        for i, (p, buf, e, gi) in enumerate(entries):
            sec = self.second[i].data_ptr() if self.RULE and i < self.n_sgd else None
            tab[i] = OptEntry(p.data_ptr(), buf.data_ptr() if buf is not None else None, e.data_ptr() if e is not None else None, p.numel(), gi, 0, sec, None)
            first_chunk_of.append(len(cmap))
            for c in range((p.numel() + chunk - 1) // chunk):
                cmap.append((i, c))
        first_chunk_of.append(len(cmap))
        # sgd=False (stand-alone ModelEMA.update): every entry goes through the gradient path with no gradients, so the
        # counter / decay kernel runs and the update kernel leaves parameters alone
        for lo, hi, has in ((0, self.n_sgd, True), (self.n_sgd, len(entries), not self.sgd)):
            for a in range(lo, hi, OPT_MAX_GRADS):
                b = min(a + OPT_MAX_GRADS, hi)
                self.ranges.append((a, b - a, first_chunk_of[a], first_chunk_of[b] - first_chunk_of[a], has))
        self._entries = entries
        self._table = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(dev)
        self._cmap = torch.tensor(cmap, dtype=torch.int32).reshape(-1, 2).contiguous().to(dev)
        self.n_grad_chunks = first_chunk_of[self.n_sgd] if self.sgd else len(cmap)
        self._partials = torch.zeros(max(self.n_grad_chunks, 1), dtype=torch.float32, device=dev)
        had = getattr(self, "_state", None) is not None
        steps = int(self._state.view(torch.int64)[3]) if had else 0  # rebuild: keep Adam's t ...
        mu_product = float(self._state.view(torch.float64)[7]) if had else 0.0  # ... and NAdam's running product
        self._state = torch.zeros(64, dtype=torch.uint8, device=dev)
        if steps:
            self._state.view(torch.int64)[3] = steps
            self._state.view(torch.float64)[7] = mu_product
        self._dev_updates = 0  # host mirror of the device-side EMA update counter (_state[2]); see _sync_updates
        self._hyper = torch.zeros(20, dtype=torch.float32, device=dev)
        self._hyper_host = None  # a fresh device array: the next sync_hyper() must fill it
        self._ptrs = [p.data_ptr() for p, *_ in entries]

    def _stale(self):
        return self._table is None or any(p.data_ptr() != q for (p, *_), q in zip(self._entries, self._ptrs))

    def sync_hyper(self):
        """push lr / momentum / weight decay to the device if a scheduler changed them (call before replaying a graph
        that captured step(); step() calls it itself)."""
        g = self.param_groups
        ema = self.ema
        host = [g[0]["lr"], g[1]["lr"], g[2]["lr"], g[0]["weight_decay"], g[1]["weight_decay"], g[2]["weight_decay"], self._beta1(),
                self.max_norm, ema.decay_max if ema is not None else 0.0, ema.tau if ema is not None else 1.0, 1.0 if g[0].get("nesterov") else 0.0,
                1.0 / self.world, *self._rule_hyper(), *self._rule_extra()]
        host = [float(v) for v in host]
        if host != self._hyper_host:
            self._hyper.copy_(torch.tensor(host, dtype=torch.float32))
            self._hyper_host = host
        self._sync_updates()

    def _beta1(self):
        g = self.param_groups
        if any(grp["momentum"] != g[0]["momentum"] or grp["nesterov"] != g[0]["nesterov"] for grp in g):
            raise RuntimeError("FusedSGD: momentum / nesterov are shared by the three groups (as the reference sets them)")
        return g[0]["momentum"]

    def _rule_hyper(self):
        """hyper[12..16]: beta2, eps, rule, 1 - beta2, 1 - beta1."""
        return 0.0, 0.0, float(self.RULE), 0.0, 0.0

    def _rule_extra(self):
        """hyper[17..19]."""
        return 0.0, 0.0, 0.0

    def _sync_updates(self):
        """the kernel derives the EMA decay from a DEVICE counter; `ema.updates` is the host's view of it.  A caller may set
        the host value at any time - checkpoint.resume, or the reference's idiom `ema.updates = ckpt["updates"]`
        (trainer.py:771) - possibly after the tables were built: push it whenever it differs from what the device holds
        (runs before every step and before every graph replay, outside captured regions)."""
        if self.ema is not None and self._table is not None and int(self.ema.updates) != self._dev_updates:
            self._state.view(torch.int64)[2] = int(self.ema.updates)
            self._dev_updates = int(self.ema.updates)

    def count_updates(self, delta):
        """bookkeeping of HIP-graph replays (engine.trainer.TrainStep): a replayed step advanced the device counter (+1); a
        captured, not executed, step did not (-1).  Keeps the host count and the mirror of the device counter together."""
        if self.ema is not None:
            self.ema.updates += delta
            self._dev_updates += delta

    # ---- the step --------------------------------------------------------------------------------------------------
    def step(self, grads_of=None):
        """grads_of: optional {param: gradient tensor} (default: p.grad).  Parameters without a gradient keep their
        value and momentum (torch.optim.SGD skips them) but still enter the EMA."""
        if self._stale():
            self._build()
        self.sync_hyper()
        L = _lib.lib()
        tab, cmap = ctypes.c_void_p(self._table.data_ptr()), self._cmap
        st = stream_ptr()
        grad_arrays = []
        for first, n, c0, nc, has in self.ranges:
            if not has:
                grad_arrays.append(None)
                continue
            arr = (ctypes.c_void_p * n)()
            for i in range(n if first < self.n_sgd else 0):
                p = self.params[first + i]
                g = grads_of.get(p) if grads_of is not None else p.grad
                if g is not None:
                    if g.dtype != torch.float32 or not g.is_contiguous() or g.shape != p.shape:
                        g = g.to(torch.float32).contiguous()
                        self._keep.append(g)
                    arr[i] = g.data_ptr()
            grad_arrays.append(arr)
        last_grad = max((i for i, r in enumerate(self.ranges) if r[4]), default=-1)
        for i, (first, n, c0, nc, has) in enumerate(self.ranges):
            if has and nc:
                check(L.ymi_opt_grad_norm(tab, ctypes.c_void_p(cmap.data_ptr() + c0 * 8), first, n, nc, grad_arrays[i], ctypes.c_void_p(self._hyper.data_ptr()),
                                          ctypes.c_void_p(self._partials.data_ptr()), c0, self.n_grad_chunks, ctypes.c_void_p(self._state.data_ptr()),
                                          1 if i == last_grad else 0, st), "opt_grad_norm")
        if last_grad < 0:
            raise RuntimeError("FusedSGD.step(): nothing to update")
        for i, (first, n, c0, nc, has) in enumerate(self.ranges):
            if nc:
                check(L.ymi_opt_update(tab, ctypes.c_void_p(cmap.data_ptr() + c0 * 8), first, n, nc, grad_arrays[i] if has else None,
                                       ctypes.c_void_p(self._hyper.data_ptr()), ctypes.c_void_p(self._state.data_ptr()), int(self.RULE), st), "opt_update")
        self._keep.clear()
        if self.ema is not None:
            self.ema.updates += 1
            self._dev_updates += 1

    def grad_norm(self):
        """total gradient norm of the last step (before clipping), as clip_grad_norm_ returns it: one device->host read."""
        return float(self._state.view(torch.float32)[1])

    def zero_grad(self, set_to_none=True):
        for grp in self.param_groups:
            for p in grp["params"]:
                if set_to_none:
                    p.grad = None
                elif p.grad is not None:
                    p.grad.zero_()

    # ---- torch.optim.SGD-compatible state (checkpoint contract: trainer.py:546 stores optimizer.state_dict()) ---------
    def state_dict(self):
        if self._table is None:
            self._build()
        state, groups, idx = {}, [], 0
        for grp in self.param_groups:
            ids = []
            for _ in grp["params"]:
                state[idx] = {"momentum_buffer": self.momentum[idx]}
                ids.append(idx)
                idx += 1
            groups.append({k: v for k, v in grp.items() if k != "params"} | {"params": ids, "maximize": False, "foreach": None, "differentiable": False, "fused": None})
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        if self._table is None:
            self._build()
        for grp, saved in zip(self.param_groups, sd["param_groups"]):
            for k in ("lr", "initial_lr", "momentum", "nesterov", "weight_decay"):
                if k in saved:
                    grp[k] = saved[k]
        for idx, st in sd["state"].items():
            buf = st.get("momentum_buffer")
            if buf is not None:
                self.momentum[int(idx)].copy_(buf.to(self.momentum[int(idx)].dtype))


class FusedAdamW(FusedSGD):
    """torch.optim.AdamW over the reference's three parameter groups (trainer.py:829-830 `optim.AdamW(g[2], lr=lr,
    betas=(momentum, 0.999), weight_decay=0.0)` + the two add_param_group calls; what optimizer='auto' picks for runs of at most
    10000 iterations, :812) + gradient clipping (+ EMA), in the same three launches as FusedSGD.  `decoupled=False` is
    torch.optim.Adam (weight decay added to the gradient).  `param_groups` carry torch's Adam keys (lr, betas, eps, weight_decay);
    state_dict() / load_state_dict() use torch.optim.Adam's layout (step, exp_avg, exp_avg_sq per parameter)."""

    RULE = 1
    SECOND_KEY = "exp_avg_sq"  # (Adamax: exp_inf)

    def __init__(self, model, lr=0.001, betas=(0.9, 0.999), eps=1e-8, decay=5e-4, max_norm=10.0, ema=None, decoupled=True):
        super().__init__(model, lr=lr, momentum=betas[0], decay=decay, nesterov=False, max_norm=max_norm, ema=ema)
        self.RULE = 1 if decoupled else 2
        for grp in self.param_groups:
            for k in ("momentum", "nesterov", "dampening"):
                grp.pop(k)
            grp.update(betas=(float(betas[0]), float(betas[1])), eps=float(eps), amsgrad=False)

    def _beta1(self):
        g = self.param_groups
        if any(tuple(grp["betas"]) != tuple(g[0]["betas"]) or grp["eps"] != g[0]["eps"] for grp in g):
            raise RuntimeError("FusedAdamW: betas / eps are shared by the three groups (as the reference sets them)")
        return g[0]["betas"][0]

    def _rule_hyper(self):
        g = self.param_groups[0]
        return g["betas"][1], g["eps"], float(self.RULE), 1 - g["betas"][1], 1 - g["betas"][0]

    @property
    def exp_avg(self):
        return self.momentum

    def steps_taken(self):
        """Adam's t (one device->host read)."""
        return int(self._state.view(torch.int64)[3]) if self._table is not None else 0

    def state_dict(self):
        if self._table is None:
            self._build()
        t = float(self.steps_taken())
        state, groups, idx = {}, [], 0
        for grp in self.param_groups:
            ids = []
            for p in grp["params"]:
                if t > 0 and p.requires_grad:  # torch creates a parameter's state at its first step with a gradient
                    state[idx] = {"step": torch.tensor(t), "exp_avg": self.momentum[idx], self.SECOND_KEY: self.second[idx]}
                    if self.RULE == 4:
                        state[idx]["mu_product"] = torch.tensor(self.mu_product())
                ids.append(idx)
                idx += 1
            extra = {"fused": None, "decoupled_weight_decay": self.RULE == 1} if self.RULE in (1, 2) else {}
            groups.append({k: v for k, v in grp.items() if k != "params"} | {"params": ids, "maximize": False, "foreach": None, "capturable": False,
                                                                             "differentiable": False} | extra)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        if self._table is None:
            self._build()
        for grp, saved in zip(self.param_groups, sd["param_groups"]):
            for k in ("lr", "initial_lr", "betas", "eps", "weight_decay"):
                if k in saved:
                    grp[k] = tuple(saved[k]) if k == "betas" else saved[k]
        steps = 0
        for idx, st in sd["state"].items():
            self.momentum[int(idx)].copy_(st["exp_avg"].to(torch.float32))
            self.second[int(idx)].copy_(st[self.SECOND_KEY].to(torch.float32))
            steps = max(steps, int(float(st["step"])))
            if self.RULE == 4 and "mu_product" in st:
                self._state.view(torch.float64)[7] = float(st["mu_product"])
        self._state.view(torch.int64)[3] = steps  # one t for every parameter (they step together)


class FusedAdamax(FusedAdamW):
    """torch.optim.Adamax (reference trainer.py:829-830 with name 'Adamax'): exp_inf = max(beta2 * exp_inf, |g| + eps),
    p -= lr / (1 - beta1^t) * exp_avg / exp_inf; weight decay added to the gradient.  State layout: step, exp_avg, exp_inf."""

    SECOND_KEY = "exp_inf"

    def __init__(self, model, lr=0.002, betas=(0.9, 0.999), eps=1e-8, decay=5e-4, max_norm=10.0, ema=None):
        super().__init__(model, lr=lr, betas=betas, eps=eps, decay=decay, max_norm=max_norm, ema=ema, decoupled=False)
        self.RULE = 3
        for grp in self.param_groups:
            grp.pop("amsgrad", None)


class FusedNAdam(FusedAdamW):
    """torch.optim.NAdam (momentum_decay 4e-3, weight decay added to the gradient): the Nesterov momentum schedule
    mu_t = beta1 (1 - 0.5 * 0.96^(t * momentum_decay)) and its running product are kept on the device.  State: step, mu_product, exp_avg, exp_avg_sq."""

    def __init__(self, model, lr=0.002, betas=(0.9, 0.999), eps=1e-8, decay=5e-4, max_norm=10.0, ema=None, momentum_decay=4e-3):
        super().__init__(model, lr=lr, betas=betas, eps=eps, decay=decay, max_norm=max_norm, ema=ema, decoupled=False)
        self.RULE = 4
        for grp in self.param_groups:
            grp.pop("amsgrad", None)
            grp.update(momentum_decay=float(momentum_decay), decoupled_weight_decay=False)

    def _rule_extra(self):
        """momentum_decay, and the float32 tails of beta1 and momentum_decay (the device rebuilds both to double precision)."""
        f32 = lambda v: float(torch.tensor(v, dtype=torch.float32))
        b1, md = float(self.param_groups[0]["betas"][0]), float(self.param_groups[0]["momentum_decay"])
        return md, b1 - f32(b1), md - f32(md)

    def mu_product(self):
        return float(self._state.view(torch.float64)[7]) if self.steps_taken() else 1.0


class FusedRAdam(FusedAdamW):
    """torch.optim.RAdam (weight decay added to the gradient): the variance-rectified step once rho_t > 5, plain bias-corrected momentum before."""

    def __init__(self, model, lr=0.001, betas=(0.9, 0.999), eps=1e-8, decay=5e-4, max_norm=10.0, ema=None):
        super().__init__(model, lr=lr, betas=betas, eps=eps, decay=decay, max_norm=max_norm, ema=ema, decoupled=False)
        self.RULE = 5
        for grp in self.param_groups:
            grp.pop("amsgrad", None)
            grp.update(decoupled_weight_decay=False)


class FusedRMSprop(FusedSGD):
    """torch.optim.RMSprop(lr, momentum=momentum) as the reference builds it (trainer.py:831-832): alpha 0.99, eps 1e-8, not centered;
    square_avg = alpha * square_avg + (1 - alpha) g^2, buf = momentum * buf + g / (sqrt(square_avg) + eps), p -= lr * buf.
    State layout: step, square_avg, momentum_buffer."""

    RULE = 6

    def __init__(self, model, lr=0.01, momentum=0.937, alpha=0.99, eps=1e-8, decay=5e-4, max_norm=10.0, ema=None):
        super().__init__(model, lr=lr, momentum=momentum, decay=decay, nesterov=False, max_norm=max_norm, ema=ema)
        for grp in self.param_groups:
            for k in ("nesterov", "dampening"):
                grp.pop(k)
            grp.update(alpha=float(alpha), eps=float(eps), centered=False)

    def _beta1(self):
        g = self.param_groups
        if any(grp["momentum"] != g[0]["momentum"] or grp["alpha"] != g[0]["alpha"] or grp["eps"] != g[0]["eps"] for grp in g):
            raise RuntimeError("FusedRMSprop: momentum / alpha / eps are shared by the three groups (as the reference sets them)")
        return g[0]["momentum"]

    def _rule_hyper(self):
        g = self.param_groups[0]
        return g["alpha"], g["eps"], float(self.RULE), 1 - g["alpha"], 0.0

    def steps_taken(self):
        return int(self._state.view(torch.int64)[3]) if self._table is not None else 0

    def state_dict(self):
        if self._table is None:
            self._build()
        t = float(self.steps_taken())
        state, groups, idx = {}, [], 0
        for grp in self.param_groups:
            ids = []
            for p in grp["params"]:
                if t > 0 and p.requires_grad:
                    state[idx] = {"step": torch.tensor(t), "square_avg": self.second[idx], "momentum_buffer": self.momentum[idx]}
                ids.append(idx)
                idx += 1
            groups.append({k: v for k, v in grp.items() if k != "params"} | {"params": ids, "maximize": False, "foreach": None, "capturable": False,
                                                                             "differentiable": False})
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        if self._table is None:
            self._build()
        for grp, saved in zip(self.param_groups, sd["param_groups"]):
            for k in ("lr", "initial_lr", "momentum", "alpha", "eps", "weight_decay"):
                if k in saved:
                    grp[k] = saved[k]
        steps = 0
        for idx, st in sd["state"].items():
            self.second[int(idx)].copy_(st["square_avg"].to(torch.float32))
            self.momentum[int(idx)].copy_(st["momentum_buffer"].to(torch.float32))
            steps = max(steps, int(float(st["step"])))
        self._state.view(torch.int64)[3] = steps
