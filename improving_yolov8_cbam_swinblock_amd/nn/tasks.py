"""Model graph: YAML -> operator modules -> forward / loss (reference: ultralytics/nn/tasks.py).

Drop-in for the detection path of the reference: the same YAML grammar (`[from, repeats, module, args]`
rows, `scales`, `nc`), the same module-name lookup, width/depth rules and attribute conventions
(`.i .f .type .np`, `model.save`, `model.stride`, state-dict keys `model.N....`), so a reference
state_dict loads with `load_state_dict`.  What runs underneath is libyolo_mi355.so.
"""
import ast
import re
from copy import deepcopy
from pathlib import Path

import torch
import torch.nn as nn
import yaml

from .. import ops
from ..utils.loss import DEFAULT_HYP, SplitPreds, v8DetectionLoss
from ..utils.ops import make_divisible
from ..utils.torch_utils import fuse_conv_and_bn, initialize_weights, intersect_dicts
from .modules import C2f, CBAM, SPPF, Bottleneck, Concat, Conv, Detect, SwinBlock, Upsample

CFG_DIR = Path(__file__).resolve().parents[1] / "cfg" / "models" / "v8"

# name lookup of reference tasks.py:1433-1439 (`globals()[m]`), restricted to what has kernels
MODULES = {
    "Conv": Conv,
    "C2f": C2f,
    "SPPF": SPPF,
    "Bottleneck": Bottleneck,
    "CBAM": CBAM,
    "SwinBlock": SwinBlock,
    "Concat": Concat,
    "Detect": Detect,
    "nn.Upsample": Upsample,  # the YAML's torch.nn.Upsample row runs as the HIP nearest-2x kernel
}
BASE_MODULES = frozenset({Conv, C2f, SPPF, Bottleneck})  # width-scaled (c1, c2, ...) constructors: tasks.py:1376-1413
REPEAT_MODULES = frozenset({C2f})  # repeats passed as an argument: tasks.py:1414-1432


def guess_model_scale(model_path):
    """'yolov8s.yaml' -> 's' (reference tasks.py:1544-1557)."""
    m = re.search(r"yolo(e-)?[v]?\d+([nslmx])", Path(model_path).stem)
    return m.group(2) if m else ""


def yaml_model_load(path):
    """load a model YAML; 'yolov8s-x.yaml' resolves to 'yolov8-x.yaml' + scale 's' (reference tasks.py:1520-1541).
    Bare names are looked up in this package's cfg/models/v8."""
    path = Path(path)
    unified = Path(re.sub(r"(\d+)([nslmx])(.+)?$", r"\1\3", str(path)))
    for cand in (unified, CFG_DIR / unified.name, path, CFG_DIR / path.name):
        if cand.is_file():
            d = yaml.safe_load(cand.read_text())
            break
    else:
        raise FileNotFoundError(f"model YAML '{path}' not found (also looked in {CFG_DIR})")
    d["scale"] = guess_model_scale(path)
    d["yaml_file"] = str(path)
    return d


def parse_model(d, ch, verbose=False):
    """model dict -> (nn.Sequential of layers, sorted save list).  Reference tasks.py:1340-1517."""
    nc, scales = d.get("nc"), d.get("scales")
    depth, width = d.get("depth_multiple", 1.0), d.get("width_multiple", 1.0)
    max_channels = float("inf")
    if scales:
        scale = d.get("scale")
        if not scale:
            scale = tuple(scales.keys())[0]  # reference warns and assumes the first scale
        depth, width, max_channels = scales[scale]
    ch = [ch]
    layers, save, c2 = [], [], ch[-1]
    for i, (f, n, m, args) in enumerate(d["backbone"] + d["head"]):
        name = m
        if name in MODULES:
            m = MODULES[name]
        elif name.startswith("nn."):
            m = getattr(nn, name[3:])  # other torch.nn rows are allowed but are not part of the accelerated path
        else:
            raise KeyError(f"module '{name}' (layer {i}) is outside the accelerated YOLOv8-CBAM-Swin path")
        args = list(args)
        for j, a in enumerate(args):
            if isinstance(a, str):
                if a == "nc":
                    args[j] = nc
                else:
                    try:
                        args[j] = ast.literal_eval(a)
                    except ValueError:
                        pass
        n = n_ = max(round(n * depth), 1) if n > 1 else n
        if m in BASE_MODULES:
            c1, c2 = ch[f], args[0]
            if c2 != nc:
                c2 = make_divisible(min(c2, max_channels) * width, 8)
            args = [c1, c2, *args[1:]]
            if m in REPEAT_MODULES:
                args.insert(2, n)
                n = 1
        elif m is Concat:
            c2 = sum(ch[x] for x in f)
        elif m is Detect:
            args.append([ch[x] for x in f])
            m.legacy = True  # v8 YAMLs never reach the C3k2/A2C2f rows that clear it (tasks.py:1355,1457-1465,1488)
        else:  # CBAM, SwinBlock, Upsample: arguments untouched, channels pass through (tasks.py:1503-1504)
            c2 = ch[f]
        m_ = nn.Sequential(*(m(*args) for _ in range(n))) if n > 1 else m(*args)
        t = str(m)[8:-2].replace("__main__.", "")
        m_.np = sum(x.numel() for x in m_.parameters())
        m_.i, m_.f, m_.type = i, f, t
        if verbose:
            print(f"{i:>3}{str(f):>20}{n_:>3}{m_.np:10.0f}  {t:<45}{str(args):<30}")
        save.extend(x % i for x in ([f] if isinstance(f, int) else f) if x != -1)
        layers.append(m_)
        if i == 0:
            ch = []
        ch.append(c2)
    return nn.Sequential(*layers), sorted(save)


class BaseModel(nn.Module):
    """forward(tensor) -> predictions, forward(dict) -> loss (reference tasks.py:113-311)."""

    def forward(self, x, *args, **kwargs):
        if isinstance(x, dict):
            return self.loss(x, *args, **kwargs)
        return self.predict(x, *args, **kwargs)

    def predict(self, x, profile=False, visualize=False, augment=False, embed=None):
        return self._predict_once(x)

    def _predict_once(self, x, profile=False, visualize=False, embed=None, split_head=False):
        """the 27-step module loop of reference tasks.py:152-179.  split_head: return Detect's maps before
        its channel concat (training loss fast path).
        Training forwards use two facts of the layer graph (`_graph_plan`): the producers of a Concat layer write
        straight into their slice of its buffer (no copies for conv.py:683 `torch.cat`), and a layer output with several
        consumers carries an ops.GradJoin, so its gradient sum forms in a consumer's kernel instead of autograd adds."""
        y = []
        # engine.trainer's split backward: `_taps` = {boundary layer: number of head layers that read it}.  The head then reads a DETACHED
        # leaf of every boundary tensor, so the loss's autograd graph ends there (first pass: head only); the backbone's pass starts from
        # the originals with the leaves' gradients.  Filled with {layer: (original, leaf)}.
        taps = getattr(self, "_taps", None)
        nb_layers = len(self.yaml["backbone"]) if taps is not None else 0
        leaf_of = {}
        # (weights used twice within ONE forward are shared: their gradients are never deferred; the BatchNorm statistics accumulators are zeroed)
        ops.new_forward_epoch(x.device if (torch.is_tensor(x) and x.is_cuda and self.training) else None)
        self._begin_weight_arena(x)
        plan = self._graph_plan() if (self.training and torch.is_grad_enabled() and torch.is_tensor(x) and x.is_cuda) else None
        bufs = {}
        with ops.deferred_bn_counters():
            for m in self.model:
                if m.f != -1:
                    x = y[m.f] if isinstance(m.f, int) else [x if j == -1 else y[j] for j in m.f]
                if leaf_of and m.i >= nb_layers:  # a head layer: boundary tensors are read through their detached leaves
                    x = [leaf_of.get(id(t), t) for t in x] if isinstance(x, list) else leaf_of.get(id(x), x)
                if split_head and isinstance(m, Detect):
                    return SplitPreds(*m.forward_split(x))
                if plan is None:
                    x = m(x)
                elif isinstance(m, Concat):
                    lazy = bufs.get(m.i)
                    x = m(x, buf=lazy.buf if lazy is not None else None)
                elif m.i in plan["slot"]:
                    cidx, off, total = plan["slot"][m.i]
                    lazy = bufs.get(cidx)
                    if lazy is None:
                        lazy = bufs[cidx] = ops.LazyConcatBuffer(total, x.device if torch.is_tensor(x) else x[0].device)
                    x = m(x, out=ops.OutSlot(None, off, lazy))
                else:
                    x = m(x)
                if plan is not None and split_head and torch.is_tensor(x):
                    # only on the loss path, where every Detect input is known to receive a gradient: a join waits for ALL
                    # its consumers, and a caller of model(img) may back-propagate through some of the outputs only
                    ops.mark_join(x, plan["consumers"].get(m.i, 1))
                if taps is not None and m.i in taps and torch.is_tensor(x) and x.requires_grad:
                    heads = taps[m.i]
                    leaf = x.detach().requires_grad_(True)
                    ops.mark_join(leaf, heads)
                    j = ops.join_of(x)
                    if j is not None:
                        j.n = j.n - heads + 1  # the head's consumers arrive as ONE deposit: the leaf's gradient (TrainStep._backbone_pass)
                    leaf_of[id(x)] = leaf
                    taps[m.i] = (x, leaf)
                y.append(x if m.i in self.save else None)
        return x

    def boundary_layers(self):
        """indices of the backbone layers whose output a head layer reads (the YAML's `backbone` / `head` lists, reference
        yolov8.yaml:736-776): every edge from the backbone into the head.  engine.trainer splits the backward pass there, so that the
        head's gradients can be on their way over xGMI while the backbone's are still being computed."""
        nb = len(self.yaml["backbone"])
        out = {}
        for m in self.model:
            if m.i >= nb:
                for j in ([m.f] if isinstance(m.f, int) else m.f):
                    j = m.i - 1 if j == -1 else j
                    if j < nb:
                        out[j] = out.get(j, 0) + (2 if (isinstance(m, Detect) and not m.pair_ok) else 1)
        return dict(sorted(out.items()))  # {boundary layer: number of head consumers of its output}

    def _graph_plan(self):
        """static facts of the layer graph, computed once: {"slot": {producer layer: (concat layer, channel offset, concat
        channels)}, "consumers": {layer: number of join-aware consumers of its output (only where all consumers are)}}."""
        plan = getattr(self, "_plan", None)
        if plan is not None:
            return plan
        n = len(self.model)
        srcs = []
        for m in self.model:
            f = [m.f] if isinstance(m.f, int) else list(m.f)
            srcs.append([m.i - 1 if j == -1 else j for j in f])
        outs = self._channel_trace(self.yaml.get("ch", 3))
        slot_ok = (Conv, C2f, SPPF, CBAM, SwinBlock, Upsample)
        join_ok = (Conv, C2f, SPPF, SwinBlock, Upsample, Concat, Detect)
        slot, consumers, ok = {}, {}, {}
        for m in self.model:
            for j in srcs[m.i]:
                if j < 0:
                    continue
                # Detect reads each input with two branches - as ONE convolution when its sibling first convolutions run as a pair
                consumers[j] = consumers.get(j, 0) + ((1 if m.pair_ok else 2) if isinstance(m, Detect) else 1)
                ok[j] = ok.get(j, True) and isinstance(m, join_ok)
            if isinstance(m, Concat):
                off, total = 0, sum(outs[j][0] for j in srcs[m.i])
                for j in srcs[m.i]:
                    c = outs[j][0]
                    if j >= 0 and j not in slot and isinstance(self.model[j], slot_ok) and off % 8 == 0 and c % 8 == 0:
                        slot[j] = (m.i, off, total)
                    off += c
        plan = self._plan = {"slot": slot, "consumers": {j: c for j, c in consumers.items() if c > 1 and ok[j]}}
        return plan

    def _begin_weight_arena(self, x):
        """training forwards pack every weight with one launch (ops.WeightArena): the first forward+backward records
        the uses, the second forward builds the arena; eval / no-grad forwards keep the per-call packers."""
        if not (self.training and torch.is_grad_enabled() and x.is_cuda):
            ops.set_weight_arena(None)
            return
        dt = ops.compute_dtype(x)
        arena = getattr(self, "_arena", None)
        if arena is None or (arena.dtype is not None and arena.dtype != dt):
            arena = self._arena = ops.WeightArena()
        elif arena.built and arena.stale():  # parameter storage moved (model.to(), .float(), ...): start over
            arena = self._arena = ops.WeightArena()
        elif not arena.built and arena.specs:
            arena.build()
        ops.set_weight_arena(arena)
        if arena.built:
            arena.pack()

    def fuse(self, verbose=False):
        """fold BN into conv for inference (reference tasks.py:210-238)."""
        if not self.is_fused():
            for m in self.model.modules():
                if isinstance(m, Conv) and hasattr(m, "bn"):
                    m.conv = fuse_conv_and_bn(m.conv, m.bn)
                    delattr(m, "bn")
                    m.forward = m.forward_fuse
            self._arena = None
        return self

    def is_fused(self, thresh=10):
        bn = tuple(v for k, v in nn.__dict__.items() if "Norm" in k)
        return sum(isinstance(v, bn) for v in self.modules()) < thresh

    def _apply(self, fn):
        """keep Detect's stride / anchors on the model's device (reference tasks.py:264-282)."""
        self = super()._apply(fn)
        self._arena = None  # packed operands and their descriptor table refer to the old parameter storage
        m = self.model[-1]
        if isinstance(m, Detect):
            m.stride = fn(m.stride)
            m.anchors = fn(m.anchors)
            m.strides = fn(m.strides)
        return self

    def load(self, weights, verbose=False):
        """load weights by intersecting keys and shapes (reference tasks.py:284-297).  `weights`: a state_dict, a module,
        a checkpoint dict in the reference's layout (utils/checkpoint.py) or a path to one (read with weights_only=True)."""
        from ..utils.checkpoint import load_checkpoint, state_dict_of

        if isinstance(weights, (str, bytes)) or hasattr(weights, "__fspath__"):
            csd, _ = load_checkpoint(weights)
        else:
            csd = state_dict_of(weights)
        csd = {k: v.float() if v.dtype.is_floating_point else v for k, v in csd.items()}
        csd = intersect_dicts(csd, self.state_dict())
        self.load_state_dict(csd, strict=False)
        self._arena = None
        return len(csd)

    def loss(self, batch, preds=None):
        if getattr(self, "criterion", None) is None:
            self.criterion = self.init_criterion()
        if preds is None:
            preds = self._predict_once(batch["img"], split_head=self.training)
        return self.criterion(preds, batch)

    def init_criterion(self):
        raise NotImplementedError


class DetectionModel(BaseModel):
    """YOLOv8 detection model (reference tasks.py:318-443).

    The stride probe of the reference (a forward of zeros(1, ch, 256, 256) at construction,
    tasks.py:351-364) needs a device; the kernels only exist on the GPU, so strides are derived from
    the graph instead (the product of the strides on the path to each Detect input), which is what the
    probe measures.  CBAM's lazy MLP, which the reference creates during that probe, is created here
    from the channel count parse_model already knows.
    """

    def __init__(self, cfg="yolov8s.yaml", ch=3, nc=None, verbose=False):
        super().__init__()
        self.yaml = cfg if isinstance(cfg, dict) else yaml_model_load(cfg)
        ch = self.yaml["ch"] = self.yaml.get("ch", ch)
        if nc and nc != self.yaml["nc"]:
            self.yaml["nc"] = nc
        self.model, self.save = parse_model(deepcopy(self.yaml), ch=ch, verbose=verbose)
        self.names = {i: f"{i}" for i in range(self.yaml["nc"])}
        self.inplace = self.yaml.get("inplace", True)
        self.end2end = False
        self.args = DEFAULT_HYP
        self._materialise_lazy_modules(ch)
        m = self.model[-1]
        if isinstance(m, Detect):
            m.inplace = self.inplace
            m.stride = torch.tensor(self._graph_strides(), dtype=torch.float32)
            self.stride = m.stride
            m.bias_init()
        else:
            self.stride = torch.Tensor([32])
        initialize_weights(self)

    def _channel_trace(self, ch):
        """(channels, stride) of every layer output, from module attributes only."""
        outs = []
        cur = (ch, 1)
        for m in self.model:
            src = cur if m.f == -1 else (outs[m.f] if isinstance(m.f, int) else [cur if j == -1 else outs[j] for j in m.f])
            mods = list(m) if isinstance(m, nn.Sequential) and not isinstance(m, (Conv,)) else [m]
            for mod in mods:
                if isinstance(mod, Conv):
                    src = (mod.conv.out_channels, src[1] * mod.conv.stride[0])
                elif isinstance(mod, (C2f, SPPF)):
                    src = (mod.cv2.conv.out_channels, src[1])
                elif isinstance(mod, Bottleneck):
                    src = (mod.cv2.conv.out_channels, src[1])
                elif isinstance(mod, Upsample):
                    src = (src[0], src[1] / 2)
                elif isinstance(mod, Concat):
                    src = (sum(s[0] for s in src), src[0][1])
                elif isinstance(mod, Detect):
                    src = [s for s in src]
                # CBAM, SwinBlock and other shape-preserving rows: unchanged
            outs.append(src)
            cur = src
        return outs

    def _materialise_lazy_modules(self, ch):
        outs = self._channel_trace(ch)
        prev = (ch, 1)
        for m, o in zip(self.model, outs):
            src = prev if m.f == -1 else (outs[m.f] if isinstance(m.f, int) else None)
            if isinstance(m, CBAM) and m.ca.shared_MLP is None:
                m.ca.create_mlp(src[0])  # reference creates it with ratio 16 on first forward (cbam.py:31-33,59)
            prev = o

    def _graph_strides(self):
        outs = self._channel_trace(self.yaml["ch"])
        return [float(s[1]) for s in outs[-1]]

    @torch.no_grad()
    def stride_probe(self, s=256):
        """Reproduce the SIDE EFFECTS of the reference's construction-time stride probe (tasks.py:351-364): one train-mode forward
        of zeros(1, ch, s, s), run BEFORE initialize_weights (:367) sets BatchNorm's eps = 1e-3 / momentum = 0.03 - i.e. with
        nn.BatchNorm2d's defaults eps = 1e-5, momentum = 0.1.  It leaves every BatchNorm of a freshly built reference model with
        running_var = 0.9 + 0.1 * (unbiased batch variance of that forward), running_mean = 0.1 * (batch mean) and
        num_batches_tracked = 1 (ahead of the first SwinBlock all activations are zero, so there running_var = 0.9 exactly).
        Here the strides come from the graph and construction touches no buffer; call this once, with the model on the GPU, to
        get the reference's buffers (tests/test_gpu_e2e_golden.py::test_stride_probe_side_effects_match_the_reference).
        Returns the measured strides, which equal self.stride."""
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("stride_probe runs the libyolo_mi355 kernels: move the model to the MI355X (cuda) device first; there is no CPU path")
        bns = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
        saved = [(m.eps, m.momentum) for m in bns]
        was_training = self.training
        try:
            for m in bns:
                m.eps, m.momentum = 1e-5, 0.1
            self.train()
            outs = self.forward(torch.zeros(1, self.yaml["ch"], s, s, device=dev))
        finally:
            for m, (eps, mom) in zip(bns, saved):
                m.eps, m.momentum = eps, mom
            self.train(was_training)
        return torch.tensor([s / o.shape[-2] for o in outs], dtype=torch.float32)

    def init_criterion(self):
        return v8DetectionLoss(self)

    def predict(self, x, profile=False, visualize=False, augment=False, embed=None):
        if augment:
            raise NotImplementedError("test-time augmentation is outside the accelerated path")
        return self._predict_once(x)
