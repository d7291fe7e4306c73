"""ctypes binding of libyolo_mi355.so (the C ABI declared in include/ymi.h).

The library is the product: there is NO fallback.  If it cannot be loaded (not built, wrong arch)
every op raises; nothing in this package computes on the CPU or through another backend.
"""
import ctypes
import os
import subprocess
import threading
from pathlib import Path

import torch

PKG_DIR = Path(__file__).resolve().parent
CSRC_DIR = PKG_DIR / "csrc"
# YMI_LIB=<path>: development aid for same-box A/B runs of two builds of the library (same C ABI); unset: the in-tree build
LIB_PATH = Path(os.environ["YMI_LIB"]).resolve() if os.environ.get("YMI_LIB") else PKG_DIR / "libyolo_mi355.so"

YMI_F32, YMI_BF16 = 0, 1
ACT_NONE, ACT_SILU, ACT_GELU = 0, 1, 2

_c_i64 = ctypes.c_int64
_c_i32 = ctypes.c_int32
_c_f32 = ctypes.c_float
_vp = ctypes.c_void_p
_sz = ctypes.c_size_t


class YmiTensor(ctypes.Structure):
    _fields_ = [
        ("data", _vp),
        ("n", _c_i64),
        ("h", _c_i64),
        ("w", _c_i64),
        ("c", _c_i64),
        ("ld", _c_i64),
        ("dtype", _c_i32),
        ("_pad", _c_i32),
    ]


_TP = ctypes.POINTER(YmiTensor)

# name -> (restype, argtypes).  Must list every function include/ymi.h declares (tests check it).
_SIGNATURES = {
    "ymi_version": (_c_i32, []),
    "ymi_last_error": (ctypes.c_char_p, []),
    "ymi_set_option": (_c_i32, [ctypes.c_char_p, _c_i64]),
    "ymi_get_option": (_c_i64, [ctypes.c_char_p]),
    "ymi_profile_begin": (_c_i32, [_c_i64]),
    "ymi_profile_end": (_c_i32, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_c_i64)]),
    "ymi_profile_end_ex": (_c_i32, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_c_i64),
                                    ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "ymi_nchw_to_nhwc": (_c_i32, [_vp, _c_i64, _c_i64, _c_i64, _c_i64, _TP, _vp]),
    "ymi_nhwc_to_nchw": (_c_i32, [_TP, _vp, _vp]),
    "ymi_copy": (_c_i32, [_TP, _TP, _vp]),
    "ymi_upsample2x": (_c_i32, [_TP, _TP, _vp]),
    "ymi_upsample2x_bwd": (_c_i32, [_TP, _TP, _vp]),
    "ymi_upsample2x_bwd_acc": (_c_i32, [_TP, _TP, _vp]),
    "ymi_add_inplace": (_c_i32, [_TP, _TP, _vp]),
    "ymi_pack_conv_weight_fwd": (_c_i32, [_vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i32, _vp, _vp]),
    "ymi_pack_conv_weight_dgrad": (_c_i32, [_vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i32, _vp, _vp]),
    "ymi_pack_conv_weight_dgrad_ex": (_c_i32, [_vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i32, _vp, _vp]),
    "ymi_conv_dgrad_pack_elems": (_c_i64, [_c_i64, _c_i64, _c_i64, _c_i64, _c_i64]),
    "ymi_pack_conv_weights_batch": (_c_i32, [_vp, _vp, _c_i32, _c_i32, _c_i32, _vp]),
    "ymi_pack_matrix": (_c_i32, [_vp, _c_i64, _c_i64, _c_i32, _c_i32, _vp, _vp]),
    "ymi_conv2d_fwd": (_c_i32, [_TP, _vp, _c_i64, _c_i64, _c_i64, _c_i64, _vp, _vp, _c_i32, _TP, _TP, _vp, ctypes.POINTER(_c_i64), _vp]),
    "ymi_conv2d_stat_blocks": (_c_i64, [_c_i64, _c_i64]),
    "ymi_conv2d_fwd_multi": (_c_i32, [_vp, _c_i32, _vp]),
    "ymi_conv2d_bwd_data_multi": (_c_i32, [_vp, _c_i32, _vp]),
    "ymi_bn_finalize_pair": (_c_i32, [_vp, _c_i64, _c_i64, _c_i64, _c_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _c_f32, _c_f32, _vp, _vp, _vp, _vp, _vp]),
    "ymi_bn_finalize": (_c_i32, [_vp, _c_i64, _c_i64, _c_i64, _vp, _vp, _vp, _vp, _c_f32, _c_f32, _vp, _vp, _vp, _vp, _vp]),
    "ymi_scale_shift_act": (_c_i32, [_TP, _vp, _vp, _c_i32, _TP, _TP, _vp]),
    "ymi_conv2d_bn_silu_fwd": (
        _c_i32,
        [_TP, _vp, _c_i64, _c_i64, _c_i64, _c_i64, _vp, _vp, _vp, _vp, _c_f32, _c_f32, _c_i32, _TP, _TP, _TP, _vp, _vp, _vp, _sz, _vp],
    ),
    "ymi_conv2d_bn_silu_fwd_acc_ok": (_c_i32, [_TP, _TP, _TP]),
    "ymi_conv2d_bn_silu_fwd_acc": (_c_i32, [_TP, _vp, _c_i64, _c_i64, _c_i64, _c_i64, _vp, _vp, _vp, _vp, _c_f32, _c_f32, _c_i32, _TP, _TP, _TP, _vp, _vp, _vp, _vp]),
    "ymi_bn_act_bwd": (_c_i32, [_TP, _TP, _vp, _vp, _vp, _vp, _c_i32, _TP, _vp, _vp, _vp, _sz, _vp]),
    "ymi_conv2d_bn_silu_fwd_pair": (
        _c_i32,
        [_TP, _vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _c_f32, _c_f32, _c_i32, _TP, _TP, _vp, _vp, _vp, _sz, _vp],
    ),
    "ymi_first_conv_stat_blocks": (_c_i64, [_c_i64, _c_i64, _c_i64]),
    "ymi_first_conv_bn_act_fwd": (_c_i32, [_vp, _c_i64, _c_i64, _c_i64, _c_i64, _vp, _c_i64, _vp, _vp, _vp, _vp, _c_f32, _c_f32, _c_i32, _TP, _TP, _vp, _vp,
                                           _vp, _sz, _vp]),
    "ymi_first_conv_bwd_workspace": (_c_i64, [_c_i64, _c_i64, _c_i64, _c_i64]),
    "ymi_first_conv_bn_act_bwd": (_c_i32, [_TP, _vp, _c_i64, _c_i64, _vp, _vp, _vp, _vp, _c_i32, _TP, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "ymi_bn_act_bwd_pair": (_c_i32, [_TP, _TP, _vp, _vp, _vp, _vp, _c_i64, _vp, _vp, _c_i32, _TP, _vp, _vp, _vp, _sz, _vp]),
    "ymi_conv2d_bwd_data": (_c_i32, [_TP, _vp, _c_i64, _c_i64, _c_i64, _c_i64, _TP, _vp]),
    "ymi_conv2d_bwd_data_add": (_c_i32, [_TP, _vp, _c_i64, _c_i64, _c_i64, _c_i64, _TP, _TP, _TP, _vp]),
    "ymi_swin_mlp_fwd": (_c_i32, [_TP, _vp, _vp, _c_i64, _vp, _vp, _TP, _TP, _TP, _TP, _vp]),
    "ymi_swin_ln_mlp_supported": (_c_i32, [_c_i64, _c_i64, _c_i32]),
    "ymi_swin_ln_mlp_pack_elems": (_c_i64, [_c_i64, _c_i64]),
    "ymi_swin_ln_mlp_pre_elems": (_c_i64, [_c_i64, _c_i64]),
    "ymi_swin_ln_mlp_pack": (_c_i32, [_vp, _vp, _c_i64, _c_i64, _vp, _vp]),
    "ymi_swin_ln_mlp_fwd": (_c_i32, [_TP, _vp, _vp, _c_f32, _vp, _vp, _vp, _c_i64, _TP, _vp, _vp, _vp, _TP, _vp]),
    "ymi_swin_ln_mlp_bwd_data": (_c_i32, [_TP, _vp, _vp, _c_i64, _TP, _TP, _TP, _vp]),
    "ymi_swin_mlp_bwd_data": (_c_i32, [_TP, _vp, _TP, _TP, _vp, _TP, _TP, _TP, _vp]),
    "ymi_conv2d_bwd_weight": (_c_i32, [_TP, _TP, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _vp, _vp, _vp, _sz, _vp]),
    "ymi_conv2d_bwd_weight_workspace": (_sz, [_c_i64, _c_i64, _c_i64, _c_i64, _c_i64]),
    "ymi_conv2d_bwd_weight_deferred": (_c_i32, [_TP, _TP, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _vp, _vp, _vp, _sz, _vp, _vp]),
    "ymi_wgrad_hold": (_c_i32, [_c_i32]),
    "ymi_wgrad_reduce_batch": (_c_i32, [_vp, _c_i32, _vp, _vp]),
    "ymi_sppf_pool3_fwd": (_c_i32, [_TP, _c_i64, _TP, _TP, _TP, _vp]),
    "ymi_sppf_pool3_bwd_workspace": (_c_i64, [_c_i64, _c_i64, _c_i64, _c_i64, _c_i32]),
    "ymi_sppf_pool3_bwd": (_c_i32, [_TP, _TP, _TP, _c_i64, _TP, _TP, _TP, _TP, _TP, _vp, _c_i64, _vp]),
    "ymi_cbam_fwd": (_c_i32, [_TP, _vp, _vp, _c_i64, _vp, _c_i64, _TP, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ymi_cbam_bwd": (_c_i32, [_TP, _TP, _vp, _vp, _c_i64, _vp, _c_i64, _vp, _vp, _vp, _vp, _vp, _vp, _TP, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ymi_cbam_bwd_workspace": (_sz, [_c_i64, _c_i64, _c_i64, _c_i64, _c_i64]),
    "ymi_window_partition_index": (_c_i32, [_c_i64, _c_i64, _c_i64, _c_i64, _vp, _vp]),
    "ymi_window_partition": (_c_i32, [_TP, _c_i64, _TP, _vp]),
    "ymi_window_reverse": (_c_i32, [_TP, _c_i64, _TP, _vp]),
    "ymi_layernorm_fwd": (_c_i32, [_TP, _c_i64, _vp, _vp, _c_f32, _TP, _vp, _vp, _vp]),
    "ymi_layernorm_bwd": (_c_i32, [_TP, _c_i64, _TP, _vp, _vp, _vp, _TP, _c_i32, _vp, _vp, _vp, _sz, _vp]),
    "ymi_layernorm_bwd_add": (_c_i32, [_TP, _c_i64, _TP, _vp, _vp, _vp, _TP, _TP, _vp, _vp, _vp, _sz, _vp]),
    "ymi_window_attention_fwd": (_c_i32, [_TP, _c_i64, _c_i64, _TP, _vp, _vp]),
    "ymi_window_attention_bwd": (_c_i32, [_TP, _TP, _TP, _vp, _c_i64, _c_i64, _TP, _vp]),
    "ymi_colsum": (_c_i32, [_TP, _vp, _vp, _sz, _vp]),
    "ymi_gelu_bwd": (_c_i32, [_TP, _TP, _TP, _vp]),
    "ymi_detect_targets": (_c_i32, [_vp, _vp, _vp, _c_i64, _c_i64, _c_i64, ctypes.c_float, ctypes.c_float, _vp, _vp]),
    "ymi_detect_loss_sizes": (_c_i32, [_c_i64, _c_i64, _c_i64, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]),
    "ymi_detect_loss_fwd": (_c_i32, [_c_i32, _TP, _TP, ctypes.POINTER(ctypes.c_float), _vp, _c_i64, _c_i32, ctypes.c_float, ctypes.c_float,
                                     _vp, _vp, _vp, _sz, _vp, _sz, _vp]),
    "ymi_detect_loss_bwd": (_c_i32, [_c_i32, _TP, _TP, ctypes.POINTER(ctypes.c_float), _vp, _sz, _vp, _vp, _TP, _TP, _vp]),
    "ymi_detect_decode": (_c_i32, [_c_i32, _TP, _TP, ctypes.POINTER(ctypes.c_float), _vp, _vp]),
    "ymi_opt_chunk_elems": (_c_i64, []),
    "ymi_opt_grad_norm": (_c_i32, [_vp, _vp, _c_i32, _c_i32, _c_i64, ctypes.POINTER(_vp), _vp, _vp, _c_i64, _c_i64, _vp, _c_i32, _vp]),
    "ymi_opt_update": (_c_i32, [_vp, _vp, _c_i32, _c_i32, _c_i64, ctypes.POINTER(_vp), _vp, _vp, _c_i32, _vp]),
}

OPT_MAX_GRADS = 448  # YMI_OPT_MAX_GRADS


class ConvProblem(ctypes.Structure):  # ymi_conv_problem
    _fields_ = [("x", _TP), ("w_packed", _vp), ("cout", _c_i64), ("kh", _c_i64), ("kw", _c_i64), ("stride", _c_i64), ("scale", _vp), ("bias", _vp),
                ("act", _c_i32), ("_pad", _c_i32), ("residual", _TP), ("y", _TP), ("stat_partials", _vp), ("stat_blocks", _c_i64), ("stat_stride", _c_i64),
                ("stat_offset", _c_i64)]


class DgradProblem(ctypes.Structure):  # ymi_dgrad_problem
    _fields_ = [("dy", _TP), ("w_dgrad_packed", _vp), ("cin", _c_i64), ("k", _c_i64), ("add1", _TP), ("add2", _TP), ("dx", _TP)]


class WgradPending(ctypes.Structure):
    _fields_ = [("slab", _vp), ("dw", _vp), ("elems", _c_i64), ("splits", _c_i32), ("ng", _c_i32), ("cin", _c_i32), ("cout_real", _c_i32),
                ("cin_real", _c_i32), ("ntaps", _c_i32), ("lanes", _c_i32), ("first_block", _c_i32), ("blocks", _c_i32), ("slab_bf16", _c_i32),
                ("bias_slab", _vp), ("dbias", _vp)]


class OptEntry(ctypes.Structure):
    _fields_ = [("param", _vp), ("momentum", _vp), ("ema", _vp), ("numel", _c_i64), ("group", _c_i32), ("_pad", _c_i32), ("second", _vp), ("_pad2", _vp)]

_lib = None
_lock = threading.Lock()


def build(verbose=False):
    """Compile csrc/*.hip for gfx950 into libyolo_mi355.so (hipcc cross-compiles without a GPU)."""
    jobs = str(min(8, os.cpu_count() or 1))
    proc = subprocess.run(["make", "-C", str(CSRC_DIR), "-j", jobs], capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"building libyolo_mi355.so failed:\n{proc.stdout[-4000:]}\n{proc.stderr[-4000:]}")
    if verbose:
        print(proc.stdout[-2000:])
    return LIB_PATH


def lib():
    """The loaded library (loads on first use). Raises if it is missing: there is no fallback path."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not LIB_PATH.exists():
                    raise RuntimeError(
                        f"{LIB_PATH} is missing: build it with improving_yolov8_cbam_swinblock_amd._lib.build() "
                        "(make -C csrc); this package has no CPU or PyTorch fallback"
                    )
                handle = ctypes.CDLL(str(LIB_PATH))
                missing = []
                for name, (res, args) in _SIGNATURES.items():
                    try:
                        fn = getattr(handle, name)
                    except AttributeError:
                        missing.append(name)
                        continue
                    fn.restype = res
                    fn.argtypes = args
                if missing:
                    raise RuntimeError(f"{LIB_PATH} is stale, missing symbols {missing}: rebuild with make -C csrc")
                _lib = handle
    return _lib


def exported_symbols():
    return list(_SIGNATURES)


def check(rc, what=""):
    if rc != 0:
        msg = lib().ymi_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libyolo_mi355 {what} failed ({rc}): {msg}")


def ymi_dtype(dt):
    if dt == torch.float32:
        return YMI_F32
    if dt == torch.bfloat16:
        return YMI_BF16
    raise TypeError(f"libyolo_mi355 computes in float32 or bfloat16, got {dt}")


def chunk_elems(dt):
    """elements per 16-byte chunk: the channel granularity of the implicit-GEMM operand loader."""
    return 8 if dt == torch.bfloat16 else 4


def is_nhwc(t):
    """logical [N,C,H,W] tensor whose memory is dense NHWC with pixel stride ld >= C."""
    if t.dim() != 4:
        return False
    n, c, h, w = t.shape
    sn, sc, sh, sw = t.stride()
    if c > 1 and sc != 1:
        return False
    ld = sw if w > 1 else (sh if h > 1 else (sn if n > 1 else c))
    if w > 1 and h > 1 and sh != w * ld:
        return False
    if (h > 1 or w > 1) and n > 1 and sn != h * w * ld:
        return False
    return ld >= c


def as_ymi(t):
    """torch tensor -> YmiTensor.  4-D: logical NCHW with NHWC memory; 2-D: [rows, C] token matrix."""
    if not t.is_cuda:
        raise RuntimeError("libyolo_mi355 kernels need tensors on the MI355X (cuda) device; there is no CPU path")
    if t.dim() == 4:
        if not is_nhwc(t):
            raise RuntimeError(f"expected NHWC memory for a logical NCHW tensor, got shape {tuple(t.shape)} strides {t.stride()}")
        n, c, h, w = t.shape
        sn, sc, sh, sw = t.stride()
        ld = sw if w > 1 else (sh if h > 1 else (sn if n > 1 else c))
        return YmiTensor(t.data_ptr(), n, h, w, c, ld, ymi_dtype(t.dtype), 0)
    if t.dim() == 2:
        rows, c = t.shape
        if c > 1 and t.stride(1) != 1:
            raise RuntimeError("token matrix must have unit channel stride")
        ld = t.stride(0) if rows > 1 else c
        return YmiTensor(t.data_ptr(), 1, 1, rows, c, ld, ymi_dtype(t.dtype), 0)
    raise RuntimeError(f"unsupported tensor rank {t.dim()}")


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def empty_nhwc(n, c, h, w, dtype, device, ld=None):
    """uninitialised logical [N,C,H,W] tensor in NHWC memory (optionally with a wider pixel stride)."""
    ld = c if ld is None else ld
    buf = torch.empty((n, h, w, ld), dtype=dtype, device=device)
    return buf.permute(0, 3, 1, 2)[:, :c]


_workspaces = {}


def workspace(nbytes, device, tag="default"):
    """grow-only scratch buffer per (device, stream, tag); contents are undefined between calls."""
    if os.environ.get("YMI_WS_NOCACHE") == "1":
        return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
    key = (device.index, torch.cuda.current_stream().cuda_stream, tag)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


def set_option(name, value):
    """development option of the library (include/ymi.h: ymi_set_option; csrc/common.h lists the names).  Tests and A/B scripts only."""
    check(lib().ymi_set_option(name.encode(), int(value)), "set_option")


def get_option(name):
    v = lib().ymi_get_option(name.encode())
    if v < 0:
        raise RuntimeError(f"unknown option {name!r}")
    return int(v)
