"""torch.autograd.Function wrappers over the C ABI of libyolo_mi355.so.

PyTorch is plumbing here: it owns device memory, streams and the autograd tape; every tensor
operation on the hot path is a hand-written gfx950 kernel reached through `_lib` (ctypes).  There is
no fallback: without the library, or with CPU tensors, these functions raise.

Tensor convention between ops: logical [N, C, H, W] tensors whose MEMORY is NHWC (channels_last),
possibly a channel slice of a wider buffer (pixel stride ld > C); token matrices are plain [T, C].
Compute dtype is bfloat16 under `torch.autocast("cuda", dtype=torch.bfloat16)` and the input's
dtype (float32 = parity mode) otherwise; parameters stay float32 and are packed per call.
"""
# The op layer as a package (round 5: one 2,300-line module before): base -> weights -> conv -> blocks, each importing only from the ones before it.
# Every name is re-exported here, so `from .. import ops; ops.conv_bn_act(...)` and the tests' `ops._dgrad(...)` read as they always did.  State that is
# REBOUND at run time (the weight arena, the gradient arena, the BatchNorm counter list) lives with its readers inside one module.
from .._lib import ACT_GELU, ACT_NONE, ACT_SILU, ConvProblem, DgradProblem, as_ymi, check, chunk_elems, empty_nhwc, is_nhwc, ptr, stream_ptr, workspace, ymi_dtype  # noqa: F401
from .base import (  # noqa: F401
    GradJoin, HOOKS, L, LazyConcatBuffer, OutSlot, _GradBuffer, _GradSlot, _ToInternal, _accumulate, _as4d, _byref, _conv_out_hw, _deferred_twice,
    _dense_ok, _in_backward, _join_plain, _note_use, _prep_adds, _stat_acc, _stat_arena, _use_epoch, compute_dtype, grad_nhwc, join_of, mark_join,
    new_forward_epoch, round_up, to_internal, to_nchw_float,
)
from .weights import (  # noqa: F401
    WeightArena, _PackDesc, _adoptable, _async, _deferred, _flush_wgrads, _new_dw, _side_stream, _side_streams, _wgrad,
    _wgrad_deferred, _wgrad_maybe_async, async_wgrad, deferred_wgrad, grad_arena, join_side_stream, pack_conv_dgrad, pack_conv_dgrad_pair,
    pack_conv_fwd, pack_conv_fwd_pair, set_weight_arena, set_wgrad_deferred, wgrad_riders,
)
from .conv import (  # noqa: F401
    _ChanSlice, _ChanSplit2, _ConvAffineAct, _ConvBnAct, _ConvBnActPair, _DT, _DetectTrain, _FirstConvBnAct, _conv_fwd_multi,
    _dgrad, _dgrad_finish, _dgrad_joined, _dgrad_joined_finish, _dgrad_joined_prepare, _dgrad_launch, _dgrad_multi, _dgrad_prepare, _width_class,
    _zero_padded, chan_split2, conv_affine_act, conv_bn_act, conv_bn_act_pair, deferred_bn_counters, detect_train, detect_train_ok, first_conv_bn_act,
    first_conv_ok, linear, padded_grad_like,
)
from .blocks import (  # noqa: F401
    _Act, _AddResidual, _C2fSplit, _Cbam, _Concat, _DetectLoss, _LayerNorm, _SppfPool, _SwinLnMlp, _SwinMlp, _Upsample2x, _WindowAttention,
    _WindowReverse, _map_array, add_residual, c2f_split, cbam, concat, detect_decode, detect_loss, detect_targets, gelu, layernorm, sppf_pool_cat,
    swin_ln_mlp, swin_ln_mlp_ok, swin_mlp, upsample2x, window_attention, window_pad, window_partition, window_partition_index, window_reverse,
)
