/*
 * ymi.h - C ABI of libyolo_mi355.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * YOLOv8-CBAM-Swin forward/backward hot path.
 *
 * The reference (mazouziwissem/improving_yolov8_CBAM_SwinBlock, a fork of Ultralytics 8.3.108) has no
 * native kernels: its operator modules call torch.nn.functional, i.e. ATen.  Each entry point below
 * therefore cites the reference *call site* whose ATen work it replaces (paths relative to the
 * reference's ultralytics/ directory).  The Python-side binding is
 * improving_yolov8_cbam_swinblock_amd/_lib.py (ctypes); INTEGRATION.md shows the stub a maintainer
 * of the reference would add.
 *
 * Conventions
 *  - plain pointers and sizes only; all pointers are DEVICE pointers unless named host_*;
 *  - activations are NHWC ("channels last"): element (n,h,w,c) of a ymi_tensor lives at
 *    data[((n*h_dim + h)*w_dim + w)*ld + c]; ld >= c lets a tensor be a channel slice of a wider
 *    buffer (this is how chunk/concat cost nothing);
 *  - dtype of activations: YMI_F32 (parity mode, exact-f32 MFMA) or YMI_BF16 (fast mode, bf16 MFMA
 *    with f32 accumulation); per-channel vectors (bias, BN affine, statistics) are always f32;
 *  - trainable weights cross the ABI in the reference's own layouts (conv OIHW f32, linear [out,in]
 *    f32); ymi_pack_* turn them into the kernels' K-contiguous operand layouts;
 *  - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*); nothing allocates,
 *    frees or synchronises, so calls are graph-capturable; workspaces are caller-provided;
 *  - return 0 on success, a negative YMI_E* otherwise; ymi_last_error() gives thread-local text.
 */
#ifndef YMI_H
#define YMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YMI_VERSION 1

enum { YMI_F32 = 0, YMI_BF16 = 1 };
enum { YMI_ACT_NONE = 0, YMI_ACT_SILU = 1, YMI_ACT_GELU = 2 };
enum { YMI_OK = 0, YMI_EINVAL = -1, YMI_EALIGN = -2, YMI_ELAUNCH = -3, YMI_EWORKSPACE = -4 };

typedef struct ymi_tensor {
    void* data;
    int64_t n, h, w, c;
    int64_t ld;    /* elements between consecutive pixels, >= c */
    int32_t dtype; /* YMI_F32 | YMI_BF16 */
    int32_t _pad;
} ymi_tensor;

int ymi_version(void);
const char* ymi_last_error(void);
/* Development options: named integers that select the "before" arm of a measured change, a grid size for an in-step sweep or a
 * forced code path for a test (csrc/common.h: YmiOpt lists them with their defaults; an environment variable YMI_<NAME> present at
 * process start sets the initial value).  The reference has no counterpart (its knobs are cfg/default.yaml keys); production code sets
 * none.  set: YMI_EINVAL for an unknown name; get: -1 for an unknown name. */
int ymi_set_option(const char* name, int64_t value);
int64_t ymi_get_option(const char* name);

/* Optional live timing of the MFMA GEMM kernels (HIP events on the launch stream) for bench.py's roofline
 * line.  begin(): pre-create events for `capacity` launches and start recording; end(): synchronise the
 * device and return per-family totals (family 0: implicit-GEMM conv forward / data-gradient / token GEMMs,
 * family 1: weight-gradient GEMM): elapsed ms, algorithmic FLOP (2*M*N*K of the launch), launch count. */
int ymi_profile_begin(int64_t capacity);
int ymi_profile_end(double* ms_by_family, double* flop_by_family, int64_t* launches_by_family);
/* as ymi_profile_end, plus per family: the algorithmic HBM bytes (operands read once, result written once) and the sum
 * over launches of each launch's own roofline time max(flop / MFMA peak of its dtype, bytes / 8 TB/s), in ms. */
int ymi_profile_end_ex(double* ms_by_family, double* flop_by_family, int64_t* launches_by_family, double* bytes_by_family,
                       double* bound_ms_by_family);

/* ---------------------------------------------------------------- layout / data movement ---- */

/* NCHW f32 image -> NHWC (dst.dtype), channels c..dst.c-1 zero-filled.  Replaces the implicit
 * layout of the first conv's input: nn/tasks.py:171 feeding nn/modules/conv.py:79. */
int ymi_nchw_to_nhwc(const float* src, int64_t n, int64_t c, int64_t h, int64_t w, const ymi_tensor* dst, void* stream);
/* NHWC (any dtype) -> NCHW f32.  API edge for callers that want the reference's memory format. */
int ymi_nhwc_to_nchw(const ymi_tensor* src, float* dst, void* stream);
/* dst[n,h,w,0:c] = src (dtype may differ: cast).  torch.cat of nn/modules/conv.py:683 and
 * nn/modules/block.py:226,304 when the producer could not write into the slice itself. */
int ymi_copy(const ymi_tensor* src, const ymi_tensor* dst, void* stream);
/* dst[n,2h+i,2w+j,:] = src[n,h,w,:] : nn.Upsample(None, 2, "nearest"), cfg yolov8.yaml:759,764. */
int ymi_upsample2x(const ymi_tensor* src, const ymi_tensor* dst, void* stream);
/* adjoint: dst[n,h,w,:] = sum of the 2x2 block of src. */
int ymi_upsample2x_bwd(const ymi_tensor* src, const ymi_tensor* dst, void* stream);
/* adjoint accumulated onto dst: dst[n,h,w,:] += sum of the 2x2 block of src (dst already holds the other consumers' gradient). */
int ymi_upsample2x_bwd_acc(const ymi_tensor* src, const ymi_tensor* dst, void* stream);
/* dst += src (f32 accumulate, elementwise over NHWC tensors of equal shape). */
int ymi_add_inplace(const ymi_tensor* src, const ymi_tensor* dst, void* stream);

/* ------------------------------------------------------------------------- weight packing ---- */

/* OIHW f32 -> [O][kh][kw][Ipad] in `dtype` (K contiguous): operand of ymi_conv2d_fwd.
 * Ipad >= I, a multiple of 8 (bf16) / 4 (f32); padded channels are zero. */
int ymi_pack_conv_weight_fwd(const float* w_oihw, int64_t o, int64_t i, int64_t kh, int64_t kw, int64_t ipad, int32_t dtype, void* dst, void* stream);
/* OIHW f32 -> operand of ymi_conv2d_bwd_data: for stride 1 one block [I][kh][kw][O] with taps
 * flipped; for stride 2 the four output-parity classes back to back (sizes from
 * ymi_conv_dgrad_pack_elems). */
int ymi_pack_conv_weight_dgrad(const float* w_oihw, int64_t o, int64_t i, int64_t kh, int64_t kw, int64_t stride, int32_t dtype, void* dst, void* stream);
/* same, with the K axis (dy channels) zero-padded from o_real to o_pad rows. */
int ymi_pack_conv_weight_dgrad_ex(const float* w_oihw, int64_t o_real, int64_t o_pad, int64_t i, int64_t kh, int64_t kw, int64_t stride, int32_t dtype, void* dst, void* stream);
int64_t ymi_conv_dgrad_pack_elems(int64_t o, int64_t i, int64_t kh, int64_t kw, int64_t stride);
/* All weights of a model in ONE launch.  descs_device: array of ymi_pack_desc in device memory; block_start_device:
 * int32[count+1] prefix sums of the workgroups per tensor = ceil(max(o, opad if dst_dgrad) / 32) * ceil(max(i, ipad if dst_fwd) / 32)
 * (a workgroup reads a 32 x 32 channel tile of the source once and writes both operands from LDS); either destination may be
 * NULL; kh * kw <= 9.
 * Same layouts as ymi_pack_conv_weight_fwd / _dgrad_ex (nn.Linear weights: kh = kw = 1). */
typedef struct ymi_pack_desc {
    const float* src;
    void* dst_fwd;
    void* dst_dgrad;
    int32_t o, i, kh, kw, ipad, opad, stride;
    int32_t ostride; /* row length of the data-gradient operand; 0 = opad.  With ostride > opad several weights fill one operand side by side: */
    int32_t o_off;   /* ... this weight's first column (the merged operand of Detect's sibling convolutions cv2[i][0] / cv3[i][0], head.py:71-72) */
    int32_t _pad;
} ymi_pack_desc;
int ymi_pack_conv_weights_batch(const void* descs_device, const int32_t* block_start_device, int32_t count, int32_t total_blocks,
                                int32_t dtype, void* stream);
/* [rows][cols] f32 -> `dtype`, optionally transposed ([cols][rows]): nn.Linear / in_proj weights. */
int ymi_pack_matrix(const float* src, int64_t rows, int64_t cols, int32_t transpose, int32_t dtype, void* dst, void* stream);

/* ------------------------------------------------------------ convolution (implicit GEMM) ---- */

/* y = act(scale[c] * conv(x, w) + bias[c]) + residual        (MFMA implicit GEMM, NHWC)
 *   w: packed by ymi_pack_conv_weight_fwd with ipad == x->c;  pad = k/2 ("autopad", conv.py:28-34).
 *   scale, bias, residual may be NULL.  If stat_partials != NULL the kernel also writes per-M-block
 *   partial sums [blocks][2][cout] (sum, sum of squares of the ROUNDED outputs) for train-mode BN;
 *   *host_stat_blocks receives `blocks`.  ymi_conv2d_stat_blocks() gives an upper bound for sizing.
 * Replaces F.conv2d at nn/modules/conv.py:79,91 (Conv), nn/modules/head.py:45-59 (Detect stacks),
 * nn/modules/cbam.py:24-26 is NOT routed here (tiny, fused in ymi_cbam_*).
 * With kh=kw=1 and n*h*w = rows it is the token GEMM of nn/modules/swin_block.py:31-35,51,53. */
int ymi_conv2d_fwd(const ymi_tensor* x, const void* w_packed, int64_t cout, int64_t kh, int64_t kw, int64_t stride,
                   const float* scale, const float* bias, int32_t act, const ymi_tensor* residual, const ymi_tensor* y,
                   float* stat_partials, int64_t* host_stat_blocks, void* stream);
int64_t ymi_conv2d_stat_blocks(int64_t m_rows, int64_t cout);
/* Several independent convolutions (any shapes; same dtype; statistics mode for all or none; input channels in whole K steps of the same
 * width class; with statistics: the same output-pixel count per BatchNorm group) in ONE launch, at most 8: the same stage of Detect's three levels (nn/modules/head.py:66-74 loops over them), whose small
 * levels cannot fill the chip alone.  Arguments per problem as ymi_conv2d_fwd; stat_blocks is written by the call. */
typedef struct ymi_conv_problem {
    const ymi_tensor* x;
    const void* w_packed;
    int64_t cout, kh, kw, stride;
    const float* scale;
    const float* bias;
    int32_t act, _pad;
    const ymi_tensor* residual;
    const ymi_tensor* y;
    float* stat_partials;
    int64_t stat_blocks;
    int64_t stat_stride, stat_offset; /* statistics rows [block][2][stat_stride] with this problem's channels at column stat_offset: the problems of
                                       * one BatchNorm group share a row array (all problems of a launch use the same row tile); 0: [2][cout] */
} ymi_conv_problem;
int ymi_conv2d_fwd_multi(const ymi_conv_problem* problems, int32_t n, void* stream);

/* Train-mode BatchNorm2d statistics from the partials above (nn/modules/conv.py:66,79 with
 * eps/momentum of utils/torch_utils.py:468-470): writes scale = gamma*invstd, shift = beta - mean*scale,
 * save_mean, save_invstd, and updates running_mean / running_var (unbiased) in place. */
int ymi_bn_finalize(const float* stat_partials, int64_t blocks, int64_t count, int64_t c, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float momentum, float eps, float* scale, float* shift,
                    float* save_mean, float* save_invstd, void* stream);
/* the same for two BatchNorms whose channels lie side by side (channels >= split read gamma2 / beta2 and update running_*2): the
 * BatchNorms of two convolutions that ran as one, or side by side into one buffer (Detect's branches, head.py:44-59). */
int ymi_bn_finalize_pair(const float* stat_partials, int64_t blocks, int64_t count, int64_t c, int64_t split, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, const float* gamma2, const float* beta2, float* running_mean2, float* running_var2,
                         float momentum, float eps, float* scale, float* shift, float* save_mean, float* save_invstd, void* stream);
/* out = act(raw*scale[c] + shift[c]) + residual : BN-affine + SiLU (+ Bottleneck add, block.py:488). */
int ymi_scale_shift_act(const ymi_tensor* raw, const float* scale, const float* shift, int32_t act, const ymi_tensor* residual,
                        const ymi_tensor* out, void* stream);

/* Conv + train-mode BN + SiLU in one call: conv (raw, statistics) -> finalize -> affine+SiLU.
 * `raw` (saved for backward) and `out` have y's shape; workspace >= ymi_conv2d_stat_blocks*2*cout*4 B
 * + 2*cout*4 B.  Replaces Conv.forward, nn/modules/conv.py:69-79. */
int ymi_conv2d_bn_silu_fwd(const ymi_tensor* x, const void* w_packed, int64_t cout, int64_t kh, int64_t kw, int64_t stride,
                           const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                           float eps, int32_t act, const ymi_tensor* residual, const ymi_tensor* raw, const ymi_tensor* out,
                           float* save_mean, float* save_invstd, void* workspace, size_t workspace_bytes, void* stream);

/* The same Conv block in TWO launches: the GEMM's statistics epilogue adds its per-block sums as 64-bit fixed-point atomics (exact, order-free:
 * the statistics stay bit-reproducible) into stat_acc = [4][2][cout] int64, ZERO on entry, and the affine + activation pass finalizes them in its
 * prologue (scale / shift per thread; saved mean / inverse deviation and the running statistics by its first workgroup) - no finalize launch.
 * `_acc_ok`: whether the affine pass takes these tensors in that form (cout a multiple of 4 up to 1024, 4-element-aligned rows); the per-channel
 * vectors must be 16-byte aligned.  Replaces Conv.forward, nn/modules/conv.py:69-79, as ymi_conv2d_bn_silu_fwd does. */
int ymi_conv2d_bn_silu_fwd_acc_ok(const ymi_tensor* raw, const ymi_tensor* out, const ymi_tensor* residual);
int ymi_conv2d_bn_silu_fwd_acc(const ymi_tensor* x, const void* w_packed, int64_t cout, int64_t kh, int64_t kw, int64_t stride,
                               const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                               float eps, int32_t act, const ymi_tensor* residual, const ymi_tensor* raw, const ymi_tensor* out,
                               float* save_mean, float* save_invstd, void* stat_acc, void* stream);

/* Backward of act(BN_train(raw)) w.r.t. raw, two passes:
 *  reduce: partial sums of dz and dz*xhat per channel, dz = dout * act'(raw*scale+shift)
 *  apply : draw = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)); also dgamma, dbeta. */
int ymi_bn_act_bwd(const ymi_tensor* dout, const ymi_tensor* raw, const float* gamma, const float* save_mean,
                   const float* save_invstd, const float* beta, int32_t act, const ymi_tensor* draw, float* dgamma, float* dbeta,
                   void* workspace, size_t workspace_bytes, void* stream);

/* Two Conv blocks that read the SAME input as ONE convolution - the first convolutions of Detect's two branches, cv2[i][0] and cv3[i][0]
 * (nn/modules/head.py:44-59,71-72): w_packed holds the first convolution's packed weights followed by the second's (output channels
 * [0, split) / [split, cout)); BatchNorm is per channel, so with each half reading its own gamma / beta and updating its own running
 * statistics the result equals the two separate blocks.  raw / out / save_mean / save_invstd cover all cout channels.  split % 4 == 0. */
int ymi_conv2d_bn_silu_fwd_pair(const ymi_tensor* x, const void* w_packed, int64_t cout, int64_t split, int64_t kh, int64_t kw, int64_t stride,
                                const float* gamma, const float* beta, float* running_mean, float* running_var, const float* gamma2,
                                const float* beta2, float* running_mean2, float* running_var2, float momentum, float eps, int32_t act,
                                const ymi_tensor* raw, const ymi_tensor* out, float* save_mean, float* save_invstd, void* workspace,
                                size_t workspace_bytes, void* stream);
/* ... and the backward of its BatchNorm + activation (as ymi_bn_act_bwd; dgamma / dbeta hold all channels, the first block's first). */
int ymi_bn_act_bwd_pair(const ymi_tensor* dout, const ymi_tensor* raw, const float* gamma, const float* beta, const float* gamma2,
                        const float* beta2, int64_t split, const float* save_mean, const float* save_invstd, int32_t act,
                        const ymi_tensor* draw, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, void* stream);
/* dx = conv_transpose(dy, w): operand packed by ymi_pack_conv_weight_dgrad.  Adjoint of conv.py:79. */
int ymi_conv2d_bwd_data(const ymi_tensor* dy, const void* w_dgrad_packed, int64_t cin, int64_t kh, int64_t kw, int64_t stride,
                        const ymi_tensor* dx, void* stream);
/* the same with up to two addends in the epilogue: dx = dgrad(dy) + add1 (+ add2).  An addend may BE dx (in-place
 * accumulation).  This is how the gradient sums of tensors with several consumers (Bottleneck shortcut block.py:488, the
 * two Detect branches head.py:72, neck skip connections yolov8.yaml:760-773, SwinBlock residuals swin_block.py:52-53) are
 * formed without separate add kernels.  kh = kw = 1 with stride 2: only the pixels on the stride grid receive a gradient; the others
 * are left as the caller prepared them (zeros) and addends must be NULL (EINVAL otherwise). */
int ymi_conv2d_bwd_data_add(const ymi_tensor* dy, const void* w_dgrad_packed, int64_t cin, int64_t kh, int64_t kw, int64_t stride,
                            const ymi_tensor* add1, const ymi_tensor* add2, const ymi_tensor* dx, void* stream);
/* ... and several independent STRIDE-1 data gradients in one launch (k x k kernel, k in {1, 3}; addends as above). */
typedef struct ymi_dgrad_problem {
    const ymi_tensor* dy;
    const void* w_dgrad_packed;
    int64_t cin, k;
    const ymi_tensor* add1;
    const ymi_tensor* add2;
    const ymi_tensor* dx;
} ymi_dgrad_problem;
int ymi_conv2d_bwd_data_multi(const ymi_dgrad_problem* problems, int32_t n, void* stream);
/* dw (OIHW f32 [cout_real][cin_real][kh][kw], overwritten) = sum over pixels dy (x) x ; optional
 * dbias = column sums of dy: a buffer of dy->c floats (the PADDED channel count), of which the first cout_real are the bias gradient.  x / dy may carry zero-padded channels (x->c >= cin_real,
 * dy->c >= cout_real).  Split-K MFMA GEMM + ordered slab reduce (deterministic); workspace from
 * ymi_conv2d_bwd_weight_workspace. */
int ymi_conv2d_bwd_weight(const ymi_tensor* x, const ymi_tensor* dy, int64_t cout_real, int64_t cin_real, int64_t kh, int64_t kw,
                          int64_t stride, float* dw_oihw, float* dbias, void* workspace, size_t workspace_bytes, void* stream);
size_t ymi_conv2d_bwd_weight_workspace(int64_t m_rows, int64_t cout, int64_t cin, int64_t kh, int64_t kw);
/* The same GEMM with the ordered slab sum DEFERRED: `pending` (host memory) receives what ymi_wgrad_reduce_batch needs.
 * The caller keeps `workspace` (the slabs) alive until that launch.  One reduce launch then serves every layer of a
 * backward pass (the 73 per-layer reduce launches of YOLOv8s were pure latency). */
typedef struct ymi_wgrad_pending {
    const void* slab;    /* [splits][elems], float32 or (slab_bf16) bfloat16 first-level partials */
    float* dw;           /* OIHW destination */
    int64_t elems;       /* padded Cout * (taps * padded Cin) */
    int32_t splits, ng, cin, cout_real, cin_real, ntaps;
    int32_t lanes;       /* interleaved split chains per output element (4, 8, 16 or 32) */
    int32_t first_block; /* set by ymi_wgrad_reduce_batch */
    int32_t blocks;      /* workgroups this record needs: ceil(elems / e / (256 / lanes)), e = 8 elements per lane for bfloat16 slabs, 4 for float32 */
    int32_t slab_bf16;   /* 1: the slabs hold bfloat16 (the bf16 path), 0: float32 (parity mode) */
    const float* bias_slab; /* optional: [splits][padded Cout] float32 column sums of dY per split (the bias gradient's partials, formed by the */
    float* dbias;           /* weight-gradient GEMM itself); their sum over the splits goes to dbias [padded Cout].  NULL: no bias */
} ymi_wgrad_pending;
int ymi_conv2d_bwd_weight_deferred(const ymi_tensor* x, const ymi_tensor* dy, int64_t cout_real, int64_t cin_real, int64_t kh, int64_t kw,
                                   int64_t stride, float* dw_oihw, float* dbias, void* workspace, size_t workspace_bytes,
                                   ymi_wgrad_pending* pending, void* stream);
/* Riders.  mode 1: a deferred weight-gradient launch (ymi_conv2d_bwd_weight_deferred) is HELD BACK - at most one - until the next BatchNorm
 * backward on the same stream (ymi_bn_act_bwd / _pair) has issued its reduce pass; it is then launched with that layer's final pass (the sum of
 * the partial rows, the apply pass's coefficients: a 1-16 workgroup kernel at a dependent-launch latency otherwise) as extra workgroups of its own
 * grid.  The next deferred call, ymi_wgrad_reduce_batch and mode 0 issue a held launch as it is; mode 2 forgets it (after a backward pass that
 * raised).  The caller keeps operands and workspace of a deferred launch alive until the batched slab sum in any case.  No reference counterpart. */
int ymi_wgrad_hold(int32_t mode);
/* host_records: the n records of the pending layers (HOST memory; first_block is filled in here); device_table: device
 * scratch for n records.  The records reach the device inside kernel arguments (no host staging: graph-capturable). */
int ymi_wgrad_reduce_batch(const ymi_wgrad_pending* host_records, int32_t n, ymi_wgrad_pending* device_table, void* stream);

/* The model's FIRST Conv block on the float32 NCHW image the caller hands over (cfg/models/v8/yolov8.yaml:738 `Conv [64, 3, 2]` through
 * nn/modules/conv.py:50-79): act(BatchNorm_train(conv 3x3, stride 2, pad 1)) as DIRECT kernels that never store the raw convolution output
 * (27 products per output: recomputing is cheaper than reading back; csrc/first_conv.hip).  c <= 4, cout in {16, 32, 48, 64}, h even,
 * w % 4 == 0, bfloat16 outputs.
 *   forward : statistics pass over the image, finalize (running statistics updated), apply pass -> out.  x4 receives the image as NHWC
 *             bfloat16 [n, 4, h, w] (dense) - all the backward pass needs of the input.
 *             workspace >= (2 * cout + (ymi_first_conv_stat_blocks(n, h, w) + 64) * 2 * cout) * 4 bytes.
 *   backward: dgamma, dbeta [cout] and the weight gradient dw_oihw [cout, c, 3, 3] (the image receives none).  pending == NULL: dw is summed
 *             before the call returns control of the stream; else the slab sum is left to ymi_wgrad_reduce_batch (record in *pending).
 *             workspace >= ymi_first_conv_bwd_workspace bytes. */
int64_t ymi_first_conv_stat_blocks(int64_t n, int64_t h, int64_t w);
int ymi_first_conv_bn_act_fwd(const float* img_nchw, int64_t n, int64_t c, int64_t h, int64_t w, const float* weight_oihw, int64_t cout,
                              const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                              int32_t act, const ymi_tensor* x4, const ymi_tensor* out, float* save_mean, float* save_invstd,
                              void* workspace, size_t workspace_bytes, void* stream);
int64_t ymi_first_conv_bwd_workspace(int64_t n, int64_t h, int64_t w, int64_t cout);
int ymi_first_conv_bn_act_bwd(const ymi_tensor* x4, const float* weight_oihw, int64_t c, int64_t cout, const float* gamma, const float* beta,
                              const float* save_mean, const float* save_invstd, int32_t act, const ymi_tensor* dout, float* dgamma, float* dbeta,
                              float* dw_oihw, void* workspace, size_t workspace_bytes, ymi_wgrad_pending* pending, void* stream);

/* -------------------------------------------------------------------- SPPF pooling cascade ---- */

/* y1 = maxpool_k(y0), y2 = maxpool_k(y1), y3 = maxpool_k(y2), stride 1, pad k/2 (-inf):
 * nn/modules/block.py:220,225.  One LDS-staged kernel; y1..y3 are usually channel slices of the
 * concat buffer that SPPF.cv2 reads. */
int ymi_sppf_pool3_fwd(const ymi_tensor* y0, int64_t k, const ymi_tensor* y1, const ymi_tensor* y2, const ymi_tensor* y3, void* stream);
/* Adjoint of the cascade: dx = dy0 + route(dy1 + route(dy2 + route(dy3 | y2) | y1) | y0), a deterministic gather in which every
 * window's gradient goes to its first arg-max in row-major order (PyTorch's max_pool2d tie rule).  No input is modified; dx may
 * not alias them.  Maps that fit LDS whole (<= ~2100 pixels in bf16) run in one launch with the running gradient in f32 and need
 * no workspace; larger maps run stage by stage through `workspace` (ymi_sppf_pool3_bwd_workspace bytes, 16-byte aligned), the
 * intermediate gradients rounded to the tensor dtype like the reference's. */
int64_t ymi_sppf_pool3_bwd_workspace(int64_t n, int64_t h, int64_t w, int64_t c, int dtype);
int ymi_sppf_pool3_bwd(const ymi_tensor* y0, const ymi_tensor* y1, const ymi_tensor* y2, int64_t k, const ymi_tensor* dy0,
                       const ymi_tensor* dy1, const ymi_tensor* dy2, const ymi_tensor* dy3, const ymi_tensor* dx, void* workspace,
                       int64_t workspace_bytes, void* stream);

/* -------------------------------------------------------------------------------- CBAM ---- */

/* out = x * ca * sa with ca = sigmoid(MLP(avg) + MLP(max)) (nn/modules/cbam.py:29-38),
 * sa = sigmoid(conv7x7([mean_c(x*ca), max_c(x*ca)])) (cbam.py:48-53), composed as cbam.py:62-71.
 * w1: [hidden][C] f32, w2: [C][hidden] f32, wsa: [2][k][k] f32.  Saved for backward: ca [N][C] f32,
 * smap [N][H][W][2] f32 (mean,max of x*ca), sa [N][H][W] f32, and the arg-max indices. */
int ymi_cbam_fwd(const ymi_tensor* x, const float* w1, const float* w2, int64_t hidden, const float* wsa, int64_t ksa,
                 const ymi_tensor* out, float* ca, float* pooled /*[N][2][C]*/, int32_t* pool_argmax /*[N][C]*/, float* smap,
                 int32_t* smap_argmax /*[N][H][W]*/, float* sa, void* stream);
int ymi_cbam_bwd(const ymi_tensor* x, const ymi_tensor* dout, const float* w1, const float* w2, int64_t hidden, const float* wsa,
                 int64_t ksa, const float* ca, const float* pooled, const int32_t* pool_argmax, const float* smap,
                 const int32_t* smap_argmax, const float* sa, const ymi_tensor* dx, float* dw1, float* dw2, float* dwsa,
                 void* workspace, size_t workspace_bytes, void* stream);
size_t ymi_cbam_bwd_workspace(int64_t n, int64_t h, int64_t w, int64_t c, int64_t hidden);

/* ------------------------------------------------------------------------------ SwinBlock ---- */

/* Integer index maps of nn/modules/swin_block.py:8-20 (bit-exact requirement): for window-ordered
 * token t the flat padded-grid pixel index, tokens = n*hp*wp. */
int ymi_window_partition_index(int64_t n, int64_t hp, int64_t wp, int64_t ws, int32_t* index, void* stream);
/* tokens[t,:] = x[pixel(t),:] (zero where the pixel is padding) / inverse with crop. */
int ymi_window_partition(const ymi_tensor* x, int64_t ws, const ymi_tensor* tokens /*n=1,h=1,w=T*/, void* stream);
int ymi_window_reverse(const ymi_tensor* tokens, int64_t ws, const ymi_tensor* x, void* stream);
/* t = LayerNorm(gather(x)) : pad + 'b c h w -> b h w c' + window_partition + norm1 fused
 * (swin_block.py:41-50).  With ws == 0 no gather: plain row-wise LayerNorm (norm2, :53).
 * Saves mean / rstd per token for backward. */
int ymi_layernorm_fwd(const ymi_tensor* x, int64_t ws, const float* gamma, const float* beta, float eps, const ymi_tensor* out,
                      float* mean, float* rstd, void* stream);
/* dx = LayerNorm adjoint (written through the window map when ws != 0; pad tokens are dropped =
 * the crop of :58); accumulate != 0 adds into dx instead (LayerNorm input that also feeds a skip). */
int ymi_layernorm_bwd(const ymi_tensor* x, int64_t ws, const ymi_tensor* dout, const float* gamma, const float* mean,
                      const float* rstd, const ymi_tensor* dx, int32_t accumulate, float* dgamma, float* dbeta, void* workspace,
                      size_t workspace_bytes, void* stream);
/* the same with an addend: dx = LayerNorm gradient + add (the gradient the LN input receives from its other consumers,
 * e.g. the residual branch of swin_block.py:52-53); add may be dx itself. */
int ymi_layernorm_bwd_add(const ymi_tensor* x, int64_t ws, const ymi_tensor* dout, const float* gamma, const float* mean, const float* rstd,
                          const ymi_tensor* add, const ymi_tensor* dx, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                          void* stream);
/* Window attention core of nn.MultiheadAttention as used at swin_block.py:29,51:
 * qkv [T][3C] (q pre-scaled is NOT assumed: the kernel scales by 1/sqrt(hd)); per (window, head)
 * P = softmax(q k^T / sqrt(hd)) over all `wlen` keys, o = P v; out [T][C].  lse [T][heads] saved. */
int ymi_window_attention_fwd(const ymi_tensor* qkv, int64_t wlen, int64_t heads, const ymi_tensor* out, float* lse, void* stream);
int ymi_window_attention_bwd(const ymi_tensor* qkv, const ymi_tensor* out, const ymi_tensor* dout, const float* lse, int64_t wlen,
                             int64_t heads, const ymi_tensor* dqkv, void* stream);
/* Generic token GEMM backward helpers (nn.Linear adjoint): column sums for bias gradients. */
int ymi_colsum(const ymi_tensor* x, float* out /*[c]*/, void* workspace, size_t workspace_bytes, void* stream);
/* dx = dy * gelu'(pre) (exact erf GELU, swin_block.py:33). */
int ymi_gelu_bwd(const ymi_tensor* pre, const ymi_tensor* dy, const ymi_tensor* dx, void* stream);

/* ---- v8 detection loss on the per-level Detect maps ----------------------------------------------------------
 * Replaces v8DetectionLoss.__call__ (utils/loss.py:201-255) with its helpers: preprocess (loss.py:176-191),
 * bbox_decode (loss.py:193-199), TaskAlignedAssigner.forward (utils/tal.py:45-104), BboxLoss / DFLoss
 * (loss.py:65-120) and bbox_iou(CIoU=True) (utils/metrics.py:74-134).  box_maps[l] is the NHWC map
 * [B,H_l,W_l,64] of Detect.cv2[l] (reg_max 16), cls_maps[l] the map [B,H_l,W_l,nc] of Detect.cv3[l]; anchors are
 * ordered level by level, row-major, as the reference's concat.  All arithmetic is f32; maps may be f32 or bf16. */

/* ragged label rows (image index, class, xywh normalised) -> dense out[batch, max_boxes, 5] = (class, xyxy in pixels);
 * rows keep their order inside an image, unused slots are zero, rows beyond max_boxes of an image are dropped. */
int ymi_detect_targets(const float* batch_idx, const float* cls, const float* bboxes_xywhn, int64_t n, int64_t batch, int64_t max_boxes,
                       float img_w, float img_h, float* out, void* stream);
/* bytes of the state (kept from forward to backward) and of the scratch workspace */
int ymi_detect_loss_sizes(int64_t batch, int64_t anchors, int64_t max_boxes, size_t* state_bytes, size_t* workspace_bytes);
/* raw[3] = (box, cls, dfl) sums divided by max(sum of target scores, 1), before the hyper-parameter gains.
 * out_scale == NULL: loss_out[3] = raw.  out_scale (device [6]): loss_out[6], loss_out[k] = raw[k] * out_scale[k] and
 * loss_out[3 + k] = raw[k] * out_scale[3 + k]: the criterion's two results (loss * gains * batch, loss * gains; reference
 * utils/loss.py:250-255) from this one launch.
 * strides: host array [nl]; targets: device [batch, max_boxes, 5] as written by ymi_detect_targets. */
int ymi_detect_loss_fwd(int32_t nl, const ymi_tensor* box_maps, const ymi_tensor* cls_maps, const float* strides, const float* targets,
                        int64_t max_boxes, int32_t topk, float alpha, float beta, const float* out_scale, float* loss_out, void* state,
                        size_t state_bytes, void* workspace, size_t workspace_bytes, void* stream);
/* gradients of sum_k grad_loss[k] * grad_scale[k] * raw[k] with respect to the maps (grad_loss: device [3]; grad_scale: device [3]
 * or NULL = 1: the out_scale[0..2] the forward multiplied its differentiable result by).  dcls_maps may be channel slices of
 * wider buffers (ld > c); they may also have MORE channels than the class maps (the same count on every level): the extra
 * channels are written as zeros (channel-padded gradient buffers need no separate fill). */
int ymi_detect_loss_bwd(int32_t nl, const ymi_tensor* box_maps, const ymi_tensor* cls_maps, const float* strides, const void* state,
                        size_t state_bytes, const float* grad_loss, const float* grad_scale, const ymi_tensor* dbox_maps,
                        const ymi_tensor* dcls_maps, void* stream);

/* Detect._inference (nn/modules/head.py:103-142, non-export branch; DFL nn/modules/block.py:58-77; make_anchors /
 * dist2bbox utils/tal.py:364-388): per level l a box map [B,H,W,64] and a class map [B,H,W,nc] (NHWC, channel slices of
 * the concatenated Detect output are fine) -> y [B][4+nc][A] float32: rows 0..3 = (cx, cy, w, h) in pixels, rows 4.. =
 * sigmoid(class logits); anchors in the reference's order (level, y, x), centre offset 0.5. */
int ymi_detect_decode(int32_t nl, const ymi_tensor* box_maps, const ymi_tensor* cls_maps, const float* strides, float* y, void* stream);

/* ------------------------------------------------------------------------- optimizer step ---- */
/* The update either side of backward, reference engine/trainer.py:614-622 (optimizer_step: clip_grad_norm_(10.0),
 * SGD-nesterov step, EMA update), :788-849 (three parameter groups) and utils/torch_utils.py:657-673 (ModelEMA.update),
 * as multi-tensor launches over a DEVICE table of every tensor (built once: parameters, momentum and EMA buffers do not
 * move).  Gradient tensors move from step to step in eager mode, so their device addresses are passed as a HOST array
 * and travel in the kernel arguments (graph-capturable, no staging buffer).
 *   table      [n] ymi_opt_entry; group: 0 biases, 1 decayed weights, 2 norm weights (index into hyper's lr / wd);
 *              momentum / ema may be NULL (no momentum buffer: EMA-only entry; no ema: EMA not attached)
 *   chunk_map  [n_chunks][2] int32 (tensor index, chunk index): one workgroup per ymi_opt_chunk_elems() elements
 *   hyper      [20] float on the device: lr[3], weight_decay[3], momentum (Adam: beta1), max_norm (<= 0: no clipping), ema decay,
 *              ema tau, nesterov (0/1), gradient scale (1 / world size), beta2 (RMSprop: alpha), eps, rule (0 SGD-momentum - reference
 *              trainer.py:832-833; 1 AdamW, 2 Adam, 3 Adamax, 4 NAdam, 5 RAdam - trainer.py:829-830; 6 RMSprop - :831), 1 - beta2,
 *              1 - beta1 (differences taken in double on the host, as torch passes them), NAdam's momentum_decay, the float32 tails of beta1 and momentum_decay (NAdam)
 *   state      64 bytes on the device, zero before the first step: float clip, float total_norm, float ema_d, float 1-ema_d,
 *              int64 updates, int64 steps (Adam's t), float 1/(1-beta1^t), float sqrt(1-beta2^t), 4 floats of per-step scalars
 *              (NAdam's two weights and bias correction, RAdam's rectification), double mu_product (NAdam)
 * A launch covers tensors [first_tensor, first_tensor + n_tensors), n_tensors <= YMI_OPT_MAX_GRADS, whose chunks are the
 * n_chunks entries at chunk_map; host_grads[i] is the gradient of tensor first_tensor + i or NULL (no gradient this step). */
#define YMI_OPT_MAX_GRADS 448
typedef struct ymi_opt_entry {
    float* param;
    float* momentum; /* SGD / RMSprop: momentum buffer; Adam family: exp_avg */
    float* ema;
    int64_t numel;
    int32_t group;
    int32_t _pad;
    float* second;   /* Adam / AdamW / NAdam / RAdam: exp_avg_sq; Adamax: exp_inf; RMSprop: square_avg; NULL for SGD and EMA-only entries */
    void* _pad2;
} ymi_opt_entry;
int64_t ymi_opt_chunk_elems(void);
/* partial sums of squares of (gradient * scale) into partials[partials_offset ..]; finalize != 0 (last launch of a step):
 * fixed-order sum of all partials_total partials -> state.total_norm, state.clip = min(1, max_norm / (norm + 1e-6)),
 * state.updates += 1, state.ema_d = decay * (1 - exp(-updates / tau)). */
int ymi_opt_grad_norm(const ymi_opt_entry* table, const int32_t* chunk_map, int32_t first_tensor, int32_t n_tensors, int64_t n_chunks,
                      const float* const* host_grads, const float* hyper, float* partials, int64_t partials_offset, int64_t partials_total,
                      void* state, int32_t finalize, void* stream);
/* rule 0: g = grad*scale*clip (+ wd*p); buf = momentum*buf + g; g = nesterov ? g + momentum*buf : buf; p -= lr*g;
 * rule 1 / 2 (torch.optim.AdamW / Adam): p *= 1 - lr*wd (Adam: g += wd*p); m = m + (g-m)(1-beta1); v = beta2 v + (1-beta2) g g;
 *   p -= lr/(1-beta1^t) * m / (sqrt(v)/sqrt(1-beta2^t) + eps);
 * rule 3 .. 6 (torch.optim.Adamax / NAdam / RAdam / RMSprop with momentum, torch's default hyper-parameters): the single-tensor
 *   update of each, weight decay added to the gradient;
 * then ema = d*ema + (1-d)*p.  `rule` must equal hyper[14] (it selects the compiled kernel; pass 2 reads hyper[14] for the bias corrections).  host_grads == NULL: EMA-only pass over the given tensors (buffers, frozen parameters). */
int ymi_opt_update(const ymi_opt_entry* table, const int32_t* chunk_map, int32_t first_tensor, int32_t n_tensors, int64_t n_chunks,
                   const float* const* host_grads, const float* hyper, const void* state, int32_t rule, void* stream);

/* SwinBlock MLP: Linear(C,4C) -> exact GELU -> Linear(4C,C) (+ skip).  Reference: ultralytics/nn/modules/swin_block.py:33 (definition)
 * and :53 (`x = x + self.mlp(self.norm2(x))`).  Token matrices are ymi_tensors with n = h = 1, w = tokens.
 * fwd: pre = u W1^T + b1 and post = gelu(pre) are both written by fc1's epilogue; out = post W2^T + b2 (+ residual).
 * bwd_data: dpre = (dout W2) * gelu'(pre) from fc2's data-gradient epilogue; du = dpre W1 (+ add1 + add2); du may be NULL.
 * Weights are the ymi_pack_conv_weight_fwd / _dgrad images of the two nn.Linear weights viewed as 1x1 convolutions. */
int ymi_swin_mlp_fwd(const ymi_tensor* u, const void* w1_packed, const float* b1, int64_t hidden, const void* w2_packed, const float* b2,
                     const ymi_tensor* residual, const ymi_tensor* pre, const ymi_tensor* post, const ymi_tensor* out, void* stream);
int ymi_swin_mlp_bwd_data(const ymi_tensor* dout, const void* w2_dgrad_packed, const ymi_tensor* pre, const ymi_tensor* dpre,
                          const void* w1_dgrad_packed, const ymi_tensor* add1, const ymi_tensor* add2, const ymi_tensor* du, void* stream);

/* SwinBlock's second half as one kernel per direction: out = x + fc2(gelu(fc1(LayerNorm(x)))) - swin_block.py:53 with the modules of
 * swin_block.py:30-35 (norm2, mlp) - for bfloat16 tokens of 256 channels (csrc/swin_mlp.hip; `supported` says whether a shape takes this
 * path, anything else keeps ymi_layernorm_fwd + ymi_swin_mlp_fwd).  The [T, hidden] activations never reach HBM in the forward except, when
 * training, the bf16 pre-activations once, in a private register-order layout of ymi_swin_ln_mlp_pre_elems(T, hidden) elements.
 *   pack    : w1 [hidden][c], w2 [c][hidden] (float32, nn.Linear layouts) -> `packed`, ymi_swin_ln_mlp_pack_elems bfloat16 elements (four images)
 *   fwd     : u / mean / rstd / pre non-null = training (u: LayerNorm output [T][c], saved for fc1's weight gradient and LayerNorm's backward)
 *   bwd_data: post = gelu(pre) and dpre = (dout W2) * gelu'(pre), both [T][hidden] row-major for the two weight-gradient GEMMs
 *             (ymi_conv2d_bwd_weight on (post, dout) and (u, dpre)); du = dpre W1 [T][c], LayerNorm's incoming gradient.  The kernel stores whole
 *             256-token tiles: post / dpre are dense (ld == hidden) views of the first T rows of buffers of ymi_swin_ln_mlp_pre_elems elements. */
int ymi_swin_ln_mlp_supported(int64_t c, int64_t hidden, int32_t dtype);
int64_t ymi_swin_ln_mlp_pack_elems(int64_t c, int64_t hidden);
int64_t ymi_swin_ln_mlp_pre_elems(int64_t tokens, int64_t hidden);
int ymi_swin_ln_mlp_pack(const float* w1, const float* w2, int64_t c, int64_t hidden, void* packed, void* stream);
int ymi_swin_ln_mlp_fwd(const ymi_tensor* x, const float* gamma, const float* beta, float eps, const void* packed, const float* b1, const float* b2,
                        int64_t hidden, const ymi_tensor* u, float* mean, float* rstd, void* pre, const ymi_tensor* out, void* stream);
int ymi_swin_ln_mlp_bwd_data(const ymi_tensor* dout, const void* packed, const void* pre, int64_t hidden, const ymi_tensor* post, const ymi_tensor* dpre,
                             const ymi_tensor* du, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* YMI_H */
