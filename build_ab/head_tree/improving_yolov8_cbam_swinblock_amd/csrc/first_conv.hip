// The model's FIRST Conv block (reference yolov8.yaml:738 `Conv [64, 3, 2]` through nn/modules/conv.py:50-79): 3 input channels, 3x3,
// stride 2, on the float32 NCHW image the caller hands over - as direct kernels, forward AND backward, that never store the raw
// convolution output.
//
// Why: this layer has 27 products per output and 3.3 M x 32 outputs at bs 32, 640 x 640: it is pure memory traffic.  Through the generic
// path it cost a layout pass (NCHW float32 -> NHWC bfloat16, channels padded to 8), an implicit GEMM whose K axis is 62 % padding, and the
// four BatchNorm passes of every Conv block over a 210 MB raw output - 515 us of a 13 ms step for 0.4 % of its FLOPs.  Recomputing the
// convolution is cheaper than reading it back, so here the raw output exists only in registers:
//   forward   K1  image (float32 NCHW, 157 MB) -> BatchNorm partial sums of the bfloat16-rounded outputs; by-product: the image as
//                 NHWC bfloat16 with 4 channels (26 MB, "x4"), the form every later kernel reads
//             --  finalize (ymi_bn_finalize, csrc/elementwise.hip)
//             K2  x4 -> convolution again -> scale / shift / SiLU -> the block's output (210 MB)
//   backward  K3  x4, dout -> convolution again -> partial sums of dz and dz * xhat  (dz = dout * act'(z))
//             --  final sums + apply coefficients (ymi_bn_bwd_final, csrc/reduce_bwd.hip)
//             K4  x4, dout -> convolution again -> d(raw) = dz c0 - raw c1 - c2 (rounded to bfloat16 as the generic path stores it)
//                 -> dW[k][co] += x[pixel, k] * d(raw)[pixel, co] by MFMA with the pixels as the K axis -> float32 slabs per workgroup,
//                 summed by the batched slab sum of csrc/wgrad.hip (deterministic).  The image gets no gradient.
// A workgroup owns a tile of TH x TW output pixels of one image; the (2 TH + 1) x (2 TW + 1) input patch sits in LDS as bfloat16
// [row][col][4] (8 bytes per pixel, the 4th channel zero).  K is ordered (kh, kw, c') with c' in 0..3, so every 8-byte LDS read is one
// input pixel's channels and K = 36, padded to 64: two 16x16x32 MFMAs per 16 pixels x 16 channels (the MFMA pipe is idle in these
// HBM-bound kernels anyway).  Arithmetic: bfloat16 products accumulated in float32 over the 27 taps, rounded to bfloat16 - exactly
// what the generic path computes and stores, so every consumer of the recomputed value sees the stored value's bits.
// (Handing dz from K3 to K4 instead of recomputing SiLU' there was measured: as bfloat16 it puts 2e-2 on dW - d(raw) cancels - and as
// float32 its 420 MB cost K3 more than K4 gained, 167 + 191 against 100 + 190 us.)
#include "common.h"

int ymi_bn_bwd_final(const float* part, int blocks, int C, const float* gamma, const float* beta, const float* mean, const float* inv, float inv_count,
                     float* dgamma, float* dbeta, float* coef, hipStream_t stream);

namespace {

constexpr int FC_TH = 8, FC_TW = 64;                   // output tile: two rows per wave
constexpr int FC_PR = 2 * FC_TH + 1;                   // patch rows
constexpr int FC_ROWB = (2 * FC_TW + 2) * 8;           // bytes per patch row: 8 bytes per pixel; col 0 unused, col 1 = left halo, cols 2.. = the tile's 2 TW columns
                                                       // (so that the aligned groups of the interior start on 16-byte boundaries)
#ifndef YMI_FC_WG_TILES
#define YMI_FC_WG_TILES 4
#endif
constexpr int FC_WG_TILES = YMI_FC_WG_TILES;           // tiles per workgroup of the weight-gradient kernel (one slab per workgroup)

enum { FC_STATS = 0, FC_APPLY = 1, FC_BWD_REDUCE = 2, FC_BWD_WGRAD = 3 };

struct FirstConvArgs {
    const float* img;      // FC_STATS: float32 NCHW image
    const float* w;        // float32 OIHW weight
    void* x4;              // NHWC bfloat16, 4 channels: written by FC_STATS, read by the others
    void* out;             // FC_APPLY: block output; FC_BWD_*: dout (read)
    int64_t ldout;
    float* partials;       // FC_STATS / FC_BWD_REDUCE: [unit][2][CO]
    const float* v0;       // FC_APPLY: scale          FC_BWD_REDUCE: gamma   FC_BWD_WGRAD: coef [5][CO]
    const float* v1;       // FC_APPLY: shift          FC_BWD_REDUCE: beta
    const float* v2;       //                          FC_BWD_REDUCE: mean
    const float* v3;       //                          FC_BWD_REDUCE: inv-std
    float* slab;           // FC_BWD_WGRAD: [workgroup][CO][72] float32
    int act;
    int N, C, H, W, Ho, Wo;
    int tiles_h, tiles_w, total;
};

template <int CTRL> __device__ __forceinline__ float fc_dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float fc_row16_sum(float v) {  // sum over the 16 lanes of a DPP row, in every lane (as igemm.hip)
    v = fc_dpp_add<0xB1>(v);
    v = fc_dpp_add<0x4E>(v);
    v = fc_dpp_add<0x141>(v);
    v = fc_dpp_add<0x140>(v);
    return v;
}
// SiLU or identity (the host refuses other activations: GELU's erf would double these kernels' code for a case no model has)
__device__ __forceinline__ float fc_act(float x, int act) { return act == YMI_ACT_SILU ? silu_f(x) : x; }
__device__ __forceinline__ float fc_act_grad(float x, int act) { return act == YMI_ACT_SILU ? silu_grad_f(x) : 1.0f; }
__device__ __forceinline__ uint32_t fc_pack2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    const bf16x2 v = {(bf16_t)lo, (bf16_t)hi};
    return __builtin_bit_cast(uint32_t, v);
}

// ---- transposed fragment read (as csrc/wgrad.hip's WFrag): 8 consecutive ROWS (pixels 8 l4 .. 8 l4 + 7 of a 32-row block) of column c0 + l15 of a
// [row][column] bfloat16 image with `rowb` bytes per row, by two ds_read_b64_tr_b16.  The 8-byte unit index is XORed with a per-row value
// (SW = 2: 64-byte rows, SW = 0: 128-byte rows) so that the 8 rows a 32-lane half addresses fall into different banks; writers store unit u
// of row r at unit u ^ (fc_swz<SW>(r) << 2).
template <int SW> __device__ __forceinline__ int fc_swz(int row) { return SW == 2 ? ((row >> 3) & 1) : (((row >> 1) & 1) | (((row >> 3) & 1) << 1)); }
template <int SW> __device__ __forceinline__ bf16x8 fc_tr_load(const char* img, int rowb, int c0, int lane) {
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int r_lo = 8 * g + q, r_hi = r_lo + 4;
    const int u = (c0 >> 2) + p;
    const int u_lo = u ^ (fc_swz<SW>(r_lo) << 2), u_hi = u ^ (fc_swz<SW>(r_hi) << 2);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + r_lo * rowb + u_lo * 8));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + r_hi * rowb + u_hi * 8));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// ---- the input patch of tile (n, oh0, ow0) into LDS -----------------------------------------------------------------------------------
// from the float32 NCHW image (and, by the way, the tile's own pixels out to x4) ...
__device__ __forceinline__ void fc_patch_from_image(const FirstConvArgs& a, char* patch, int n, int oh0, int ow0, int tid) {
    const float* src = a.img + (int64_t)n * a.C * a.H * a.W;
    const int64_t plane = (int64_t)a.H * a.W;
    const int ih0 = 2 * oh0 - 1;
    constexpr int ITEMS = FC_PR * (2 * FC_TW / 4);     // an item = one patch row x 4 consecutive image columns (16-byte aligned in the planes)
    constexpr int PASSES = (ITEMS + 255) / 256;
    f32x4 v[PASSES][3];
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int it = tid + 256 * p;
        const int r = it / (2 * FC_TW / 4), g = it % (2 * FC_TW / 4);
        const int ih = ih0 + r, iw = 2 * ow0 + 4 * g;
        const bool ok = it < ITEMS && (unsigned)ih < (unsigned)a.H && iw < a.W;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[p][c] = (ok && c < a.C) ? *reinterpret_cast<const f32x4*>(src + c * plane + (int64_t)ih * a.W + iw) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float halo = 0.f;                                  // left halo column (iw = 2 ow0 - 1): scalar loads by the first threads
    const int hr = tid / 4, hc = tid % 4;
    if (tid < FC_PR * 4) {
        const int ih = ih0 + hr, iw = 2 * ow0 - 1;
        if (hc < a.C && (unsigned)ih < (unsigned)a.H && iw >= 0) halo = src[hc * plane + (int64_t)ih * a.W + iw];
    }
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int it = tid + 256 * p;
        if (it < ITEMS) {
            const int r = it / (2 * FC_TW / 4), g = it % (2 * FC_TW / 4);
            char* dst = patch + r * FC_ROWB + (4 * g + 2) * 8;
            *reinterpret_cast<u32x4*>(dst) = u32x4{fc_pack2(v[p][0][0], v[p][1][0]), fc_pack2(v[p][2][0], 0.f), fc_pack2(v[p][0][1], v[p][1][1]), fc_pack2(v[p][2][1], 0.f)};
            *reinterpret_cast<u32x4*>(dst + 16) = u32x4{fc_pack2(v[p][0][2], v[p][1][2]), fc_pack2(v[p][2][2], 0.f), fc_pack2(v[p][0][3], v[p][1][3]), fc_pack2(v[p][2][3], 0.f)};
        }
    }
    if (tid < FC_PR * 4) *reinterpret_cast<bf16_t*>(patch + hr * FC_ROWB + 1 * 8 + hc * 2) = (bf16_t)halo;
    if (a.C == 4) {  // (rare: a fourth input channel - filled in with scalar loads; the 16-byte stores above wrote zeros there)
        __syncthreads();
        for (int e = tid; e < FC_PR * 2 * FC_TW; e += 256) {
            const int r = e / (2 * FC_TW), cc = e % (2 * FC_TW);
            const int ih = ih0 + r, iw = 2 * ow0 + cc;
            if ((unsigned)ih < (unsigned)a.H && iw < a.W) *reinterpret_cast<bf16_t*>(patch + r * FC_ROWB + (cc + 2) * 8 + 6) = (bf16_t)src[3 * plane + (int64_t)ih * a.W + iw];
        }
    }
    __syncthreads();
    // x4: the image pixels this tile owns (rows 2 oh0 .. 2 oh0 + 2 TH - 1, cols 2 ow0 .. 2 ow0 + 2 TW - 1), two pixels per 16-byte store
    bf16_t* x4 = reinterpret_cast<bf16_t*>(a.x4);
#pragma unroll 4
    for (int e = tid; e < 2 * FC_TH * FC_TW; e += 256) {
        const int r = e / FC_TW, cc = 2 * (e - r * FC_TW);
        const int ih = 2 * oh0 + r, iw = 2 * ow0 + cc;
        if (ih < a.H && iw < a.W) *reinterpret_cast<u32x4*>(x4 + (((int64_t)n * a.H + ih) * a.W + iw) * 4) = *reinterpret_cast<const u32x4*>(patch + (r + 1) * FC_ROWB + (cc + 2) * 8);
    }
}
// ... or from x4 (already in the patch's own form: 8 bytes per pixel)
__device__ __forceinline__ void fc_patch_from_x4(const FirstConvArgs& a, char* patch, int n, int oh0, int ow0, int tid) {
    const bf16_t* x4 = reinterpret_cast<const bf16_t*>(a.x4);
    const int ih0 = 2 * oh0 - 1;
    constexpr int ITEMS = FC_PR * FC_TW;               // an item = one patch row x 2 consecutive image columns (16 bytes)
    constexpr int PASSES = (ITEMS + 255) / 256;
    u32x4 v[PASSES];
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int it = tid + 256 * p;
        const int r = it / FC_TW, g = it % FC_TW;
        const int ih = ih0 + r, iw = 2 * ow0 + 2 * g;
        const bool ok = it < ITEMS && (unsigned)ih < (unsigned)a.H && iw < a.W;
        v[p] = ok ? *reinterpret_cast<const u32x4*>(x4 + (((int64_t)n * a.H + ih) * a.W + iw) * 4) : u32x4{0u, 0u, 0u, 0u};
    }
    u32x2 halo = u32x2{0u, 0u};
    if (tid < FC_PR) {
        const int ih = ih0 + tid, iw = 2 * ow0 - 1;
        if ((unsigned)ih < (unsigned)a.H && iw >= 0) halo = *reinterpret_cast<const u32x2*>(x4 + (((int64_t)n * a.H + ih) * a.W + iw) * 4);
    }
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int it = tid + 256 * p;
        if (it < ITEMS) *reinterpret_cast<u32x4*>(patch + (it / FC_TW) * FC_ROWB + (2 * (it % FC_TW) + 2) * 8) = v[p];
    }
    if (tid < FC_PR) *reinterpret_cast<u32x2*>(patch + tid * FC_ROWB + 8) = halo;
    __syncthreads();
}

// ---- 16 pixels x CO channels of the convolution from the patch: acc[ct][r] = channel ct*16 + 4*l4 + r of pixel mt*16 + l15 of the wave's
// output row, float32 -----------------------------------------------------------------------------------------------------------------
template <int NCT>
__device__ __forceinline__ void fc_conv_group(const char* patch, int ohl, int mt, int l15, int l4, const bf16x8 (&wf)[NCT][2], f32x4 (&acc)[NCT]) {
    const int owl = mt * 16 + l15;
    // K half 0: half-groups q = 2 l4, 2 l4 + 1 (all < 9); K half 1: q = 8 + 2 l4, 9 + 2 l4 - only q = 8 exists.  Patch column of
    // (output column owl, tap kw): image column 2 (ow0 + owl) + kw - 1 = patch column 2 owl + kw + 1
    const int q0 = 2 * l4, q1 = 2 * l4 + 1;
    const u32x2 lo0 = *reinterpret_cast<const u32x2*>(patch + (2 * ohl + q0 / 3) * FC_ROWB + (2 * owl + q0 % 3 + 1) * 8);
    const u32x2 hi0 = *reinterpret_cast<const u32x2*>(patch + (2 * ohl + q1 / 3) * FC_ROWB + (2 * owl + q1 % 3 + 1) * 8);
    u32x2 lo1 = u32x2{0u, 0u};
    if (l4 == 0) lo1 = *reinterpret_cast<const u32x2*>(patch + (2 * ohl + 2) * FC_ROWB + (2 * owl + 2 + 1) * 8);
    const bf16x8 x0 = __builtin_bit_cast(bf16x8, u32x4{lo0[0], lo0[1], hi0[0], hi0[1]});
    const bf16x8 x1 = __builtin_bit_cast(bf16x8, u32x4{lo1[0], lo1[1], 0u, 0u});
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ct][0], x0, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ct][1], x1, acc[ct], 0, 0, 0);
    }
}

template <int CO, int MODE>
__global__ __launch_bounds__(256, 4) void first_conv_kernel(FirstConvArgs a) {  // (4 waves per SIMD where LDS allows: these kernels live on latency hiding)
    constexpr int NCT = CO / 16;                       // channel tiles of 16
    constexpr int OUTB = FC_TW * CO * 2;               // bytes of one wave's output row segment
    constexpr int CPP = CO / 8;                        // 16-byte chunks per output pixel
    constexpr int DROWB = CO <= 32 ? 64 : 128, DSW = CO <= 32 ? 2 : 0;  // FC_BWD_WGRAD: row bytes / swizzle mode of the wave's [pixel][channel] image of d(raw)
    constexpr int STAGEB = MODE == FC_BWD_WGRAD ? FC_TW * DROWB : OUTB;
    __shared__ __attribute__((aligned(16))) char patch[FC_PR * FC_ROWB];
    __shared__ __attribute__((aligned(16))) char stage[(MODE == FC_APPLY || MODE == FC_BWD_WGRAD) ? 4 * STAGEB : 16];
    __shared__ __attribute__((aligned(16))) bf16_t wtab[CO * 64];  // [co][k], k = (kh * 3 + kw) * 4 + c'
    __shared__ __attribute__((aligned(16))) float ctab[5 * CO];    // per-channel constants (below)
    __shared__ float red[(MODE == FC_BWD_WGRAD) ? 48 * CO : 4 * 2 * CO];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;

    // ---- the weight table (float32 OIHW parameter -> bfloat16 [co][k]) and the per-channel constants [5][CO] (fetched from LDS per
    // 16-pixel group: as five register arrays per lane they cost two waves per SIMD)
#pragma unroll
    for (int i = 0; i < CO * 64 / 256; ++i) {
        const int e = tid + 256 * i, co = e >> 6, k = e & 63, q = k >> 2, c = k & 3;
        wtab[e] = (bf16_t)((q < 9 && c < a.C) ? a.w[(co * a.C + c) * 9 + q] : 0.f);
    }
    if (tid < CO) {
        float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f, c4 = 0.f;
        if constexpr (MODE == FC_APPLY) {
            c0 = a.v0[tid];                            // scale
            c1 = a.v1[tid];                            // shift
        } else if constexpr (MODE == FC_BWD_REDUCE) {
            const float g = a.v0 ? a.v0[tid] : 1.f, bb = a.v1 ? a.v1[tid] : 0.f, mu = a.v2[tid], iv = a.v3[tid];
            c0 = iv;                                   // xhat = raw * iv + c1
            c1 = -mu * iv;
            c2 = g;                                    // z = xhat * g + b
            c3 = bb;
        } else if constexpr (MODE == FC_BWD_WGRAD) {
            c0 = a.v0[tid];                            // z = raw a0 + a1 ; d(raw) = dz c0 - raw c1 - c2   (coef = [a0 | a1 | c0 | c1 | c2][CO])
            c1 = a.v0[CO + tid];
            c2 = a.v0[2 * CO + tid];
            c3 = a.v0[3 * CO + tid];
            c4 = a.v0[4 * CO + tid];
        }
        ctab[tid] = c0; ctab[CO + tid] = c1; ctab[2 * CO + tid] = c2; ctab[3 * CO + tid] = c3; ctab[4 * CO + tid] = c4;
    }
    __syncthreads();
    bf16x8 wf[NCT][2];                                 // this lane's weight fragments (A operand: rows = output channels)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int h = 0; h < 2; ++h) wf[ct][h] = *reinterpret_cast<const bf16x8*>(wtab + (ct * 16 + l15) * 64 + 32 * h + 8 * l4);

    float s1[NCT][4], s2[NCT][4];                      // FC_STATS / FC_BWD_REDUCE: this lane's partial sums
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[ct][r] = s2[ct][r] = 0.f;
    f32x4 dwa[3][NCT];                                 // FC_BWD_WGRAD: dW[k = kt*16 + 4*l4 + r][co = ct*16 + l15]
#pragma unroll
    for (int kt = 0; kt < 3; ++kt)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) dwa[kt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

    char* my = stage + ((MODE == FC_APPLY || MODE == FC_BWD_WGRAD) ? wave * STAGEB : 0);
    constexpr int TILES = MODE == FC_BWD_WGRAD ? FC_WG_TILES : 1;
    // XCD ownership of the pixel order (common.h): consecutive units - tiles of one image, image after image - run on one XCD
    const int wg_unit = xcd_unit(blockIdx.x, gridDim.x);
#pragma unroll 1
    for (int ti = 0; ti < TILES; ++ti) {
        const int unit = wg_unit * TILES + ti;
        if (unit >= a.total) break;                    // (workgroup-uniform)
        const int tw = unit % a.tiles_w;
        const int t2 = unit / a.tiles_w;
        const int th = t2 % a.tiles_h, n = t2 / a.tiles_h;
        const int oh0 = th * FC_TH, ow0 = tw * FC_TW;
        if (ti > 0) __syncthreads();                   // everybody has finished reading the previous tile's patch
        if constexpr (MODE == FC_STATS) fc_patch_from_image(a, patch, n, oh0, ow0, tid);
        else fc_patch_from_x4(a, patch, n, oh0, ow0, tid);

#pragma unroll 1
        for (int rr = 0; rr < FC_TH / 4; ++rr) {       // this wave's output rows: wave, wave + 4
            const int ohl = wave + 4 * rr;
            const int oh = oh0 + ohl;
            const bool row_ok = oh < a.Ho;
            const int64_t rowbase = (((int64_t)n * a.Ho + oh) * a.Wo + ow0) * a.ldout;
            // dout of a 16-pixel group (8-byte pieces: this lane's 4 channels of its pixel, per channel tile) is requested one group ahead
            // of its use.  The group loop is NOT unrolled: unrolled, the compiler interleaves the four groups and needs 211 registers
            // (two waves per SIMD; a scheduling barrier per group did not stop it) - rolled, 90.
            [[maybe_unused]] bf16x4 dy_cur[NCT], dy_nxt[NCT];
            [[maybe_unused]] auto load_dy = [&](int mt, bf16x4 (&dst)[NCT]) {
                const bf16_t* dyp = reinterpret_cast<const bf16_t*>(a.out);
                const int owl = mt * 16 + l15;
                const bool ok = row_ok && ow0 + owl < a.Wo;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
                    dst[ct] = ok ? *reinterpret_cast<const bf16x4*>(dyp + rowbase + (int64_t)owl * a.ldout + ct * 16 + 4 * l4) : bf16x4{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
            };
            if constexpr (MODE == FC_BWD_REDUCE || MODE == FC_BWD_WGRAD) load_dy(0, dy_cur);
#pragma unroll 1
            for (int mt = 0; mt < 4; ++mt) {
                const int owl = mt * 16 + l15;
                const bool ok = row_ok && ow0 + owl < a.Wo;
                if constexpr (MODE == FC_BWD_REDUCE || MODE == FC_BWD_WGRAD) {
                    if (mt + 1 < 4) load_dy(mt + 1, dy_nxt);
                }
                f32x4 acc[NCT];
                fc_conv_group<NCT>(patch, ohl, mt, l15, l4, wf, acc);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    int ch0 = ct * 16 + 4 * l4;
                    asm volatile("" : "+v"(ch0));  // the constants are re-read from LDS per group: hoisted out of the loop they are 40 registers per lane
                    float raw[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) raw[r] = (float)(bf16_t)acc[ct][r];  // the value the generic path stores
                    if constexpr (MODE == FC_STATS) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float v = ok ? raw[r] : 0.f;
                            s1[ct][r] += v;
                            s2[ct][r] += v * v;
                        }
                    } else if constexpr (MODE == FC_APPLY) {
                        const f32x4 sc = *reinterpret_cast<const f32x4*>(ctab + ch0), sh = *reinterpret_cast<const f32x4*>(ctab + CO + ch0);
                        bf16x4 o;
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = (bf16_t)fc_act(raw[r] * sc[r] + sh[r], a.act);
                        *reinterpret_cast<bf16x4*>(my + (owl * CO + ch0) * 2) = o;
                    } else if constexpr (MODE == FC_BWD_REDUCE) {
                        const f32x4 iv = *reinterpret_cast<const f32x4*>(ctab + ch0), of = *reinterpret_cast<const f32x4*>(ctab + CO + ch0);
                        const f32x4 g = *reinterpret_cast<const f32x4*>(ctab + 2 * CO + ch0), bb = *reinterpret_cast<const f32x4*>(ctab + 3 * CO + ch0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float xh = raw[r] * iv[r] + of[r];
                            const float dz = (float)dy_cur[ct][r] * fc_act_grad(xh * g[r] + bb[r], a.act);  // (dout = 0 outside the image)
                            s1[ct][r] += dz;
                            s2[ct][r] += dz * xh;
                        }
                    } else {
                        // d(raw), rounded to bfloat16 as the generic path stores it, into this wave's [pixel][channel] image (one 8-byte store;
                        // the weight-gradient MFMAs read it transposed: fc_tr_load)
                        const f32x4 a0 = *reinterpret_cast<const f32x4*>(ctab + ch0), a1 = *reinterpret_cast<const f32x4*>(ctab + CO + ch0);
                        const f32x4 c0 = *reinterpret_cast<const f32x4*>(ctab + 2 * CO + ch0), c1 = *reinterpret_cast<const f32x4*>(ctab + 3 * CO + ch0);
                        const f32x4 c2 = *reinterpret_cast<const f32x4*>(ctab + 4 * CO + ch0);
                        bf16x4 o;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float dz = (float)dy_cur[ct][r] * fc_act_grad(raw[r] * a0[r] + a1[r], a.act);
                            o[r] = (bf16_t)(ok ? dz * c0[r] - (raw[r] * c1[r] + c2[r]) : 0.f);
                        }
                        *reinterpret_cast<bf16x4*>(my + owl * DROWB + ((ch0 >> 2) ^ (fc_swz<DSW>(owl) << 2)) * 8) = o;
                    }
                }
                if constexpr (MODE == FC_BWD_REDUCE || MODE == FC_BWD_WGRAD) {
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) dy_cur[ct] = dy_nxt[ct];
                }
            }
            if constexpr (MODE == FC_APPLY) {
                // the wave's 64 pixels x CO channels leave as 16-byte chunks: consecutive lanes, consecutive bytes of the NHWC row
                // (the staging area is this wave's own: its LDS writes and reads are ordered by the wave's program order)
                if (row_ok) {
                    bf16_t* outp = reinterpret_cast<bf16_t*>(a.out);
#pragma unroll
                    for (int it = 0; it < FC_TW * CPP / 64; ++it) {
                        const int q = it * 64 + lane;
                        const int px = q / CPP, cc = q % CPP;
                        if (ow0 + px < a.Wo) *reinterpret_cast<u32x4*>(outp + rowbase + (int64_t)px * a.ldout + cc * 8) = *reinterpret_cast<const u32x4*>(my + q * 16);
                    }
                }
            } else if constexpr (MODE == FC_BWD_WGRAD) {
                // dW[k][co] += sum over the row's 64 pixels of x[pixel, k] * d(raw)[pixel, co]: pixels are the MFMA's K axis (two halves of 32).
                // A fragment (rows = k): lane (k = kt*16 + l15, pixels 32 h + 8 l4 + j): patch[(2 ohl + kh)][2 pixel + kw + 1][c'], k = (kh*3 + kw)*4 + c'
                // B fragment (cols = co): lane (co = ct*16 + l15, the same 8 pixels): transposed read of the [pixel][channel] image
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    bf16x8 df[NCT];
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) df[ct] = fc_tr_load<DSW>(my + 32 * h * DROWB, DROWB, ct * 16, lane);
#pragma unroll
                    for (int kt = 0; kt < 3; ++kt) {
                        const int k = kt * 16 + l15, q = k >> 2, c = k & 3;
                        bf16x8 xf;
                        if (q < 9) {
                            const char* base = patch + (2 * ohl + q / 3) * FC_ROWB + (q % 3 + 1) * 8 + c * 2 + (32 * h + 8 * l4) * 16;
#pragma unroll
                            for (int j = 0; j < 8; ++j) xf[j] = *reinterpret_cast<const bf16_t*>(base + j * 16);
                        } else {
#pragma unroll
                            for (int j = 0; j < 8; ++j) xf[j] = (bf16_t)0.f;
                        }
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct) dwa[kt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, df[ct], dwa[kt][ct], 0, 0, 0);
                    }
                }
            }
        }
        if constexpr (MODE == FC_STATS || MODE == FC_BWD_REDUCE) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float t1 = fc_row16_sum(s1[ct][r]), t2s = fc_row16_sum(s2[ct][r]);
                    if (l15 == 0) {
                        red[(wave * 2 + 0) * CO + ct * 16 + 4 * l4 + r] = t1;
                        red[(wave * 2 + 1) * CO + ct * 16 + 4 * l4 + r] = t2s;
                    }
                }
            __syncthreads();
            if (tid < 2 * CO) {
                const int which = tid / CO, ch = tid % CO;
                a.partials[((int64_t)unit * 2 + which) * CO + ch] = (red[(0 * 2 + which) * CO + ch] + red[(1 * 2 + which) * CO + ch]) + (red[(2 * 2 + which) * CO + ch] + red[(3 * 2 + which) * CO + ch]);
            }
        }
    }
    if constexpr (MODE == FC_BWD_WGRAD) {
        // the four waves' sums -> one float32 slab [co][tap * 8 + c] per workgroup (the layout of wgrad_kernel's slabs for 8 padded input
        // channels: the batched slab sum of csrc/wgrad.hip adds the workgroups' slabs in a fixed order and scatters into OIHW)
        for (int wv = 0; wv < 4; ++wv) {  // wave after wave into one [k][co] image (fixed order: deterministic)
            if (wave == wv) {
#pragma unroll
                for (int kt = 0; kt < 3; ++kt)
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float* dst = &red[(kt * 16 + 4 * l4 + r) * CO + ct * 16 + l15];
                            *dst = wv == 0 ? dwa[kt][ct][r] : *dst + dwa[kt][ct][r];
                        }
            }
            __syncthreads();
        }
        float* slab = a.slab + (int64_t)wg_unit * CO * 72;
        for (int e = tid; e < CO * 72; e += 256) {
            const int co = e / 72, col = e % 72, tap = col >> 3, c = col & 7;
            slab[e] = c < 4 ? red[(tap * 4 + c) * CO + co] : 0.f;
        }
    }
}

template <int MODE>
int fc_launch(const FirstConvArgs& a, int64_t cout, unsigned blocks, hipStream_t s) {
    dim3 grid(blocks), blk(256);
    if (cout == 16) hipLaunchKernelGGL((first_conv_kernel<16, MODE>), grid, blk, 0, s, a);
    else if (cout == 32) hipLaunchKernelGGL((first_conv_kernel<32, MODE>), grid, blk, 0, s, a);
    else if (cout == 48) hipLaunchKernelGGL((first_conv_kernel<48, MODE>), grid, blk, 0, s, a);
    else hipLaunchKernelGGL((first_conv_kernel<64, MODE>), grid, blk, 0, s, a);
    YMI_CHECK_LAUNCH("first_conv");
    return YMI_OK;
}

int fc_geometry(FirstConvArgs& a, int64_t n, int64_t c, int64_t h, int64_t w, int64_t cout) {
    YMI_CHECK_ARG(c >= 1 && c <= 4 && (cout == 16 || cout == 32 || cout == 48 || cout == 64), "first_conv: c <= 4 input channels, 16 / 32 / 48 / 64 output channels");
    YMI_CHECK_ARG(a.act == YMI_ACT_SILU || a.act == YMI_ACT_NONE, "first_conv: SiLU or no activation");
    YMI_CHECK_ARG(w % 4 == 0 && h % 2 == 0 && n > 0, "first_conv: even height, width a multiple of 4");
    const int64_t ho = h / 2, wo = w / 2;
    YMI_CHECK_ARG(n * ho * wo * cout < (1ll << 31) && n * h * w * 4 < (1ll << 31), "first_conv: too large for 32-bit indexing");
    a.N = (int)n; a.C = (int)c; a.H = (int)h; a.W = (int)w; a.Ho = (int)ho; a.Wo = (int)wo;
    a.tiles_h = (int)((ho + FC_TH - 1) / FC_TH); a.tiles_w = (int)((wo + FC_TW - 1) / FC_TW);
    a.total = (int)(n * a.tiles_h * a.tiles_w);
    return YMI_OK;
}

}  // namespace

extern "C" int64_t ymi_first_conv_stat_blocks(int64_t n, int64_t h, int64_t w) {
    const int64_t ho = h / 2, wo = w / 2;
    return n * ((ho + FC_TH - 1) / FC_TH) * ((wo + FC_TW - 1) / FC_TW);
}

// forward: statistics pass, finalize, apply pass.  x4: receives the image as NHWC bfloat16 [n, 4, h, w] (saved for the backward pass, which
// reads nothing else of the input).  out: bfloat16 [n, cout, h/2, w/2] NHWC.  workspace: (2 cout + (blocks + 64) 2 cout) floats.
extern "C" int ymi_first_conv_bn_act_fwd(const float* img_nchw, int64_t n, int64_t c, int64_t h, int64_t w, const float* weight_oihw, int64_t cout,
                                         const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                         int32_t act, const ymi_tensor* x4, const ymi_tensor* out, float* save_mean, float* save_invstd,
                                         void* workspace, size_t workspace_bytes, void* stream) {
    YMI_CHECK_ARG(img_nchw && weight_oihw && ymi_tensor_ok(out) && ymi_tensor_ok(x4) && workspace, "first_conv: bad argument");
    YMI_CHECK_ARG(((uintptr_t)img_nchw & 15) == 0, "first_conv: 16-byte aligned image");
    FirstConvArgs a{};
    a.act = act;
    int rc = fc_geometry(a, n, c, h, w, cout);
    if (rc) return rc;
    YMI_CHECK_ARG(out->dtype == YMI_BF16 && out->n == n && out->h == a.Ho && out->w == a.Wo && out->c == cout && out->ld % 8 == 0 && ((uintptr_t)out->data & 15) == 0,
                  "first_conv: output must be bfloat16 [n, cout, h/2, w/2] NHWC with 16-byte aligned rows");
    YMI_CHECK_ARG(x4->dtype == YMI_BF16 && x4->n == n && x4->h == h && x4->w == w && x4->c == 4 && x4->ld == 4 && ((uintptr_t)x4->data & 15) == 0,
                  "first_conv: the image copy must be dense bfloat16 [n, 4, h, w] NHWC");
    const int64_t blocks = a.total;
    const size_t need = (size_t)(2 * cout + (blocks + 64) * 2 * cout) * sizeof(float);
    if (workspace_bytes < need) {
        ymi_set_error("first_conv: workspace %zu < %zu bytes", workspace_bytes, need);
        return YMI_EWORKSPACE;
    }
    float* scale = reinterpret_cast<float*>(workspace);
    float* shift = scale + cout;
    a.img = img_nchw; a.w = weight_oihw; a.x4 = x4->data; a.partials = shift + cout;
    hipStream_t s = (hipStream_t)stream;
    rc = fc_launch<FC_STATS>(a, cout, (unsigned)blocks, s);
    if (rc) return rc;
    rc = ymi_bn_finalize(a.partials, blocks, n * a.Ho * a.Wo, cout, gamma, beta, running_mean, running_var, momentum, eps, scale, shift, save_mean, save_invstd, stream);
    if (rc) return rc;
    a.out = out->data; a.ldout = out->ld; a.v0 = scale; a.v1 = shift;
    return fc_launch<FC_APPLY>(a, cout, (unsigned)blocks, s);
}

extern "C" int64_t ymi_first_conv_bwd_workspace(int64_t n, int64_t h, int64_t w, int64_t cout) {
    const int64_t blocks = ymi_first_conv_stat_blocks(n, h, w), wgs = (blocks + FC_WG_TILES - 1) / FC_WG_TILES;
    return (int64_t)sizeof(float) * (blocks * 2 * cout + 5 * cout + wgs * cout * 72) + 1024;
}

// backward of the block w.r.t. its parameters (the image receives no gradient): dgamma, dbeta [cout] and the weight gradient.
// pending == NULL: dw_oihw [cout, c, 3, 3] is complete when the stream reaches this point; else the slab sum is left to
// ymi_wgrad_reduce_batch with the record written to *pending (as ymi_conv2d_bwd_weight_deferred).  workspace: ymi_first_conv_bwd_workspace.
extern "C" int ymi_first_conv_bn_act_bwd(const ymi_tensor* x4, const float* weight_oihw, int64_t c, int64_t cout, const float* gamma, const float* beta,
                                         const float* save_mean, const float* save_invstd, int32_t act, const ymi_tensor* dout, float* dgamma,
                                         float* dbeta, float* dw_oihw, void* workspace, size_t workspace_bytes, ymi_wgrad_pending* pending, void* stream) {
    YMI_CHECK_ARG(ymi_tensor_ok(x4) && ymi_tensor_ok(dout) && weight_oihw && save_mean && save_invstd && dgamma && dbeta && dw_oihw && workspace, "first_conv_bwd: bad argument");
    FirstConvArgs a{};
    a.act = act;
    int rc = fc_geometry(a, x4->n, c, x4->h, x4->w, cout);
    if (rc) return rc;
    YMI_CHECK_ARG(x4->dtype == YMI_BF16 && x4->c == 4 && x4->ld == 4 && ((uintptr_t)x4->data & 15) == 0, "first_conv_bwd: the image copy must be dense bfloat16 [n, 4, h, w] NHWC");
    YMI_CHECK_ARG(dout->dtype == YMI_BF16 && dout->n == x4->n && dout->h == a.Ho && dout->w == a.Wo && dout->c == cout && dout->ld % 4 == 0 && ((uintptr_t)dout->data & 7) == 0,
                  "first_conv_bwd: dout must be bfloat16 [n, cout, h/2, w/2] NHWC");
    const int64_t blocks = a.total, wgs = (blocks + FC_WG_TILES - 1) / FC_WG_TILES;
    if ((int64_t)workspace_bytes < ymi_first_conv_bwd_workspace(x4->n, x4->h, x4->w, cout)) {
        ymi_set_error("first_conv_bwd: workspace too small");
        return YMI_EWORKSPACE;
    }
    float* part = reinterpret_cast<float*>(workspace);
    float* coef = part + blocks * 2 * cout;
    float* slab = coef + 5 * cout;
    ymi_wgrad_pending* table = reinterpret_cast<ymi_wgrad_pending*>(slab + wgs * cout * 72);
    table = reinterpret_cast<ymi_wgrad_pending*>(((uintptr_t)table + 63) & ~(uintptr_t)63);
    hipStream_t s = (hipStream_t)stream;
    a.w = weight_oihw; a.x4 = x4->data; a.out = dout->data; a.ldout = dout->ld;
    a.partials = part; a.v0 = gamma; a.v1 = beta; a.v2 = save_mean; a.v3 = save_invstd;
    rc = fc_launch<FC_BWD_REDUCE>(a, cout, (unsigned)blocks, s);
    if (rc) return rc;
    rc = ymi_bn_bwd_final(part, (int)blocks, (int)cout, gamma, beta, save_mean, save_invstd, 1.0f / (float)((int64_t)x4->n * a.Ho * a.Wo), dgamma, dbeta, coef, s);
    if (rc) return rc;
    a.v0 = coef; a.slab = slab;
    rc = fc_launch<FC_BWD_WGRAD>(a, cout, (unsigned)wgs, s);
    if (rc) return rc;
    const int64_t elems = cout * 72;
    const int lanes = wgs > 128 ? 32 : wgs > 32 ? 16 : wgs > 8 ? 8 : 4;
    ymi_wgrad_pending rec{slab, dw_oihw, elems, (int32_t)wgs, 72, 8, (int32_t)cout, (int32_t)c, 9, lanes, 0, (int32_t)((elems / 4 + 256 / lanes - 1) / (256 / lanes)), 0};
    if (pending) {
        *pending = rec;
        return YMI_OK;
    }
    return ymi_wgrad_reduce_batch(&rec, 1, table, stream);
}
