// The model's FIRST convolution (reference yolov8.yaml:738 `Conv [64, 3, 2]` through nn/modules/conv.py:50-79): 3 input channels, 3x3,
// stride 2, on the float32 NCHW image the caller hands over - as a direct kernel.
//
// Through the implicit-GEMM path this layer cost a layout pass (NCHW float32 -> NHWC bfloat16 with the 3 channels zero-padded to a
// 16-byte chunk: 157 MB read, 52 MB written) and a GEMM whose K axis is 9 taps x 8 padded channels = 72 for 27 real products, i.e.
// 62 % padding, with 16-byte gathers of which 6 bytes are data (112 + 71 us at bs 32, 640 x 640).  Here one workgroup owns a tile of
// TH x TW output pixels of one image: it reads the (2 TH + 1) x (2 TW + 1) x 3 input patch straight from the NCHW planes (rows of
// 516 contiguous bytes), keeps it in LDS as bfloat16 [row][col][4] (8 bytes per pixel, the 4th channel zero), and
//   * writes the NHWC bfloat16 8-channel copy of its part of the image as a by-product (the weight-gradient GEMM of the backward pass
//     reads that copy: the separate layout pass disappears);
//   * forms the 16 x 16 x 32 MFMA fragments from the patch: K is ordered (kh, kw, c') with c' in 0..3, so every 8-byte LDS read is one
//     input pixel's channels and K = 36 (padded to 64: two MFMAs per 16 pixels x 16 channels; the MFMA pipe is idle in this
//     HBM-bound kernel anyway);
//   * rounds to bfloat16, accumulates the BatchNorm partial sums of the rounded values (one row per workgroup, deterministic) and
//     leaves through LDS so that every store instruction writes whole lines of the NHWC output.
// Arithmetic: bfloat16 products accumulated in float32 over the 27 taps, as the GEMM path computes them.
#include "common.h"

namespace {

constexpr int FC_TH = 8, FC_TW = 64;                   // output tile: two rows per wave
constexpr int FC_PR = 2 * FC_TH + 1;                   // patch rows
constexpr int FC_ROWB = (2 * FC_TW + 2) * 8;           // bytes per patch row: 8 bytes per pixel; col 0 unused, col 1 = left halo, cols 2.. = the tile's 2 TW columns
                                                       // (so that the aligned 4-pixel groups of the interior start on 16-byte boundaries)

struct FirstConvArgs {
    const float* img;
    const float* w;
    void* raw;
    void* x8;
    float* partials;
    int64_t ldraw, ldx8;
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
__device__ __forceinline__ uint32_t fc_pack2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    const bf16x2 v = {(bf16_t)lo, (bf16_t)hi};
    return __builtin_bit_cast(uint32_t, v);
}

template <int CO>
__global__ __launch_bounds__(256) void first_conv_kernel(FirstConvArgs a) {
    constexpr int NCT = CO / 16;                       // channel tiles of 16
    constexpr int OUTB = FC_TW * CO * 2;               // bytes of one wave's output row segment
    __shared__ __attribute__((aligned(16))) char patch[FC_PR * FC_ROWB];
    __shared__ __attribute__((aligned(16))) char stage[4 * OUTB];
    __shared__ __attribute__((aligned(16))) bf16_t wtab[CO * 64];  // [co][k], k = (kh * 3 + kw) * 4 + c'
    __shared__ float red[4][2][CO];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    // XCD ownership of the pixel order (common.h): consecutive units - tiles of one image, image after image - run on one XCD
    const int unit = xcd_unit(blockIdx.x, a.total);
    const int tw = unit % a.tiles_w;
    const int t2 = unit / a.tiles_w;
    const int th = t2 % a.tiles_h, n = t2 / a.tiles_h;
    const int oh0 = th * FC_TH, ow0 = tw * FC_TW;
    const int ih0 = 2 * oh0 - 1;

    // ---- input patch -> LDS as bfloat16 [row][col][4 channels]: an item = one patch row x 4 consecutive image columns (16-byte aligned
    // in the NCHW planes: W % 4 == 0), its up to three channels fetched by three 16-byte loads, stored as two 16-byte LDS writes
    {
        const float* src = a.img + (int64_t)n * a.C * a.H * a.W;
        const int64_t plane = (int64_t)a.H * a.W;
        constexpr int ITEMS = FC_PR * (2 * FC_TW / 4);
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
        // left halo column (iw = 2 ow0 - 1) and, for C == 4, the fourth channel: scalar loads by the first threads
        float halo = 0.f;
        const int hr = tid / 4, hc = tid % 4;  // (row, channel) of this thread's halo element
        if (tid < FC_PR * 4) {
            const int ih = ih0 + hr, iw = 2 * ow0 - 1;
            if (hc < a.C && (unsigned)ih < (unsigned)a.H && iw >= 0) halo = src[hc * plane + (int64_t)ih * a.W + iw];
        }
        // the weight table beside them (float32 OIHW parameter -> bfloat16 [co][k])
        float wv[CO * 64 / 256];
#pragma unroll
        for (int i = 0; i < CO * 64 / 256; ++i) {
            const int e = tid + 256 * i, co = e >> 6, k = e & 63, q = k >> 2, c = k & 3;
            wv[i] = (q < 9 && c < a.C) ? a.w[(co * a.C + c) * 9 + q] : 0.f;
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
#pragma unroll
        for (int i = 0; i < CO * 64 / 256; ++i) wtab[tid + 256 * i] = (bf16_t)wv[i];
        if (a.C == 4) {  // (rare: a fourth input channel - fill it in with scalar loads; the 16-byte stores above wrote zeros there)
            __syncthreads();
            for (int e = tid; e < FC_PR * 2 * FC_TW; e += 256) {
                const int r = e / (2 * FC_TW), cc = e % (2 * FC_TW);
                const int ih = ih0 + r, iw = 2 * ow0 + cc;
                if ((unsigned)ih < (unsigned)a.H && iw < a.W) *reinterpret_cast<bf16_t*>(patch + r * FC_ROWB + (cc + 2) * 8 + 6) = (bf16_t)src[3 * plane + (int64_t)ih * a.W + iw];
            }
        }
    }
    __syncthreads();

    // ---- by-product: the NHWC bfloat16 8-channel copy of the image pixels this tile owns (rows 2 oh0 .. 2 oh0 + 2 TH - 1, cols likewise)
    if (a.x8) {
        bf16_t* x8 = reinterpret_cast<bf16_t*>(a.x8);
#pragma unroll 4
        for (int e = tid; e < 2 * FC_TH * 2 * FC_TW; e += 256) {
            const int r = e / (2 * FC_TW), cc = e - r * (2 * FC_TW);
            const int ih = 2 * oh0 + r, iw = 2 * ow0 + cc;
            if (ih < a.H && iw < a.W) {
                const u32x2 px = *reinterpret_cast<const u32x2*>(patch + (r + 1) * FC_ROWB + (cc + 2) * 8);
                *reinterpret_cast<u32x4*>(x8 + (((int64_t)n * a.H + ih) * a.W + iw) * a.ldx8) = u32x4{px[0], px[1], 0u, 0u};
            }
        }
    }

    // ---- weight fragments (A operand: rows = output channels, K = (kh, kw, c'))
    bf16x8 wf[NCT][2];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int h = 0; h < 2; ++h) wf[ct][h] = *reinterpret_cast<const bf16x8*>(wtab + (ct * 16 + l15) * 64 + 32 * h + 8 * l4);

    char* my = stage + wave * OUTB;
    float s1[NCT][4], s2[NCT][4];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[ct][r] = s2[ct][r] = 0.f;
    bf16_t* raw = reinterpret_cast<bf16_t*>(a.raw);
    constexpr int CPP = CO / 8;                        // 16-byte chunks per output pixel

#pragma unroll 1
    for (int rr = 0; rr < FC_TH / 4; ++rr) {           // this wave's output rows: wave, wave + 4
        const int ohl = wave + 4 * rr;
        const int oh = oh0 + ohl;
        const bool row_ok = oh < a.Ho;
        // ---- MFMA: four 16-pixel groups of the row
        f32x4 acc[4][NCT];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[mt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
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
                acc[mt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ct][0], x0, acc[mt][ct], 0, 0, 0);
                acc[mt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ct][1], x1, acc[mt][ct], 0, 0, 0);
            }
        }
        // ---- round, statistics of the rounded values over the valid pixels, stage the row segment in LDS
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int owl = mt * 16 + l15;
            const bool ok = row_ok && ow0 + owl < a.Wo;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    o[r] = (bf16_t)acc[mt][ct][r];
                    const float v = ok ? (float)o[r] : 0.f;
                    s1[ct][r] += v;
                    s2[ct][r] += v * v;
                }
                *reinterpret_cast<bf16x4*>(my + (owl * CO + ct * 16 + 4 * l4) * 2) = o;
            }
        }
        // the wave's 64 pixels x CO channels leave as 16-byte chunks: consecutive lanes, consecutive bytes of the NHWC row
        // (the staging area is this wave's own: its LDS writes and reads are ordered by the wave's program order + lgkmcnt)
        if (row_ok) {
            const int64_t rowbase = (((int64_t)n * a.Ho + oh) * a.Wo + ow0) * a.ldraw;
#pragma unroll
            for (int it = 0; it < FC_TW * CPP / 64; ++it) {
                const int q = it * 64 + lane;
                const int px = q / CPP, cc = q % CPP;
                if (ow0 + px < a.Wo) *reinterpret_cast<u32x4*>(raw + rowbase + (int64_t)px * a.ldraw + cc * 8) = *reinterpret_cast<const u32x4*>(my + q * 16);
            }
        }
    }
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float t1 = fc_row16_sum(s1[ct][r]), t2s = fc_row16_sum(s2[ct][r]);
            if (l15 == 0) {
                red[wave][0][ct * 16 + 4 * l4 + r] = t1;
                red[wave][1][ct * 16 + 4 * l4 + r] = t2s;
            }
        }
    __syncthreads();
    if (tid < 2 * CO) {
        const int which = tid / CO, ch = tid % CO;
        a.partials[((int64_t)unit * 2 + which) * CO + ch] = (red[0][which][ch] + red[1][which][ch]) + (red[2][which][ch] + red[3][which][ch]);
    }
}

}  // namespace

extern "C" int64_t ymi_first_conv_stat_blocks(int64_t n, int64_t h, int64_t w) {
    const int64_t ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    return n * ((ho + FC_TH - 1) / FC_TH) * ((wo + FC_TW - 1) / FC_TW);
}

// act(BatchNorm_train(conv3x3 stride 2 pad 1 (img))) for the float32 NCHW image: direct convolution (raw output + statistics), finalize,
// affine + activation.  x8 (optional): receives the NHWC bfloat16 copy of the image with the channels zero-padded to 8 - what the
// weight-gradient GEMM of the backward pass reads.  bfloat16 outputs only; cout in {16, 32, 48, 64}; c <= 4.
// workspace: (2 * cout + (ymi_first_conv_stat_blocks + 64) * 2 * cout) floats.
extern "C" int ymi_first_conv_bn_act_fwd(const float* img_nchw, int64_t n, int64_t c, int64_t h, int64_t w, const float* weight_oihw, int64_t cout,
                                         const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                         int32_t act, const ymi_tensor* x8, const ymi_tensor* raw, const ymi_tensor* out, float* save_mean,
                                         float* save_invstd, void* workspace, size_t workspace_bytes, void* stream) {
    YMI_CHECK_ARG(img_nchw && weight_oihw && ymi_tensor_ok(raw) && ymi_tensor_ok(out) && workspace, "first_conv: bad argument");
    YMI_CHECK_ARG(c >= 1 && c <= 4 && (cout == 16 || cout == 32 || cout == 48 || cout == 64), "first_conv: c <= 4 input channels, 16 / 32 / 48 / 64 output channels");
    const int64_t ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    YMI_CHECK_ARG(raw->dtype == YMI_BF16 && out->dtype == YMI_BF16 && raw->n == n && raw->h == ho && raw->w == wo && raw->c == cout && ymi_same_shape(raw, out),
                  "first_conv: output shape / dtype (bfloat16 [n, cout, h/2, w/2])");
    YMI_CHECK_ARG(raw->ld % 8 == 0 && ((uintptr_t)raw->data & 15) == 0, "first_conv: 16-byte aligned output rows");
    if (x8) YMI_CHECK_ARG(ymi_tensor_ok(x8) && x8->dtype == YMI_BF16 && x8->n == n && x8->h == h && x8->w == w && x8->c == 8 && x8->ld % 8 == 0 && ((uintptr_t)x8->data & 15) == 0,
                          "first_conv: the image copy must be bfloat16 [n, 8, h, w] NHWC");
    YMI_CHECK_ARG(n * ho * wo < (1ll << 31) && n * h * w * 8 < (1ll << 31) * 8, "first_conv: too large");
    YMI_CHECK_ARG(w % 4 == 0 && ((uintptr_t)img_nchw & 15) == 0, "first_conv: image rows must be whole 16-byte groups (w %% 4 == 0, 16-byte aligned base)");
    FirstConvArgs a{};
    a.img = img_nchw; a.w = weight_oihw; a.raw = raw->data; a.x8 = x8 ? x8->data : nullptr;
    a.ldraw = raw->ld; a.ldx8 = x8 ? x8->ld : 0;
    a.N = (int)n; a.C = (int)c; a.H = (int)h; a.W = (int)w; a.Ho = (int)ho; a.Wo = (int)wo;
    a.tiles_h = (int)((ho + FC_TH - 1) / FC_TH); a.tiles_w = (int)((wo + FC_TW - 1) / FC_TW);
    const int64_t blocks = n * a.tiles_h * a.tiles_w;
    a.total = (int)blocks;
    const size_t need = (size_t)(2 * cout + (blocks + 64) * 2 * cout) * sizeof(float);
    if (workspace_bytes < need) {
        ymi_set_error("first_conv: workspace %zu < %zu bytes", workspace_bytes, need);
        return YMI_EWORKSPACE;
    }
    float* scale = reinterpret_cast<float*>(workspace);
    float* shift = scale + cout;
    a.partials = shift + cout;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)blocks), blk(256);
    if (cout == 16) hipLaunchKernelGGL(first_conv_kernel<16>, grid, blk, 0, s, a);
    else if (cout == 32) hipLaunchKernelGGL(first_conv_kernel<32>, grid, blk, 0, s, a);
    else if (cout == 48) hipLaunchKernelGGL(first_conv_kernel<48>, grid, blk, 0, s, a);
    else hipLaunchKernelGGL(first_conv_kernel<64>, grid, blk, 0, s, a);
    YMI_CHECK_LAUNCH("first_conv");
    int rc = ymi_bn_finalize(a.partials, blocks, n * ho * wo, cout, gamma, beta, running_mean, running_var, momentum, eps, scale, shift, save_mean, save_invstd, stream);
    if (rc) return rc;
    return ymi_scale_shift_act(raw, scale, shift, act, nullptr, out, stream);
}
