// SwinBlock pieces (fork's nn/modules/swin_block.py) for NHWC tensors on gfx950.
//
//  * window_partition / window_reverse are pure index maps; since activations are already NHWC
//    ('b c h w -> b h w c' is free) they are folded into the LayerNorm-1 load (gather, zero for
//    padding pixels) and into the final store (scatter + crop).  Stand-alone copies and the integer
//    index map exist for the bit-exact test.
//  * LayerNorm: one wave per token, 16 bytes per lane, two-pass statistics in registers.
//  * window attention: one workgroup (4 waves) per window, heads in sequence; every product is a
//    16x16-tile "NT" MFMA (both operands K-contiguous rows in LDS, rows padded by 16 B so
//    ds_read_b128 fragment reads are conflict-free); 49 tokens pad to 64 with zero rows, pad keys
//    masked to -inf before the softmax; V (and, in backward, dO / Q / K) are transposed on the way
//    into LDS.  Token GEMMs (QKV, proj, MLP) are ymi_conv2d_fwd with a 1x1 "kernel".
#include <stdlib.h>

#include "common.h"

int ymi_chan_reduce_final(const float* part, int blocks, int C, float* out0, float* out1, hipStream_t stream);

struct SV {
    const void* p;
    int64_t ld;
};

// token t (window order) -> pixel of the padded grid, -1 if outside the real image.  swin_block.py:8-13
__device__ __forceinline__ int64_t token_pixel(int64_t t64, int H, int W, int Hp, int Wp, int ws, bool* real) {
    // token counts are < 2^31 (checked by the launchers): 32-bit division only
    const uint32_t t = (uint32_t)t64, L = (uint32_t)(ws * ws), uws = (uint32_t)ws;
    const uint32_t nww = (uint32_t)Wp / uws, nwh = (uint32_t)Hp / uws;
    const uint32_t win = t / L, tok = t - win * L;
    const uint32_t wrow = win / nww, ww = win - wrow * nww;
    const uint32_t b = wrow / nwh, wh = wrow - b * nwh;
    const uint32_t tr = tok / uws;
    const int h = (int)(wh * uws + tr), w = (int)(ww * uws + (tok - tr * uws));
    *real = (h < H) && (w < W);
    return ((int64_t)b * H + h) * W + w;  // index in the UNPADDED image (valid only when *real)
}

__global__ void window_index_kernel(int64_t T, int Hp, int Wp, int ws, int* __restrict__ out) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < T; t += (int64_t)gridDim.x * blockDim.x) {
        bool real;
        out[t] = (int)token_pixel(t, Hp, Wp, Hp, Wp, ws, &real);  // padded grid: every token is "real"
    }
}

extern "C" int ymi_window_partition_index(int64_t n, int64_t hp, int64_t wp, int64_t ws, int32_t* index, void* stream) {
    YMI_CHECK_ARG(index && ws > 0 && hp % ws == 0 && wp % ws == 0 && n > 0, "window_partition_index: args");
    const int64_t T = n * hp * wp;
    YMI_CHECK_ARG(T < (1ll << 31), "window_partition_index: too many tokens");
    hipLaunchKernelGGL(window_index_kernel, dim3((unsigned)((T + 255) / 256 > 4096 ? 4096 : (T + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       T, (int)hp, (int)wp, (int)ws, index);
    YMI_CHECK_LAUNCH("window_partition_index");
    return YMI_OK;
}

// DIR 0: tokens[t] = x[pixel(t)] (zeros for padding)   DIR 1: x[pixel(t)] = tokens[t] (padding dropped)
template <typename T, int DIR>
__global__ void window_move_kernel(SV x, SV tok, int64_t Tn, int H, int W, int Hp, int Wp, int ws, int C) {
    const int groups = C / 4;
    const uint32_t total = (uint32_t)(Tn * groups), ugroups = (uint32_t)groups;  // launcher: Tn * groups < 2^31
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint32_t tu = i / ugroups;
        const int g = (int)(i - tu * ugroups);
        const int64_t t = tu;
        bool real;
        const int64_t px = token_pixel(t, H, W, Hp, Wp, ws, &real);
        if (DIR == 0) {
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (real) Pack<T, 4>::load(reinterpret_cast<const T*>(x.p) + px * x.ld + g * 4, v);
            Pack<T, 4>::store(reinterpret_cast<T*>(const_cast<void*>(tok.p)) + t * tok.ld + g * 4, v);
        } else if (real) {
            float v[4];
            Pack<T, 4>::load(reinterpret_cast<const T*>(tok.p) + t * tok.ld + g * 4, v);
            Pack<T, 4>::store(reinterpret_cast<T*>(const_cast<void*>(x.p)) + px * x.ld + g * 4, v);
        }
    }
}

static int window_geometry(const ymi_tensor* x, int64_t ws, int64_t* Hp, int64_t* Wp) {
    *Hp = (x->h + ws - 1) / ws * ws;
    *Wp = (x->w + ws - 1) / ws * ws;
    return 0;
}

template <int DIR>
static int launch_window_move(const ymi_tensor* x, int64_t ws, const ymi_tensor* tokens, const char* what, void* stream) {
    YMI_CHECK_ARG(ymi_tensor_ok(x) && ymi_tensor_ok(tokens) && ws > 0, "%s: bad tensor", what);
    int64_t Hp, Wp;
    window_geometry(x, ws, &Hp, &Wp);
    const int64_t T = x->n * Hp * Wp;
    YMI_CHECK_ARG(ymi_pixels(tokens) == T && tokens->c == x->c && tokens->dtype == x->dtype, "%s: tokens must be [%lld, %lld]", what, (long long)T, (long long)x->c);
    YMI_CHECK_ARG(x->c % 4 == 0 && x->ld % 4 == 0 && tokens->ld % 4 == 0, "%s: channels multiple of 4", what);
    const int64_t total = T * (x->c / 4);
    YMI_CHECK_ARG(total < (1ll << 31), "%s: tensor too large for 32-bit indexing", what);
    const unsigned gb = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    SV xv{x->data, x->ld}, tv{tokens->data, tokens->ld};
    if (x->dtype == YMI_BF16)
        hipLaunchKernelGGL((window_move_kernel<bf16_t, DIR>), dim3(gb), dim3(256), 0, (hipStream_t)stream, xv, tv, T, (int)x->h, (int)x->w, (int)Hp, (int)Wp, (int)ws, (int)x->c);
    else
        hipLaunchKernelGGL((window_move_kernel<float, DIR>), dim3(gb), dim3(256), 0, (hipStream_t)stream, xv, tv, T, (int)x->h, (int)x->w, (int)Hp, (int)Wp, (int)ws, (int)x->c);
    YMI_CHECK_LAUNCH(what);
    return YMI_OK;
}
extern "C" int ymi_window_partition(const ymi_tensor* x, int64_t ws, const ymi_tensor* tokens, void* stream) {
    return launch_window_move<0>(x, ws, tokens, "window_partition", stream);
}
extern "C" int ymi_window_reverse(const ymi_tensor* tokens, int64_t ws, const ymi_tensor* x, void* stream) {
    return launch_window_move<1>(x, ws, tokens, "window_reverse", stream);
}

// -------------------------------------------------------------------------------- LayerNorm
// C <= 64 lanes * 4 channels * 4 groups = 1024
// the kernels are instantiated per number of ACTIVE 256-channel groups (C = 256 -> 1): the per-token loop is a serial
// load -> reduce -> store chain hidden only by resident waves, and register arrays sized for 1024 channels cut those
#define YMI_LN_G(LAUNCH, T)                     \
    do {                                        \
        const int g_ = ((int)x->c + 255) / 256; \
        if (g_ <= 1) LAUNCH(T, 1);              \
        else if (g_ == 2) LAUNCH(T, 2);         \
        else if (g_ == 3) LAUNCH(T, 3);         \
        else LAUNCH(T, 4);                      \
    } while (0)
#define YMI_LNF(T, G) hipLaunchKernelGGL((layernorm_fwd_kernel<T, G>), grid, dim3(256), 0, (hipStream_t)stream, xv, ov, T_, (int)x->h, (int)x->w, (int)Hp, (int)Wp, (int)ws, (int)x->c, gamma, beta, eps, mean, rstd)

// one wave per token; ws > 0: rows are gathered from the NHWC image through the window map
template <typename T, int G>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(SV x, SV out, int64_t Tn, int H, int W, int Hp, int Wp, int ws, int C,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                            float* __restrict__ mean, float* __restrict__ rstd) {
    const int lane = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= Tn) return;
    bool real = true;
    int64_t row = t;
    if (ws > 0) row = token_pixel(t, H, W, Hp, Wp, ws, &real);
    const T* xp = reinterpret_cast<const T*>(x.p) + row * x.ld;
    float v[G][4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const int c = lane * 4 + 256 * i;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[i][r] = 0.f;
        if (c < C && real) Pack<T, 4>::load(xp + c, v[i]);
#pragma unroll
        for (int r = 0; r < 4; ++r) s += v[i][r];
    }
    const float mu = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < C)
#pragma unroll
            for (int r = 0; r < 4; ++r) q += (v[i][r] - mu) * (v[i][r] - mu);
    }
    const float rs = rsqrtf(wave_sum(q) / (float)C + eps);
    T* op = reinterpret_cast<T*>(const_cast<void*>(out.p)) + t * out.ld;
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < C) {
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (v[i][r] - mu) * rs * gamma[c + r] + beta[c + r];
            Pack<T, 4>::store(op + c, o);
        }
    }
    if (lane == 0) {
        mean[t] = mu;
        rstd[t] = rs;
    }
}

// bf16, C <= 256 (a multiple of 8): HALF a wave per token, 16-byte accesses (8 channels per lane) - two tokens in flight per wave where the
// kernel above has one.  The per-token chain (load -> mean -> variance -> store) is pure latency; this form halves the number of chains per
// CU (19.9 -> 14.3 us for the 56448 x 256 token matrix of the model).
__global__ __launch_bounds__(256) void layernorm_fwd_half_kernel(SV x, SV out, int64_t Tn, int H, int W, int Hp, int Wp, int ws, int C,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                                 float* __restrict__ mean, float* __restrict__ rstd) {
    typedef bf16_t T;
    const int lane = threadIdx.x & 63, sub = lane & 31;
    const int64_t t = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + (lane >> 5);
    const int c = sub * 8;
    const bool tok = t < Tn, cok = c < C;
    bool real = true;
    int64_t row = tok ? t : 0;
    if (ws > 0 && tok) row = token_pixel(t, H, W, Hp, Wp, ws, &real);
    float v[8], g[8], b[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) { v[r] = 0.f; g[r] = 0.f; b[r] = 0.f; }
    if (tok && cok && real) Pack<T, 8>::load(reinterpret_cast<const T*>(x.p) + row * x.ld + c, v);
    if (cok) {
        const float4 g0 = *reinterpret_cast<const float4*>(gamma + c), g1 = *reinterpret_cast<const float4*>(gamma + c + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(beta + c), b1 = *reinterpret_cast<const float4*>(beta + c + 4);
        g[0] = g0.x; g[1] = g0.y; g[2] = g0.z; g[3] = g0.w; g[4] = g1.x; g[5] = g1.y; g[6] = g1.z; g[7] = g1.w;
        b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w; b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) s += v[r];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);  // within the half-wave (xor < 32)
    const float mu = s / (float)C;
    float q = 0.f;
    if (cok)
#pragma unroll
        for (int r = 0; r < 8; ++r) q += (v[r] - mu) * (v[r] - mu);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rs = rsqrtf(q / (float)C + eps);
    if (tok && cok) {
        float o8[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) o8[r] = (v[r] - mu) * rs * g[r] + b[r];
        Pack<T, 8>::store(reinterpret_cast<T*>(const_cast<void*>(out.p)) + t * out.ld + c, o8);
    }
    if (tok && sub == 0) {
        mean[t] = mu;
        rstd[t] = rs;
    }
}

extern "C" int ymi_layernorm_fwd(const ymi_tensor* x, int64_t ws, const float* gamma, const float* beta, float eps, const ymi_tensor* out,
                                 float* mean, float* rstd, void* stream) {
    YMI_CHECK_ARG(ymi_tensor_ok(x) && ymi_tensor_ok(out) && gamma && beta && mean && rstd && x->dtype == out->dtype, "layernorm_fwd: args");
    YMI_CHECK_ARG(x->c == out->c && x->c % 4 == 0 && x->c <= 1024 && x->ld % 4 == 0 && out->ld % 4 == 0, "layernorm_fwd: channels multiple of 4, <= 1024");
    int64_t Hp = x->h, Wp = x->w;
    if (ws > 0) window_geometry(x, ws, &Hp, &Wp);
    const int64_t T = ws > 0 ? x->n * Hp * Wp : ymi_pixels(x);
    const int64_t T_ = T;  // (name used by the launch macros)
    YMI_CHECK_ARG(ymi_pixels(out) == T, "layernorm_fwd: output must hold %lld tokens", (long long)T);
    SV xv{x->data, x->ld}, ov{out->data, out->ld};
    dim3 grid((unsigned)((T + 3) / 4));
    const bool half = x->dtype == YMI_BF16 && x->c <= 256 && x->c % 8 == 0 && x->ld % 8 == 0 && out->ld % 8 == 0 &&
                      ((((uintptr_t)x->data) | ((uintptr_t)out->data) | ((uintptr_t)gamma) | ((uintptr_t)beta)) & 15) == 0;
    if (half)
        hipLaunchKernelGGL(layernorm_fwd_half_kernel, dim3((unsigned)((T + 7) / 8)), dim3(256), 0, (hipStream_t)stream, xv, ov, T_, (int)x->h, (int)x->w, (int)Hp,
                           (int)Wp, (int)ws, (int)x->c, gamma, beta, eps, mean, rstd);
    else if (x->dtype == YMI_BF16)
        YMI_LN_G(YMI_LNF, bf16_t);
    else
        YMI_LN_G(YMI_LNF, float);
    YMI_CHECK_LAUNCH("layernorm_fwd");
    return YMI_OK;
}

// backward: dx = rstd*(g*dy - mean(g*dy) - xhat*mean(g*dy*xhat)); per-block partials of dgamma/dbeta.
// ACCUM: dx += (used when the LayerNorm input also feeds a residual branch).
template <typename T, int G>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(SV x, SV dy, SV dx, SV add, int64_t Tn, int H, int W, int Hp, int Wp, int ws, int C,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* __restrict__ part) {
    const bool accumulate = add.p != nullptr;  // dx = LN gradient + add (the other consumers' gradient of the LN input; may be dx itself)
    extern __shared__ float red_dyn[];  // [4 waves][2][Cr], Cr = C rounded up to 256: sized by the launcher, so 8+ workgroups fit a CU
    const int Cr = (C + 255) / 256 * 256;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float ag[G][4], ab[G][4], g[G][4];
#pragma unroll
    for (int i = 0; i < G; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ag[i][r] = 0.f;
            ab[i][r] = 0.f;
            const int c = lane * 4 + 256 * i + r;
            g[i][r] = c < C ? gamma[c] : 0.f;
        }
    for (int64_t t = (int64_t)blockIdx.x * 4 + wv; t < Tn; t += (int64_t)gridDim.x * 4) {
        bool real = true;
        int64_t row = t;
        if (ws > 0) row = token_pixel(t, H, W, Hp, Wp, ws, &real);
        const T* xp = reinterpret_cast<const T*>(x.p) + row * x.ld;
        const T* dp = reinterpret_cast<const T*>(dy.p) + t * dy.ld;
        const float mu = mean[t], rs = rstd[t];
        float xh[G][4], d[G][4], prev[G][4];
        float s1 = 0.f, s2 = 0.f;
        T* op = reinterpret_cast<T*>(const_cast<void*>(dx.p)) + row * dx.ld;
        const T* ap = reinterpret_cast<const T*>(add.p) + row * add.ld;
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int c = lane * 4 + 256 * i;
#pragma unroll
            for (int r = 0; r < 4; ++r) { xh[i][r] = 0.f; d[i][r] = 0.f; prev[i][r] = 0.f; }
            if (c < C) {
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (real) Pack<T, 4>::load(xp + c, v);
                Pack<T, 4>::load(dp + c, d[i]);
                // the value this pass accumulates onto: fetched WITH the operands, not after the reductions (it was a
                // second dependent memory round trip per token)
                if (accumulate && real) Pack<T, 4>::load(ap + c, prev[i]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    xh[i][r] = (v[r] - mu) * rs;
                    ag[i][r] += d[i][r] * xh[i][r];
                    ab[i][r] += d[i][r];
                    const float gd = g[i][r] * d[i][r];
                    s1 += gd;
                    s2 += gd * xh[i][r];
                }
            }
        }
        s1 = wave_sum(s1) / (float)C;
        s2 = wave_sum(s2) / (float)C;
        if (real) {
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int c = lane * 4 + 256 * i;
                if (c < C) {
                    float o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = rs * (g[i][r] * d[i][r] - s1 - xh[i][r] * s2) + prev[i][r];
                    Pack<T, 4>::store(op + c, o);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < G; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = lane * 4 + 256 * i + r;
            if (c < Cr) {
                red_dyn[(wv * 2 + 0) * Cr + c] = ab[i][r];
                red_dyn[(wv * 2 + 1) * Cr + c] = ag[i][r];
            }
        }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        part[((int64_t)blockIdx.x * 2 + 0) * C + c] = red_dyn[0 * Cr + c] + red_dyn[2 * Cr + c] + red_dyn[4 * Cr + c] + red_dyn[6 * Cr + c];
        part[((int64_t)blockIdx.x * 2 + 1) * C + c] = red_dyn[1 * Cr + c] + red_dyn[3 * Cr + c] + red_dyn[5 * Cr + c] + red_dyn[7 * Cr + c];
    }
}

#define YMI_LNB(T, G) hipLaunchKernelGGL((layernorm_bwd_kernel<T, G>), dim3(blocks), dim3(256), red_bytes, s, xv, dv, ov, av, T_, (int)x->h, (int)x->w, (int)Hp, (int)Wp, (int)ws, (int)x->c, gamma, mean, rstd, (float*)workspace)

static int ln_bwd_blocks(int64_t T) {
    int64_t b = (T + 31) / 32;  // >= 8 tokens per wave
    if (b > 2048) b = 2048;     // the token loop is a serial load -> reduce -> store chain: many resident waves hide it
    if (b < 1) b = 1;
    return (int)b;
}

extern "C" int ymi_layernorm_bwd(const ymi_tensor* x, int64_t ws, const ymi_tensor* dout, const float* gamma, const float* mean,
                                 const float* rstd, const ymi_tensor* dx, int32_t accumulate, float* dgamma, float* dbeta, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    return ymi_layernorm_bwd_add(x, ws, dout, gamma, mean, rstd, accumulate ? dx : nullptr, dx, dgamma, dbeta, workspace, workspace_bytes, stream);
}

extern "C" int ymi_layernorm_bwd_add(const ymi_tensor* x, int64_t ws, const ymi_tensor* dout, const float* gamma, const float* mean,
                                     const float* rstd, const ymi_tensor* add, const ymi_tensor* dx, float* dgamma, float* dbeta, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    YMI_CHECK_ARG(!add || (ymi_tensor_ok(add) && ymi_same_shape(add, dx) && add->dtype == dx->dtype && add->ld % 4 == 0), "layernorm_bwd_add: addend");
    YMI_CHECK_ARG(ymi_tensor_ok(x) && ymi_tensor_ok(dout) && ymi_tensor_ok(dx) && gamma && mean && rstd && dgamma && dbeta && workspace, "layernorm_bwd: args");
    YMI_CHECK_ARG(x->dtype == dout->dtype && x->dtype == dx->dtype && ymi_same_shape(x, dx), "layernorm_bwd: dtypes/shapes");
    YMI_CHECK_ARG(x->c == dout->c && x->c % 4 == 0 && x->c <= 1024 && x->ld % 4 == 0 && dout->ld % 4 == 0 && dx->ld % 4 == 0, "layernorm_bwd: channels");
    int64_t Hp = x->h, Wp = x->w;
    if (ws > 0) window_geometry(x, ws, &Hp, &Wp);
    const int64_t T = ws > 0 ? x->n * Hp * Wp : ymi_pixels(x);
    const int64_t T_ = T;  // (name used by the launch macros)
    YMI_CHECK_ARG(ymi_pixels(dout) == T, "layernorm_bwd: dout must hold %lld tokens", (long long)T);
    const int blocks = ln_bwd_blocks(T);
    const size_t need = (size_t)blocks * 2 * x->c * sizeof(float);
    if (workspace_bytes < need) {
        ymi_set_error("layernorm_bwd: workspace %zu < %zu", workspace_bytes, need);
        return YMI_EWORKSPACE;
    }
    SV xv{x->data, x->ld}, dv{dout->data, dout->ld}, ov{dx->data, dx->ld}, av{add ? add->data : nullptr, add ? add->ld : 0};
    hipStream_t s = (hipStream_t)stream;
    const size_t red_bytes = (size_t)8 * ((x->c + 255) / 256 * 256) * sizeof(float);
    if (x->dtype == YMI_BF16)
        YMI_LN_G(YMI_LNB, bf16_t);
    else
        YMI_LN_G(YMI_LNB, float);
    YMI_CHECK_LAUNCH("layernorm_bwd");
    return ymi_chan_reduce_final((const float*)workspace, blocks, (int)x->c, dbeta, dgamma, s);
}

// --------------------------------------------------------------------------- window attention
template <typename T> struct AT;
template <> struct AT<bf16_t> {
    static constexpr int PAD = 8;   // elements (16 B)
    static constexpr int KS = 32;   // k per MFMA step
    // acc += A[arow + i][k0..] . B[brow + j][k0..]   (i = 4*(lane>>4)+r, j = lane&15)
    static __device__ __forceinline__ void mma(const char* A, int astride, int arow, const char* B, int bstride, int brow, int klen, int lane, f32x4& acc) {
        const int l15 = lane & 15, l4 = lane >> 4;
        for (int k0 = 0; k0 < klen; k0 += 32) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(A + (size_t)(arow + l15) * astride + (k0 + 8 * l4) * 2);
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(B + (size_t)(brow + l15) * bstride + (k0 + 8 * l4) * 2);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
        }
    }
};
template <> struct AT<float> {
    static constexpr int PAD = 4;
    static constexpr int KS = 4;
    static __device__ __forceinline__ void mma(const char* A, int astride, int arow, const char* B, int bstride, int brow, int klen, int lane, f32x4& acc) {
        const int l15 = lane & 15, l4 = lane >> 4;
        for (int k0 = 0; k0 < klen; k0 += 4) {
            const float a = *reinterpret_cast<const float*>(A + (size_t)(arow + l15) * astride + (k0 + l4) * 4);
            const float b = *reinterpret_cast<const float*>(B + (size_t)(brow + l15) * bstride + (k0 + l4) * 4);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        }
    }
};

// G[r][c] = src[(r)*ld + c] for r < L, c < hd; zero elsewhere.  G rows: 64, cols: hdp, stride (hdp+PAD)
template <typename T>
__device__ __forceinline__ void lds_load_rows(char* G, const T* src, int64_t ld, int L, int hd, int hdp) {
    const int stride = (hdp + AT<T>::PAD) * (int)sizeof(T);
    const int g4 = hdp / 4;
    for (int i = threadIdx.x; i < 64 * g4; i += 256) {
        const int r = i / g4, c = (i % g4) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < L && c < hd) Pack<T, 4>::load(src + (int64_t)r * ld + c, v);
        Pack<T, 4>::store(reinterpret_cast<T*>(G + (size_t)r * stride) + c, v);
    }
}
// G[c][r] = src[r*ld + c]: rows hdp, cols 64, stride (64+PAD)
template <typename T>
__device__ __forceinline__ void lds_load_transposed(char* G, const T* src, int64_t ld, int L, int hd, int hdp) {
    const int stride = (64 + AT<T>::PAD) * (int)sizeof(T);
    const int g4 = hdp / 4;
    for (int i = threadIdx.x; i < 64 * g4; i += 256) {
        const int r = i / g4, c = (i % g4) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < L && c < hd) Pack<T, 4>::load(src + (int64_t)r * ld + c, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) reinterpret_cast<T*>(G + (size_t)(c + e) * stride)[r] = from_f32<T>(v[e]);
    }
}

struct AttnArgs {
    SV qkv, out, dout, dqkv;
    float* lse;
    int L, heads, C, hd, hdp;
    float scale;
};

constexpr int ATT_MAXDT = 12;  // head_dim <= 192

template <typename T>
__global__ __launch_bounds__(256) void window_attn_fwd_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ES = (int)sizeof(T);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int hdp = a.hdp, L = a.L;
    const int rstride = (hdp + AT<T>::PAD) * ES, tstride = (64 + AT<T>::PAD) * ES;
    const size_t gsz = (size_t)((64 * rstride > hdp * tstride) ? 64 * rstride : hdp * tstride);
    char* G1 = smem;
    char* G2 = G1 + gsz;
    char* P = G2 + gsz;  // [64][64+PAD]
    const int64_t t0 = (int64_t)blockIdx.x * L;
    const T* qkv = reinterpret_cast<const T*>(a.qkv.p) + t0 * a.qkv.ld;
    T* out = reinterpret_cast<T*>(const_cast<void*>(a.out.p)) + t0 * a.out.ld;
    const int ndt = (a.hd + 15) / 16;

    {  // one (window, head) pair per workgroup: blockIdx.y = head
        const int head = blockIdx.y;
        const int co = head * a.hd;
        lds_load_rows<T>(G1, qkv + co, a.qkv.ld, L, a.hd, hdp);            // Q
        lds_load_rows<T>(G2, qkv + a.C + co, a.qkv.ld, L, a.hd, hdp);      // K
        __syncthreads();
        // S^T tiles: A = K rows (i = key), B = Q rows (j = query): this wave owns queries 16*wv..+15
        f32x4 s[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            AT<T>::mma(G2, rstride, t * 16, G1, rstride, wv * 16, hdp, lane, s[t]);
        }
        // lane holds query m = 16*wv + l15, keys j = 16t + 4*l4 + r
        float mx = -__builtin_inff();
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * t + 4 * l4 + r;
                s[t][r] = j < L ? s[t][r] * a.scale : -__builtin_inff();
                mx = fmaxf(mx, s[t][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[t][r] = __expf(s[t][r] - mx);
                sum += s[t][r];
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;
        const int m = 16 * wv + l15;
        if (l4 == 0 && m < L && a.lse) a.lse[(t0 + m) * a.heads + head] = mx + __logf(sum);
        __syncthreads();  // everyone is done reading G1 (Q) / G2 (K)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float p[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) p[r] = s[t][r] * inv;
            Pack<T, 4>::store(reinterpret_cast<T*>(P + (size_t)m * tstride) + 16 * t + 4 * l4, p);
        }
        lds_load_transposed<T>(G1, qkv + 2 * a.C + co, a.qkv.ld, L, a.hd, hdp);  // V^T
        __syncthreads();
        // O tiles: A = P rows (i = query), B = V^T rows (j = d)
#pragma unroll
        for (int dt = 0; dt < ATT_MAXDT; ++dt) {
            if (dt < ndt) {
                f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
                AT<T>::mma(P, tstride, wv * 16, G1, tstride, dt * 16, 64, lane, o);
                const int d = dt * 16 + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int mq = wv * 16 + 4 * l4 + r;
                    if (mq < L && d < a.hd) out[(int64_t)mq * a.out.ld + co + d] = from_f32<T>(o[r]);
                }
            }
        }
        __syncthreads();
    }
}

template <typename T>
__global__ __launch_bounds__(256) void window_attn_bwd_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ES = (int)sizeof(T);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int hdp = a.hdp, L = a.L;
    const int rstride = (hdp + AT<T>::PAD) * ES, tstride = (64 + AT<T>::PAD) * ES;
    const size_t gsz = (size_t)((64 * rstride > hdp * tstride) ? 64 * rstride : hdp * tstride);
    char* G1 = smem;
    char* G2 = G1 + gsz;
    char* PT = G2 + gsz;                     // P^T   [key][query]
    char* DS = PT + (size_t)64 * tstride;    // dS    [query][key]   (scale folded in)
    char* DST = DS + (size_t)64 * tstride;   // dS^T  [key][query]
    float* delta = reinterpret_cast<float*>(DST + (size_t)64 * tstride);  // [64]
    const int64_t t0 = (int64_t)blockIdx.x * L;
    const T* qkv = reinterpret_cast<const T*>(a.qkv.p) + t0 * a.qkv.ld;
    const T* o = reinterpret_cast<const T*>(a.out.p) + t0 * a.out.ld;
    const T* dO = reinterpret_cast<const T*>(a.dout.p) + t0 * a.dout.ld;
    T* dqkv = reinterpret_cast<T*>(const_cast<void*>(a.dqkv.p)) + t0 * a.dqkv.ld;
    const int ndt = (a.hd + 15) / 16;

    {  // one (window, head) pair per workgroup: blockIdx.y = head
        const int head = blockIdx.y;
        const int co = head * a.hd;
        lds_load_rows<T>(G1, qkv + co, a.qkv.ld, L, a.hd, hdp);        // Q
        lds_load_rows<T>(G2, qkv + a.C + co, a.qkv.ld, L, a.hd, hdp);  // K
        // delta[m] = sum_d dO[m][d] * O[m][d] : one wave per row, 16 rows per wave
        for (int m = wv; m < 64; m += 4) {
            float acc = 0.f;
            if (m < L)
                for (int d = lane; d < a.hd; d += 64) acc += to_f32(dO[(int64_t)m * a.dout.ld + co + d]) * to_f32(o[(int64_t)m * a.out.ld + co + d]);
            acc = wave_sum(acc);
            if (lane == 0) delta[m] = acc;
        }
        __syncthreads();
        // S tiles: A = Q rows (i = query m), B = K rows (j = key): this wave owns queries 16*wv..+15
        f32x4 s[4], dp[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            AT<T>::mma(G1, rstride, wv * 16, G2, rstride, t * 16, hdp, lane, s[t]);
        }
        __syncthreads();
        lds_load_rows<T>(G1, dO + co, a.dout.ld, L, a.hd, hdp);             // dO
        lds_load_rows<T>(G2, qkv + 2 * a.C + co, a.qkv.ld, L, a.hd, hdp);   // V
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            dp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            AT<T>::mma(G1, rstride, wv * 16, G2, rstride, t * 16, hdp, lane, dp[t]);
        }
        // lane holds queries m = 16*wv + 4*l4 + r, key j = 16t + l15
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int j = 16 * t + l15;
            float p[4], ds[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = 16 * wv + 4 * l4 + r;
                const bool ok = (j < L) && (m < L);
                const float lse = ok ? a.lse[(t0 + m) * a.heads + head] : 0.f;
                p[r] = ok ? __expf(s[t][r] * a.scale - lse) : 0.f;
                ds[r] = ok ? p[r] * (dp[t][r] - delta[m]) * a.scale : 0.f;
                reinterpret_cast<T*>(DS + (size_t)m * tstride)[j] = from_f32<T>(ds[r]);
            }
            Pack<T, 4>::store(reinterpret_cast<T*>(PT + (size_t)j * tstride) + 16 * wv + 4 * l4, p);
            Pack<T, 4>::store(reinterpret_cast<T*>(DST + (size_t)j * tstride) + 16 * wv + 4 * l4, ds);
        }
        __syncthreads();
        // dV[j][d] = sum_m P^T[j][m] dO^T[d][m]
        lds_load_transposed<T>(G1, dO + co, a.dout.ld, L, a.hd, hdp);
        __syncthreads();
#pragma unroll
        for (int dt = 0; dt < ATT_MAXDT; ++dt)
            if (dt < ndt) {
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                AT<T>::mma(PT, tstride, wv * 16, G1, tstride, dt * 16, 64, lane, acc);
                const int d = dt * 16 + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = wv * 16 + 4 * l4 + r;
                    if (j < L && d < a.hd) dqkv[(int64_t)j * a.dqkv.ld + 2 * a.C + co + d] = from_f32<T>(acc[r]);
                }
            }
        __syncthreads();
        // dK[j][d] = sum_m dS^T[j][m] Q^T[d][m]
        lds_load_transposed<T>(G1, qkv + co, a.qkv.ld, L, a.hd, hdp);
        __syncthreads();
#pragma unroll
        for (int dt = 0; dt < ATT_MAXDT; ++dt)
            if (dt < ndt) {
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                AT<T>::mma(DST, tstride, wv * 16, G1, tstride, dt * 16, 64, lane, acc);
                const int d = dt * 16 + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = wv * 16 + 4 * l4 + r;
                    if (j < L && d < a.hd) dqkv[(int64_t)j * a.dqkv.ld + a.C + co + d] = from_f32<T>(acc[r]);
                }
            }
        __syncthreads();
        // dQ[m][d] = sum_j dS[m][j] K^T[d][j]
        lds_load_transposed<T>(G1, qkv + a.C + co, a.qkv.ld, L, a.hd, hdp);
        __syncthreads();
#pragma unroll
        for (int dt = 0; dt < ATT_MAXDT; ++dt)
            if (dt < ndt) {
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                AT<T>::mma(DS, tstride, wv * 16, G1, tstride, dt * 16, 64, lane, acc);
                const int d = dt * 16 + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = wv * 16 + 4 * l4 + r;
                    if (m < L && d < a.hd) dqkv[(int64_t)m * a.dqkv.ld + co + d] = from_f32<T>(acc[r]);
                }
            }
        __syncthreads();
    }
}

// ---- bf16 backward with hardware-transposed fragment reads -------------------------------------------------------
// Q, K, V, dO of one (window, head) are staged ONCE, row-major; P and dS are written row-major; every product whose
// reduction index is the slow (row) index of an LDS image takes its fragments with ds_read_b64_tr_b16, so no operand
// is loaded from HBM a second time in transposed form and the kernel has two workgroup barriers instead of twelve.
typedef __attribute__((ext_vector_type(4))) short at_s16x4;
typedef __attribute__((ext_vector_type(8))) short at_s16x8;
// 8 k-values (rows k0 + 8*(lane>>4) .. +7) of column c0 + (lane&15) from a K-major image img[k][col]
__device__ __forceinline__ bf16x8 attn_tr_frag(const char* img, int rowb, int k0, int c0, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int r_lo = k0 + 8 * g + q, r_hi = r_lo + 4;
    const int u = (c0 >> 2) + p;  // 8-byte unit holding this lane's 4 columns
    const at_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) at_s16x4*)(img + (size_t)r_lo * rowb + u * 8));
    const at_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) at_s16x4*)(img + (size_t)r_hi * rowb + u * 8));
    const at_s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// stage NM [L][hd] bf16 matrices as zero-padded [64][hdp+8] LDS images: all global loads of a round are issued before
// the first LDS store (these kernels are latency-bound: a load->store round per matrix costs more than the arithmetic)
template <int NM>
__device__ __forceinline__ void attn_stage_rows(const bf16_t* const (&srcs)[NM], const int64_t (&lds_)[NM], char* const (&dsts)[NM], int L, int hd, int hdp) {
    const int rstride = (hdp + 8) * 2;
    bool wide = (hd & 7) == 0 && (hdp & 7) == 0;  // 16-byte chunks (8 channels): half the load instructions of the 8-byte form
#pragma unroll
    for (int mtx = 0; mtx < NM; ++mtx) wide = wide && (lds_[mtx] & 7) == 0 && (((uintptr_t)srcs[mtx]) & 15) == 0;
    if (wide) {  // (workgroup-uniform)
        const int g8 = hdp / 8, items = 64 * g8;
        for (int i = threadIdx.x; i < items; i += 256) {
            const int r = i / g8, c = (i - r * g8) * 8;
            const bool ok = r < L && c < hd;
            bf16x8 v[NM];
#pragma unroll
            for (int mtx = 0; mtx < NM; ++mtx) {
                if (ok) v[mtx] = *reinterpret_cast<const bf16x8*>(srcs[mtx] + (int64_t)r * lds_[mtx] + c);
                else
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[mtx][e] = (bf16_t)0.f;
            }
#pragma unroll
            for (int mtx = 0; mtx < NM; ++mtx) *reinterpret_cast<bf16x8*>(dsts[mtx] + (size_t)r * rstride + c * 2) = v[mtx];
        }
        return;
    }
    const int g4 = hdp / 4, items = 64 * g4;
    for (int base = threadIdx.x; base < items; base += 512) {
        bf16x4 v[2][NM];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = base + 256 * u;
            const int r = i / g4, c = (i - r * g4) * 4;
            const bool ok = i < items && r < L && c < hd;
#pragma unroll
            for (int mtx = 0; mtx < NM; ++mtx)
                v[u][mtx] = ok ? *reinterpret_cast<const bf16x4*>(srcs[mtx] + (int64_t)r * lds_[mtx] + c) : bf16x4{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = base + 256 * u;
            const int r = i / g4, c = (i - r * g4) * 4;
            if (i < items) {
#pragma unroll
                for (int mtx = 0; mtx < NM; ++mtx) *reinterpret_cast<bf16x4*>(dsts[mtx] + (size_t)r * rstride + c * 2) = v[u][mtx];
            }
        }
    }
}

// bf16 forward: Q, K, V staged once row-major; O = P V takes V through transposed fragment reads
__global__ __launch_bounds__(256) void window_attn_fwd_tr_kernel(AttnArgs a) {
    typedef bf16_t T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int hdp = a.hdp, L = a.L;
    const int rstride = (hdp + 8) * 2, tstride = (64 + 8) * 2;
    char* Qs = smem;
    char* Ks = Qs + (size_t)64 * rstride;
    char* Vs = Ks + (size_t)64 * rstride;
    char* P = Vs + (size_t)64 * rstride;  // [query][key]
    const int head = blockIdx.y;
    const int co = head * a.hd;
    const int64_t t0 = (int64_t)blockIdx.x * L;
    const T* qkv = reinterpret_cast<const T*>(a.qkv.p) + t0 * a.qkv.ld;
    T* out = reinterpret_cast<T*>(const_cast<void*>(a.out.p)) + t0 * a.out.ld;
    const int ndt = (a.hd + 15) / 16;
    const bool vec4 = (a.hd & 3) == 0 && (a.out.ld & 3) == 0 && (((uintptr_t)a.out.p) & 7) == 0;  // 8-byte stores of 4 channels
    {
        const bf16_t* srcs[3] = {qkv + co, qkv + a.C + co, qkv + 2 * a.C + co};
        const int64_t lds_[3] = {a.qkv.ld, a.qkv.ld, a.qkv.ld};
        char* dsts[3] = {Qs, Ks, Vs};
        attn_stage_rows<3>(srcs, lds_, dsts, L, a.hd, hdp);
    }
    __syncthreads();
    // S^T tiles: A = K rows (i = key), B = Q rows (j = query): this wave owns queries 16*wv..+15
    f32x4 s[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        AT<T>::mma(Ks, rstride, t * 16, Qs, rstride, wv * 16, hdp, lane, s[t]);
    }
    // lane holds query m = 16*wv + l15, keys j = 16t + 4*l4 + r
    float mx = -__builtin_inff();
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * t + 4 * l4 + r;
            s[t][r] = j < L ? s[t][r] * a.scale : -__builtin_inff();
            mx = fmaxf(mx, s[t][r]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s[t][r] = __expf(s[t][r] - mx);
            sum += s[t][r];
        }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
    const int m = 16 * wv + l15;
    if (l4 == 0 && m < L && a.lse) a.lse[(t0 + m) * a.heads + head] = mx + __logf(sum);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float p[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] = s[t][r] * inv;
        Pack<T, 4>::store(reinterpret_cast<T*>(P + (size_t)m * tstride) + 16 * t + 4 * l4, p);
    }
    __syncthreads();
    // O[m][d] = sum_j P[m][j] V[j][d]: A = rows of P (k = j contiguous), B = transposed read of V
#pragma unroll
    for (int dt = 0; dt < ATT_MAXDT; ++dt)
        if (dt < ndt) {
            f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            // operands swapped (first = V^T fragment: rows d, second = P rows: columns m), so a lane's four values are four CONSECUTIVE
            // channels of one query: one 8-byte store instead of four 2-byte stores to four rows
            for (int k0 = 0; k0 < 64; k0 += 32) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(P + (size_t)(16 * wv + l15) * tstride + (k0 + 8 * l4) * 2);
                o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(attn_tr_frag(Vs, rstride, k0, 16 * dt, lane), af, o, 0, 0, 0);
            }
            const int d = dt * 16 + 4 * l4, mq = wv * 16 + l15;
            if (mq < L) {
                T* dst = out + (int64_t)mq * a.out.ld + co + d;
                if (vec4 && d + 3 < a.hd) {
                    *reinterpret_cast<bf16x4*>(dst) = bf16x4{(T)o[0], (T)o[1], (T)o[2], (T)o[3]};
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (d + r < a.hd) dst[r] = (T)o[r];
                }
            }
        }
}

__global__ __launch_bounds__(256) void window_attn_bwd_tr_kernel(AttnArgs a) {
    typedef bf16_t T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int hdp = a.hdp, L = a.L;
    const int rstride = (hdp + 8) * 2, tstride = (64 + 8) * 2;
    char* Qs = smem;
    char* Ks = Qs + (size_t)64 * rstride;
    char* Vs = Ks + (size_t)64 * rstride;
    char* Os = Vs + (size_t)64 * rstride;   // dO
    char* Ps = Os + (size_t)64 * rstride;   // P  [query][key]
    char* Ds = Ps + (size_t)64 * tstride;   // dS [query][key] (scale folded in)
    const int head = blockIdx.y;
    const int co = head * a.hd;
    const int64_t t0 = (int64_t)blockIdx.x * L;
    const T* qkv = reinterpret_cast<const T*>(a.qkv.p) + t0 * a.qkv.ld;
    const T* dO = reinterpret_cast<const T*>(a.dout.p) + t0 * a.dout.ld;
    T* dqkv = reinterpret_cast<T*>(const_cast<void*>(a.dqkv.p)) + t0 * a.dqkv.ld;
    const int ndt = (a.hd + 15) / 16;
    const bool vec4 = (a.hd & 3) == 0 && (a.C & 3) == 0 && (a.dqkv.ld & 3) == 0 && (((uintptr_t)a.dqkv.p) & 7) == 0;  // 8-byte stores of 4 channels

    // the saved log-sum-exp of this lane's query row: requested before the operand staging (it was a dependent global round trip
    // between the first MFMAs and the softmax)
    const int mrow = 16 * wv + l15;  // this lane's query in the transposed tiles below
    const float lse = mrow < L ? a.lse[(t0 + mrow) * a.heads + head] : 0.f;
    {
        const bf16_t* srcs[4] = {qkv + co, qkv + a.C + co, qkv + 2 * a.C + co, dO + co};
        const int64_t lds_[4] = {a.qkv.ld, a.qkv.ld, a.qkv.ld, a.dout.ld};
        char* dsts[4] = {Qs, Ks, Vs, Os};
        attn_stage_rows<4>(srcs, lds_, dsts, L, a.hd, hdp);
    }
    __syncthreads();
    // TRANSPOSED tiles, as in the forward: S^T = K Q^T and dP^T = V dO^T, so a lane holds keys j = 16 t + 4 l4 + r (r = 0..3) of ONE
    // query m = 16 wv + l15: one log-sum-exp per lane, the row sum delta needs two shuffles instead of sixteen, and P / dS go to LDS
    // as 8-byte stores of four consecutive keys instead of 2-byte stores
    f32x4 s[4], dp[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        dp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        AT<T>::mma(Ks, rstride, t * 16, Qs, rstride, wv * 16, hdp, lane, s[t]);
        AT<T>::mma(Vs, rstride, t * 16, Os, rstride, wv * 16, hdp, lane, dp[t]);
    }
    float pr[4][4], delta = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = (16 * t + 4 * l4 + r < L) && (mrow < L);
            pr[t][r] = ok ? __expf(s[t][r] * a.scale - lse) : 0.f;
            delta += pr[t][r] * dp[t][r];  // sum_j P dP = dO . O
        }
    delta += __shfl_xor(delta, 16, 64);
    delta += __shfl_xor(delta, 32, 64);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) ds[r] = pr[t][r] * (dp[t][r] - delta) * a.scale;
        Pack<T, 4>::store(reinterpret_cast<T*>(Ps + (size_t)mrow * tstride) + 16 * t + 4 * l4, pr[t]);
        Pack<T, 4>::store(reinterpret_cast<T*>(Ds + (size_t)mrow * tstride) + 16 * t + 4 * l4, ds);
    }
    __syncthreads();
#pragma unroll
    for (int dt = 0; dt < ATT_MAXDT; ++dt)
        if (dt < ndt) {
            f32x4 dv = f32x4{0.f, 0.f, 0.f, 0.f}, dk = dv, dq = dv;
#pragma unroll
            for (int k0 = 0; k0 < 64; k0 += 32) {
                // dV[j][d] = sum_m P[m][j] dO[m][d] ; dK[j][d] = sum_m dS[m][j] Q[m][d]   (k = m: both operands transposed reads)
                // (operands in this order - the d-indexed fragment first - so that a lane's four values are four CONSECUTIVE channels of one
                // row: one 8-byte store per matrix and tile instead of four 2-byte stores to four rows)
                dv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(attn_tr_frag(Os, rstride, k0, 16 * dt, lane), attn_tr_frag(Ps, tstride, k0, 16 * wv, lane), dv, 0, 0, 0);
                dk = __builtin_amdgcn_mfma_f32_16x16x32_bf16(attn_tr_frag(Qs, rstride, k0, 16 * dt, lane), attn_tr_frag(Ds, tstride, k0, 16 * wv, lane), dk, 0, 0, 0);
                // dQ[m][d] = sum_j dS[m][j] K[j][d]   (rows of dS, k = j contiguous; transposed read of K)
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(Ds + (size_t)(16 * wv + l15) * tstride + (k0 + 8 * l4) * 2);
                dq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(attn_tr_frag(Ks, rstride, k0, 16 * dt, lane), af, dq, 0, 0, 0);
            }
            const int d = dt * 16 + 4 * l4, row = wv * 16 + l15;  // row: key j for dV / dK, query m for dQ
            if (row < L) {
                T* dst = dqkv + (int64_t)row * a.dqkv.ld + co + d;
                if (vec4 && d + 3 < a.hd) {
                    *reinterpret_cast<bf16x4*>(dst + 2 * a.C) = bf16x4{(T)dv[0], (T)dv[1], (T)dv[2], (T)dv[3]};
                    *reinterpret_cast<bf16x4*>(dst + a.C) = bf16x4{(T)dk[0], (T)dk[1], (T)dk[2], (T)dk[3]};
                    *reinterpret_cast<bf16x4*>(dst) = bf16x4{(T)dq[0], (T)dq[1], (T)dq[2], (T)dq[3]};
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (d + r < a.hd) {
                            dst[2 * a.C + r] = (T)dv[r];
                            dst[a.C + r] = (T)dk[r];
                            dst[r] = (T)dq[r];
                        }
                }
            }
        }
}

// ---- windows of more than 64 tokens (any window_size: swin_block.py:24) and head dims the one-tile kernels cannot hold ----
// Flash-style tiling: 64-query x 64-key tiles, online softmax in the forward, P recomputed from the saved log-sum-exp in the
// backward (one kernel accumulates dQ over key tiles, one accumulates dK / dV over query tiles: no atomics, deterministic).
// The head dimension is walked in chunks of ATT_HC columns through two LDS staging buffers, so float32 at head_dim 192
// fits the 160 KiB of a CU.  This path serves config 5's optional ws = 14 row (196 tokens); the 49-token windows of the
// benchmark stay on the one-tile kernels above.
constexpr int ATT_HC = 96;

template <typename T> struct TiledLds {
    static constexpr int ES = (int)sizeof(T);
    static constexpr int PAD = AT<T>::PAD;
    static constexpr int RS = (ATT_HC + PAD) * ES;  // row stride of a [64][chunk] image
    static constexpr int TS = (64 + PAD) * ES;      // row stride of a [chunk][64] / [64][64] image
    static constexpr size_t STAGE = (size_t)(64 * RS > ATT_HC * TS ? 64 * RS : ATT_HC * TS);
};

// delta[m] = sum_d dO[m][d] * O[m][d] for the 64 rows starting at row r0 of the window (zero beyond the window)
template <typename T>
__device__ __forceinline__ void attn_delta_rows(float* delta, const T* dO, int64_t ldd, const T* o, int64_t ldo, int rows, int hd) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int m = wv; m < 64; m += 4) {
        float acc = 0.f;
        if (m < rows)
            for (int d = lane; d < hd; d += 64) acc += to_f32(dO[(int64_t)m * ldd + d]) * to_f32(o[(int64_t)m * ldo + d]);
        acc = wave_sum(acc);
        if (lane == 0) delta[m] = acc;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void window_attn_tiled_fwd_kernel(AttnArgs a) {
    typedef TiledLds<T> LD;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* GA = smem;
    char* GB = GA + LD::STAGE;
    char* P = GB + LD::STAGE;                                          // [64 queries][64 keys]
    float* al = reinterpret_cast<float*>(P + (size_t)64 * LD::TS);     // [64] rescale factor of this key tile / final 1/l
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int L = a.L, hd = a.hd;
    const int head = blockIdx.y, co = head * hd;
    const int q0 = blockIdx.z * 64, qrows = (L - q0 < 64) ? L - q0 : 64;
    const int64_t t0 = (int64_t)blockIdx.x * L;
    const T* qkv = reinterpret_cast<const T*>(a.qkv.p) + t0 * a.qkv.ld;
    T* out = reinterpret_cast<T*>(const_cast<void*>(a.out.p)) + t0 * a.out.ld;
    const int ndt = (hd + 15) / 16;
    f32x4 o[ATT_MAXDT];
#pragma unroll
    for (int dt = 0; dt < ATT_MAXDT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -__builtin_inff(), l_run = 0.f;  // of query q0 + 16*wv + l15 (replicated in the 4 lanes that share it)
    for (int k0 = 0; k0 < L; k0 += 64) {
        const int krows = (L - k0 < 64) ? L - k0 : 64;
        // S^T tile: A = K rows (i = key), B = Q rows (j = query)
        f32x4 s[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int d0 = 0; d0 < hd; d0 += ATT_HC) {
            const int hc = (hd - d0 < ATT_HC) ? hd - d0 : ATT_HC, hcp = (hc + 31) / 32 * 32;
            __syncthreads();
            lds_load_rows<T>(GA, qkv + (int64_t)q0 * a.qkv.ld + co + d0, a.qkv.ld, qrows, hc, hcp);
            lds_load_rows<T>(GB, qkv + (int64_t)k0 * a.qkv.ld + a.C + co + d0, a.qkv.ld, krows, hc, hcp);
            __syncthreads();
            const int rs = (hcp + AT<T>::PAD) * LD::ES;
#pragma unroll
            for (int t = 0; t < 4; ++t) AT<T>::mma(GB, rs, t * 16, GA, rs, wv * 16, hcp, lane, s[t]);
        }
        float mx = -__builtin_inff();
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * t + 4 * l4 + r;
                s[t][r] = j < krows ? s[t][r] * a.scale : -__builtin_inff();
                mx = fmaxf(mx, s[t][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);  // first tile: exp(-inf) = 0
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[t][r] = __expf(s[t][r] - m_new);
                sum += s[t][r];
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        l_run = l_run * alpha + sum;
        m_run = m_new;
        const int m = 16 * wv + l15;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float p[4] = {s[t][0], s[t][1], s[t][2], s[t][3]};
            Pack<T, 4>::store(reinterpret_cast<T*>(P + (size_t)m * LD::TS) + 16 * t + 4 * l4, p);
        }
        if (l4 == 0) al[m] = alpha;
        __syncthreads();
        float ar[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) ar[r] = al[16 * wv + 4 * l4 + r];
#pragma unroll
        for (int dt = 0; dt < ATT_MAXDT; ++dt)
            if (dt < ndt)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[dt][r] *= ar[r];
        // O += P V: A = P rows (i = query), B = V^T rows (j = d)
        for (int d0 = 0; d0 < hd; d0 += ATT_HC) {
            const int hc = (hd - d0 < ATT_HC) ? hd - d0 : ATT_HC, hcp = (hc + 31) / 32 * 32;
            __syncthreads();
            lds_load_transposed<T>(GA, qkv + (int64_t)k0 * a.qkv.ld + 2 * a.C + co + d0, a.qkv.ld, krows, hc, hcp);
            __syncthreads();
#pragma unroll
            for (int dt = 0; dt < ATT_MAXDT; ++dt)
                if (dt < ndt && dt * 16 >= d0 && dt * 16 < d0 + ATT_HC) AT<T>::mma(P, LD::TS, wv * 16, GA, LD::TS, dt * 16 - d0, 64, lane, o[dt]);
        }
        __syncthreads();  // P / al are rewritten by the next key tile
    }
    const int m = 16 * wv + l15;
    if (l4 == 0) {
        al[m] = 1.f / l_run;
        if (m < qrows && a.lse) a.lse[(t0 + q0 + m) * a.heads + head] = m_run + __logf(l_run);
    }
    __syncthreads();
#pragma unroll
    for (int dt = 0; dt < ATT_MAXDT; ++dt)
        if (dt < ndt) {
            const int d = dt * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mq = 16 * wv + 4 * l4 + r;
                if (mq < qrows && d < hd) out[(int64_t)(q0 + mq) * a.out.ld + co + d] = from_f32<T>(o[dt][r] * al[mq]);
            }
        }
}

// S and dP of one (query tile, key tile) pair, then P = exp(S*scale - lse), dS = P (dP - delta) scale.
// Lane layout of the results: queries m = 16*wv + 4*l4 + r, key j = 16*t + l15.
template <typename T>
__device__ __forceinline__ void attn_tile_p_ds(const AttnArgs& a, char* GA, char* GB, const T* qkv, const T* dO, int q0, int qrows, int k0, int krows, int co,
                                               int64_t trow0, int head, const float* delta, float (&p)[4][4], float (&ds)[4][4]) {
    typedef TiledLds<T> LD;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int hd = a.hd;
    f32x4 s[4], dp[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        dp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int d0 = 0; d0 < hd; d0 += ATT_HC) {
        const int hc = (hd - d0 < ATT_HC) ? hd - d0 : ATT_HC, hcp = (hc + 31) / 32 * 32;
        const int rs = (hcp + AT<T>::PAD) * LD::ES;
        __syncthreads();
        lds_load_rows<T>(GA, qkv + (int64_t)q0 * a.qkv.ld + co + d0, a.qkv.ld, qrows, hc, hcp);          // Q
        lds_load_rows<T>(GB, qkv + (int64_t)k0 * a.qkv.ld + a.C + co + d0, a.qkv.ld, krows, hc, hcp);  // K
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t) AT<T>::mma(GA, rs, wv * 16, GB, rs, t * 16, hcp, lane, s[t]);
        __syncthreads();
        lds_load_rows<T>(GA, dO + (int64_t)q0 * a.dout.ld + co + d0, a.dout.ld, qrows, hc, hcp);           // dO
        lds_load_rows<T>(GB, qkv + (int64_t)k0 * a.qkv.ld + 2 * a.C + co + d0, a.qkv.ld, krows, hc, hcp);  // V
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t) AT<T>::mma(GA, rs, wv * 16, GB, rs, t * 16, hcp, lane, dp[t]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = 16 * wv + 4 * l4 + r;
        const float lse = m < qrows ? a.lse[(trow0 + q0 + m) * a.heads + head] : 0.f;
        const float dl = delta[m];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bool ok = (16 * t + l15 < krows) && (m < qrows);
            p[t][r] = ok ? __expf(s[t][r] * a.scale - lse) : 0.f;
            ds[t][r] = ok ? p[t][r] * (dp[t][r] - dl) * a.scale : 0.f;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void window_attn_tiled_dq_kernel(AttnArgs a) {
    typedef TiledLds<T> LD;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* GA = smem;
    char* GB = GA + LD::STAGE;
    char* DS = GB + LD::STAGE;                                          // [64 queries][64 keys]
    float* delta = reinterpret_cast<float*>(DS + (size_t)64 * LD::TS);  // [64]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int L = a.L, hd = a.hd;
    const int head = blockIdx.y, co = head * hd;
    const int q0 = blockIdx.z * 64, qrows = (L - q0 < 64) ? L - q0 : 64;
    const int64_t t0 = (int64_t)blockIdx.x * L;
    const T* qkv = reinterpret_cast<const T*>(a.qkv.p) + t0 * a.qkv.ld;
    const T* ov = reinterpret_cast<const T*>(a.out.p) + t0 * a.out.ld;
    const T* dO = reinterpret_cast<const T*>(a.dout.p) + t0 * a.dout.ld;
    T* dqkv = reinterpret_cast<T*>(const_cast<void*>(a.dqkv.p)) + t0 * a.dqkv.ld;
    const int ndt = (hd + 15) / 16;
    attn_delta_rows<T>(delta, dO + (int64_t)q0 * a.dout.ld + co, a.dout.ld, ov + (int64_t)q0 * a.out.ld + co, a.out.ld, qrows, hd);
    f32x4 dq[ATT_MAXDT];
#pragma unroll
    for (int dt = 0; dt < ATT_MAXDT; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < L; k0 += 64) {
        const int krows = (L - k0 < 64) ? L - k0 : 64;
        float p[4][4], ds[4][4];
        attn_tile_p_ds<T>(a, GA, GB, qkv, dO, q0, qrows, k0, krows, co, t0, head, delta, p, ds);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) reinterpret_cast<T*>(DS + (size_t)(16 * wv + 4 * l4 + r) * LD::TS)[16 * t + l15] = from_f32<T>(ds[t][r]);
        // dQ[m][d] += sum_j dS[m][j] K^T[d][j]
        for (int d0 = 0; d0 < hd; d0 += ATT_HC) {
            const int hc = (hd - d0 < ATT_HC) ? hd - d0 : ATT_HC, hcp = (hc + 31) / 32 * 32;
            __syncthreads();
            lds_load_transposed<T>(GA, qkv + (int64_t)k0 * a.qkv.ld + a.C + co + d0, a.qkv.ld, krows, hc, hcp);
            __syncthreads();
#pragma unroll
            for (int dt = 0; dt < ATT_MAXDT; ++dt)
                if (dt < ndt && dt * 16 >= d0 && dt * 16 < d0 + ATT_HC) AT<T>::mma(DS, LD::TS, wv * 16, GA, LD::TS, dt * 16 - d0, 64, lane, dq[dt]);
        }
    }
#pragma unroll
    for (int dt = 0; dt < ATT_MAXDT; ++dt)
        if (dt < ndt) {
            const int d = dt * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = 16 * wv + 4 * l4 + r;
                if (m < qrows && d < hd) dqkv[(int64_t)(q0 + m) * a.dqkv.ld + co + d] = from_f32<T>(dq[dt][r]);
            }
        }
}

template <typename T>
__global__ __launch_bounds__(256) void window_attn_tiled_dkv_kernel(AttnArgs a) {
    typedef TiledLds<T> LD;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* GA = smem;
    char* GB = GA + LD::STAGE;
    char* PT = GB + LD::STAGE;                                           // P^T  [64 keys][64 queries]
    char* DST = PT + (size_t)64 * LD::TS;                                // dS^T [64 keys][64 queries]
    float* delta = reinterpret_cast<float*>(DST + (size_t)64 * LD::TS);  // [64]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int L = a.L, hd = a.hd;
    const int head = blockIdx.y, co = head * hd;
    const int k0 = blockIdx.z * 64, krows = (L - k0 < 64) ? L - k0 : 64;
    const int64_t t0 = (int64_t)blockIdx.x * L;
    const T* qkv = reinterpret_cast<const T*>(a.qkv.p) + t0 * a.qkv.ld;
    const T* ov = reinterpret_cast<const T*>(a.out.p) + t0 * a.out.ld;
    const T* dO = reinterpret_cast<const T*>(a.dout.p) + t0 * a.dout.ld;
    T* dqkv = reinterpret_cast<T*>(const_cast<void*>(a.dqkv.p)) + t0 * a.dqkv.ld;
    const int ndt = (hd + 15) / 16;
    f32x4 dk[ATT_MAXDT], dv[ATT_MAXDT];
#pragma unroll
    for (int dt = 0; dt < ATT_MAXDT; ++dt) {
        dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int q0 = 0; q0 < L; q0 += 64) {
        const int qrows = (L - q0 < 64) ? L - q0 : 64;
        __syncthreads();  // delta / PT / DST of the previous query tile are no longer read
        attn_delta_rows<T>(delta, dO + (int64_t)q0 * a.dout.ld + co, a.dout.ld, ov + (int64_t)q0 * a.out.ld + co, a.out.ld, qrows, hd);
        float p[4][4], ds[4][4];
        attn_tile_p_ds<T>(a, GA, GB, qkv, dO, q0, qrows, k0, krows, co, t0, head, delta, p, ds);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int j = 16 * t + l15;
            float pv[4] = {p[t][0], p[t][1], p[t][2], p[t][3]}, dv4[4] = {ds[t][0], ds[t][1], ds[t][2], ds[t][3]};
            Pack<T, 4>::store(reinterpret_cast<T*>(PT + (size_t)j * LD::TS) + 16 * wv + 4 * l4, pv);
            Pack<T, 4>::store(reinterpret_cast<T*>(DST + (size_t)j * LD::TS) + 16 * wv + 4 * l4, dv4);
        }
        for (int d0 = 0; d0 < hd; d0 += ATT_HC) {
            const int hc = (hd - d0 < ATT_HC) ? hd - d0 : ATT_HC, hcp = (hc + 31) / 32 * 32;
            __syncthreads();
            lds_load_transposed<T>(GA, dO + (int64_t)q0 * a.dout.ld + co + d0, a.dout.ld, qrows, hc, hcp);  // dO^T [d][m]
            lds_load_transposed<T>(GB, qkv + (int64_t)q0 * a.qkv.ld + co + d0, a.qkv.ld, qrows, hc, hcp);   // Q^T  [d][m]
            __syncthreads();
#pragma unroll
            for (int dt = 0; dt < ATT_MAXDT; ++dt)
                if (dt < ndt && dt * 16 >= d0 && dt * 16 < d0 + ATT_HC) {
                    AT<T>::mma(PT, LD::TS, wv * 16, GA, LD::TS, dt * 16 - d0, 64, lane, dv[dt]);   // dV[j][d] += sum_m P^T[j][m] dO^T[d][m]
                    AT<T>::mma(DST, LD::TS, wv * 16, GB, LD::TS, dt * 16 - d0, 64, lane, dk[dt]);  // dK[j][d] += sum_m dS^T[j][m] Q^T[d][m]
                }
        }
    }
#pragma unroll
    for (int dt = 0; dt < ATT_MAXDT; ++dt)
        if (dt < ndt) {
            const int d = dt * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * wv + 4 * l4 + r;
                if (j < krows && d < hd) {
                    T* dst = dqkv + (int64_t)(k0 + j) * a.dqkv.ld + co + d;
                    dst[a.C] = from_f32<T>(dk[dt][r]);
                    dst[2 * a.C] = from_f32<T>(dv[dt][r]);
                }
            }
        }
}

template <typename T> static size_t tiled_lds(bool bwd_kv) {
    return 2 * TiledLds<T>::STAGE + (size_t)(bwd_kv ? 2 : 1) * 64 * TiledLds<T>::TS + 64 * sizeof(float);
}

static int attn_common(const ymi_tensor* qkv, int64_t wlen, int64_t heads, AttnArgs* a, size_t* lds, bool bwd, const char* what) {
    YMI_CHECK_ARG(ymi_tensor_ok(qkv) && wlen > 0 && heads > 0, "%s: args", what);
    YMI_CHECK_ARG(qkv->c % 3 == 0, "%s: qkv must have 3C channels", what);
    const int64_t C = qkv->c / 3;
    YMI_CHECK_ARG(C % heads == 0, "%s: C %% heads", what);
    const int64_t hd = C / heads;
    YMI_CHECK_ARG(hd % 4 == 0 && hd <= 16 * ATT_MAXDT && qkv->ld % 4 == 0, "%s: head_dim must be a multiple of 4 and <= %d", what, 16 * ATT_MAXDT);
    YMI_CHECK_ARG(ymi_pixels(qkv) % wlen == 0, "%s: token count not a multiple of the window length", what);
    a->L = (int)wlen; a->heads = (int)heads; a->C = (int)C; a->hd = (int)hd; a->hdp = (int)((hd + 31) / 32 * 32);
    a->scale = 1.0f / sqrtf((float)hd);
    const int es = (int)ymi_esize(qkv->dtype), pad = qkv->dtype == YMI_BF16 ? 8 : 4;
    const size_t rs = (size_t)(a->hdp + pad) * es, ts = (size_t)(64 + pad) * es;
    const size_t gsz = 64 * rs > a->hdp * ts ? 64 * rs : a->hdp * ts;
    *lds = 2 * gsz + (bwd ? 3 : 1) * 64 * ts + (bwd ? 64 * sizeof(float) : 0);  // one-tile kernels; > 160 KiB or wlen > 64: tiled kernels
    return YMI_OK;
}

extern "C" int ymi_window_attention_fwd(const ymi_tensor* qkv, int64_t wlen, int64_t heads, const ymi_tensor* out, float* lse, void* stream) {
    AttnArgs a{};
    size_t lds = 0;
    int rc = attn_common(qkv, wlen, heads, &a, &lds, false, "window_attention_fwd");
    if (rc) return rc;
    YMI_CHECK_ARG(ymi_tensor_ok(out) && out->c == a.C && ymi_pixels(out) == ymi_pixels(qkv) && out->dtype == qkv->dtype, "window_attention_fwd: out");
    a.qkv = SV{qkv->data, qkv->ld}; a.out = SV{out->data, out->ld}; a.lse = lse;
    dim3 grid((unsigned)(ymi_pixels(qkv) / wlen), (unsigned)heads);
    const int attn_tiled = ymi_opt(OPT_ATTN_TILED);  // 1: tiled kernels for every window size (tests)
    const size_t lds_tr = (size_t)3 * 64 * (a.hdp + 8) * 2 + (size_t)64 * (64 + 8) * 2;
    if (wlen > 64 || lds > 160 * 1024 || attn_tiled) {
        dim3 tg(grid.x, grid.y, (unsigned)((wlen + 63) / 64));
        if (qkv->dtype == YMI_BF16) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(window_attn_tiled_fwd_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipLaunchKernelGGL(window_attn_tiled_fwd_kernel<bf16_t>, tg, dim3(256), tiled_lds<bf16_t>(false), (hipStream_t)stream, a);
        } else {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(window_attn_tiled_fwd_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipLaunchKernelGGL(window_attn_tiled_fwd_kernel<float>, tg, dim3(256), tiled_lds<float>(false), (hipStream_t)stream, a);
        }
    } else if (qkv->dtype == YMI_BF16 && lds_tr <= 160 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(window_attn_fwd_tr_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(window_attn_fwd_tr_kernel, grid, dim3(256), lds_tr, (hipStream_t)stream, a);
    } else if (qkv->dtype == YMI_BF16) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(window_attn_fwd_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(window_attn_fwd_kernel<bf16_t>, grid, dim3(256), lds, (hipStream_t)stream, a);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(window_attn_fwd_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(window_attn_fwd_kernel<float>, grid, dim3(256), lds, (hipStream_t)stream, a);
    }
    YMI_CHECK_LAUNCH("window_attention_fwd");
    return YMI_OK;
}

extern "C" int ymi_window_attention_bwd(const ymi_tensor* qkv, const ymi_tensor* out, const ymi_tensor* dout, const float* lse, int64_t wlen,
                                        int64_t heads, const ymi_tensor* dqkv, void* stream) {
    AttnArgs a{};
    size_t lds = 0;
    int rc = attn_common(qkv, wlen, heads, &a, &lds, true, "window_attention_bwd");
    if (rc) return rc;
    YMI_CHECK_ARG(ymi_tensor_ok(out) && ymi_tensor_ok(dout) && ymi_tensor_ok(dqkv) && lse, "window_attention_bwd: args");
    YMI_CHECK_ARG(out->c == a.C && dout->c == a.C && dqkv->c == 3 * a.C && out->dtype == qkv->dtype && dout->dtype == qkv->dtype && dqkv->dtype == qkv->dtype,
                  "window_attention_bwd: shapes/dtypes");
    YMI_CHECK_ARG(ymi_pixels(out) == ymi_pixels(qkv) && ymi_pixels(dout) == ymi_pixels(qkv) && ymi_pixels(dqkv) == ymi_pixels(qkv), "window_attention_bwd: token counts");
    a.qkv = SV{qkv->data, qkv->ld}; a.out = SV{out->data, out->ld}; a.dout = SV{dout->data, dout->ld}; a.dqkv = SV{dqkv->data, dqkv->ld};
    a.lse = const_cast<float*>(lse);
    dim3 grid((unsigned)(ymi_pixels(qkv) / wlen), (unsigned)heads);
    const int attn_tiled = ymi_opt(OPT_ATTN_TILED);
    const size_t lds_tr = (size_t)4 * 64 * (a.hdp + 8) * 2 + (size_t)2 * 64 * (64 + 8) * 2;
    if (wlen > 64 || lds > 160 * 1024 || attn_tiled) {
        dim3 tg(grid.x, grid.y, (unsigned)((wlen + 63) / 64));
#define YMI_TILED_BWD(T)                                                                                                                          \
    do {                                                                                                                                          \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(window_attn_tiled_dq_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);  \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(window_attn_tiled_dkv_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        hipLaunchKernelGGL(window_attn_tiled_dq_kernel<T>, tg, dim3(256), tiled_lds<T>(false), (hipStream_t)stream, a);                             \
        hipLaunchKernelGGL(window_attn_tiled_dkv_kernel<T>, tg, dim3(256), tiled_lds<T>(true), (hipStream_t)stream, a);                             \
    } while (0)
        if (qkv->dtype == YMI_BF16) YMI_TILED_BWD(bf16_t);
        else YMI_TILED_BWD(float);
#undef YMI_TILED_BWD
    } else if (qkv->dtype == YMI_BF16 && lds_tr <= 160 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(window_attn_bwd_tr_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(window_attn_bwd_tr_kernel, grid, dim3(256), lds_tr, (hipStream_t)stream, a);
    } else if (qkv->dtype == YMI_BF16) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(window_attn_bwd_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(window_attn_bwd_kernel<bf16_t>, grid, dim3(256), lds, (hipStream_t)stream, a);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(window_attn_bwd_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(window_attn_bwd_kernel<float>, grid, dim3(256), lds, (hipStream_t)stream, a);
    }
    YMI_CHECK_LAUNCH("window_attention_bwd");
    return YMI_OK;
}
