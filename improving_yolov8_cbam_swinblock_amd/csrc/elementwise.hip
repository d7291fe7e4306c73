// Memory-bound helpers: layout changes, casts, weight packing, BatchNorm finalize / affine+activation.
// All are grid-stride kernels over 4-element groups (8 B bf16 / 16 B f32 per lane access); tensors
// whose ld or base is not 4-element aligned take a scalar path.
#include <math.h>

#include "common.h"

static inline dim3 ew_grid(int64_t work) {
    int64_t b = (work + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;  // 16 workgroups per CU, grid-stride the rest
    if (b < 1) b = 1;
    return dim3((unsigned)b);
}
static inline bool vec4_ok(const ymi_tensor* t) {
    return t->c % 4 == 0 && t->ld % 4 == 0 && ((uintptr_t)t->data % (4 * ymi_esize(t->dtype))) == 0;
}

struct TV {  // device view of a ymi_tensor
    void* p;
    int64_t ld;
    int n, h, w, c;
};
static inline TV tv(const ymi_tensor* t) { return TV{t->data, t->ld, (int)t->n, (int)t->h, (int)t->w, (int)t->c}; }

// ---------------------------------------------------------------------------------- NCHW <-> NHWC
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, int C, int64_t HW, int64_t NP, TV d) {
    // one thread per (pixel, 4-channel group); reads are coalesced over pixels for each channel
    const int groups = d.c / 4;
    const int64_t total = NP * groups;
    T* dst = reinterpret_cast<T*>(d.p);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i % NP;  // pixel fastest -> coalesced source reads
        const int g = (int)(i / NP);
        const int64_t n = p / HW, hw = p - n * HW;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = g * 4 + r;
            v[r] = c < C ? src[(n * C + c) * HW + hw] : 0.0f;
        }
        Pack<T, 4>::store(dst + p * d.ld + g * 4, v);
    }
}

// image path (C <= 8 into 8 bf16 channels): one thread per pixel reads the C planes (coalesced over pixels) and
// writes ONE 16-byte chunk, so a wave stores 1 KiB contiguously
__global__ __launch_bounds__(256) void nchw_to_nhwc8_bf16_kernel(const float* __restrict__ src, int C, uint32_t HW, uint32_t NP, TV d) {
    bf16_t* dst = reinterpret_cast<bf16_t*>(d.p);
    for (uint32_t p = blockIdx.x * 256u + threadIdx.x; p < NP; p += gridDim.x * 256u) {
        const uint32_t n = p / HW, hw = p - n * HW;
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = c < C ? src[((uint64_t)n * C + c) * HW + hw] : 0.0f;
        Pack<bf16_t, 8>::store(dst + (uint64_t)p * d.ld, v);
    }
}

extern "C" int ymi_nchw_to_nhwc(const float* src, int64_t n, int64_t c, int64_t h, int64_t w, const ymi_tensor* dst, void* stream) {
    YMI_CHECK_ARG(src && ymi_tensor_ok(dst), "nchw_to_nhwc: bad tensor");
    YMI_CHECK_ARG(dst->n == n && dst->h == h && dst->w == w && dst->c >= c, "nchw_to_nhwc: shape");
    YMI_CHECK_ARG(vec4_ok(dst), "nchw_to_nhwc: destination must be 4-channel aligned");
    const int64_t np = n * h * w, total = np * (dst->c / 4);
    if (dst->dtype == YMI_BF16 && dst->c == 8 && dst->ld % 8 == 0 && ((uintptr_t)dst->data & 15) == 0 && np < (1ll << 31)) {
        hipLaunchKernelGGL(nchw_to_nhwc8_bf16_kernel, ew_grid(np), dim3(256), 0, (hipStream_t)stream, src, (int)c, (uint32_t)(h * w), (uint32_t)np, tv(dst));
        YMI_CHECK_LAUNCH("nchw_to_nhwc");
        return YMI_OK;
    }
    if (dst->dtype == YMI_BF16)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, ew_grid(total), dim3(256), 0, (hipStream_t)stream, src, (int)c, h * w, np, tv(dst));
    else
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, ew_grid(total), dim3(256), 0, (hipStream_t)stream, src, (int)c, h * w, np, tv(dst));
    YMI_CHECK_LAUNCH("nchw_to_nhwc");
    return YMI_OK;
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(TV s, float* __restrict__ dst, int64_t HW, int64_t total) {
    const T* src = reinterpret_cast<const T*>(s.p);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t hw = i % HW;
        const int64_t nc = i / HW;
        const int c = (int)(nc % s.c);
        const int64_t n = nc / s.c;
        dst[i] = to_f32(src[(n * HW + hw) * s.ld + c]);
    }
}

extern "C" int ymi_nhwc_to_nchw(const ymi_tensor* src, float* dst, void* stream) {
    YMI_CHECK_ARG(ymi_tensor_ok(src) && dst, "nhwc_to_nchw: bad tensor");
    const int64_t hw = src->h * src->w, total = src->n * src->c * hw;
    if (src->dtype == YMI_BF16)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, ew_grid(total), dim3(256), 0, (hipStream_t)stream, tv(src), dst, hw, total);
    else
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, ew_grid(total), dim3(256), 0, (hipStream_t)stream, tv(src), dst, hw, total);
    YMI_CHECK_LAUNCH("nhwc_to_nchw");
    return YMI_OK;
}

// ------------------------------------------------------------------- copy / upsample / accumulate
// MODE 0: dst = src ; 1: dst(2h+i,2w+j) = src(h,w) ; 2: dst(h,w) = sum src(2h+i,2w+j) ; 3: dst += src ; 4: dst(h,w) += sum src(2h+i,2w+j)
template <typename TS, typename TD, int MODE, bool VEC>
__global__ void move_kernel(TV s, TV d) {
    constexpr int G = VEC ? 4 : 1;
    const int groups = d.c / G;
    const int64_t total = (int64_t)d.n * d.h * d.w * groups;
    const TS* src = reinterpret_cast<const TS*>(s.p);
    TD* dst = reinterpret_cast<TD*>(d.p);
    // 32-bit index arithmetic (launch_move checks total < 2^31): 64-bit division would dominate this kernel
    const uint32_t total32 = (uint32_t)total, ugroups = (uint32_t)groups, uw = (uint32_t)d.w, uh = (uint32_t)d.h;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total32; i += gridDim.x * blockDim.x) {
        const uint32_t pu = i / ugroups;
        const int g = (int)(i - pu * ugroups);
        const int64_t p = pu;
        int w = 0, h = 0, n = 0;
        if (MODE == 1 || MODE == 2 || MODE == 4) {
            const uint32_t t = pu / uw;
            w = (int)(pu - t * uw);
            n = (int)(t / uh);
            h = (int)(t - (uint32_t)n * uh);
        }
        float v[G];
        auto ld = [&](int64_t sp, float (&o)[G]) {
            if constexpr (VEC) Pack<TS, 4>::load(src + sp * s.ld + g * 4, o);
            else o[0] = to_f32(src[sp * s.ld + g]);
        };
        if (MODE == 0 || MODE == 3) {
            ld(p, v);
        } else if (MODE == 1) {
            ld(((int64_t)n * s.h + (h >> 1)) * s.w + (w >> 1), v);
        } else {
            float a[G], b[G], c2[G], e[G];
            const int64_t base = ((int64_t)n * s.h + 2 * h) * s.w + 2 * w;
            ld(base, a); ld(base + 1, b); ld(base + s.w, c2); ld(base + s.w + 1, e);
#pragma unroll
            for (int r = 0; r < G; ++r) v[r] = (a[r] + b[r]) + (c2[r] + e[r]);
        }
        if (MODE == 3 || MODE == 4) {
            float o[G];
            if constexpr (VEC) Pack<TD, 4>::load(dst + p * d.ld + g * 4, o);
            else o[0] = to_f32(dst[p * d.ld + g]);
#pragma unroll
            for (int r = 0; r < G; ++r) v[r] += o[r];
        }
        if constexpr (VEC) Pack<TD, 4>::store(dst + p * d.ld + g * 4, v);
        else dst[p * d.ld + g] = from_f32<TD>(v[0]);
    }
}

template <int MODE>
static int launch_move(const ymi_tensor* src, const ymi_tensor* dst, const char* what, hipStream_t stream) {
    const bool vec = vec4_ok(src) && vec4_ok(dst);
    const int64_t total = ymi_pixels(dst) * (vec ? dst->c / 4 : dst->c);
    YMI_CHECK_ARG(total < (1ll << 31), "%s: tensor too large for 32-bit indexing", what);
    dim3 g = ew_grid(total), b(256);
#define YMI_MV(TS, TD)                                                                              \
    do {                                                                                            \
        if (vec) hipLaunchKernelGGL((move_kernel<TS, TD, MODE, true>), g, b, 0, stream, tv(src), tv(dst));  \
        else hipLaunchKernelGGL((move_kernel<TS, TD, MODE, false>), g, b, 0, stream, tv(src), tv(dst));     \
    } while (0)
    if (src->dtype == YMI_BF16 && dst->dtype == YMI_BF16) YMI_MV(bf16_t, bf16_t);
    else if (src->dtype == YMI_F32 && dst->dtype == YMI_F32) YMI_MV(float, float);
    else if (src->dtype == YMI_F32 && dst->dtype == YMI_BF16) YMI_MV(float, bf16_t);
    else YMI_MV(bf16_t, float);
#undef YMI_MV
    YMI_CHECK_LAUNCH(what);
    return YMI_OK;
}

extern "C" int ymi_copy(const ymi_tensor* src, const ymi_tensor* dst, void* stream) {
    YMI_CHECK_ARG(ymi_tensor_ok(src) && ymi_tensor_ok(dst) && ymi_same_shape(src, dst), "copy: shapes");
    return launch_move<0>(src, dst, "copy", (hipStream_t)stream);
}
extern "C" int ymi_upsample2x(const ymi_tensor* src, const ymi_tensor* dst, void* stream) {
    YMI_CHECK_ARG(ymi_tensor_ok(src) && ymi_tensor_ok(dst), "upsample2x: bad tensor");
    YMI_CHECK_ARG(dst->n == src->n && dst->h == 2 * src->h && dst->w == 2 * src->w && dst->c == src->c, "upsample2x: shapes");
    return launch_move<1>(src, dst, "upsample2x", (hipStream_t)stream);
}
extern "C" int ymi_upsample2x_bwd(const ymi_tensor* src, const ymi_tensor* dst, void* stream) {
    YMI_CHECK_ARG(ymi_tensor_ok(src) && ymi_tensor_ok(dst), "upsample2x_bwd: bad tensor");
    YMI_CHECK_ARG(dst->n == src->n && src->h == 2 * dst->h && src->w == 2 * dst->w && dst->c == src->c, "upsample2x_bwd: shapes");
    return launch_move<2>(src, dst, "upsample2x_bwd", (hipStream_t)stream);
}
extern "C" int ymi_upsample2x_bwd_acc(const ymi_tensor* src, const ymi_tensor* dst, void* stream) {
    YMI_CHECK_ARG(ymi_tensor_ok(src) && ymi_tensor_ok(dst), "upsample2x_bwd_acc: bad tensor");
    YMI_CHECK_ARG(dst->n == src->n && src->h == 2 * dst->h && src->w == 2 * dst->w && dst->c == src->c, "upsample2x_bwd_acc: shapes");
    return launch_move<4>(src, dst, "upsample2x_bwd_acc", (hipStream_t)stream);
}
extern "C" int ymi_add_inplace(const ymi_tensor* src, const ymi_tensor* dst, void* stream) {
    YMI_CHECK_ARG(ymi_tensor_ok(src) && ymi_tensor_ok(dst) && ymi_same_shape(src, dst), "add_inplace: shapes");
    return launch_move<3>(src, dst, "add_inplace", (hipStream_t)stream);
}

// --------------------------------------------------------------------------------- weight packing
template <typename T>
__global__ void pack_fwd_kernel(const float* __restrict__ w, int O, int I, int KH, int KW, int IP, T* __restrict__ dst) {
    // dst[o][kh][kw][ip]  <-  w[o][i][kh][kw]
    const int64_t total = (int64_t)O * KH * KW * IP;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int ip = (int)(idx % IP);
        int64_t t = idx / IP;
        const int kw = (int)(t % KW); t /= KW;
        const int kh = (int)(t % KH);
        const int o = (int)(t / KH);
        const float v = ip < I ? w[(((int64_t)o * I + ip) * KH + kh) * KW + kw] : 0.0f;
        dst[idx] = from_f32<T>(v);
    }
}

extern "C" int ymi_pack_conv_weight_fwd(const float* w, int64_t o, int64_t i, int64_t kh, int64_t kw, int64_t ipad, int32_t dtype,
                                        void* dst, void* stream) {
    YMI_CHECK_ARG(w && dst && ipad >= i && o > 0 && i > 0, "pack_conv_weight_fwd: args");
    const int64_t total = o * kh * kw * ipad;
    if (dtype == YMI_BF16)
        hipLaunchKernelGGL(pack_fwd_kernel<bf16_t>, ew_grid(total), dim3(256), 0, (hipStream_t)stream, w, (int)o, (int)i, (int)kh, (int)kw, (int)ipad, (bf16_t*)dst);
    else
        hipLaunchKernelGGL(pack_fwd_kernel<float>, ew_grid(total), dim3(256), 0, (hipStream_t)stream, w, (int)o, (int)i, (int)kh, (int)kw, (int)ipad, (float*)dst);
    YMI_CHECK_LAUNCH("pack_conv_weight_fwd");
    return YMI_OK;
}

struct DgradPack {
    int ntaps;          // taps of this class
    int kh[9], kw[9];   // original kernel coordinates of each tap
    int64_t off;        // element offset of the class block
};
template <typename T>
__global__ void pack_dgrad_kernel(const float* __restrict__ w, int O, int OP, int I, int KH, int KW, DgradPack d, T* __restrict__ dst) {
    // class block: dst[i][t][op]  <-  w[o][i][kh_t][kw_t]
    const int64_t total = (int64_t)I * d.ntaps * OP;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int o = (int)(idx % OP);
        int64_t t = idx / OP;
        const int tp = (int)(t % d.ntaps);
        const int i = (int)(t / d.ntaps);
        const float v = o < O ? w[(((int64_t)o * I + i) * KH + d.kh[tp]) * KW + d.kw[tp]] : 0.0f;
        dst[d.off + idx] = from_f32<T>(v);
    }
}

// `o` is the PADDED channel count of dy (multiple of 8 bf16 / 4 f32); o_real rows exist in w.
extern "C" int ymi_pack_conv_weight_dgrad_ex(const float* w, int64_t o_real, int64_t o, int64_t i, int64_t kh, int64_t kw, int64_t stride,
                                             int32_t dtype, void* dst, void* stream) {
    YMI_CHECK_ARG(w && dst && o >= o_real && (stride == 1 || stride == 2), "pack_conv_weight_dgrad: args");
    const int pad = (int)kh / 2;
    const int nclass = stride == 1 ? 1 : 4;
    int64_t off = 0;
    for (int cls = 0; cls < nclass; ++cls) {
        const int ph = stride == 1 ? 0 : cls / 2, pw = stride == 1 ? 0 : cls % 2;
        DgradPack d{};
        for (int a = 0; a < kh; ++a)
            for (int b = 0; b < kw; ++b) {
                const int nh = ph + pad - a, nw = pw + pad - b;
                if (nh % (int)stride != 0 || nw % (int)stride != 0) continue;
                d.kh[d.ntaps] = a; d.kw[d.ntaps] = b; ++d.ntaps;
            }
        d.off = off;
        const int64_t total = i * d.ntaps * o;
        if (total > 0) {
            if (dtype == YMI_BF16)
                hipLaunchKernelGGL(pack_dgrad_kernel<bf16_t>, ew_grid(total), dim3(256), 0, (hipStream_t)stream, w, (int)o_real, (int)o, (int)i, (int)kh, (int)kw, d, (bf16_t*)dst);
            else
                hipLaunchKernelGGL(pack_dgrad_kernel<float>, ew_grid(total), dim3(256), 0, (hipStream_t)stream, w, (int)o_real, (int)o, (int)i, (int)kh, (int)kw, d, (float*)dst);
        }
        off += total;
    }
    YMI_CHECK_LAUNCH("pack_conv_weight_dgrad");
    return YMI_OK;
}
extern "C" int ymi_pack_conv_weight_dgrad(const float* w, int64_t o, int64_t i, int64_t kh, int64_t kw, int64_t stride, int32_t dtype,
                                          void* dst, void* stream) {
    return ymi_pack_conv_weight_dgrad_ex(w, o, o, i, kh, kw, stride, dtype, dst, stream);
}

template <typename T>
__global__ void pack_matrix_kernel(const float* __restrict__ src, int R, int C, int transpose, T* __restrict__ dst) {
    const int64_t total = (int64_t)R * C;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        if (!transpose) {
            dst[idx] = from_f32<T>(src[idx]);
        } else {  // dst[c][r] = src[r][c]; idx walks dst
            const int r = (int)(idx % R);
            const int c = (int)(idx / R);
            dst[idx] = from_f32<T>(src[(int64_t)r * C + c]);
        }
    }
}
extern "C" int ymi_pack_matrix(const float* src, int64_t rows, int64_t cols, int32_t transpose, int32_t dtype, void* dst, void* stream) {
    YMI_CHECK_ARG(src && dst && rows > 0 && cols > 0, "pack_matrix: args");
    const int64_t total = rows * cols;
    if (dtype == YMI_BF16)
        hipLaunchKernelGGL(pack_matrix_kernel<bf16_t>, ew_grid(total), dim3(256), 0, (hipStream_t)stream, src, (int)rows, (int)cols, transpose, (bf16_t*)dst);
    else
        hipLaunchKernelGGL(pack_matrix_kernel<float>, ew_grid(total), dim3(256), 0, (hipStream_t)stream, src, (int)rows, (int)cols, transpose, (float*)dst);
    YMI_CHECK_LAUNCH("pack_matrix");
    return YMI_OK;
}

// ------------------------------------------------------------------------------ BatchNorm pieces
// 32 channels x 32 row slices per 1024-thread block (short dependent chains: this kernel is pure latency);
// sums in double so that the cross-block reduction adds nothing to the error of the per-block f32 partials.
// second parameter set (p2): channels >= split read / update gamma2[c - split] ... - the BatchNorms of two convolutions that ran as one
// (Detect's sibling branches, head.py:71-72); p2.gamma == nullptr: one set
struct BnParams2 {
    const float* gamma;
    const float* beta;
    float* rmean;
    float* rvar;
    int split;
};
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ part, int blocks, double count, int C,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ rmean, float* __restrict__ rvar, float momentum, float eps,
                                                           float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ smean,
                                                           float* __restrict__ sinv, BnParams2 p2) {
    __shared__ double red[2][32][33];
    const int cl = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    int pc = c;  // index into this channel's parameter arrays
    if (p2.split > 0 && c >= p2.split) {
        gamma = p2.gamma; beta = p2.beta; rmean = p2.rmean; rvar = p2.rvar;
        pc = c - p2.split;
    }
    double s1 = 0.0, s2 = 0.0;
    // parameters and running statistics are fetched first, beside the partials (a chain of load latencies otherwise)
    float g_ = 1.0f, b_ = 0.0f, rm_ = 0.f, rv_ = 0.f;
    if (slice == 0 && c < C) {
        if (gamma) g_ = gamma[pc];
        if (beta) b_ = beta[pc];
        if (rmean) rm_ = rmean[pc];
        if (rvar) rv_ = rvar[pc];
    }
    if (c < C) {
        // four independent chains per sum: the loads of a trip are all in flight before the first add
        float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f, c0 = 0.f, c1 = 0.f, d0 = 0.f, d1 = 0.f;
        int b = slice;
        for (; b + 96 < blocks; b += 128) {
            a0 += part[((int64_t)b * 2 + 0) * C + c];
            a1 += part[((int64_t)b * 2 + 1) * C + c];
            b0 += part[((int64_t)(b + 32) * 2 + 0) * C + c];
            b1 += part[((int64_t)(b + 32) * 2 + 1) * C + c];
            c0 += part[((int64_t)(b + 64) * 2 + 0) * C + c];
            c1 += part[((int64_t)(b + 64) * 2 + 1) * C + c];
            d0 += part[((int64_t)(b + 96) * 2 + 0) * C + c];
            d1 += part[((int64_t)(b + 96) * 2 + 1) * C + c];
        }
        s1 = ((double)a0 + (double)b0) + ((double)c0 + (double)d0);
        s2 = ((double)a1 + (double)b1) + ((double)c1 + (double)d1);
        for (; b < blocks; b += 32) {
            s1 += (double)part[((int64_t)b * 2 + 0) * C + c];
            s2 += (double)part[((int64_t)b * 2 + 1) * C + c];
        }
    }
    red[0][slice][cl] = s1;
    red[1][slice][cl] = s2;
    __syncthreads();
    if (slice == 0 && c < C) {
        s1 = 0.0;
        s2 = 0.0;
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            s1 += red[0][q][cl];
            s2 += red[1][q][cl];
        }
        const double mean = s1 / count;
        double var = s2 / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float inv = (float)(1.0 / sqrt(var + (double)eps));
        const float g = g_, b = b_;
        const float sc = g * inv;
        scale[c] = sc;
        shift[c] = b - (float)mean * sc;
        if (smean) smean[c] = (float)mean;
        if (sinv) sinv[c] = inv;
        if (rmean) rmean[pc] = (1.0f - momentum) * rm_ + momentum * (float)mean;
        if (rvar) {
            const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            rvar[pc] = (1.0f - momentum) * rv_ + momentum * (float)unb;
        }
    }
}

// First stage for long partial lists (layer 0 at bs 32 has 25,600 rows): R row-groups in parallel, each
// summed in double and stored as float, IN PLACE over the first R rows' slots of a separate region.
__global__ __launch_bounds__(256) void stat_rows_reduce_kernel(const float* __restrict__ part, int blocks, int C, int R, float* __restrict__ out) {
    __shared__ double red[2][4][64];
    const int cl = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int g = blockIdx.y;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
        int b = g + R * slice;
        for (; b + 12 * R < blocks; b += 16 * R) {  // four rows per trip in flight
            const float x0 = part[((int64_t)b * 2 + 0) * C + c], y0 = part[((int64_t)b * 2 + 1) * C + c];
            const float x1 = part[((int64_t)(b + 4 * R) * 2 + 0) * C + c], y1 = part[((int64_t)(b + 4 * R) * 2 + 1) * C + c];
            const float x2 = part[((int64_t)(b + 8 * R) * 2 + 0) * C + c], y2 = part[((int64_t)(b + 8 * R) * 2 + 1) * C + c];
            const float x3 = part[((int64_t)(b + 12 * R) * 2 + 0) * C + c], y3 = part[((int64_t)(b + 12 * R) * 2 + 1) * C + c];
            s1 += ((double)x0 + (double)x1) + ((double)x2 + (double)x3);
            s2 += ((double)y0 + (double)y1) + ((double)y2 + (double)y3);
        }
        for (; b < blocks; b += 4 * R) {
            s1 += (double)part[((int64_t)b * 2 + 0) * C + c];
            s2 += (double)part[((int64_t)b * 2 + 1) * C + c];
        }
    }
    red[0][slice][cl] = s1;
    red[1][slice][cl] = s2;
    __syncthreads();
    if (slice == 0 && c < C) {
        out[((int64_t)g * 2 + 0) * C + c] = (float)(red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl]);
        out[((int64_t)g * 2 + 1) * C + c] = (float)(red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl]);
    }
}

constexpr int BN_STAGE_ROWS = 64;

// `part` may be overwritten beyond row `blocks` (callers size it with ymi_conv2d_stat_blocks + BN_STAGE_ROWS rows).
static int bn_finalize_impl(const float* part, int64_t blocks, int64_t count, int64_t c, const float* gamma, const float* beta,
                            float* rmean, float* rvar, float momentum, float eps, float* scale, float* shift, float* smean,
                            float* sinv, BnParams2 p2, void* stream);
extern "C" int ymi_bn_finalize(const float* part, int64_t blocks, int64_t count, int64_t c, const float* gamma, const float* beta,
                               float* rmean, float* rvar, float momentum, float eps, float* scale, float* shift, float* smean,
                               float* sinv, void* stream) {
    return bn_finalize_impl(part, blocks, count, c, gamma, beta, rmean, rvar, momentum, eps, scale, shift, smean, sinv, BnParams2{nullptr, nullptr, nullptr, nullptr, 0}, stream);
}
extern "C" int ymi_bn_finalize_pair(const float* part, int64_t blocks, int64_t count, int64_t c, int64_t split, const float* gamma, const float* beta,
                                    float* rmean, float* rvar, const float* gamma2, const float* beta2, float* rmean2, float* rvar2, float momentum,
                                    float eps, float* scale, float* shift, float* smean, float* sinv, void* stream) {
    YMI_CHECK_ARG(split > 0 && split < c && gamma2 && beta2, "bn_finalize_pair: split");
    return bn_finalize_impl(part, blocks, count, c, gamma, beta, rmean, rvar, momentum, eps, scale, shift, smean, sinv, BnParams2{gamma2, beta2, rmean2, rvar2, (int)split}, stream);
}
static int bn_finalize_impl(const float* part, int64_t blocks, int64_t count, int64_t c, const float* gamma, const float* beta,
                            float* rmean, float* rvar, float momentum, float eps, float* scale, float* shift, float* smean,
                            float* sinv, BnParams2 p2, void* stream) {
    YMI_CHECK_ARG(part && scale && shift && blocks > 0 && count > 0 && c > 0, "bn_finalize: args");
    const float* src = part;
    int rows = (int)blocks;
    if (blocks > 16 * BN_STAGE_ROWS) {
        float* stage = const_cast<float*>(part) + blocks * 2 * c;  // scratch rows after the partials
        hipLaunchKernelGGL(stat_rows_reduce_kernel, dim3((unsigned)((c + 63) / 64), BN_STAGE_ROWS), dim3(256), 0, (hipStream_t)stream, part,
                           (int)blocks, (int)c, BN_STAGE_ROWS, stage);
        src = stage;
        rows = BN_STAGE_ROWS;
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)((c + 31) / 32)), dim3(1024), 0, (hipStream_t)stream, src, rows,
                       (double)count, (int)c, gamma, beta, rmean, rvar, momentum, eps, scale, shift, smean, sinv, p2);
    YMI_CHECK_LAUNCH("bn_finalize");
    return YMI_OK;
}

template <typename T, bool VEC>
__global__ void scale_shift_act_kernel(TV x, const float* __restrict__ scale, const float* __restrict__ shift, int act, TV res, TV o) {
    constexpr int G = VEC ? 4 : 1;
    const int groups = x.c / G;
    const int64_t total = (int64_t)x.n * x.h * x.w * groups;
    const T* xp = reinterpret_cast<const T*>(x.p);
    const T* rp = reinterpret_cast<const T*>(res.p);
    T* op = reinterpret_cast<T*>(o.p);
    const uint32_t total32 = (uint32_t)total, ugroups = (uint32_t)groups;  // host: total < 2^31
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total32; i += gridDim.x * blockDim.x) {
        const uint32_t pu = i / ugroups;
        const int g = (int)(i - pu * ugroups);
        const int64_t p = pu;
        float v[G];
        if constexpr (VEC) Pack<T, 4>::load(xp + p * x.ld + g * 4, v);
        else v[0] = to_f32(xp[p * x.ld + g]);
#pragma unroll
        for (int r = 0; r < G; ++r) {
            const int c = g * G + r;
            const float s = scale ? scale[c] : 1.0f, b = shift ? shift[c] : 0.0f;
            v[r] = apply_act_rt(v[r] * s + b, act);
        }
        if (rp) {
            float rr[G];
            if constexpr (VEC) Pack<T, 4>::load(rp + p * res.ld + g * 4, rr);
            else rr[0] = to_f32(rp[p * res.ld + g]);
#pragma unroll
            for (int r = 0; r < G; ++r) v[r] += rr[r];
        }
        if constexpr (VEC) Pack<T, 4>::store(op + p * o.ld + g * 4, v);
        else op[p * o.ld + g] = from_f32<T>(v[0]);
    }
}

// Fast path: the number of 4-channel groups divides the block size, so a thread keeps ONE channel group for
// all its pixels and holds scale/shift in registers; 8-byte (bf16) / 16-byte (f32) accesses, coalesced along C.
// BatchNorm finalize folded into the affine pass (round 5): `fin.acc` holds the statistics of this layer as fixed-point sums
// [4 replicas][2][C] (igemm.hip's statistics epilogue).  Every thread derives scale / shift of ITS four channels in the prologue (eight
// 16-byte loads, all in flight together, double arithmetic as bn_finalize_kernel); workgroup 0 also writes the saved mean / inverse
// deviation and updates the running statistics (conv.py:66-67; torch_utils.py:468-470 sets eps / momentum).  nullptr: scale / shift arrays.
struct BnFin {
    const long long* acc;
    const float* gamma;
    const float* beta;
    float* rmean;
    float* rvar;
    float* smean;
    float* sinv;
    double count;
    double inv_scale;  // 2^-shift of the fixed-point sums (common.h: ymi_stat_fixed_point_shift)
    float momentum, eps;
};
#ifndef YMI_EW_U  // pixels a thread of the affine pass keeps in flight per trip
#define YMI_EW_U 4
#endif
template <typename T, int ACT, bool FIN = false>
__global__ __launch_bounds__(256) void scale_shift_act_fixed_kernel(TV x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                                    TV res, TV o, int groups, int64_t Pall, int64_t span, BnFin fin = BnFin{}) {
    // pixels of this workgroup's XCD only (common.h, XCD ownership of the pixel axis): the GEMM that wrote `x` and the one that will read
    // `o` give this XCD the same eighth
    const XcdRange xr = xcd_range(Pall, span);
    const int64_t P = xr.hi;
    const int g = threadIdx.x % groups;
    const int rows_per_block = 256 / groups;
    if ((int)threadIdx.x >= rows_per_block * groups) return;  // group counts that do not divide 256 (192 channels: 48 groups, 5 rows, 16 idle threads); no barriers below
    float sc[4], sh[4];
    if constexpr (FIN) {
        const int C = groups * 4;
        typedef __attribute__((ext_vector_type(2))) long long i64x2;
        i64x2 q[4][2][2];  // [replica][sum | sum of squares][channel pair]
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int w = 0; w < 2; ++w)
#pragma unroll
                for (int h = 0; h < 2; ++h) q[rep][w][h] = *reinterpret_cast<const i64x2*>(fin.acc + (int64_t)(rep * 2 + w) * C + g * 4 + 2 * h);
        const f32x4 ga = fin.gamma ? *reinterpret_cast<const f32x4*>(fin.gamma + g * 4) : f32x4{1.f, 1.f, 1.f, 1.f};
        const f32x4 be = fin.beta ? *reinterpret_cast<const f32x4*>(fin.beta + g * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        const bool writer = blockIdx.x == 0 && (int)threadIdx.x < groups;  // one thread per channel group of workgroup 0
        f32x4 rm = {0.f, 0.f, 0.f, 0.f}, rv = {0.f, 0.f, 0.f, 0.f};
        if (writer && fin.rmean) rm = *reinterpret_cast<const f32x4*>(fin.rmean + g * 4);
        if (writer && fin.rvar) rv = *reinterpret_cast<const f32x4*>(fin.rvar + g * 4);
        f32x4 mean4, inv4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long long s1 = (q[0][0][r >> 1][r & 1] + q[1][0][r >> 1][r & 1]) + (q[2][0][r >> 1][r & 1] + q[3][0][r >> 1][r & 1]);
            const long long s2 = (q[0][1][r >> 1][r & 1] + q[1][1][r >> 1][r & 1]) + (q[2][1][r >> 1][r & 1] + q[3][1][r >> 1][r & 1]);
            const double mean = (double)s1 * fin.inv_scale / fin.count;
            double var = (double)s2 * fin.inv_scale / fin.count - mean * mean;
            if (var < 0.0) var = 0.0;
            const float inv = (float)(1.0 / sqrt(var + (double)fin.eps));
            sc[r] = ga[r] * inv;
            sh[r] = be[r] - (float)mean * sc[r];
            mean4[r] = (float)mean;
            inv4[r] = inv;
            if (writer) {
                rm[r] = (1.0f - fin.momentum) * rm[r] + fin.momentum * (float)mean;
                const double unb = fin.count > 1.0 ? var * fin.count / (fin.count - 1.0) : var;
                rv[r] = (1.0f - fin.momentum) * rv[r] + fin.momentum * (float)unb;
            }
        }
        if (writer) {
            if (fin.smean) *reinterpret_cast<f32x4*>(fin.smean + g * 4) = mean4;
            if (fin.sinv) *reinterpret_cast<f32x4*>(fin.sinv + g * 4) = inv4;
            if (fin.rmean) *reinterpret_cast<f32x4*>(fin.rmean + g * 4) = rm;
            if (fin.rvar) *reinterpret_cast<f32x4*>(fin.rvar + g * 4) = rv;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sc[r] = scale ? scale[g * 4 + r] : 1.0f;
            sh[r] = shift ? shift[g * 4 + r] : 0.0f;
        }
    }
    const T* xp = reinterpret_cast<const T*>(x.p);
    const T* rp = reinterpret_cast<const T*>(res.p);
    T* op = reinterpret_cast<T*>(o.p);
    auto one = [&](float (&v)[4], const float (&rr)[4], int64_t p) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = apply_act<ACT>(v[r] * sc[r] + sh[r]) + rr[r];
        Pack<T, 4>::store(op + p * o.ld + g * 4, v);
    };
    const int64_t step = (int64_t)xr.nbx * rows_per_block;
    int64_t p = xr.lo + (int64_t)xr.bi * rows_per_block + threadIdx.x / groups;
    // 4 pixels per trip: the loads of all four are in flight before the first use, and the NEXT trip's (raw) loads are issued before this
    // trip's arithmetic (as in the BatchNorm backward passes, csrc/reduce_bwd.hip)
    typedef typename Raw4<T>::type R4;
    constexpr int U = YMI_EW_U;  // pixels per trip
    R4 rv[U], rres[U];
    bool have = p + (U - 1) * step < P;
    if (have) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            rv[u] = *reinterpret_cast<const R4*>(xp + (p + u * step) * x.ld + g * 4);
            if (rp) rres[u] = *reinterpret_cast<const R4*>(rp + (p + u * step) * res.ld + g * 4);
        }
    }
    while (have) {
        float v[U][4], rr[U][4] = {};
#pragma unroll
        for (int u = 0; u < U; ++u) {
            Raw4<T>::to_f32(rv[u], v[u]);
            if (rp) Raw4<T>::to_f32(rres[u], rr[u]);
        }
        const int64_t pc = p;
        p += U * step;
        have = p + (U - 1) * step < P;
        if (have) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                rv[u] = *reinterpret_cast<const R4*>(xp + (p + u * step) * x.ld + g * 4);
                if (rp) rres[u] = *reinterpret_cast<const R4*>(rp + (p + u * step) * res.ld + g * 4);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) one(v[u], rr[u], pc + u * step);
    }
    for (; p < P; p += step) {
        float v[4], rr[4] = {0.f, 0.f, 0.f, 0.f};
        Pack<T, 4>::load(xp + p * x.ld + g * 4, v);
        if (rp) Pack<T, 4>::load(rp + p * res.ld + g * 4, rr);
        one(v, rr, p);
    }
}

template <typename T>
static void launch_ssa_fixed(const ymi_tensor* raw, const float* scale, const float* shift, int act, TV r, const ymi_tensor* out, hipStream_t s, const BnFin* fin = nullptr) {
    const int groups = (int)raw->c / 4;
    const int64_t P = ymi_pixels(raw);
    const int rows = 256 / groups;
    // every thread reloads its group's scale/shift: give it ~8 pixels when the tensor allows.  Small maps (<= 13 MB here) are pure
    // latency: with 256 blocks a thread walked 3-6 pixels one dependent round trip after the other (10 us for 1.6 MB); give them
    // up to one resident round of blocks, i.e. 1-2 pixels per thread
    int64_t gb = (P + (int64_t)rows * ew_ppt() - 1) / ((int64_t)rows * ew_ppt());
    if (gb < 1024) gb = (P + rows - 1) / rows < 1024 ? (P + rows - 1) / rows : 1024;
    if (gb > ew_cap()) gb = ew_cap();  // default 1024: four 256-thread workgroups per CU (runtime.hip)
    gb = (gb + 7) / 8 * 8;             // the same number of workgroups on every XCD
    const int64_t span = ymi_xcd_span_arg(P);
    dim3 g((unsigned)gb), b(256);
    if (fin) {
        if (act == YMI_ACT_SILU) hipLaunchKernelGGL((scale_shift_act_fixed_kernel<T, YMI_ACT_SILU, true>), g, b, 0, s, tv(raw), scale, shift, r, tv(out), groups, P, span, *fin);
        else if (act == YMI_ACT_GELU) hipLaunchKernelGGL((scale_shift_act_fixed_kernel<T, YMI_ACT_GELU, true>), g, b, 0, s, tv(raw), scale, shift, r, tv(out), groups, P, span, *fin);
        else hipLaunchKernelGGL((scale_shift_act_fixed_kernel<T, YMI_ACT_NONE, true>), g, b, 0, s, tv(raw), scale, shift, r, tv(out), groups, P, span, *fin);
        return;
    }
    if (act == YMI_ACT_SILU) hipLaunchKernelGGL((scale_shift_act_fixed_kernel<T, YMI_ACT_SILU>), g, b, 0, s, tv(raw), scale, shift, r, tv(out), groups, P, span);
    else if (act == YMI_ACT_GELU) hipLaunchKernelGGL((scale_shift_act_fixed_kernel<T, YMI_ACT_GELU>), g, b, 0, s, tv(raw), scale, shift, r, tv(out), groups, P, span);
    else hipLaunchKernelGGL((scale_shift_act_fixed_kernel<T, YMI_ACT_NONE>), g, b, 0, s, tv(raw), scale, shift, r, tv(out), groups, P, span);
}

extern "C" int ymi_scale_shift_act(const ymi_tensor* raw, const float* scale, const float* shift, int32_t act, const ymi_tensor* residual,
                                   const ymi_tensor* out, void* stream) {
    YMI_CHECK_ARG(ymi_tensor_ok(raw) && ymi_tensor_ok(out) && ymi_same_shape(raw, out) && raw->dtype == out->dtype, "scale_shift_act: shapes");
    if (residual) YMI_CHECK_ARG(ymi_tensor_ok(residual) && ymi_same_shape(residual, out) && residual->dtype == out->dtype, "scale_shift_act: residual");
    const bool vec = vec4_ok(raw) && vec4_ok(out) && (!residual || vec4_ok(residual));
    const int64_t total = ymi_pixels(raw) * (vec ? raw->c / 4 : raw->c);
    YMI_CHECK_ARG(total < (1ll << 31), "scale_shift_act: tensor too large for 32-bit indexing");
    TV r = residual ? tv(residual) : TV{nullptr, 0, 0, 0, 0, 0};
    dim3 g = ew_grid(total), b(256);
    hipStream_t s = (hipStream_t)stream;
    if (vec && raw->c / 4 <= 256) {
        if (raw->dtype == YMI_BF16) launch_ssa_fixed<bf16_t>(raw, scale, shift, act, r, out, s);
        else launch_ssa_fixed<float>(raw, scale, shift, act, r, out, s);
        YMI_CHECK_LAUNCH("scale_shift_act");
        return YMI_OK;
    }
    if (raw->dtype == YMI_BF16) {
        if (vec) hipLaunchKernelGGL((scale_shift_act_kernel<bf16_t, true>), g, b, 0, s, tv(raw), scale, shift, act, r, tv(out));
        else hipLaunchKernelGGL((scale_shift_act_kernel<bf16_t, false>), g, b, 0, s, tv(raw), scale, shift, act, r, tv(out));
    } else {
        if (vec) hipLaunchKernelGGL((scale_shift_act_kernel<float, true>), g, b, 0, s, tv(raw), scale, shift, act, r, tv(out));
        else hipLaunchKernelGGL((scale_shift_act_kernel<float, false>), g, b, 0, s, tv(raw), scale, shift, act, r, tv(out));
    }
    YMI_CHECK_LAUNCH("scale_shift_act");
    return YMI_OK;
}

// Conv + train-mode BN + activation: three launches, one C call.
static int conv_bn_act_fwd_impl(const ymi_tensor* x, const void* w_packed, int64_t cout, int64_t kh, int64_t kw, int64_t stride,
                                const float* gamma, const float* beta, float* running_mean, float* running_var, BnParams2 p2, float momentum,
                                float eps, int32_t act, const ymi_tensor* residual, const ymi_tensor* raw, const ymi_tensor* out,
                                float* save_mean, float* save_invstd, void* workspace, size_t workspace_bytes, void* stream);
extern "C" int ymi_conv2d_bn_silu_fwd(const ymi_tensor* x, const void* w_packed, int64_t cout, int64_t kh, int64_t kw, int64_t stride,
                                      const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                                      float eps, int32_t act, const ymi_tensor* residual, const ymi_tensor* raw, const ymi_tensor* out,
                                      float* save_mean, float* save_invstd, void* workspace, size_t workspace_bytes, void* stream) {
    return conv_bn_act_fwd_impl(x, w_packed, cout, kh, kw, stride, gamma, beta, running_mean, running_var, BnParams2{nullptr, nullptr, nullptr, nullptr, 0}, momentum,
                                eps, act, residual, raw, out, save_mean, save_invstd, workspace, workspace_bytes, stream);
}
// Two convolutions of the SAME input as one (their packed weights lie back to back: output channels [0, split) are the first's,
// [split, cout) the second's), each with its own BatchNorm parameters and running statistics.  BatchNorm is per channel, so this is
// exactly the two separate Conv blocks; `raw` / `out` hold both results side by side.  Reference: Detect.forward reads x[i] with
// cv2[i] and cv3[i] (nn/modules/head.py:71-72).
extern "C" int ymi_conv2d_bn_silu_fwd_pair(const ymi_tensor* x, const void* w_packed, int64_t cout, int64_t split, int64_t kh, int64_t kw, int64_t stride,
                                           const float* gamma, const float* beta, float* running_mean, float* running_var, const float* gamma2,
                                           const float* beta2, float* running_mean2, float* running_var2, float momentum, float eps, int32_t act,
                                           const ymi_tensor* raw, const ymi_tensor* out, float* save_mean, float* save_invstd, void* workspace,
                                           size_t workspace_bytes, void* stream) {
    YMI_CHECK_ARG(split > 0 && split < cout && split % 4 == 0 && gamma2 && beta2, "conv2d_bn_silu_fwd_pair: split");
    return conv_bn_act_fwd_impl(x, w_packed, cout, kh, kw, stride, gamma, beta, running_mean, running_var,
                                BnParams2{gamma2, beta2, running_mean2, running_var2, (int)split}, momentum, eps, act, nullptr, raw, out, save_mean, save_invstd,
                                workspace, workspace_bytes, stream);
}
static int conv_bn_act_fwd_impl(const ymi_tensor* x, const void* w_packed, int64_t cout, int64_t kh, int64_t kw, int64_t stride,
                                const float* gamma, const float* beta, float* running_mean, float* running_var, BnParams2 p2, float momentum,
                                float eps, int32_t act, const ymi_tensor* residual, const ymi_tensor* raw, const ymi_tensor* out,
                                float* save_mean, float* save_invstd, void* workspace, size_t workspace_bytes, void* stream) {
    YMI_CHECK_ARG(ymi_tensor_ok(raw) && ymi_tensor_ok(out) && workspace, "conv2d_bn_silu_fwd: bad tensor");
    const int64_t m = ymi_pixels(raw);
    const int64_t maxblk = ymi_conv2d_stat_blocks(m, cout);
    const size_t need = (size_t)(maxblk * 2 * cout + 2 * cout) * sizeof(float);
    if (workspace_bytes < need) {
        ymi_set_error("conv2d_bn_silu_fwd: workspace %zu < %zu bytes", workspace_bytes, need);
        return YMI_EWORKSPACE;
    }
    float* scale = reinterpret_cast<float*>(workspace);
    float* shift = scale + cout;
    float* part = shift + cout;
    int64_t blocks = 0;
    int rc = ymi_conv2d_fwd(x, w_packed, cout, kh, kw, stride, nullptr, nullptr, YMI_ACT_NONE, nullptr, raw, part, &blocks, stream);
    if (rc) return rc;
    rc = bn_finalize_impl(part, blocks, m, cout, gamma, beta, running_mean, running_var, momentum, eps, scale, shift, save_mean, save_invstd, p2, stream);
    if (rc) return rc;
    return ymi_scale_shift_act(raw, scale, shift, act, residual, out, stream);
}

// The same Conv block with the BatchNorm statistics as fixed-point atomic sums and the finalize inside the affine pass: TWO launches
// (igemm.hip's statistics epilogue; scale_shift_act_fixed_kernel<.., FIN>).  stat_acc: [4][2][cout] int64, ZERO on entry (the caller zeroes
// its arena of them once per forward), garbage afterwards.  Shapes the fixed-group affine kernel does not take (cout not a multiple of 4 or
// > 1024, unaligned tensors) are refused: the caller keeps ymi_conv2d_bn_silu_fwd for them (ymi_conv2d_bn_silu_fwd_acc_ok).
int ymi_conv2d_fwd_statacc(const ymi_tensor* x, const void* w_packed, int64_t cout, int64_t kh, int64_t kw, int64_t stride, const ymi_tensor* y,
                           long long* stat_acc, void* stream);
extern "C" int ymi_conv2d_bn_silu_fwd_acc_ok(const ymi_tensor* raw, const ymi_tensor* out, const ymi_tensor* residual) {
    if (!ymi_tensor_ok(raw) || !ymi_tensor_ok(out) || !ymi_same_shape(raw, out) || raw->dtype != out->dtype) return 0;
    if (residual && (!ymi_tensor_ok(residual) || !ymi_same_shape(residual, out) || residual->dtype != out->dtype)) return 0;
    const bool vec = vec4_ok(raw) && vec4_ok(out) && (!residual || vec4_ok(residual));
    return vec && raw->c / 4 <= 256 && ymi_pixels(raw) * (raw->c / 4) < (1ll << 31);
}
extern "C" int ymi_conv2d_bn_silu_fwd_acc(const ymi_tensor* x, const void* w_packed, int64_t cout, int64_t kh, int64_t kw, int64_t stride,
                                          const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                                          float eps, int32_t act, const ymi_tensor* residual, const ymi_tensor* raw, const ymi_tensor* out,
                                          float* save_mean, float* save_invstd, void* stat_acc, void* stream) {
    YMI_CHECK_ARG(stat_acc && ((uintptr_t)stat_acc & 15) == 0 && ymi_conv2d_bn_silu_fwd_acc_ok(raw, out, residual) && raw->c == cout,
                  "conv2d_bn_silu_fwd_acc: shapes the fused finalize does not take (see ymi_conv2d_bn_silu_fwd_acc_ok)");
    YMI_CHECK_ARG((!gamma || ((uintptr_t)gamma & 15) == 0) && (!beta || ((uintptr_t)beta & 15) == 0) && (!running_mean || ((uintptr_t)running_mean & 15) == 0) &&
                      (!running_var || ((uintptr_t)running_var & 15) == 0) && (!save_mean || ((uintptr_t)save_mean & 15) == 0) &&
                      (!save_invstd || ((uintptr_t)save_invstd & 15) == 0),
                  "conv2d_bn_silu_fwd_acc: 16-byte aligned per-channel vectors");
    int rc = ymi_conv2d_fwd_statacc(x, w_packed, cout, kh, kw, stride, raw, (long long*)stat_acc, stream);
    if (rc) return rc;
    const BnFin fin{(const long long*)stat_acc, gamma, beta, running_mean, running_var, save_mean, save_invstd, (double)ymi_pixels(raw),
                    ldexp(1.0, -ymi_stat_fixed_point_shift(ymi_pixels(raw))), momentum, eps};
    TV r = residual ? tv(residual) : TV{nullptr, 0, 0, 0, 0, 0};
    if (raw->dtype == YMI_BF16) launch_ssa_fixed<bf16_t>(raw, nullptr, nullptr, act, r, out, (hipStream_t)stream, &fin);
    else launch_ssa_fixed<float>(raw, nullptr, nullptr, act, r, out, (hipStream_t)stream, &fin);
    YMI_CHECK_LAUNCH("conv2d_bn_silu_fwd_acc");
    return YMI_OK;
}
