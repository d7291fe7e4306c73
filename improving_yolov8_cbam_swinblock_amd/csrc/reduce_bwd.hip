// Per-channel reductions over NHWC tensors and the backward of act(BatchNorm_train(raw)).
//
// All reductions are two-stage and deterministic: stage 1 gives every workgroup a contiguous pixel
// range and a fixed channel group per thread (coalesced 8/16-byte reads along C), accumulates in f32
// registers and writes [block][2][C] partials; stage 2 sums the partials per channel in double.
#include <stdlib.h>

#include "common.h"

struct RV {
    const void* p;
    int64_t ld;
};

// FN: 0 colsum(a)            -> s0 = sum a
//     1 bn-act backward      -> dz = a * act'(z), z = (b-mean)*inv*gamma+beta ; s0 = sum dz, s1 = sum dz*xhat
//     2 layernorm param grads-> s0 = sum a (dbeta), s1 = sum a * (b - mean_row)*rstd_row (dgamma); mean/rstd per ROW
// gamma / beta of the channels >= split come from a second pair of arrays (two BatchNorms behind one convolution: Detect's sibling
// branches run as one, head.py:71-72); split == 0: one pair
struct GammaBeta2 {
    const float* gamma;
    const float* beta;
    int split;
};
// coefficients of the BN backward apply pass, from the finished sums (one set per channel):
//   z = x*a0 + a1 ;  draw = dy*act'(z)*c0 - x*c1 - c2
struct BnCoefArgs {
    const float* gamma;
    const float* beta;
    const float* mean;
    const float* inv;
    float inv_count;
    float* coef;  // [5][C] or nullptr
    GammaBeta2 g2;
};
__device__ __forceinline__ void bn_coef_write(const BnCoefArgs& bn, int C, int c, float s0, float s1) {
    const bool second = bn.g2.split > 0 && c >= bn.g2.split;
    const float* gp = second ? bn.g2.gamma : bn.gamma;
    const float* bp = second ? bn.g2.beta : bn.beta;
    const int pc = second ? c - bn.g2.split : c;
    const float ga = gp ? gp[pc] : 1.0f, be = bp ? bp[pc] : 0.0f;
    const float p0 = bn.inv[c], p1 = -bn.mean[c] * p0;
    const float k1 = s0 * bn.inv_count, k2 = s1 * bn.inv_count;
    bn.coef[0 * C + c] = p0 * ga;
    bn.coef[1 * C + c] = p1 * ga + be;
    bn.coef[2 * C + c] = ga * p0;
    bn.coef[3 * C + c] = ga * p0 * p0 * k2;
    bn.coef[4 * C + c] = ga * p0 * (k1 + p1 * k2);
}
// Tail of stage 1 (round 5): the workgroup that finishes LAST does stage 2 itself, so the separate final launch (a 1-16 workgroup kernel
// at a dependent-launch latency, ~57 times a step) disappears.  Two levels, both deterministic (fixed rows in fixed order, whoever runs
// them): the last workgroup of each of the 8 id classes (blockIdx.x & 7) sums that class's rows into xrows[class]; the last of those 8
// sums the 8 rows and writes the results / the apply pass's coefficients.  The hand-off is common.h's write-through form: partial rows
// and class rows leave by sc1 stores, one lane per workgroup draws a ticket, only the last arriver reads (sc1 loads).  Round 4 built this
// with __threadfence() in every workgroup: 17.6 -> 148 us per launch (profiles/r04_bn_tail_ticket_ab.txt).  tickets == nullptr: no tail.
struct ReduceTail {
    unsigned* tickets;  // ncls + 1 counters (the last at [32]), zero between launches
    int ncls;           // id classes: 32 or 8 (divides the grid)
    double* xrows;      // [ncls][2][C]
    float* out0;
    float* out1;
    BnCoefArgs bn;
};
#ifndef YMI_RED_U  // pixels a thread of the reduce pass keeps in flight per trip
#define YMI_RED_U 4
#endif
#ifndef YMI_APPLY_U
#define YMI_APPLY_U 4
#endif
template <typename T, int FN, int ACT>
__global__ void chan_reduce_kernel(RV a, RV b, int64_t P, int64_t span, int C, int TG, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   const float* __restrict__ mean, const float* __restrict__ inv, float* __restrict__ part, GammaBeta2 g2, ReduceTail tail) {
    extern __shared__ float red[];  // [rows][TG][8]
    const int rows = 256 / TG;
    const int tx = threadIdx.x % TG, ty = threadIdx.x / TG;
    const int c0 = blockIdx.y * 1024;  // channels beyond 1024 go to further grid rows
    const int groups = ((C - c0 < 1024) ? C - c0 : 1024) / 4;
    // a contiguous pixel range per workgroup, inside the eighth of the pixel order this workgroup's XCD owns (common.h: the kernel
    // that wrote `a` - a data-gradient GEMM, the loss - and the apply pass / GEMMs that follow use the same eighths).  gridDim.x is a
    // multiple of 8; the partial row of a workgroup stays blockIdx.x, and the final pass sums rows in index order: deterministic.
    const XcdRange xr = xcd_range(P, span);
    const int64_t per = (xr.hi - xr.lo + xr.nbx - 1) / xr.nbx;
    const int64_t p0 = (xr.lo + xr.bi * per < xr.hi) ? xr.lo + xr.bi * per : xr.hi, p1 = (p0 + per < xr.hi) ? p0 + per : xr.hi;
    const T* ap = reinterpret_cast<const T*>(a.p) + c0;
    const T* bp = reinterpret_cast<const T*>(b.p) + c0;
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    if (tx < groups) {
        float g[4], be[4], mu[4], iv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = c0 + tx * 4 + r;
            const bool second = g2.split > 0 && c >= g2.split;
            const float* gp = second ? g2.gamma : gamma;
            const float* bp = second ? g2.beta : beta;
            const int pc = second ? c - g2.split : c;
            g[r] = (FN == 1 && gp) ? gp[pc] : 1.0f;
            be[r] = (FN == 1 && bp) ? bp[pc] : 0.0f;
            mu[r] = (FN == 1) ? mean[c] : 0.0f;
            iv[r] = (FN == 1) ? inv[c] : 1.0f;
        }
        auto accum = [&](const float (&va)[4], const float (&vb)[4], int64_t p) {
            if (FN == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) s0[r] += va[r];
            } else if (FN == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float xh = (vb[r] - mu[r]) * iv[r];
                    const float dz = va[r] * act_grad<ACT>(xh * g[r] + be[r]);
                    s0[r] += dz;
                    s1[r] += dz * xh;
                }
            } else {
                const float m = mean[p], rs = inv[p];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s0[r] += va[r];
                    s1[r] += va[r] * (vb[r] - m) * rs;
                }
            }
        };
        int64_t p = p0 + ty;
        // 4 pixels per trip: 8 independent 8/16-byte loads in flight per lane before the first use; the NEXT trip's loads are issued
        // (raw, unconverted: 2 VGPRs each in bf16) before this trip's arithmetic, so a workgroup with several trips does not pay one
        // full memory round trip per trip (4 waves per SIMD are not enough to hide it: the arithmetic of a trip is as long as its loads)
        typedef typename Raw4<T>::type R4;
        constexpr int U = YMI_RED_U;  // pixels per trip
        R4 ra[U], rb[U];
        bool have = p + (U - 1) * (int64_t)rows < p1;
        if (have) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                ra[u] = *reinterpret_cast<const R4*>(ap + (p + u * (int64_t)rows) * a.ld + tx * 4);
                if (FN != 0) rb[u] = *reinterpret_cast<const R4*>(bp + (p + u * (int64_t)rows) * b.ld + tx * 4);
            }
        }
        while (have) {
            float va[U][4], vb[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                Raw4<T>::to_f32(ra[u], va[u]);
                if (FN != 0) Raw4<T>::to_f32(rb[u], vb[u]);
            }
            const int64_t pc = p;
            p += U * (int64_t)rows;
            have = p + (U - 1) * (int64_t)rows < p1;
            if (have) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    ra[u] = *reinterpret_cast<const R4*>(ap + (p + u * (int64_t)rows) * a.ld + tx * 4);
                    if (FN != 0) rb[u] = *reinterpret_cast<const R4*>(bp + (p + u * (int64_t)rows) * b.ld + tx * 4);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) accum(va[u], vb[u], pc + u * (int64_t)rows);
        }
        for (; p < p1; p += rows) {
            float va[4], vb[4];
            Pack<T, 4>::load(ap + p * a.ld + tx * 4, va);
            if (FN != 0) Pack<T, 4>::load(bp + p * b.ld + tx * 4, vb);
            accum(va, vb, p);
        }
    }
    float* my = red + ((size_t)ty * TG + tx) * 8;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        my[r] = s0[r];
        my[4 + r] = s1[r];
    }
    __syncthreads();
    if (ty == 0 && tx < groups) {
        for (int q = 1; q < rows; ++q) {
            const float* o = red + ((size_t)q * TG + tx) * 8;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s0[r] += o[r];
                s1[r] += o[4 + r];
            }
        }
        float* r0 = part + ((int64_t)blockIdx.x * 2 + 0) * C + c0 + tx * 4;
        float* r1 = part + ((int64_t)blockIdx.x * 2 + 1) * C + c0 + tx * 4;
        if (tail.tickets) {  // (kernel argument: uniform) write-through: the last workgroup of the id class reads these rows in this launch
            st_wt2(r0, s0[0], s0[1]);
            st_wt2(r0 + 2, s0[2], s0[3]);
            st_wt2(r1, s1[0], s1[1]);
            st_wt2(r1 + 2, s1[2], s1[3]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                r0[r] = s0[r];
                r1[r] = s1[r];
            }
        }
    }
    if (tail.tickets == nullptr) return;
    // ---- stage 2 by the last workgroups (host: one grid row, C <= 1024) -----------------------------------------------------------------
    // NC id classes (blockIdx.x % NC; 32 when the grid allows, else 8): a class has gridDim.x / NC rows - 16 for the usual 512-row grid -
    // so its last workgroup has ALL its loads in flight at once, and the last of the NC class leaders reads NC rows the same way
    int* flag = reinterpret_cast<int*>(red);
    const int NC = tail.ncls;
    const int xg = blockIdx.x % NC, nbx = gridDim.x / NC;
    if (!ticket_is_last(&tail.tickets[xg], (unsigned)nbx - 1u, flag)) return;
    for (int col = threadIdx.x * 2; col < 2 * C; col += 512) {  // rows xg, xg + NC, ... : [2][C] floats each, a lane sums two columns
        const float* src = part + (int64_t)xg * 2 * C + col;
        const int64_t rs = (int64_t)NC * 2 * C;  // NC rows on
        double a0 = 0.0, a1 = 0.0;
        int j = 0;
        for (; j + 15 < nbx; j += 16) {  // sixteen loads in flight
            float x[16], y[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) ld_wt2(src + (int64_t)(j + u) * rs, x[u], y[u]);
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                a0 += (double)x[u];
                a1 += (double)y[u];
            }
        }
        for (; j < nbx; ++j) {
            float x, y;
            ld_wt2(src + (int64_t)j * rs, x, y);
            a0 += (double)x;
            a1 += (double)y;
        }
        st_wt(&tail.xrows[(int64_t)xg * 2 * C + col], a0);
        st_wt(&tail.xrows[(int64_t)xg * 2 * C + col + 1], a1);
    }
    if (!ticket_is_last(&tail.tickets[32], (unsigned)NC - 1u, flag)) return;
    for (int c = threadIdx.x; c < C; c += 256) {
        double sa = 0.0, sb = 0.0;
        for (int x0 = 0; x0 < NC; x0 += 8) {
            double v0[8], v1[8];
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                v0[x] = ld_wt(&tail.xrows[((int64_t)(x0 + x) * 2 + 0) * C + c]);
                v1[x] = ld_wt(&tail.xrows[((int64_t)(x0 + x) * 2 + 1) * C + c]);
            }
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                sa += v0[x];
                sb += v1[x];
            }
        }
        if (tail.out0) tail.out0[c] = (float)sa;
        if (tail.out1) tail.out1[c] = (float)sb;
        if (tail.bn.coef) bn_coef_write(tail.bn, C, c, (float)sa, (float)sb);
    }
}

// stage 2: out0[c] = sum_b part[b][0][c], out1[c] = sum_b part[b][1][c]
// (Its own launch, ~70 a step at a dependent-launch latency each.  Round 4 measured the alternative - the LAST workgroup of stage 1, found
// with ticket counters behind __threadfence(), does stage 2 - parity-green and 12.85 -> 19.3 ms/step: a device-scope fence is an L2
// write-back + invalidate on this 8-XCD part and every one of up to 1024 workgroups pays it (17.6 -> 148 us per reduce launch);
// profiles/r04_bn_tail_ticket_ab.txt, code in git.)
// 32 channels x 32 row slices per block, 4 independent accumulator pairs per thread (loads in flight)
__global__ __launch_bounds__(1024) void chan_reduce_final_kernel(const float* __restrict__ part, int blocks, int C, float* __restrict__ out0, float* __restrict__ out1,
                                                                 BnCoefArgs bn) {
    __shared__ double red[2][32][33];
    const int cl = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double s0 = 0.0, s1 = 0.0;
    // the per-channel parameters of the coefficient pass are fetched FIRST, beside the partials (this kernel is a chain of load
    // latencies: behind the barrier they were a round trip of their own)
    float ga = 1.0f, be = 0.0f, p_inv = 0.f, p_mean = 0.f;
    if (slice == 0 && c < C && bn.coef) {
        const bool second = bn.g2.split > 0 && c >= bn.g2.split;
        const float* gp = second ? bn.g2.gamma : bn.gamma;
        const float* bp = second ? bn.g2.beta : bn.beta;
        const int pc = second ? c - bn.g2.split : c;
        ga = gp ? gp[pc] : 1.0f;
        be = bp ? bp[pc] : 0.0f;
        p_inv = bn.inv[c];
        p_mean = bn.mean[c];
    }
    if (c < C) {
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
        for (; b < blocks; b += 32) {
            a0 += part[((int64_t)b * 2 + 0) * C + c];
            a1 += part[((int64_t)b * 2 + 1) * C + c];
        }
        s0 = ((double)a0 + (double)b0) + ((double)c0 + (double)d0);
        s1 = ((double)a1 + (double)b1) + ((double)c1 + (double)d1);
    }
    red[0][slice][cl] = s0;
    red[1][slice][cl] = s1;
    __syncthreads();
    if (slice == 0 && c < C) {
        double a = 0.0, b2 = 0.0;
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            a += red[0][q][cl];
            b2 += red[1][q][cl];
        }
        if (out0) out0[c] = (float)a;
        if (out1) out1[c] = (float)b2;
        if (bn.coef) {
            const float p0 = p_inv, p1 = -p_mean * p0;
            const float k1 = (float)a * bn.inv_count, k2 = (float)b2 * bn.inv_count;
            bn.coef[0 * C + c] = p0 * ga;
            bn.coef[1 * C + c] = p1 * ga + be;
            bn.coef[2 * C + c] = ga * p0;
            bn.coef[3 * C + c] = ga * p0 * p0 * k2;
            bn.coef[4 * C + c] = ga * p0 * (k1 + p1 * k2);
        }
    }
}

// option bn_tail = 0: the final pass as its own launch (the "before" of profiles/r05_bn_tail_ab.txt)
static bool bn_tail_on() { return ymi_opt(OPT_BN_TAIL) != 0; }

static int pow2_ge(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// returns number of stage-1 blocks; partials must hold blocks*2*C floats.  A block's 256 threads cover 256 / (C/4) pixel rows at a
// time and walk their pixels four per trip; on the small maps that walk is a chain of dependent HBM round trips (a 13 MB tensor
// with 200 blocks of 64 pixels took 20 us at C = 512: 8 trips per thread), so a block gets one trip's worth of pixels
// (>= 16) until the block cap.  The cap is 512 = two workgroups per CU since round 4 (swept inside the step: 1024 was the isolated optimum
// and 0.17 ms/step slower there, 2048 slower still; 640 / 768 leave an uneven second round; profiles/r04_ew_grid_sweep.txt): half the
// partial rows for the final pass, and the pass shares the chip with nothing.
static int reduce_blocks(int64_t P, int C) {
    const int tg = pow2_ge(C / 4) > 256 ? 256 : pow2_ge(C / 4);
    int64_t ppb = (int64_t)(256 / tg) * 4;
    if (ppb < 16) ppb = 16;
    if (ppb > 64) ppb = 64;
    int64_t b = (P + ppb - 1) / ppb;
    const int rcap = ymi_opt(OPT_RED_CAP);
    if (b > rcap) b = rcap;
    if (b < 1) b = 1;
    return (int)((b + 7) / 8 * 8);  // the same number of workgroups on every XCD (chan_reduce_kernel walks XCD-owned pixel ranges)
}

template <int FN>
static int launch_chan_reduce(const ymi_tensor* a, const ymi_tensor* b, const float* gamma, const float* beta, const float* mean,
                              const float* inv, int act, float* part, int* blocks_out, hipStream_t stream, const char* what,
                              GammaBeta2 g2 = GammaBeta2{nullptr, nullptr, 0}, ReduceTail tail = ReduceTail{}) {
    const int64_t P = ymi_pixels(a);
    const int C = (int)a->c;
    YMI_CHECK_ARG(C % 4 == 0 && a->ld % 4 == 0 && (!b || (b->ld % 4 == 0)), "%s: channels must be a multiple of 4", what);
    const int TG = pow2_ge(C / 4) > 256 ? 256 : pow2_ge(C / 4);
    const int crows = (C + 1023) / 1024;
    const int blocks = reduce_blocks(P, C);
    YMI_CHECK_ARG(!tail.tickets || crows == 1, "%s: the in-kernel final pass covers at most 1024 channels", what);
    const size_t lds = (size_t)256 * 8 * sizeof(float);
    RV ra{a->data, a->ld}, rb{b ? b->data : nullptr, b ? b->ld : 0};
#define YMI_CR(T, A) hipLaunchKernelGGL((chan_reduce_kernel<T, FN, A>), dim3(blocks, crows), dim3(256), lds, stream, ra, rb, P, ymi_xcd_span_arg(P), C, TG, gamma, beta, mean, inv, part, g2, tail)
#define YMI_CR_T(T)                                                      \
    do {                                                                 \
        if (FN != 1 || act == YMI_ACT_NONE) YMI_CR(T, YMI_ACT_NONE);     \
        else if (act == YMI_ACT_SILU) YMI_CR(T, YMI_ACT_SILU);           \
        else YMI_CR(T, YMI_ACT_GELU);                                    \
    } while (0)
    if (a->dtype == YMI_BF16) YMI_CR_T(bf16_t);
    else YMI_CR_T(float);
#undef YMI_CR_T
#undef YMI_CR
    YMI_CHECK_LAUNCH(what);
    *blocks_out = blocks;
    return YMI_OK;
}

int ymi_chan_reduce_final(const float* part, int blocks, int C, float* out0, float* out1, hipStream_t stream) {
    hipLaunchKernelGGL(chan_reduce_final_kernel, dim3((C + 31) / 32), dim3(1024), 0, stream, part, blocks, C, out0, out1, BnCoefArgs{});  // (value-initialised: no coefficients)
    YMI_CHECK_LAUNCH("chan_reduce_final");
    return YMI_OK;
}

// final pass of a BatchNorm + activation backward whose first stage ran elsewhere (csrc/first_conv.hip): sums `blocks` partial rows
// [block][2][C] (dz, dz * xhat) into dbeta / dgamma and writes the apply pass's coefficients [a0 | a1 | c0 | c1 | c2][C]
int ymi_bn_bwd_final(const float* part, int blocks, int C, const float* gamma, const float* beta, const float* mean, const float* inv, float inv_count,
                     float* dgamma, float* dbeta, float* coef, hipStream_t stream) {
    hipLaunchKernelGGL(chan_reduce_final_kernel, dim3((C + 31) / 32), dim3(1024), 0, stream, part, blocks, C, dbeta, dgamma,
                       BnCoefArgs{gamma, beta, mean, inv, inv_count, coef, GammaBeta2{nullptr, nullptr, 0}});
    YMI_CHECK_LAUNCH("bn_bwd_final");
    return YMI_OK;
}

extern "C" int ymi_colsum(const ymi_tensor* x, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    YMI_CHECK_ARG(ymi_tensor_ok(x) && out && workspace, "colsum: args");
    const size_t need = (size_t)reduce_blocks(ymi_pixels(x), (int)x->c) * 2 * x->c * sizeof(float);
    if (workspace_bytes < need) {
        ymi_set_error("colsum: workspace %zu < %zu", workspace_bytes, need);
        return YMI_EWORKSPACE;
    }
    int blocks = 0;
    int rc = launch_chan_reduce<0>(x, nullptr, nullptr, nullptr, nullptr, nullptr, 0, (float*)workspace, &blocks, (hipStream_t)stream, "colsum");
    if (rc) return rc;
    return ymi_chan_reduce_final((const float*)workspace, blocks, (int)x->c, out, nullptr, (hipStream_t)stream);
}

// LayerNorm parameter gradients (used by swin.hip): dgamma = sum_rows dy*xhat, dbeta = sum_rows dy
int ymi_ln_param_grads(const ymi_tensor* dy, const ymi_tensor* x, const float* mean, const float* rstd, float* dgamma, float* dbeta,
                       void* workspace, size_t workspace_bytes, hipStream_t stream) {
    const size_t need = (size_t)reduce_blocks(ymi_pixels(dy), (int)dy->c) * 2 * dy->c * sizeof(float);
    if (workspace_bytes < need) {
        ymi_set_error("layernorm_bwd: workspace %zu < %zu", workspace_bytes, need);
        return YMI_EWORKSPACE;
    }
    int blocks = 0;
    int rc = launch_chan_reduce<2>(dy, x, nullptr, nullptr, mean, rstd, 0, (float*)workspace, &blocks, stream, "layernorm_bwd(param)");
    if (rc) return rc;
    return ymi_chan_reduce_final((const float*)workspace, blocks, (int)dy->c, dbeta, dgamma, stream);
}

// draw = gamma*inv*(dz - mean(dz) - xhat*mean(dz*xhat)), dz = dy*act'(z), written with the per-channel coefficients the
// final-reduce kernel prepared:  z = x*a0 + a1 ;  draw = dz*c0 - x*c1 - c2   (coef = [a0|a1|c0|c1|c2][C])
// FIXED: the number of G-channel groups divides 256, so a thread keeps one channel group and its coefficients in
// registers for all of its pixels.  G = 4 (8-byte bf16 / 16-byte f32 accesses) or 8 (bf16 only, 16-byte accesses).
template <typename T, int G, bool FIXED, int ACT>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(RV dout, RV raw, RV draw, int64_t Pall, int64_t span, int C, const float* __restrict__ coef) {
    const int groups = C / G;
    int64_t P = Pall;
    const T* dp = reinterpret_cast<const T*>(dout.p);
    const T* rp = reinterpret_cast<const T*>(raw.p);
    T* op = reinterpret_cast<T*>(const_cast<void*>(draw.p));
    if constexpr (FIXED) {
        const int g = threadIdx.x % groups, rows = 256 / groups;
        if ((int)threadIdx.x >= rows * groups) return;  // group counts that do not divide 256 (192 channels); no barriers in this kernel
        float a0[G], a1[G], c0[G], c1[G], c2[G];
#pragma unroll
        for (int r = 0; r < G; ++r) {
            const int c = g * G + r;
            a0[r] = coef[c];
            a1[r] = coef[C + c];
            c0[r] = coef[2 * C + c];
            c1[r] = coef[3 * C + c];
            c2[r] = coef[4 * C + c];
        }
        auto one = [&](const float (&d)[G], const float (&x)[G], int64_t p) {
            float o[G];
#pragma unroll
            for (int r = 0; r < G; ++r) {
                const float dz = d[r] * act_grad<ACT>(x[r] * a0[r] + a1[r]);
                o[r] = dz * c0[r] - (x[r] * c1[r] + c2[r]);
            }
            Pack<T, G>::store(op + p * draw.ld + g * G, o);
        };
        constexpr int U = G == 8 ? 2 : YMI_APPLY_U;  // pixels per trip: 2*U independent loads in flight per lane
        // the pixels of this workgroup's XCD (common.h, XCD ownership of the pixel axis; the grid is a multiple of 8)
        const XcdRange xr = xcd_range(Pall, span);
        P = xr.hi;
        const int64_t step = (int64_t)xr.nbx * rows;
        int64_t p = xr.lo + (int64_t)xr.bi * rows + threadIdx.x / groups;
        if constexpr (G == 4) {
            // as in chan_reduce_kernel: the next trip's (raw) loads are in flight while this trip's arithmetic runs
            typedef typename Raw4<T>::type R4;
            R4 rd[U], rx[U];
            bool have = p + (U - 1) * step < P;
            if (have) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    rd[u] = *reinterpret_cast<const R4*>(dp + (p + u * step) * dout.ld + g * G);
                    rx[u] = *reinterpret_cast<const R4*>(rp + (p + u * step) * raw.ld + g * G);
                }
            }
            while (have) {
                float d[U][G], x[U][G];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    Raw4<T>::to_f32(rd[u], d[u]);
                    Raw4<T>::to_f32(rx[u], x[u]);
                }
                const int64_t pc = p;
                p += U * step;
                have = p + (U - 1) * step < P;
                if (have) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        rd[u] = *reinterpret_cast<const R4*>(dp + (p + u * step) * dout.ld + g * G);
                        rx[u] = *reinterpret_cast<const R4*>(rp + (p + u * step) * raw.ld + g * G);
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) one(d[u], x[u], pc + u * step);
            }
        } else {
            for (; p + (U - 1) * step < P; p += U * step) {
                float d[U][G], x[U][G];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    Pack<T, G>::load(dp + (p + u * step) * dout.ld + g * G, d[u]);
                    Pack<T, G>::load(rp + (p + u * step) * raw.ld + g * G, x[u]);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) one(d[u], x[u], p + u * step);
            }
        }
        for (; p < P; p += step) {
            float d[G], x[G];
            Pack<T, G>::load(dp + p * dout.ld + g * G, d);
            Pack<T, G>::load(rp + p * raw.ld + g * G, x);
            one(d, x, p);
        }
    } else {
        const uint32_t total = (uint32_t)(P * groups), ugroups = (uint32_t)groups;  // host: P * groups < 2^31
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
            const uint32_t pu = i / ugroups;
            const int g = (int)(i - pu * ugroups);
            const int64_t p = pu;
            float d[G], x[G], o[G];
            Pack<T, G>::load(dp + p * dout.ld + g * G, d);
            Pack<T, G>::load(rp + p * raw.ld + g * G, x);
#pragma unroll
            for (int r = 0; r < G; ++r) {
                const int c = g * G + r;
                const float dz = d[r] * act_grad<ACT>(x[r] * coef[c] + coef[C + c]);
                o[r] = dz * coef[2 * C + c] - (x[r] * coef[3 * C + c] + coef[4 * C + c]);
            }
            Pack<T, G>::store(op + p * draw.ld + g * G, o);
        }
    }
}

// apply pass of the BN backward: draw = dout * act'(raw * a0 + a1) * c0 - raw * c1 - c2 with coef = [a0|a1|c0|c1|c2][C]
static int launch_bn_apply(const ymi_tensor* dout, const ymi_tensor* raw, const ymi_tensor* draw, int act, const float* coef, hipStream_t s) {
    const int64_t P = ymi_pixels(dout);
    const int C = (int)dout->c;
    const bool bf = dout->dtype == YMI_BF16;
    const int G = 4;  // 8-byte (bf16) / 16-byte (f32) channel groups (16-byte groups for bf16 measured equal in round 2: removed)
    const int groups = C / G;
    const bool fixed = groups <= 256;
    const int64_t total = P * groups;
    YMI_CHECK_ARG(total < (1ll << 31), "bn_act_bwd: tensor too large for 32-bit indexing");
    int64_t gb;
    if (fixed) {
        // every thread reloads its group's coefficients: give it at least ~8 pixels when the tensor allows
        const int rows = 256 / groups;
        gb = (P + (int64_t)rows * ew_ppt() - 1) / ((int64_t)rows * ew_ppt());
        if (gb < 1024) gb = (P + rows - 1) / rows < 1024 ? (P + rows - 1) / rows : 1024;  // small maps: latency, not bandwidth (see launch_ssa_fixed)
    } else {
        gb = (total + 255) / 256;
    }
    if (gb > ew_cap()) gb = ew_cap();
    gb = (gb + 7) / 8 * 8;
    const int64_t span = ymi_xcd_span_arg(P);
    RV a{dout->data, dout->ld}, b{raw->data, raw->ld}, o{draw->data, draw->ld};
#define YMI_BWD_APPLY(T, GG, F, A) hipLaunchKernelGGL((bn_act_bwd_apply_kernel<T, GG, F, A>), dim3((unsigned)gb), dim3(256), 0, s, a, b, o, P, span, C, coef)
#define YMI_BWD_APPLY_A(T, GG, F)                                          \
    do {                                                                   \
        if (act == YMI_ACT_SILU) YMI_BWD_APPLY(T, GG, F, YMI_ACT_SILU);      \
        else if (act == YMI_ACT_GELU) YMI_BWD_APPLY(T, GG, F, YMI_ACT_GELU); \
        else YMI_BWD_APPLY(T, GG, F, YMI_ACT_NONE);                         \
    } while (0)
    if (bf) {
        if (fixed) YMI_BWD_APPLY_A(bf16_t, 4, true);
        else YMI_BWD_APPLY_A(bf16_t, 4, false);
    } else {
        if (fixed) YMI_BWD_APPLY_A(float, 4, true);
        else YMI_BWD_APPLY_A(float, 4, false);
    }
#undef YMI_BWD_APPLY_A
#undef YMI_BWD_APPLY
    YMI_CHECK_LAUNCH("bn_act_bwd(apply)");
    return YMI_OK;
}

static int bn_act_bwd_impl(const ymi_tensor* dout, const ymi_tensor* raw, const float* gamma, const float* save_mean,
                           const float* save_invstd, const float* beta, GammaBeta2 g2, int32_t act, const ymi_tensor* draw, float* dgamma,
                           float* dbeta, void* workspace, size_t workspace_bytes, void* stream);
extern "C" int ymi_bn_act_bwd(const ymi_tensor* dout, const ymi_tensor* raw, const float* gamma, const float* save_mean,
                              const float* save_invstd, const float* beta, int32_t act, const ymi_tensor* draw, float* dgamma,
                              float* dbeta, void* workspace, size_t workspace_bytes, void* stream) {
    return bn_act_bwd_impl(dout, raw, gamma, save_mean, save_invstd, beta, GammaBeta2{nullptr, nullptr, 0}, act, draw, dgamma, dbeta, workspace, workspace_bytes, stream);
}
// backward of ymi_conv2d_bn_silu_fwd_pair's BatchNorm + activation: channels >= split use gamma2 / beta2; dgamma / dbeta hold all channels
extern "C" int ymi_bn_act_bwd_pair(const ymi_tensor* dout, const ymi_tensor* raw, const float* gamma, const float* beta, const float* gamma2,
                                   const float* beta2, int64_t split, const float* save_mean, const float* save_invstd, int32_t act,
                                   const ymi_tensor* draw, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, void* stream) {
    YMI_CHECK_ARG(dout && split > 0 && split < dout->c && split % 4 == 0 && gamma2 && beta2, "bn_act_bwd_pair: split");
    return bn_act_bwd_impl(dout, raw, gamma, save_mean, save_invstd, beta, GammaBeta2{gamma2, beta2, (int)split}, act, draw, dgamma, dbeta, workspace, workspace_bytes, stream);
}
static int bn_act_bwd_impl(const ymi_tensor* dout, const ymi_tensor* raw, const float* gamma, const float* save_mean,
                           const float* save_invstd, const float* beta, GammaBeta2 g2, int32_t act, const ymi_tensor* draw, float* dgamma,
                           float* dbeta, void* workspace, size_t workspace_bytes, void* stream) {
    YMI_CHECK_ARG(ymi_tensor_ok(dout) && ymi_tensor_ok(raw) && ymi_tensor_ok(draw) && ymi_same_shape(dout, raw) && ymi_same_shape(dout, draw),
                  "bn_act_bwd: shapes");
    YMI_CHECK_ARG(dout->dtype == raw->dtype && raw->dtype == draw->dtype, "bn_act_bwd: dtypes");
    YMI_CHECK_ARG(save_mean && save_invstd && dgamma && dbeta && workspace, "bn_act_bwd: null argument");
    YMI_CHECK_ARG(draw->ld % 4 == 0, "bn_act_bwd: draw ld");
    const int64_t P = ymi_pixels(dout);
    const int C = (int)dout->c;
    const int rblocks = reduce_blocks(P, C);
    const size_t need = ((size_t)rblocks * 2 * C + 6 * (size_t)C) * sizeof(float) + 64 * (size_t)C * sizeof(double);
    if (workspace_bytes < need) {
        ymi_set_error("bn_act_bwd: workspace %zu < %zu", workspace_bytes, need);
        return YMI_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    int blocks = 0;
    float* coef = (float*)workspace + (size_t)rblocks * 2 * C;
    double* xrows = reinterpret_cast<double*>(coef + 6 * (size_t)C);  // (8-byte aligned: the workspace is, and every term is a multiple of 8 bytes)
    if (bn_tail_on() && C <= 1024) {
        // dbeta = sum dz, dgamma = sum dz*xhat and the apply pass's coefficients from the reduce kernel's last workgroups
        unsigned* tickets = ymi_ticket_slot();
        YMI_CHECK_ARG(tickets, "bn_act_bwd: ticket counters");
        const BnCoefArgs bn{gamma, beta, save_mean, save_invstd, 1.0f / (float)P, coef, g2};
        int rc = launch_chan_reduce<1>(dout, raw, gamma, beta, save_mean, save_invstd, act, (float*)workspace, &blocks, s, "bn_act_bwd(reduce)", g2,
                                       ReduceTail{tickets, rblocks % 32 == 0 ? 32 : 8, xrows, dbeta, dgamma, bn});
        if (rc) return rc;
        return launch_bn_apply(dout, raw, draw, act, coef, s);
    }
    int rc = launch_chan_reduce<1>(dout, raw, gamma, beta, save_mean, save_invstd, act, (float*)workspace, &blocks, s, "bn_act_bwd(reduce)", g2);
    if (rc) return rc;
    {   // the final pass rides in the weight-gradient launch held back for it, when there is one (common.h: YmiBnRider)
        const YmiBnRider rider{(const float*)workspace, blocks, C, 0, dbeta, dgamma, gamma, beta, save_mean, save_invstd, g2.gamma, g2.beta, g2.split, 1.0f / (float)P, coef};
        if (C <= 1024 && ymi_wgrad_issue_held(&rider, s)) {
            YMI_CHECK_LAUNCH("bn_act_bwd(final, riding)");
            return launch_bn_apply(dout, raw, draw, act, coef, s);
        }
    }
    // dbeta = sum dz, dgamma = sum dz*xhat, and the apply pass's coefficients
    hipLaunchKernelGGL(chan_reduce_final_kernel, dim3((C + 31) / 32), dim3(1024), 0, s, (const float*)workspace, blocks, C, dbeta, dgamma,
                       BnCoefArgs{gamma, beta, save_mean, save_invstd, 1.0f / (float)P, coef, g2});
    YMI_CHECK_LAUNCH("bn_act_bwd(final)");
    return launch_bn_apply(dout, raw, draw, act, coef, s);
}

// dx = dy * gelu'(pre)
template <typename T>
__global__ void gelu_bwd_kernel(RV pre, RV dy, RV dx, int64_t P, int C) {
    const int groups = C / 4;
    const uint32_t total = (uint32_t)(P * groups), ugroups = (uint32_t)groups;  // host: P * groups < 2^31
    const T* pp = reinterpret_cast<const T*>(pre.p);
    const T* dp = reinterpret_cast<const T*>(dy.p);
    T* op = reinterpret_cast<T*>(const_cast<void*>(dx.p));
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint32_t pu = i / ugroups;
        const int g = (int)(i - pu * ugroups);
        const int64_t p = pu;
        float a[4], d[4];
        Pack<T, 4>::load(pp + p * pre.ld + g * 4, a);
        Pack<T, 4>::load(dp + p * dy.ld + g * 4, d);
#pragma unroll
        for (int r = 0; r < 4; ++r) d[r] *= gelu_grad_f(a[r]);
        Pack<T, 4>::store(op + p * dx.ld + g * 4, d);
    }
}

extern "C" int ymi_gelu_bwd(const ymi_tensor* pre, const ymi_tensor* dy, const ymi_tensor* dx, void* stream) {
    YMI_CHECK_ARG(ymi_tensor_ok(pre) && ymi_tensor_ok(dy) && ymi_tensor_ok(dx) && ymi_same_shape(pre, dy) && ymi_same_shape(pre, dx), "gelu_bwd: shapes");
    YMI_CHECK_ARG(pre->dtype == dy->dtype && dy->dtype == dx->dtype && pre->c % 4 == 0 && pre->ld % 4 == 0 && dy->ld % 4 == 0 && dx->ld % 4 == 0, "gelu_bwd: dtype/alignment");
    const int64_t P = ymi_pixels(pre), total = P * (pre->c / 4);
    YMI_CHECK_ARG(total < (1ll << 31), "gelu_bwd: tensor too large for 32-bit indexing");
    int64_t gb = (total + 255) / 256;
    if (gb > 4096) gb = 4096;
    RV a{pre->data, pre->ld}, b{dy->data, dy->ld}, o{dx->data, dx->ld};
    if (pre->dtype == YMI_BF16)
        hipLaunchKernelGGL(gelu_bwd_kernel<bf16_t>, dim3((unsigned)gb), dim3(256), 0, (hipStream_t)stream, a, b, o, P, (int)pre->c);
    else
        hipLaunchKernelGGL(gelu_bwd_kernel<float>, dim3((unsigned)gb), dim3(256), 0, (hipStream_t)stream, a, b, o, P, (int)pre->c);
    YMI_CHECK_LAUNCH("gelu_bwd");
    return YMI_OK;
}
