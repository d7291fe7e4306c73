// CBAM (fork's nn/modules/cbam.py) for NHWC tensors: memory-bound wavefront kernels.
//
//   ca  = sigmoid(W2 relu(W1 avg_p x) + W2 relu(W1 max_p x))          [N][C]
//   x1  = x * ca
//   sm  = [mean_c x1, max_c x1]                                        [N][H][W][2]
//   sa  = sigmoid(conv_kxk(sm))                                        [N][H][W]
//   out = x1 * sa
//
// Forward = 4 launches: pool (x read once, coalesced along C), MLP (one workgroup per image, weights
// in registers/LDS), spatial statistics (one wave per pixel: 64 lanes x 16 B = 512 bf16 channels per
// load, wave-reduce), apply (k*k*2-tap stencil from the tiny f32 map + final scale).  x is streamed
// 3 times (pool, stats, apply) and out written once; the second and third reads hit L2 / Infinity
// Cache for the shapes of this model (26 MB at bs 32).
#include "common.h"

struct CV {
    const void* p;
    int64_t ld;
};

// ---------------------------------------------------------------------------------- forward
// grid (ceil(C/CB), N); block 256 = (CB/4 channel groups) x pixel lanes
template <typename T>
__global__ __launch_bounds__(256) void cbam_pool_kernel(CV x, int HW, int C, float* __restrict__ pooled, int* __restrict__ amax) {
    constexpr int CB = 64;            // channels per block
    constexpr int GX = CB / 4;        // 16 channel groups of 4
    constexpr int PY = 256 / GX;      // 16 pixel lanes
    __shared__ float s_sum[PY][CB];
    __shared__ float s_max[PY][CB];
    __shared__ int s_idx[PY][CB];
    const int gx = threadIdx.x % GX, py = threadIdx.x / GX;
    const int n = blockIdx.y, c = blockIdx.x * CB + gx * 4;
    const T* xp = reinterpret_cast<const T*>(x.p) + (int64_t)n * HW * x.ld;
    float sum[4] = {0.f, 0.f, 0.f, 0.f}, mx[4];
    int ix[4] = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 4; ++r) mx[r] = -__builtin_inff();
    if (c < C) {
        int p = py;
        for (; p + 3 * PY < HW; p += 4 * PY) {  // four pixels in flight per lane (the loop is a chain of load latencies otherwise)
            float v[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) Pack<T, 4>::load(xp + (int64_t)(p + u * PY) * x.ld + c, v[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sum[r] += v[u][r];
                    if (v[u][r] > mx[r]) { mx[r] = v[u][r]; ix[r] = p + u * PY; }  // strictly greater: first maximum in scan order
                }
        }
        for (; p < HW; p += PY) {
            float v[4];
            Pack<T, 4>::load(xp + (int64_t)p * x.ld + c, v);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sum[r] += v[r];
                if (v[r] > mx[r]) { mx[r] = v[r]; ix[r] = p; }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        s_sum[py][gx * 4 + r] = sum[r];
        s_max[py][gx * 4 + r] = mx[r];
        s_idx[py][gx * 4 + r] = ix[r];
    }
    __syncthreads();
    if (threadIdx.x < CB) {
        const int cc = blockIdx.x * CB + threadIdx.x;
        if (cc < C) {
            float s = 0.f, m = -__builtin_inff();
            int bi = 0;
            for (int q = 0; q < PY; ++q) {
                s += s_sum[q][threadIdx.x];
                const float v = s_max[q][threadIdx.x];
                const int i2 = s_idx[q][threadIdx.x];
                if (v > m || (v == m && i2 < bi)) { m = v; bi = i2; }
            }
            pooled[((int64_t)n * 2 + 0) * C + cc] = s / (float)HW;
            pooled[((int64_t)n * 2 + 1) * C + cc] = m;
            amax[(int64_t)n * C + cc] = bi;
        }
    }
}

// one block (any multiple of 64 threads; launched with 1024) per image: hidden = relu(W1 avg) + relu(W1 max); ca = sigmoid(W2 hidden).
// Pure latency: three dependent rounds of loads; every round's loads are issued together (unrolled), one hidden unit or two per wave.
__global__ __launch_bounds__(1024) void cbam_mlp_kernel(const float* __restrict__ pooled, const float* __restrict__ w1, const float* __restrict__ w2,
                                                        int C, int Hd, float* __restrict__ ca) {
    extern __shared__ float sh[];  // [2][C] pooled, [Hd] hidden sum
    float* sp = sh;
    float* hs = sh + 2 * C;
    const int n = blockIdx.x, nt = blockDim.x, nw = nt >> 6;
    for (int i = threadIdx.x; i < 2 * C; i += nt) sp[i] = pooled[(int64_t)n * 2 * C + i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int h = wave; h < Hd; h += nw) {
        float a = 0.f, m = 0.f;
        const float* wr = w1 + (int64_t)h * C;
#pragma unroll 8
        for (int c = lane; c < C; c += 64) {
            const float w = wr[c];
            a += w * sp[c];
            m += w * sp[C + c];
        }
        a = wave_sum(a);
        m = wave_sum(m);
        if (lane == 0) hs[h] = fmaxf(a, 0.f) + fmaxf(m, 0.f);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += nt) {
        float z = 0.f;
        const float* wr = w2 + (int64_t)c * Hd;
#pragma unroll 8
        for (int h = 0; h < Hd; ++h) z += wr[h] * hs[h];
        ca[(int64_t)n * C + c] = sigmoidf_(z);
    }
}

// one wave per pixel: mean / max (+ first arg-max channel) over C of x*ca
template <typename T>
__global__ __launch_bounds__(256) void cbam_stats_kernel(CV x, int64_t NP, int HW, int C, const float* __restrict__ ca, float* __restrict__ smap,
                                                         int* __restrict__ sarg) {
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= NP) return;
    const int n = (int)(p / HW);
    const T* xp = reinterpret_cast<const T*>(x.p) + p * x.ld;
    const float* cap = ca + (int64_t)n * C;
    float s = 0.f, m = -__builtin_inff();
    int mi = 0x7fffffff;
    for (int c = lane * 4; c < C; c += 256) {
        float v[4];
        Pack<T, 4>::load(xp + c, v);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float t = v[r] * cap[c + r];
            s += t;
            if (t > m) { m = t; mi = c + r; }
        }
    }
    s = wave_sum(s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(m, o, 64);
        const int oi = __shfl_xor(mi, o, 64);
        if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
    }
    if (lane == 0) {
        smap[p * 2 + 0] = s / (float)C;
        smap[p * 2 + 1] = m;
        sarg[p] = mi;
    }
}

// one wave per pixel: sa = sigmoid(stencil), out = x*ca*sa
template <typename T>
__global__ __launch_bounds__(256) void cbam_apply_kernel(CV x, CV out, int N, int H, int W, int C, const float* __restrict__ ca,
                                                         const float* __restrict__ smap, const float* __restrict__ wsa, int K,
                                                         float* __restrict__ sa_out) {
    const int lane = threadIdx.x & 63;
    const int64_t NP = (int64_t)N * H * W;
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= NP) return;
    const int w = (int)(p % W), h = (int)((p / W) % H), n = (int)(p / ((int64_t)W * H));
    const int R = K / 2;
    float u = 0.f;
    for (int t = lane; t < K * K; t += 64) {
        const int hh = h + t / K - R, ww = w + t % K - R;
        if (hh >= 0 && hh < H && ww >= 0 && ww < W) {
            const int64_t q = ((int64_t)n * H + hh) * W + ww;
            u += wsa[t] * smap[q * 2 + 0] + wsa[K * K + t] * smap[q * 2 + 1];
        }
    }
    u = wave_sum(u);
    const float sa = sigmoidf_(u);
    if (lane == 0) sa_out[p] = sa;
    const T* xp = reinterpret_cast<const T*>(x.p) + p * x.ld;
    T* op = reinterpret_cast<T*>(const_cast<void*>(out.p)) + p * out.ld;
    const float* cap = ca + (int64_t)n * C;
    for (int c = lane * 4; c < C; c += 256) {
        float v[4];
        Pack<T, 4>::load(xp + c, v);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] * cap[c + r] * sa;
        Pack<T, 4>::store(op + c, v);
    }
}

extern "C" int ymi_cbam_fwd(const ymi_tensor* x, const float* w1, const float* w2, int64_t hidden, const float* wsa, int64_t ksa,
                            const ymi_tensor* out, float* ca, float* pooled, int32_t* pool_argmax, float* smap, int32_t* smap_argmax,
                            float* sa, void* stream) {
    YMI_CHECK_ARG(ymi_tensor_ok(x) && ymi_tensor_ok(out) && ymi_same_shape(x, out) && x->dtype == out->dtype, "cbam_fwd: tensors");
    YMI_CHECK_ARG(w1 && w2 && wsa && ca && pooled && pool_argmax && smap && smap_argmax && sa, "cbam_fwd: null buffer");
    YMI_CHECK_ARG(x->c % 4 == 0 && x->ld % 4 == 0 && out->ld % 4 == 0, "cbam_fwd: channels must be a multiple of 4");
    YMI_CHECK_ARG(hidden >= 1 && hidden <= 1024 && (ksa == 3 || ksa == 7), "cbam_fwd: hidden/kernel");
    hipStream_t s = (hipStream_t)stream;
    const int N = (int)x->n, H = (int)x->h, W = (int)x->w, C = (int)x->c, HW = H * W;
    const int64_t NP = (int64_t)N * HW;
    CV xv{x->data, x->ld}, ov{out->data, out->ld};
    dim3 gp((C + 63) / 64, N), gw((unsigned)((NP + 3) / 4));
    const size_t mlp_lds = (size_t)(2 * C + hidden) * sizeof(float);
    if (x->dtype == YMI_BF16) {
        hipLaunchKernelGGL(cbam_pool_kernel<bf16_t>, gp, dim3(256), 0, s, xv, HW, C, pooled, pool_argmax);
        hipLaunchKernelGGL(cbam_mlp_kernel, dim3(N), dim3(1024), mlp_lds, s, pooled, w1, w2, C, (int)hidden, ca);
        hipLaunchKernelGGL(cbam_stats_kernel<bf16_t>, gw, dim3(256), 0, s, xv, NP, HW, C, ca, smap, smap_argmax);
        hipLaunchKernelGGL(cbam_apply_kernel<bf16_t>, gw, dim3(256), 0, s, xv, ov, N, H, W, C, ca, smap, wsa, (int)ksa, sa);
    } else {
        hipLaunchKernelGGL(cbam_pool_kernel<float>, gp, dim3(256), 0, s, xv, HW, C, pooled, pool_argmax);
        hipLaunchKernelGGL(cbam_mlp_kernel, dim3(N), dim3(1024), mlp_lds, s, pooled, w1, w2, C, (int)hidden, ca);
        hipLaunchKernelGGL(cbam_stats_kernel<float>, gw, dim3(256), 0, s, xv, NP, HW, C, ca, smap, smap_argmax);
        hipLaunchKernelGGL(cbam_apply_kernel<float>, gw, dim3(256), 0, s, xv, ov, N, H, W, C, ca, smap, wsa, (int)ksa, sa);
    }
    YMI_CHECK_LAUNCH("cbam_fwd");
    return YMI_OK;
}

// --------------------------------------------------------------------------------- backward
// B1: du[p] = (sum_c dout*x*ca) * sa*(1-sa)        (one wave per pixel)
template <typename T>
__global__ __launch_bounds__(256) void cbam_bwd_du_kernel(CV x, CV dout, int64_t NP, int HW, int C, const float* __restrict__ ca,
                                                          const float* __restrict__ sa, float* __restrict__ du) {
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= NP) return;
    const int n = (int)(p / HW);
    const T* xp = reinterpret_cast<const T*>(x.p) + p * x.ld;
    const T* dp = reinterpret_cast<const T*>(dout.p) + p * dout.ld;
    const float* cap = ca + (int64_t)n * C;
    float s = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        float v[4], d[4];
        Pack<T, 4>::load(xp + c, v);
        Pack<T, 4>::load(dp + c, d);
#pragma unroll
        for (int r = 0; r < 4; ++r) s += d[r] * v[r] * cap[c + r];
    }
    s = wave_sum(s);
    if (lane == 0) {
        const float a = sa[p];
        du[p] = s * a * (1.f - a);
    }
}

// B2: dsm[p][j] = sum_t du[p - off_t] * wsa[j][t]   (adjoint stencil), then
//     dx1 = dout*sa + dsm0/C + [c==argmax]*dsm1 ;  dx = dx1*ca ;  dca partial[n][pixel-block][c] += dx1*x
// one wave per pixel; the per-image reduction of dx1*x is done by B3 from `dcap` partials written per pixel group.
template <typename T>
__global__ __launch_bounds__(256) void cbam_bwd_dx_kernel(CV x, CV dout, CV dx, int N, int H, int W, int C, const float* __restrict__ ca,
                                                          const float* __restrict__ sa, const float* __restrict__ du,
                                                          const int* __restrict__ sarg, const float* __restrict__ wsa, int K,
                                                          float* __restrict__ dcap /*[N][PB][C]*/, int PB) {
    __shared__ float part[4][1024 + 4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int HW = H * W;
    // blockIdx.x = n * PB + pb ; each block walks pixels pb*4+wv, +4*PB, ... of image n (wave per pixel)
    const int n = blockIdx.x / PB, pb = blockIdx.x % PB;
    const int R = K / 2;
    const float* cap = ca + (int64_t)n * C;
    // per-wave accumulation of dx1*x over its pixels, channel c = lane*4 + 256*i + r  (C <= 1024 -> i < 4)
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int q = pb * 4 + wv; q < HW; q += 4 * PB) {
        const int h = q / W, w = q % W;
        const int64_t p = (int64_t)n * HW + q;
        float d0 = 0.f, d1 = 0.f;
        for (int t = lane; t < K * K; t += 64) {
            // forward: u[p'] += wsa[j][t] * sm[p' + (ty-R, tx-R)]  =>  dsm[p] gets du[p - off] * wsa[j][t]
            const int hh = h - (t / K - R), ww = w - (t % K - R);
            if (hh >= 0 && hh < H && ww >= 0 && ww < W) {
                const float g = du[((int64_t)n * H + hh) * W + ww];
                d0 += g * wsa[t];
                d1 += g * wsa[K * K + t];
            }
        }
        d0 = wave_sum(d0);
        d1 = wave_sum(d1);
        const float a = sa[p];
        const int am = sarg[p];
        const T* xp = reinterpret_cast<const T*>(x.p) + p * x.ld;
        const T* dp = reinterpret_cast<const T*>(dout.p) + p * dout.ld;
        T* op = reinterpret_cast<T*>(const_cast<void*>(dx.p)) + p * dx.ld;
        const float dmean = d0 / (float)C;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = lane * 4 + 256 * i;
            if (c < C) {
                float v[4], d[4], o[4];
                Pack<T, 4>::load(xp + c, v);
                Pack<T, 4>::load(dp + c, d);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dx1 = d[r] * a + dmean + ((c + r == am) ? d1 : 0.f);
                    o[r] = dx1 * cap[c + r];
                    acc[i][r] += dx1 * v[r];
                }
                Pack<T, 4>::store(op + c, o);
            }
        }
    }
    // combine the 4 waves of the block, write one partial row per block
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = lane * 4 + 256 * i + r;
            if (c < 1024) part[wv][c] = acc[i][r];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256)
        dcap[((int64_t)n * PB + pb) * C + c] = part[0][c] + part[1][c] + part[2][c] + part[3][c];
}

// B3: per image: dca -> dz -> hidden grads -> d_avg, d_max ; stores dz[N][C], hsum[N][Hd], dpre[N][2][Hd]
__global__ __launch_bounds__(1024) void cbam_bwd_mlp_kernel(const float* __restrict__ dcap, int PB, const float* __restrict__ pooled,
                                                            const float* __restrict__ ca, const float* __restrict__ w1,
                                                            const float* __restrict__ w2, int C, int Hd, float* __restrict__ dz,
                                                            float* __restrict__ hsum, float* __restrict__ dpre, float* __restrict__ dpool) {
    extern __shared__ float sh[];  // [2C] pooled | [C] dz | [2Hd] pre (avg,max) | [Hd] dh
    float* sp = sh;
    float* sdz = sh + 2 * C;
    float* pre = sdz + C;
    float* dh = pre + 2 * Hd;
    const int n = blockIdx.x, nt = blockDim.x, nw = nt >> 6;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 2 * C; i += nt) sp[i] = pooled[(int64_t)n * 2 * C + i];
    for (int c = threadIdx.x; c < C; c += nt) {
        float s = 0.f;
#pragma unroll 8
        for (int b = 0; b < PB; ++b) s += dcap[((int64_t)n * PB + b) * C + c];
        const float a = ca[(int64_t)n * C + c];
        const float g = s * a * (1.f - a);
        sdz[c] = g;
        dz[(int64_t)n * C + c] = g;
    }
    __syncthreads();
    for (int h = wave; h < Hd; h += nw) {
        float a = 0.f, m = 0.f, g = 0.f;
#pragma unroll 8
        for (int c = lane; c < C; c += 64) {
            const float w = w1[(int64_t)h * C + c];
            a += w * sp[c];
            m += w * sp[C + c];
            g += w2[(int64_t)c * Hd + h] * sdz[c];
        }
        a = wave_sum(a); m = wave_sum(m); g = wave_sum(g);
        if (lane == 0) {
            pre[h] = a; pre[Hd + h] = m; dh[h] = g;
            hsum[(int64_t)n * Hd + h] = fmaxf(a, 0.f) + fmaxf(m, 0.f);
            dpre[((int64_t)n * 2 + 0) * Hd + h] = a > 0.f ? g : 0.f;
            dpre[((int64_t)n * 2 + 1) * Hd + h] = m > 0.f ? g : 0.f;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += nt) {
        float da = 0.f, dm = 0.f;
#pragma unroll 8
        for (int h = 0; h < Hd; ++h) {
            const float w = w1[(int64_t)h * C + c];
            da += (pre[h] > 0.f ? dh[h] : 0.f) * w;
            dm += (pre[Hd + h] > 0.f ? dh[h] : 0.f) * w;
        }
        dpool[((int64_t)n * 2 + 0) * C + c] = da;
        dpool[((int64_t)n * 2 + 1) * C + c] = dm;
    }
}

// B4: weight gradients (sum over images, fixed order): dw2[c][h] = sum_n dz[n,c]*hsum[n,h];
//     dw1[h][c] = sum_n dpre[n,0,h]*avg[n,c] + dpre[n,1,h]*max[n,c]
__global__ void cbam_bwd_w_kernel(const float* __restrict__ dz, const float* __restrict__ hsum, const float* __restrict__ dpre,
                                  const float* __restrict__ pooled, int N, int C, int Hd, float* __restrict__ dw1, float* __restrict__ dw2) {
    const int total = C * Hd;
    for (int i2 = blockIdx.x * blockDim.x + threadIdx.x; i2 < 2 * total; i2 += gridDim.x * blockDim.x) {
        const int i = i2 < total ? i2 : i2 - total;
        if (i2 < total) {
            const int c = i / Hd, h = i % Hd;  // dw2 [C][Hd]
            float s = 0.f;
#pragma unroll 8
            for (int n = 0; n < N; ++n) s += dz[(int64_t)n * C + c] * hsum[(int64_t)n * Hd + h];
            dw2[i] = s;
        } else {
            const int h = i / C, c = i % C;  // dw1 [Hd][C]
            float s = 0.f;
#pragma unroll 8
            for (int n = 0; n < N; ++n)
                s += dpre[((int64_t)n * 2 + 0) * Hd + h] * pooled[((int64_t)n * 2 + 0) * C + c] +
                     dpre[((int64_t)n * 2 + 1) * Hd + h] * pooled[((int64_t)n * 2 + 1) * C + c];
            dw1[i] = s;
        }
    }
}

// B5: dwsa[j][t] = sum_{n,p} du[p] * sm[p + off_t][j]   (one block per (j,t))
__global__ __launch_bounds__(1024) void cbam_bwd_wsa_kernel(const float* __restrict__ du, const float* __restrict__ smap, int N, int H, int W, int K,
                                                            float* __restrict__ dwsa) {
    __shared__ float red[16];
    const int t = blockIdx.x % (K * K), j = blockIdx.x / (K * K);
    const int R = K / 2, dy = t / K - R, dx = t % K - R;
    const uint32_t NP = (uint32_t)N * H * W, uw = (uint32_t)W, uh = (uint32_t)H;  // host: N*H*W < 2^31
    const int shift = dy * W + dx;
    float s0 = 0.f, s1 = 0.f;  // two chains: even / odd trips (fixed order: deterministic)
    uint32_t p = threadIdx.x;
    for (; p + blockDim.x < NP; p += 2 * blockDim.x) {
        const uint32_t q = p + blockDim.x;
        const uint32_t r0 = p / uw, r1 = q / uw;
        const int w0 = (int)(p - r0 * uw), h0 = (int)(r0 % uh), w1 = (int)(q - r1 * uw), h1 = (int)(r1 % uh);
        const bool ok0 = h0 + dy >= 0 && h0 + dy < H && w0 + dx >= 0 && w0 + dx < W;
        const bool ok1 = h1 + dy >= 0 && h1 + dy < H && w1 + dx >= 0 && w1 + dx < W;
        const float a0 = ok0 ? du[p] * smap[((int64_t)p + shift) * 2 + j] : 0.f;
        const float a1 = ok1 ? du[q] * smap[((int64_t)q + shift) * 2 + j] : 0.f;
        s0 += a0;
        s1 += a1;
    }
    if (p < NP) {
        const uint32_t r0 = p / uw;
        const int w0 = (int)(p - r0 * uw), h0 = (int)(r0 % uh);
        if (h0 + dy >= 0 && h0 + dy < H && w0 + dx >= 0 && w0 + dx < W) s0 += du[p] * smap[((int64_t)p + shift) * 2 + j];
    }
    float s = wave_sum(s0 + s1);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int q = 0; q < (int)(blockDim.x >> 6); ++q) tot += red[q];
        dwsa[blockIdx.x] = tot;
    }
}

// B6: dx[p,c] += d_avg[c]/HW + [p == argmax_p(c)] * d_max[c]
template <typename T>
__global__ void cbam_bwd_pool_kernel(CV dx, int64_t NP, int HW, int C, const float* __restrict__ dpool, const int* __restrict__ amax) {
    const uint32_t groups = (uint32_t)(C / 4), total = (uint32_t)NP * groups, uhw = (uint32_t)HW;  // host: NP * C / 4 < 2^31
    T* dp = reinterpret_cast<T*>(const_cast<void*>(dx.p));
    const float inv = 1.f / (float)HW;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint32_t p = i / groups, g = i - p * groups;
        const uint32_t n = p / uhw, q = p - n * uhw;
        float v[4];
        Pack<T, 4>::load(dp + (int64_t)p * dx.ld + g * 4, v);
        const float4 da = *reinterpret_cast<const float4*>(dpool + ((int64_t)n * 2 + 0) * C + g * 4);
        const float4 dm = *reinterpret_cast<const float4*>(dpool + ((int64_t)n * 2 + 1) * C + g * 4);
        const int4 am = *reinterpret_cast<const int4*>(amax + (int64_t)n * C + g * 4);
        v[0] += da.x * inv + (am.x == (int)q ? dm.x : 0.f);
        v[1] += da.y * inv + (am.y == (int)q ? dm.y : 0.f);
        v[2] += da.z * inv + (am.z == (int)q ? dm.z : 0.f);
        v[3] += da.w * inv + (am.w == (int)q ? dm.w : 0.f);
        Pack<T, 4>::store(dp + (int64_t)p * dx.ld + g * 4, v);
    }
}

static int cbam_pb(int HW) {
    int pb = (HW + 31) / 32;  // >= 8 pixels per wave
    if (pb > 64) pb = 64;
    if (pb < 1) pb = 1;
    return pb;
}

extern "C" size_t ymi_cbam_bwd_workspace(int64_t n, int64_t h, int64_t w, int64_t c, int64_t hidden) {
    const int pb = cbam_pb((int)(h * w));
    // du [N*H*W] | dcap [N][PB][C] | dz [N][C] | hsum [N][Hd] | dpre [N][2][Hd] | dpool [N][2][C]
    return (size_t)(n * h * w + n * pb * c + n * c + n * hidden + 2 * n * hidden + 2 * n * c + 6 * 4) * sizeof(float) + 256;
}

extern "C" int ymi_cbam_bwd(const ymi_tensor* x, const ymi_tensor* dout, const float* w1, const float* w2, int64_t hidden, const float* wsa,
                            int64_t ksa, const float* ca, const float* pooled, const int32_t* pool_argmax, const float* smap,
                            const int32_t* smap_argmax, const float* sa, const ymi_tensor* dx, float* dw1, float* dw2, float* dwsa,
                            void* workspace, size_t workspace_bytes, void* stream) {
    YMI_CHECK_ARG(ymi_tensor_ok(x) && ymi_tensor_ok(dout) && ymi_tensor_ok(dx) && ymi_same_shape(x, dout) && ymi_same_shape(x, dx), "cbam_bwd: tensors");
    YMI_CHECK_ARG(x->dtype == dout->dtype && x->dtype == dx->dtype, "cbam_bwd: dtypes");
    YMI_CHECK_ARG(x->c % 4 == 0 && x->c <= 1024 && x->ld % 4 == 0 && dout->ld % 4 == 0 && dx->ld % 4 == 0, "cbam_bwd: channels multiple of 4, <= 1024");
    YMI_CHECK_ARG(w1 && w2 && wsa && ca && pooled && pool_argmax && smap && smap_argmax && sa && dw1 && dw2 && dwsa && workspace, "cbam_bwd: null buffer");
    const int N = (int)x->n, H = (int)x->h, W = (int)x->w, C = (int)x->c, HW = H * W, Hd = (int)hidden, K = (int)ksa;
    const size_t need = ymi_cbam_bwd_workspace(N, H, W, C, Hd);
    if (workspace_bytes < need) {
        ymi_set_error("cbam_bwd: workspace %zu < %zu", workspace_bytes, need);
        return YMI_EWORKSPACE;
    }
    const int PB = cbam_pb(HW);
    const int64_t NP = (int64_t)N * HW;
    YMI_CHECK_ARG(NP * (C / 4) < ((int64_t)1 << 31), "cbam_bwd: N*H*W*C/4 must stay below 2^31");
    auto up4 = [](int64_t v) { return (v + 3) / 4 * 4; };  // sub-buffers start on 16-byte boundaries (float4 reads of dpool)
    YMI_CHECK_ARG(((uintptr_t)workspace & 15) == 0, "cbam_bwd: workspace must be 16-byte aligned");
    float* du = reinterpret_cast<float*>(workspace);
    float* dcap = du + up4(NP);
    float* dz = dcap + up4((int64_t)N * PB * C);
    float* hsum = dz + up4((int64_t)N * C);
    float* dpre = hsum + up4((int64_t)N * Hd);
    float* dpool = dpre + up4((int64_t)2 * N * Hd);
    hipStream_t s = (hipStream_t)stream;
    CV xv{x->data, x->ld}, dv{dout->data, dout->ld}, ov{dx->data, dx->ld};
    dim3 gw((unsigned)((NP + 3) / 4));
    const size_t lds3 = (size_t)(3 * C + 3 * Hd) * sizeof(float);
    int64_t ge = (NP * (C / 4) + 255) / 256;
    if (ge > 4096) ge = 4096;
    if (x->dtype == YMI_BF16) {
        hipLaunchKernelGGL(cbam_bwd_du_kernel<bf16_t>, gw, dim3(256), 0, s, xv, dv, NP, HW, C, ca, sa, du);
        hipLaunchKernelGGL(cbam_bwd_dx_kernel<bf16_t>, dim3(N * PB), dim3(256), 0, s, xv, dv, ov, N, H, W, C, ca, sa, du, smap_argmax, wsa, K, dcap, PB);
    } else {
        hipLaunchKernelGGL(cbam_bwd_du_kernel<float>, gw, dim3(256), 0, s, xv, dv, NP, HW, C, ca, sa, du);
        hipLaunchKernelGGL(cbam_bwd_dx_kernel<float>, dim3(N * PB), dim3(256), 0, s, xv, dv, ov, N, H, W, C, ca, sa, du, smap_argmax, wsa, K, dcap, PB);
    }
    hipLaunchKernelGGL(cbam_bwd_mlp_kernel, dim3(N), dim3(1024), lds3, s, dcap, PB, pooled, ca, w1, w2, C, Hd, dz, hsum, dpre, dpool);
    hipLaunchKernelGGL(cbam_bwd_w_kernel, dim3((2 * C * Hd + 255) / 256), dim3(256), 0, s, dz, hsum, dpre, pooled, N, C, Hd, dw1, dw2);
    hipLaunchKernelGGL(cbam_bwd_wsa_kernel, dim3(2 * K * K), dim3(1024), 0, s, du, smap, N, H, W, K, dwsa);
    if (x->dtype == YMI_BF16) hipLaunchKernelGGL(cbam_bwd_pool_kernel<bf16_t>, dim3((unsigned)ge), dim3(256), 0, s, ov, NP, HW, C, dpool, pool_argmax);
    else hipLaunchKernelGGL(cbam_bwd_pool_kernel<float>, dim3((unsigned)ge), dim3(256), 0, s, ov, NP, HW, C, dpool, pool_argmax);
    YMI_CHECK_LAUNCH("cbam_bwd");
    return YMI_OK;
}
