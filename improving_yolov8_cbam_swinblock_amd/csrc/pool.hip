// SPPF max-pool cascade (k x k, stride 1, pad k/2 with -inf, three times chained) for NHWC tensors.
//
// Forward: one launch produces y1, y2, y3 from y0.  A workgroup owns a spatial tile x a 64-byte channel
// slab (32 bf16 / 16 f32 channels), stages the tile plus a 3*(k/2) halo (clamped to the image) in LDS
// once and runs the three pools as separable row-max / column-max passes between two LDS images, so
// y0 is read from HBM once and y1..y3 are written once (the algorithmic minimum: 1 read + 3 writes).
// Positions outside the image are -inf for every stage, which reproduces the chained semantics of
// nn.MaxPool2d exactly (max is exact in any precision -> bit-identical to the f32 reference).
//
// Backward: per stage, g_in[s] += sum over p in window(s) of [argmax_window(p) == s] * g_out[p], as a
// GATHER (deterministic), with the arg-max of every window recomputed in LDS using PyTorch's tie rule
// (first maximum in row-major scan order).
#include "common.h"

struct PV {
    void* p;
    int64_t ld;
};

struct PoolArgs {
    PV y0, y1, y2, y3;
    int N, H, W, C;
    int k, TH, TW;
};

template <typename T> struct SlabTraits;
template <> struct SlabTraits<bf16_t> { static constexpr int CS = 32; };  // channels per 64-byte slab
template <> struct SlabTraits<float> { static constexpr int CS = 16; };

// 16-byte chunk as floats
template <typename T> struct Chunk;
template <> struct Chunk<bf16_t> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void load(const void* p, float (&v)[8]) { Pack<bf16_t, 8>::load(reinterpret_cast<const bf16_t*>(p), v); }
    static __device__ __forceinline__ void store(void* p, const float (&v)[8]) { Pack<bf16_t, 8>::store(reinterpret_cast<bf16_t*>(p), v); }
};
template <> struct Chunk<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void load(const void* p, float (&v)[4]) { Pack<float, 4>::load(reinterpret_cast<const float*>(p), v); }
    static __device__ __forceinline__ void store(void* p, const float (&v)[4]) { Pack<float, 4>::store(reinterpret_cast<float*>(p), v); }
};

#define NEG_INF (-__builtin_inff())

template <typename T>
__global__ __launch_bounds__(256) void sppf_pool3_kernel(PoolArgs a) {
    constexpr int CN = Chunk<T>::N;
    constexpr int CS = SlabTraits<T>::CS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int r = a.k / 2;
    const int tiles_w = (a.W + a.TW - 1) / a.TW;
    const int th0 = (blockIdx.x / tiles_w) * a.TH, tw0 = (blockIdx.x % tiles_w) * a.TW;
    const int th1 = min(th0 + a.TH, a.H), tw1 = min(tw0 + a.TW, a.W);
    const int c0 = blockIdx.y * CS;
    const int n = blockIdx.z;
    // staged region = tile + 3r halo, clamped to the image
    const int rh0 = max(th0 - 3 * r, 0), rh1 = min(th1 + 3 * r, a.H);
    const int rw0 = max(tw0 - 3 * r, 0), rw1 = min(tw1 + 3 * r, a.W);
    const int RH = rh1 - rh0, RW = rw1 - rw0;
    char* A = smem;
    char* B = smem + (size_t)RH * RW * 64;
    const int items = RH * RW * 4;  // (pixel, 16-byte chunk)

    const T* src = reinterpret_cast<const T*>(a.y0.p);
    for (int i = threadIdx.x; i < items; i += 256) {
        const int ch = i & 3, px = i >> 2;
        const int h = rh0 + px / RW, w = rw0 + px % RW;
        float v[CN];
        if (c0 + ch * CN < a.C) {
            Chunk<T>::load(src + (((int64_t)n * a.H + h) * a.W + w) * a.y0.ld + c0 + ch * CN, v);
        } else {
#pragma unroll
            for (int e = 0; e < CN; ++e) v[e] = NEG_INF;
        }
        Chunk<T>::store(A + (size_t)i * 16, v);
    }
    __syncthreads();

    PV outs[3] = {a.y1, a.y2, a.y3};
#pragma unroll 1
    for (int stage = 0; stage < 3; ++stage) {
        // row max: B[h][w] = max_{|dx|<=r, inside region/image} A[h][w+dx]
        for (int i = threadIdx.x; i < items; i += 256) {
            const int ch = i & 3, px = i >> 2;
            const int hh = px / RW, ww = px % RW;
            float m[CN];
#pragma unroll
            for (int e = 0; e < CN; ++e) m[e] = NEG_INF;
            const int lo = max(ww - r, 0), hi = min(ww + r, RW - 1);
            for (int x = lo; x <= hi; ++x) {
                float v[CN];
                Chunk<T>::load(A + ((size_t)(hh * RW + x) * 4 + ch) * 16, v);
#pragma unroll
                for (int e = 0; e < CN; ++e) m[e] = fmaxf(m[e], v[e]);
            }
            Chunk<T>::store(B + (size_t)i * 16, m);
        }
        __syncthreads();
        // column max: A[h][w] = max_{|dy|<=r} B[h+dy][w]; then emit the tile interior
        T* dst = reinterpret_cast<T*>(outs[stage].p);
        for (int i = threadIdx.x; i < items; i += 256) {
            const int ch = i & 3, px = i >> 2;
            const int hh = px / RW, ww = px % RW;
            float m[CN];
#pragma unroll
            for (int e = 0; e < CN; ++e) m[e] = NEG_INF;
            const int lo = max(hh - r, 0), hi = min(hh + r, RH - 1);
            for (int y = lo; y <= hi; ++y) {
                float v[CN];
                Chunk<T>::load(B + ((size_t)(y * RW + ww) * 4 + ch) * 16, v);
#pragma unroll
                for (int e = 0; e < CN; ++e) m[e] = fmaxf(m[e], v[e]);
            }
            Chunk<T>::store(A + (size_t)i * 16, m);
            const int h = rh0 + hh, w = rw0 + ww;
            if (h >= th0 && h < th1 && w >= tw0 && w < tw1 && c0 + ch * CN < a.C)
                Chunk<T>::store(dst + (((int64_t)n * a.H + h) * a.W + w) * outs[stage].ld + c0 + ch * CN, m);
        }
        __syncthreads();
    }
    // Note on halo validity: a value of stage s at region position q is exact when q is at least
    // s*r inside the clamped region edge OR that edge is the image border; the tile interior is 3r
    // inside every non-border edge, so all three emitted stages are exact.
}

static bool pool_geometry(int H, int W, int k, int es, int* TH, int* TW, size_t* lds) {
    const int r = k / 2;
    int th = H, tw = W;
    auto bytes = [&](int t_h, int t_w) {
        const int RH = (t_h + 6 * r < H) ? t_h + 6 * r : H, RW = (t_w + 6 * r < W) ? t_w + 6 * r : W;
        return (size_t)RH * RW * 64 * 2;
    };
    while (bytes(th, tw) > 150 * 1024) {
        if (th >= tw && th > 8) th = (th + 1) / 2;
        else if (tw > 8) tw = (tw + 1) / 2;
        else return false;
    }
    *TH = th; *TW = tw; *lds = bytes(th, tw);
    (void)es;
    return true;
}

extern "C" int ymi_sppf_pool3_fwd(const ymi_tensor* y0, int64_t k, const ymi_tensor* y1, const ymi_tensor* y2, const ymi_tensor* y3, void* stream) {
    YMI_CHECK_ARG(ymi_tensor_ok(y0) && ymi_tensor_ok(y1) && ymi_tensor_ok(y2) && ymi_tensor_ok(y3), "sppf_pool3_fwd: bad tensor");
    YMI_CHECK_ARG(ymi_same_shape(y0, y1) && ymi_same_shape(y0, y2) && ymi_same_shape(y0, y3), "sppf_pool3_fwd: shapes");
    YMI_CHECK_ARG(y0->dtype == y1->dtype && y0->dtype == y2->dtype && y0->dtype == y3->dtype, "sppf_pool3_fwd: dtypes");
    YMI_CHECK_ARG(k >= 1 && (k & 1) && k <= 13, "sppf_pool3_fwd: odd k <= 13");
    const int cn = y0->dtype == YMI_BF16 ? 8 : 4;
    const ymi_tensor* ts[4] = {y0, y1, y2, y3};
    for (auto t : ts)
        YMI_CHECK_ARG(t->c % cn == 0 && t->ld % cn == 0 && ((uintptr_t)t->data & 15) == 0, "sppf_pool3_fwd: channels/ld/base must be 16-byte aligned");
    PoolArgs a{};
    a.y0 = PV{y0->data, y0->ld}; a.y1 = PV{y1->data, y1->ld}; a.y2 = PV{y2->data, y2->ld}; a.y3 = PV{y3->data, y3->ld};
    a.N = (int)y0->n; a.H = (int)y0->h; a.W = (int)y0->w; a.C = (int)y0->c; a.k = (int)k;
    size_t lds = 0;
    YMI_CHECK_ARG(pool_geometry(a.H, a.W, a.k, 0, &a.TH, &a.TW, &lds), "sppf_pool3_fwd: tile does not fit LDS");
    const int cs = y0->dtype == YMI_BF16 ? 32 : 16;
    dim3 grid(((a.H + a.TH - 1) / a.TH) * ((a.W + a.TW - 1) / a.TW), (a.C + cs - 1) / cs, a.N);
    if (y0->dtype == YMI_BF16) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sppf_pool3_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(sppf_pool3_kernel<bf16_t>, grid, dim3(256), lds, (hipStream_t)stream, a);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sppf_pool3_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(sppf_pool3_kernel<float>, grid, dim3(256), lds, (hipStream_t)stream, a);
    }
    YMI_CHECK_LAUNCH("sppf_pool3_fwd");
    return YMI_OK;
}

// ---------------------------------------------------------------------------------------- backward
struct PoolBwdArgs {
    PV x, gout, gin;
    int N, H, W, C;
    int k, TH, TW;
};

// CN one-byte arg-max codes of a 16-byte channel chunk, packed: one 8-byte (bf16) / 4-byte (f32) LDS access
template <int CN> __device__ __forceinline__ void store_codes(unsigned char* p, const int (&code)[CN]) {
    uint64_t v = 0;
#pragma unroll
    for (int e = 0; e < CN; ++e) v |= (uint64_t)(code[e] & 255) << (8 * e);
    if (CN == 8) *reinterpret_cast<uint64_t*>(p) = v;
    else *reinterpret_cast<uint32_t*>(p) = (uint32_t)v;
}
template <int CN> __device__ __forceinline__ uint64_t load_codes(const unsigned char* p) {
    return CN == 8 ? *reinterpret_cast<const uint64_t*>(p) : (uint64_t)*reinterpret_cast<const uint32_t*>(p);
}

// one stage: gin[s] += sum_{p in window(s)} [argmax(x, window(p)) == s] gout[p]
// NCH = 16-byte channel chunks per pixel owned by one workgroup (4: a 64-byte slab, coalesced for large maps;
// 1: four times as many workgroups, for the small SPPF maps where the launch would not fill the chip otherwise).
// The arg-max is separable: first the row maximum (and its column code) over the k columns, then the first row
// whose row maximum is the window maximum: 2k LDS reads per window instead of k*k, and the same element as a
// row-major scan with "strictly greater wins" (PyTorch's tie rule).
template <typename T, int NCH>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(PoolBwdArgs a) {
    constexpr int CN = Chunk<T>::N;
    constexpr int CS = CN * NCH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int r = a.k / 2, k = a.k;
    const int tiles_w = (a.W + a.TW - 1) / a.TW;
    const int th0 = (blockIdx.x / tiles_w) * a.TH, tw0 = (blockIdx.x % tiles_w) * a.TW;
    const int th1 = min(th0 + a.TH, a.H), tw1 = min(tw0 + a.TW, a.W);
    const int c0 = blockIdx.y * CS, n = blockIdx.z;
    // x on tile + 2r, gout / argmax on tile + r (both clamped to the image)
    const int xh0 = max(th0 - 2 * r, 0), xh1 = min(th1 + 2 * r, a.H), xw0 = max(tw0 - 2 * r, 0), xw1 = min(tw1 + 2 * r, a.W);
    const int gh0 = max(th0 - r, 0), gh1 = min(th1 + r, a.H), gw0 = max(tw0 - r, 0), gw1 = min(tw1 + r, a.W);
    const int XH = xh1 - xh0, XW = xw1 - xw0, GH = gh1 - gh0, GW = gw1 - gw0;
    char* X = smem;                                          // [XH][XW][NCH * 16 B]
    char* G = X + (size_t)XH * XW * NCH * 16;                // [GH][GW][NCH * 16 B]
    char* RM = G + (size_t)GH * GW * NCH * 16;               // [XH][GW][NCH * 16 B] row maxima
    unsigned char* IDX = reinterpret_cast<unsigned char*>(RM + (size_t)XH * GW * NCH * 16);  // [GH][GW][CS]
    unsigned char* RC = IDX + (size_t)GH * GW * CS;          // [XH][GW][CS] column code of the row maximum

    const T* xs = reinterpret_cast<const T*>(a.x.p);
    const T* gs = reinterpret_cast<const T*>(a.gout.p);
    for (int i = threadIdx.x; i < XH * XW * NCH; i += 256) {
        const int ch = i % NCH, px = i / NCH;
        const int h = xh0 + px / XW, w = xw0 + px % XW;
        float v[CN];
        if (c0 + ch * CN < a.C) Chunk<T>::load(xs + (((int64_t)n * a.H + h) * a.W + w) * a.x.ld + c0 + ch * CN, v);
        else
#pragma unroll
            for (int e = 0; e < CN; ++e) v[e] = NEG_INF;
        Chunk<T>::store(X + (size_t)i * 16, v);
    }
    for (int i = threadIdx.x; i < GH * GW * NCH; i += 256) {
        const int ch = i % NCH, px = i / NCH;
        const int h = gh0 + px / GW, w = gw0 + px % GW;
        float v[CN];
        if (c0 + ch * CN < a.C) Chunk<T>::load(gs + (((int64_t)n * a.H + h) * a.W + w) * a.gout.ld + c0 + ch * CN, v);
        else
#pragma unroll
            for (int e = 0; e < CN; ++e) v[e] = 0.f;
        Chunk<T>::store(G + (size_t)i * 16, v);
    }
    __syncthreads();
    // row pass: for every staged row and every G column, the maximum over the k columns and its column code
    for (int i = threadIdx.x; i < XH * GW * NCH; i += 256) {
        const int ch = i % NCH, px = i / NCH;
        const int hh = px / GW, w = gw0 + px % GW;
        float best[CN];
        int code[CN];
#pragma unroll
        for (int e = 0; e < CN; ++e) { best[e] = NEG_INF; code[e] = -1; }
        for (int dx = 0; dx < k; ++dx) {
            const int ww = w + dx - r;
            if (ww < 0 || ww >= a.W) continue;
            float v[CN];
            Chunk<T>::load(X + ((size_t)(hh * XW + (ww - xw0)) * NCH + ch) * 16, v);
#pragma unroll
            for (int e = 0; e < CN; ++e)
                if (code[e] < 0 || v[e] > best[e]) {  // first in-image element initialises; then strictly greater wins
                    best[e] = v[e];
                    code[e] = dx;
                }
        }
        Chunk<T>::store(RM + (size_t)i * 16, best);
        store_codes<CN>(RC + (size_t)px * CS + ch * CN, code);
    }
    __syncthreads();
    // column pass: arg-max code (dy*k + dx) of every window centred in the G region
    for (int i = threadIdx.x; i < GH * GW * NCH; i += 256) {
        const int ch = i % NCH, px = i / NCH;
        const int h = gh0 + px / GW, wl = px % GW;
        float best[CN];
        int code[CN];
#pragma unroll
        for (int e = 0; e < CN; ++e) { best[e] = NEG_INF; code[e] = -1; }
        for (int dy = 0; dy < k; ++dy) {
            const int hh = h + dy - r;
            if (hh < 0 || hh >= a.H) continue;
            const size_t rp = (size_t)(hh - xh0) * GW + wl;
            float v[CN];
            Chunk<T>::load(RM + (rp * NCH + ch) * 16, v);
            const uint64_t rc = load_codes<CN>(RC + rp * CS + ch * CN);
#pragma unroll
            for (int e = 0; e < CN; ++e)
                if (code[e] < 0 || v[e] > best[e]) {
                    best[e] = v[e];
                    code[e] = dy * k + (int)((rc >> (8 * e)) & 255);
                }
        }
        store_codes<CN>(IDX + (size_t)px * CS + ch * CN, code);
    }
    __syncthreads();
    T* gi = reinterpret_cast<T*>(a.gin.p);
    const int TH = th1 - th0, TW = tw1 - tw0;
    for (int i = threadIdx.x; i < TH * TW * NCH; i += 256) {
        const int ch = i % NCH, px = i / NCH;
        const int h = th0 + px / TW, w = tw0 + px % TW;
        if (c0 + ch * CN >= a.C) continue;
        float acc[CN];
        T* dst = gi + (((int64_t)n * a.H + h) * a.W + w) * a.gin.ld + c0 + ch * CN;
        Chunk<T>::load(dst, acc);
        for (int ay = -r; ay <= r; ++ay) {
            const int ph = h + ay;
            if (ph < 0 || ph >= a.H) continue;
            for (int ax = -r; ax <= r; ++ax) {
                const int pw = w + ax;
                if (pw < 0 || pw >= a.W) continue;
                const int want = (r - ay) * k + (r - ax);
                const size_t gp = (size_t)(ph - gh0) * GW + (pw - gw0);
                float g[CN];
                Chunk<T>::load(G + (gp * NCH + ch) * 16, g);
                const uint64_t ix = load_codes<CN>(IDX + gp * CS + ch * CN);  // one LDS access for the CN arg-max codes
#pragma unroll
                for (int e = 0; e < CN; ++e)
                    if ((int)((ix >> (8 * e)) & 255) == want) acc[e] += g[e];
            }
        }
        Chunk<T>::store(dst, acc);
    }
}

template <typename T, int NCH>
static void launch_pool_bwd_t(const PoolBwdArgs& a, dim3 grid, size_t lds, hipStream_t stream) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(maxpool_bwd_kernel<T, NCH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((maxpool_bwd_kernel<T, NCH>), grid, dim3(256), lds, stream, a);
}

static int launch_pool_bwd(const ymi_tensor* x, int k, const ymi_tensor* gout, const ymi_tensor* gin, hipStream_t stream) {
    PoolBwdArgs a{};
    a.x = PV{x->data, x->ld}; a.gout = PV{gout->data, gout->ld}; a.gin = PV{gin->data, gin->ld};
    a.N = (int)x->n; a.H = (int)x->h; a.W = (int)x->w; a.C = (int)x->c; a.k = k;
    const int r = k / 2;
    const int cn = x->dtype == YMI_BF16 ? 8 : 4;
    const int nch = (a.H * a.W <= 1024) ? 1 : 4;  // small maps: one chunk per workgroup so the grid fills the chip
    const int cs = cn * nch;
    int th = a.H, tw = a.W;
    auto bytes = [&](int t_h, int t_w) {
        const int XH = (t_h + 4 * r < a.H) ? t_h + 4 * r : a.H, XW = (t_w + 4 * r < a.W) ? t_w + 4 * r : a.W;
        const int GH = (t_h + 2 * r < a.H) ? t_h + 2 * r : a.H, GW = (t_w + 2 * r < a.W) ? t_w + 2 * r : a.W;
        return (size_t)XH * XW * nch * 16 + (size_t)GH * GW * (nch * 16 + cs) + (size_t)XH * GW * (nch * 16 + cs);
    };
    while (bytes(th, tw) > 150 * 1024) {
        if (th >= tw && th > 8) th = (th + 1) / 2;
        else if (tw > 8) tw = (tw + 1) / 2;
        else {
            ymi_set_error("sppf_pool3_bwd: tile does not fit LDS");
            return YMI_EINVAL;
        }
    }
    a.TH = th; a.TW = tw;
    const size_t lds = bytes(th, tw);
    dim3 grid(((a.H + th - 1) / th) * ((a.W + tw - 1) / tw), (a.C + cs - 1) / cs, a.N);
    if (x->dtype == YMI_BF16) {
        if (nch == 1) launch_pool_bwd_t<bf16_t, 1>(a, grid, lds, stream);
        else launch_pool_bwd_t<bf16_t, 4>(a, grid, lds, stream);
    } else {
        if (nch == 1) launch_pool_bwd_t<float, 1>(a, grid, lds, stream);
        else launch_pool_bwd_t<float, 4>(a, grid, lds, stream);
    }
    YMI_CHECK_LAUNCH("sppf_pool3_bwd");
    return YMI_OK;
}

// ---- the three stages in ONE launch, for maps that fit LDS whole (the SPPF maps of the model: 20x20) ----------------------------
// A workgroup owns one image and one 16-byte channel chunk: it stages y0, y1, y2 and the running gradient once and runs
//   g := dy3;  g := dy2 + route(g | y2);  g := dy1 + route(g | y1);  dy0 += route(g | y0)
// between LDS images, the running gradient kept in f32 (the per-stage kernel rounds it to the tensor dtype between stages).
// Same arg-max rule and the same separable row / column search as maxpool_bwd_kernel.  Three launches of ~47 us become one.
struct PoolBwd3Args {
    PV y[3];    // y0, y1, y2
    PV dy[4];   // dy0 (in/out), dy1, dy2, dy3 (in)
    int N, H, W, C, k;
};

template <typename T>
__global__ __launch_bounds__(256) void sppf_bwd_fused_kernel(PoolBwd3Args a) {
    constexpr int CN = Chunk<T>::N;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int HW = a.H * a.W, r = a.k / 2, k = a.k;
    const int c0 = blockIdx.x * CN, n = blockIdx.y;
    float* G0 = reinterpret_cast<float*>(smem);                       // [HW][CN] running gradient (ping)
    float* G1 = G0 + (size_t)HW * CN;                                 // (pong)
    char* X = reinterpret_cast<char*>(G1 + (size_t)HW * CN);          // [HW][16 B] values of the current stage
    char* RM = X + (size_t)HW * 16;                                   // [HW][16 B] row maxima
    unsigned char* RC = reinterpret_cast<unsigned char*>(RM + (size_t)HW * 16);  // [HW][CN] column code of the row maximum
    unsigned char* IDX = RC + (size_t)HW * CN;                        // [HW][CN] arg-max code of the window centred here
    const int64_t img = (int64_t)n * HW;
    // g := dy3
    {
        const T* gp = reinterpret_cast<const T*>(a.dy[3].p);
        for (int i = threadIdx.x; i < HW; i += 256) {
            float v[CN];
            Chunk<T>::load(gp + (img + i) * a.dy[3].ld + c0, v);
#pragma unroll
            for (int e = 0; e < CN; ++e) G0[(size_t)i * CN + e] = v[e];
        }
    }
    float* gcur = G0;
    float* gnext = G1;
    for (int st = 2; st >= 0; --st) {
        const T* xp = reinterpret_cast<const T*>(a.y[st].p);
        for (int i = threadIdx.x; i < HW; i += 256) {
            float v[CN];
            Chunk<T>::load(xp + (img + i) * a.y[st].ld + c0, v);
            Chunk<T>::store(X + (size_t)i * 16, v);
        }
        __syncthreads();
        // row pass: maximum over the k columns around every position, and its column code
        for (int i = threadIdx.x; i < HW; i += 256) {
            const int h = i / a.W, w = i - h * a.W;
            float best[CN];
            int code[CN];
#pragma unroll
            for (int e = 0; e < CN; ++e) { best[e] = NEG_INF; code[e] = -1; }
            for (int dx = 0; dx < k; ++dx) {
                const int ww = w + dx - r;
                if (ww < 0 || ww >= a.W) continue;
                float v[CN];
                Chunk<T>::load(X + (size_t)(h * a.W + ww) * 16, v);
#pragma unroll
                for (int e = 0; e < CN; ++e)
                    if (code[e] < 0 || v[e] > best[e]) {  // first in-image element initialises; then strictly greater wins
                        best[e] = v[e];
                        code[e] = dx;
                    }
            }
            Chunk<T>::store(RM + (size_t)i * 16, best);
            store_codes<CN>(RC + (size_t)i * CN, code);
        }
        __syncthreads();
        // column pass: arg-max code (dy*k + dx) of the window centred at every position
        for (int i = threadIdx.x; i < HW; i += 256) {
            const int h = i / a.W, w = i - h * a.W;
            float best[CN];
            int code[CN];
#pragma unroll
            for (int e = 0; e < CN; ++e) { best[e] = NEG_INF; code[e] = -1; }
            for (int dy = 0; dy < k; ++dy) {
                const int hh = h + dy - r;
                if (hh < 0 || hh >= a.H) continue;
                const size_t rp = (size_t)hh * a.W + w;
                float v[CN];
                Chunk<T>::load(RM + rp * 16, v);
                const uint64_t rc = load_codes<CN>(RC + rp * CN);
#pragma unroll
                for (int e = 0; e < CN; ++e)
                    if (code[e] < 0 || v[e] > best[e]) {
                        best[e] = v[e];
                        code[e] = dy * k + (int)((rc >> (8 * e)) & 255);
                    }
            }
            store_codes<CN>(IDX + (size_t)i * CN, code);
        }
        __syncthreads();
        // gather: gnext[s] = dy_st[s] + sum over windows p containing s whose arg-max is s of gcur[p]
        const T* dp = reinterpret_cast<const T*>(a.dy[st].p);
        T* op = reinterpret_cast<T*>(a.dy[0].p);
        for (int i = threadIdx.x; i < HW; i += 256) {
            const int h = i / a.W, w = i - h * a.W;
            float acc[CN];
            Chunk<T>::load(dp + (img + i) * a.dy[st].ld + c0, acc);
            for (int ay = -r; ay <= r; ++ay) {
                const int ph = h + ay;
                if (ph < 0 || ph >= a.H) continue;
                for (int ax = -r; ax <= r; ++ax) {
                    const int pw = w + ax;
                    if (pw < 0 || pw >= a.W) continue;
                    const int want = (r - ay) * k + (r - ax);
                    const size_t gp = (size_t)ph * a.W + pw;
                    const uint64_t ix = load_codes<CN>(IDX + gp * CN);
#pragma unroll
                    for (int e = 0; e < CN; ++e)
                        if ((int)((ix >> (8 * e)) & 255) == want) acc[e] += gcur[gp * CN + e];
                }
            }
            if (st == 0) {
                Chunk<T>::store(op + (img + i) * a.dy[0].ld + c0, acc);
            } else {
#pragma unroll
                for (int e = 0; e < CN; ++e) gnext[(size_t)i * CN + e] = acc[e];
            }
        }
        __syncthreads();  // gnext complete; X / RM / RC / IDX free for the next stage
        float* t = gcur;
        gcur = gnext;
        gnext = t;
    }
}

static size_t sppf_bwd_fused_lds(int hw, int cn) { return (size_t)hw * (2 * cn * sizeof(float) + 16 + 16 + 2 * cn); }

// dy2 += route(dy3 | y2); dy1 += route(dy2 | y1); dy0 += route(dy1 | y0).  dy1 and dy2 are modified in place.
extern "C" int ymi_sppf_pool3_bwd(const ymi_tensor* y0, const ymi_tensor* y1, const ymi_tensor* y2, int64_t k, const ymi_tensor* dy1,
                                  const ymi_tensor* dy2, const ymi_tensor* dy3, const ymi_tensor* dy0_accum, void* stream) {
    const ymi_tensor* ts[7] = {y0, y1, y2, dy1, dy2, dy3, dy0_accum};
    const int cn = (y0 && y0->dtype == YMI_BF16) ? 8 : 4;
    for (auto t : ts) {
        YMI_CHECK_ARG(ymi_tensor_ok(t) && ymi_same_shape(t, y0) && t->dtype == y0->dtype, "sppf_pool3_bwd: tensors must share shape and dtype");
        YMI_CHECK_ARG(t->c % cn == 0 && t->ld % cn == 0 && ((uintptr_t)t->data & 15) == 0, "sppf_pool3_bwd: 16-byte alignment");
    }
    YMI_CHECK_ARG(k >= 1 && (k & 1) && k <= 13, "sppf_pool3_bwd: odd k <= 13");
    hipStream_t s = (hipStream_t)stream;
    static const int fused_env = getenv("YMI_SPPF_BWD_FUSED") ? atoi(getenv("YMI_SPPF_BWD_FUSED")) : 0;  // 1: all three stages in one launch - measured SLOWER (14.88 vs 14.76 ms/step on the same box: the gather is VALU/LDS-bound, not launch-bound, and the f32 running gradient halves the resident workgroups)
    const int hw = (int)(y0->h * y0->w);
    if (fused_env && y0->c % cn == 0 && sppf_bwd_fused_lds(hw, cn) <= 64 * 1024) {  // whole map in LDS (dy1 / dy2 are left untouched)
        PoolBwd3Args a{};
        a.y[0] = PV{y0->data, y0->ld}; a.y[1] = PV{y1->data, y1->ld}; a.y[2] = PV{y2->data, y2->ld};
        a.dy[0] = PV{dy0_accum->data, dy0_accum->ld}; a.dy[1] = PV{dy1->data, dy1->ld}; a.dy[2] = PV{dy2->data, dy2->ld}; a.dy[3] = PV{dy3->data, dy3->ld};
        a.N = (int)y0->n; a.H = (int)y0->h; a.W = (int)y0->w; a.C = (int)y0->c; a.k = (int)k;
        const size_t lds = sppf_bwd_fused_lds(hw, cn);
        dim3 grid((unsigned)(y0->c / cn), (unsigned)y0->n);
        if (y0->dtype == YMI_BF16) hipLaunchKernelGGL((sppf_bwd_fused_kernel<bf16_t>), grid, dim3(256), lds, s, a);
        else hipLaunchKernelGGL((sppf_bwd_fused_kernel<float>), grid, dim3(256), lds, s, a);
        YMI_CHECK_LAUNCH("sppf_pool3_bwd(fused)");
        return YMI_OK;
    }
    int rc = launch_pool_bwd(y2, (int)k, dy3, dy2, s);
    if (rc) return rc;
    rc = launch_pool_bwd(y1, (int)k, dy2, dy1, s);
    if (rc) return rc;
    return launch_pool_bwd(y0, (int)k, dy1, dy0_accum, s);
}
