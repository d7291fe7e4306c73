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
    // unit order: channel slab fastest, then tile, then image; consecutive units on one XCD (neighbouring slabs share 128-byte lines)
    const int unit = xcd_unit(flat_block_id(), gridDim.x * gridDim.y * gridDim.z);
    const int bslab = unit % gridDim.y, btile = (unit / gridDim.y) % gridDim.x;
    const int th0 = (btile / tiles_w) * a.TH, tw0 = (btile % tiles_w) * a.TW;
    const int th1 = min(th0 + a.TH, a.H), tw1 = min(tw0 + a.TW, a.W);
    const int c0 = bslab * CS;
    const int n = unit / (gridDim.x * gridDim.y);
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

// ---------------------------------------------------------------------------------------- backward
struct PoolBwdArgs {
    PV x, gout, gsrc, gin;  // gin = gsrc + route(gout | x)
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

// one stage (maps too large for the whole-map kernel below): gin[s] = gsrc[s] + sum_{p in window(s)} [argmax(x, window(p)) == s] gout[p]
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
    // unit order: channel chunk fastest, then tile, then image; consecutive units on one XCD: with one 16-byte chunk per workgroup
    // eight workgroups share every 128-byte line, and dealt to eight XCDs they fetched it eight times (47 -> see profiles/r03 us)
    const int unit = xcd_unit(flat_block_id(), gridDim.x * gridDim.y * gridDim.z);
    const int bslab = unit % gridDim.y, btile = (unit / gridDim.y) % gridDim.x;
    const int th0 = (btile / tiles_w) * a.TH, tw0 = (btile % tiles_w) * a.TW;
    const int th1 = min(th0 + a.TH, a.H), tw1 = min(tw0 + a.TW, a.W);
    const int c0 = bslab * CS, n = unit / (gridDim.x * gridDim.y);
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
        Chunk<T>::load(reinterpret_cast<const T*>(a.gsrc.p) + (((int64_t)n * a.H + h) * a.W + w) * a.gsrc.ld + c0 + ch * CN, acc);
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

static int launch_pool_bwd(const ymi_tensor* x, int k, PV gout, const ymi_tensor* gsrc, PV gin, hipStream_t stream) {
    PoolBwdArgs a{};
    a.x = PV{x->data, x->ld}; a.gout = gout; a.gsrc = PV{gsrc->data, gsrc->ld}; a.gin = gin;
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

// ---- whole-map kernels: a workgroup owns ONE image x ONE 16-byte channel chunk and keeps the whole map in LDS --------------------
// (the SPPF maps of the model are 20x20 - 40x40 at 1280 input.)  Work units are (chunk fastest, image) through xcd_unit(), so the
// eight chunks of a 128-byte line run on one XCD.  One chunk per workgroup gives C/8 * N workgroups (1024 for the model) where the
// tiled kernels' 64-byte slabs gave 256; loops over the K window positions are unrolled (K = 5, 7 compiled; 0: runtime k) with
// out-of-image positions folded into the compare instead of branched around, so the LDS reads of one item are issued together.
struct MapArgs {
    PV y[4];    // forward: y0 (in), y1..y3 (out).  backward: y0, y1, y2 (y[3] unused)
    PV dy[4];   // backward: dy0..dy3 (in)
    PV dx;      // backward: out
    int N, H, W, C, k;
};

template <typename T> __device__ __forceinline__ void lds_load_f32(const float* base, int planes_stride, int i, float (&v)[Chunk<T>::N]) {
    // f32 images are kept as planes of 4 floats per pixel (16-byte LDS accesses at unit stride across lanes: conflict-free)
#pragma unroll
    for (int q = 0; q < Chunk<T>::N / 4; ++q) {
        const float4 t = *reinterpret_cast<const float4*>(base + (size_t)q * planes_stride + (size_t)i * 4);
        v[q * 4 + 0] = t.x; v[q * 4 + 1] = t.y; v[q * 4 + 2] = t.z; v[q * 4 + 3] = t.w;
    }
}
template <typename T> __device__ __forceinline__ void lds_store_f32(float* base, int planes_stride, int i, const float (&v)[Chunk<T>::N]) {
#pragma unroll
    for (int q = 0; q < Chunk<T>::N / 4; ++q)
        *reinterpret_cast<float4*>(base + (size_t)q * planes_stride + (size_t)i * 4) = make_float4(v[q * 4 + 0], v[q * 4 + 1], v[q * 4 + 2], v[q * 4 + 3]);
}

template <typename T, int K>
__global__ __launch_bounds__(1024) void sppf_fwd_map_kernel(MapArgs a) {
    constexpr int CN = Chunk<T>::N;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int UN = K ? K : 1;  // unroll count of the window loops (runtime k: not unrolled)
    const int HW = a.H * a.W, k = K ? K : a.k, r = k / 2;
    const int unit = xcd_unit(flat_block_id(), gridDim.x * gridDim.y);
    const int c0 = (unit % gridDim.x) * CN, n = unit / gridDim.x;
    char* X = smem;                       // [HW][16 B] input of the current stage
    char* RM = smem + (size_t)HW * 16;    // [HW][16 B] row maxima
    const int64_t img = (int64_t)n * HW;
    {
        const T* src = reinterpret_cast<const T*>(a.y[0].p);
        for (int i = threadIdx.x; i < HW; i += blockDim.x)
            *reinterpret_cast<uint4*>(X + (size_t)i * 16) = *reinterpret_cast<const uint4*>(src + (img + i) * a.y[0].ld + c0);
    }
    __syncthreads();
#pragma unroll 1
    for (int st = 1; st <= 3; ++st) {
        for (int i = threadIdx.x; i < HW; i += blockDim.x) {
            const int h = i / a.W, w = i - h * a.W;
            float m[CN];
#pragma unroll
            for (int e = 0; e < CN; ++e) m[e] = NEG_INF;
#pragma unroll UN
            for (int d = 0; d < k; ++d) {
                const int ww = w - r + d;
                const bool ok = ww >= 0 && ww < a.W;
                float v[CN];
                Chunk<T>::load(X + (size_t)(ok ? i - r + d : i) * 16, v);  // (an out-of-image column reads the centre: max unchanged)
#pragma unroll
                for (int e = 0; e < CN; ++e) m[e] = fmaxf(m[e], v[e]);
            }
            Chunk<T>::store(RM + (size_t)i * 16, m);
        }
        __syncthreads();
        T* dst = reinterpret_cast<T*>(a.y[st].p);
        for (int i = threadIdx.x; i < HW; i += blockDim.x) {
            const int h = i / a.W;
            float m[CN];
#pragma unroll
            for (int e = 0; e < CN; ++e) m[e] = NEG_INF;
#pragma unroll UN
            for (int d = 0; d < k; ++d) {
                const int hh = h - r + d;
                const bool ok = hh >= 0 && hh < a.H;
                float v[CN];
                Chunk<T>::load(RM + (size_t)(ok ? i + (d - r) * a.W : i) * 16, v);
#pragma unroll
                for (int e = 0; e < CN; ++e) m[e] = fmaxf(m[e], v[e]);
            }
            Chunk<T>::store(X + (size_t)i * 16, m);  // (X is only read by the row pass: free since the barrier above)
            Chunk<T>::store(dst + (img + i) * a.y[st].ld + c0, m);
        }
        __syncthreads();
    }
}

// Backward of the cascade in one launch:  g := dy3;  g := dy2 + route(g | y2);  g := dy1 + route(g | y1);  dx = dy0 + route(g | y0),
// the running gradient in f32 in LDS.  route() is a deterministic gather with PyTorch's arg-max rule (first maximum in row-major
// order), and it is SEPARABLE: the arg-max of window p is (first row whose row maximum is the window maximum, that row's first
// maximal column), so  V[hh][w] = sum_{ph} [rowcode(ph, w) -> hh] g[ph][w]  followed by  out[hh][ww] = sum_{pw} [colcode(hh, pw) -> ww]
// V[hh][pw]  visits 2k positions per pixel instead of k*k (the sums associate column-first: exact for the dyadic tie fixtures,
// within f32 rounding of the row-major order otherwise).
template <typename T, int K>
__global__ __launch_bounds__(1024) void sppf_bwd_map_kernel(MapArgs a) {
    constexpr int CN = Chunk<T>::N;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int UN = K ? K : 1;  // unroll count of the window loops (runtime k: not unrolled)
    const int HW = a.H * a.W, k = K ? K : a.k, r = k / 2;
    const int unit = xcd_unit(flat_block_id(), gridDim.x * gridDim.y);
    const int c0 = (unit % gridDim.x) * CN, n = unit / gridDim.x;
    const int PS = HW * 4;                                            // floats per f32 plane
    float* G = reinterpret_cast<float*>(smem);                        // [CN/4][HW][4] running gradient
    char* X = reinterpret_cast<char*>(G + (size_t)HW * CN);           // [HW][16 B] values of the current stage   } later V: [CN/4][HW][4] f32
    char* RM = X + (size_t)HW * 16;                                   // [HW][16 B] row maxima (f32 tensors: V needs X only) }
    float* V = reinterpret_cast<float*>(X);
    unsigned char* RC = reinterpret_cast<unsigned char*>(X + (size_t)HW * 32);  // [HW][CN] column code of the row maximum
    unsigned char* RR = RC + (size_t)HW * CN;                         // [HW][CN] row code of the window maximum
    const int64_t img = (int64_t)n * HW;
    {
        const T* gp = reinterpret_cast<const T*>(a.dy[3].p);
        for (int i = threadIdx.x; i < HW; i += blockDim.x) {
            float v[CN];
            Chunk<T>::load(gp + (img + i) * a.dy[3].ld + c0, v);
            lds_store_f32<T>(G, PS, i, v);
        }
    }
#pragma unroll 1
    for (int st = 2; st >= 0; --st) {
        const T* xp = reinterpret_cast<const T*>(a.y[st].p);
        for (int i = threadIdx.x; i < HW; i += blockDim.x) {
            float v[CN];
            Chunk<T>::load(xp + (img + i) * a.y[st].ld + c0, v);
            Chunk<T>::store(X + (size_t)i * 16, v);
        }
        __syncthreads();
        // row pass: maximum over the k columns around every position and the code (0..k-1) of its first occurrence
        for (int i = threadIdx.x; i < HW; i += blockDim.x) {
            const int h = i / a.W, w = i - h * a.W;
            float best[CN];
            int code[CN];
#pragma unroll
            for (int e = 0; e < CN; ++e) { best[e] = NEG_INF; code[e] = -1; }
#pragma unroll UN
            for (int d = 0; d < k; ++d) {
                const int ww = w - r + d;
                const bool ok = ww >= 0 && ww < a.W;
                float v[CN];
                Chunk<T>::load(X + (size_t)(ok ? i - r + d : i) * 16, v);
#pragma unroll
                for (int e = 0; e < CN; ++e)
                    if (ok && (code[e] < 0 || v[e] > best[e])) {  // the first in-image element initialises; then strictly greater wins
                        best[e] = v[e];
                        code[e] = d;
                    }
            }
            Chunk<T>::store(RM + (size_t)i * 16, best);
            store_codes<CN>(RC + (size_t)i * CN, code);
        }
        __syncthreads();
        // column pass: code (0..k-1) of the first row whose row maximum is the maximum of the window centred here
        for (int i = threadIdx.x; i < HW; i += blockDim.x) {
            const int h = i / a.W;
            float best[CN];
            int code[CN];
#pragma unroll
            for (int e = 0; e < CN; ++e) { best[e] = NEG_INF; code[e] = -1; }
#pragma unroll UN
            for (int d = 0; d < k; ++d) {
                const int hh = h - r + d;
                const bool ok = hh >= 0 && hh < a.H;
                float v[CN];
                Chunk<T>::load(RM + (size_t)(ok ? i + (d - r) * a.W : i) * 16, v);
#pragma unroll
                for (int e = 0; e < CN; ++e)
                    if (ok && (code[e] < 0 || v[e] > best[e])) {
                        best[e] = v[e];
                        code[e] = d;
                    }
            }
            store_codes<CN>(RR + (size_t)i * CN, code);
        }
        __syncthreads();  // X and RM are dead from here: V takes their place
        // vertical gather: V[hh][w] = sum over window centres (ph, w), ph = hh - r + d, whose maximal row is hh (row code 2r - d)
        for (int i = threadIdx.x; i < HW; i += blockDim.x) {
            const int h = i / a.W;
            float acc[CN];
#pragma unroll
            for (int e = 0; e < CN; ++e) acc[e] = 0.f;
#pragma unroll UN
            for (int d = 0; d < k; ++d) {
                const int ph = h - r + d;
                const bool ok = ph >= 0 && ph < a.H;
                const int j = ok ? i + (d - r) * a.W : i;
                const int want = ok ? 2 * r - d : 255;
                float g[CN];
                lds_load_f32<T>(G, PS, j, g);
                const uint64_t rc = load_codes<CN>(RR + (size_t)j * CN);
#pragma unroll
                for (int e = 0; e < CN; ++e)
                    if ((int)((rc >> (8 * e)) & 255) == want) acc[e] += g[e];
            }
            lds_store_f32<T>(V, PS, i, acc);
        }
        __syncthreads();  // G is dead from here: the next running gradient is written over it
        // horizontal gather: out[hh][ww] = dy_st[hh][ww] + sum over pw = ww - r + d with column code 2r - d in row hh of V[hh][pw]
        const T* dp = reinterpret_cast<const T*>(a.dy[st].p);
        T* op = reinterpret_cast<T*>(a.dx.p);
        for (int i = threadIdx.x; i < HW; i += blockDim.x) {
            const int h = i / a.W, w = i - h * a.W;
            float acc[CN];
            Chunk<T>::load(dp + (img + i) * a.dy[st].ld + c0, acc);
#pragma unroll UN
            for (int d = 0; d < k; ++d) {
                const int pw = w - r + d;
                const bool ok = pw >= 0 && pw < a.W;
                const int j = ok ? i - r + d : i;
                const int want = ok ? 2 * r - d : 255;
                float g[CN];
                lds_load_f32<T>(V, PS, j, g);
                const uint64_t rc = load_codes<CN>(RC + (size_t)j * CN);
#pragma unroll
                for (int e = 0; e < CN; ++e)
                    if ((int)((rc >> (8 * e)) & 255) == want) acc[e] += g[e];
            }
            if (st == 0) Chunk<T>::store(op + (img + i) * a.dx.ld + c0, acc);
            else lds_store_f32<T>(G, PS, i, acc);
        }
        __syncthreads();  // V (= X, RM), RC, RR free for the next stage; G complete
    }
}

// LDS of the whole-map kernels per pixel: forward 2 x 16 B; backward G (4 CN) + X|RM / V (32 B >= 4 CN) + RC + RR (2 CN)
static size_t sppf_map_lds(int hw, int cn, bool bwd) { return (size_t)hw * (bwd ? 4 * cn + 32 + 2 * cn : 32); }
static const size_t SPPF_MAP_LDS_MAX = 150 * 1024;
static int sppf_map_threads(int hw) { const int t = (hw + 63) / 64 * 64; return t < 64 ? 64 : (t > 1024 ? 1024 : t); }

template <typename T>
static void launch_sppf_map(bool bwd, const MapArgs& a, size_t lds, hipStream_t s) {
    constexpr int CN = Chunk<T>::N;
    const dim3 grid((unsigned)(a.C / CN), (unsigned)a.N), block((unsigned)sppf_map_threads(a.H * a.W));
#define YMI_SPPF_MAP(KERN, KK)                                                                                                    \
    do {                                                                                                                          \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(KERN<T, KK>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        hipLaunchKernelGGL((KERN<T, KK>), grid, block, lds, s, a);                                                                \
    } while (0)
    if (bwd) {
        if (a.k == 5) YMI_SPPF_MAP(sppf_bwd_map_kernel, 5);
        else if (a.k == 7) YMI_SPPF_MAP(sppf_bwd_map_kernel, 7);
        else YMI_SPPF_MAP(sppf_bwd_map_kernel, 0);
    } else {
        if (a.k == 5) YMI_SPPF_MAP(sppf_fwd_map_kernel, 5);
        else if (a.k == 7) YMI_SPPF_MAP(sppf_fwd_map_kernel, 7);
        else YMI_SPPF_MAP(sppf_fwd_map_kernel, 0);
    }
#undef YMI_SPPF_MAP
}

extern "C" int64_t ymi_sppf_pool3_bwd_workspace(int64_t n, int64_t h, int64_t w, int64_t c, int dtype) {
    const int cn = dtype == YMI_BF16 ? 8 : 4;
    if (sppf_map_lds((int)(h * w), cn, true) <= SPPF_MAP_LDS_MAX) return 0;
    return 2 * n * h * w * c * (int64_t)ymi_esize(dtype);  // the two intermediate gradients of the per-stage path
}

// dx = dy0 + route(dy1 + route(dy2 + route(dy3 | y2) | y1) | y0); no input is modified.
extern "C" int ymi_sppf_pool3_bwd(const ymi_tensor* y0, const ymi_tensor* y1, const ymi_tensor* y2, int64_t k, const ymi_tensor* dy0,
                                  const ymi_tensor* dy1, const ymi_tensor* dy2, const ymi_tensor* dy3, const ymi_tensor* dx, void* workspace,
                                  int64_t workspace_bytes, void* stream) {
    const ymi_tensor* ts[8] = {y0, y1, y2, dy0, dy1, dy2, dy3, dx};
    const int cn = (y0 && y0->dtype == YMI_BF16) ? 8 : 4;
    for (auto t : ts) {
        YMI_CHECK_ARG(ymi_tensor_ok(t) && ymi_same_shape(t, y0) && t->dtype == y0->dtype, "sppf_pool3_bwd: tensors must share shape and dtype");
        YMI_CHECK_ARG(t->c % cn == 0 && t->ld % cn == 0 && ((uintptr_t)t->data & 15) == 0, "sppf_pool3_bwd: 16-byte alignment");
    }
    YMI_CHECK_ARG(k >= 1 && (k & 1) && k <= 13, "sppf_pool3_bwd: odd k <= 13");
    hipStream_t s = (hipStream_t)stream;
    const int hw = (int)(y0->h * y0->w);
    const size_t lds = sppf_map_lds(hw, cn, true);
    if (lds <= SPPF_MAP_LDS_MAX) {
        MapArgs a{};
        a.y[0] = PV{y0->data, y0->ld}; a.y[1] = PV{y1->data, y1->ld}; a.y[2] = PV{y2->data, y2->ld};
        a.dy[0] = PV{dy0->data, dy0->ld}; a.dy[1] = PV{dy1->data, dy1->ld}; a.dy[2] = PV{dy2->data, dy2->ld}; a.dy[3] = PV{dy3->data, dy3->ld};
        a.dx = PV{dx->data, dx->ld};
        a.N = (int)y0->n; a.H = (int)y0->h; a.W = (int)y0->w; a.C = (int)y0->c; a.k = (int)k;
        if (y0->dtype == YMI_BF16) launch_sppf_map<bf16_t>(true, a, lds, s);
        else launch_sppf_map<float>(true, a, lds, s);
        YMI_CHECK_LAUNCH("sppf_pool3_bwd(map)");
        return YMI_OK;
    }
    const int64_t one = y0->n * y0->h * y0->w * y0->c * (int64_t)ymi_esize(y0->dtype);
    YMI_CHECK_ARG(workspace && workspace_bytes >= 2 * one && ((uintptr_t)workspace & 15) == 0, "sppf_pool3_bwd: workspace (see ymi_sppf_pool3_bwd_workspace)");
    const PV g2{workspace, y0->c}, g1{(char*)workspace + one, y0->c};
    int rc = launch_pool_bwd(y2, (int)k, PV{dy3->data, dy3->ld}, dy2, g2, s);
    if (rc) return rc;
    rc = launch_pool_bwd(y1, (int)k, g2, dy1, g1, s);
    if (rc) return rc;
    return launch_pool_bwd(y0, (int)k, g1, dy0, PV{dx->data, dx->ld}, s);
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
    if (sppf_map_lds((int)(y0->h * y0->w), cn, false) <= SPPF_MAP_LDS_MAX) {  // whole map per (image, 16-byte chunk) workgroup
        MapArgs m{};
        m.y[0] = PV{y0->data, y0->ld}; m.y[1] = PV{y1->data, y1->ld}; m.y[2] = PV{y2->data, y2->ld}; m.y[3] = PV{y3->data, y3->ld};
        m.N = (int)y0->n; m.H = (int)y0->h; m.W = (int)y0->w; m.C = (int)y0->c; m.k = (int)k;
        const size_t mlds = sppf_map_lds((int)(y0->h * y0->w), cn, false);
        if (y0->dtype == YMI_BF16) launch_sppf_map<bf16_t>(false, m, mlds, (hipStream_t)stream);
        else launch_sppf_map<float>(false, m, mlds, (hipStream_t)stream);
        YMI_CHECK_LAUNCH("sppf_pool3_fwd(map)");
        return YMI_OK;
    }
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

