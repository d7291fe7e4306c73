// v8 detection loss on the per-level Detect maps: DFL decode, task-aligned assignment, CIoU + DFL + BCE sums and
// their gradients, in ten small launches instead of several hundred framework ops.
//
// Mathematics follows the reference (ultralytics/utils/loss.py:201-255 v8DetectionLoss.__call__, loss.py:65-120
// DFLoss/BboxLoss, utils/tal.py:14-327 TaskAlignedAssigner, utils/metrics.py:74-134 bbox_iou CIoU) on the layout the
// Detect convolutions already write: per level an NHWC box map [B,H,W,4*reg_max] and class map [B,H,W,nc].
// Anchor a of image b = level l, pixel q = a - off[l] (row-major y,x): exactly the reference's concat order.
//
// Everything is f32 and deterministic (fixed-order block reductions, no atomics).  top-k ties go to the lower anchor
// index (the reference leaves tie order to torch.topk).
#include "common.h"

namespace {
constexpr int LOSS_MAXL = 4;
constexpr int REG = 16;  // reg_max (DFL bins per side)
constexpr float TAL_EPS = 1e-9f;

struct LossGeom {
    const void* box[LOSS_MAXL];
    const void* cls[LOSS_MAXL];
    void* dbox[LOSS_MAXL];
    void* dcls[LOSS_MAXL];
    int64_t ldb[LOSS_MAXL], ldc[LOSS_MAXL], lddb[LOSS_MAXL], lddc[LOSS_MAXL];
    int H[LOSS_MAXL], W[LOSS_MAXL], off[LOSS_MAXL + 1];
    float stride[LOSS_MAXL];
    int nl, B, A, G, nc;
    int dcw;  // backward: channels of the class-gradient maps (>= nc; the ones beyond nc are written as zeros: padded gradient buffers)
};

struct Anchor {
    int l;
    int64_t pix;  // pixel row inside the level's maps (b, y, x)
    float ax, ay, s;
};

// arr[l] by a select chain: a dynamically indexed kernel-argument array would be copied to scratch memory
template <typename X> __device__ __forceinline__ X pick(const X (&arr)[LOSS_MAXL], int l) {
    X r = arr[0];
#pragma unroll
    for (int i = 1; i < LOSS_MAXL; ++i) r = (l == i) ? arr[i] : r;
    return r;
}

__device__ __forceinline__ Anchor anchor_of(const LossGeom& g, int b, int a) {
    int l = 0, off = 0, w = g.W[0], h = g.H[0];
    float s = g.stride[0];
#pragma unroll
    for (int i = 1; i < LOSS_MAXL; ++i)
        if (i < g.nl && a >= g.off[i]) {
            l = i;
            off = g.off[i];
            w = g.W[i];
            h = g.H[i];
            s = g.stride[i];
        }
    const int q = a - off;
    const int y = q / w, x = q - y * w;
    Anchor r;
    r.l = l;
    r.pix = (int64_t)b * h * w + q;
    r.ax = x + 0.5f;
    r.ay = y + 0.5f;
    r.s = s;
    return r;
}

__device__ __forceinline__ float sq(float v) { return v * v; }

// CIoU of xyxy boxes, b1 and b2 in the order the reference passes them (metrics.py:74-134, xywh=False, CIoU=True)
__device__ __forceinline__ float ciou_f(const float (&b1)[4], const float (&b2)[4]) {
    const float eps = 1e-7f;
    const float w1 = b1[2] - b1[0], h1 = b1[3] - b1[1] + eps, w2 = b2[2] - b2[0], h2 = b2[3] - b2[1] + eps;
    const float iw = fmaxf(fminf(b1[2], b2[2]) - fmaxf(b1[0], b2[0]), 0.f);
    const float ih = fmaxf(fminf(b1[3], b2[3]) - fmaxf(b1[1], b2[1]), 0.f);
    const float inter = iw * ih;
    const float uni = w1 * h1 + w2 * h2 - inter + eps;
    const float iou = inter / uni;
    const float cw = fmaxf(b1[2], b2[2]) - fminf(b1[0], b2[0]);
    const float ch = fmaxf(b1[3], b2[3]) - fminf(b1[1], b2[1]);
    const float c2 = cw * cw + ch * ch + eps;
    const float rho2 = (sq(b2[0] + b2[2] - b1[0] - b1[2]) + sq(b2[1] + b2[3] - b1[1] - b1[3])) * 0.25f;
    const float dv = atanf(w2 / h2) - atanf(w1 / h1);
    const float v = 0.40528473456935108578f * dv * dv;  // 4 / pi^2
    const float alpha = v / (v - iou + (1.0f + eps));
    return iou - (rho2 / c2 + v * alpha);
}

// CIoU(b1, b2) and its gradient with respect to b1 (alpha is a constant, as under the reference's no_grad)
__device__ __forceinline__ float ciou_grad(const float (&b1)[4], const float (&b2)[4], float (&g)[4]) {
    const float eps = 1e-7f;
    const float w1 = b1[2] - b1[0], h1 = b1[3] - b1[1] + eps, w2 = b2[2] - b2[0], h2 = b2[3] - b2[1] + eps;
    const float iwr = fminf(b1[2], b2[2]) - fmaxf(b1[0], b2[0]);
    const float ihr = fminf(b1[3], b2[3]) - fmaxf(b1[1], b2[1]);
    const float iw = fmaxf(iwr, 0.f), ih = fmaxf(ihr, 0.f);
    const float inter = iw * ih;
    const float uni = w1 * h1 + w2 * h2 - inter + eps;
    const float iou = inter / uni;
    // d inter
    const float on_w = iwr > 0.f ? 1.f : 0.f, on_h = ihr > 0.f ? 1.f : 0.f;
    float dint[4];
    dint[0] = -(b1[0] > b2[0] ? 1.f : 0.f) * on_w * ih;
    dint[1] = -(b1[1] > b2[1] ? 1.f : 0.f) * on_h * iw;
    dint[2] = (b1[2] < b2[2] ? 1.f : 0.f) * on_w * ih;
    dint[3] = (b1[3] < b2[3] ? 1.f : 0.f) * on_h * iw;
    const float duni[4] = {-h1 - dint[0], -w1 - dint[1], h1 - dint[2], w1 - dint[3]};
    const float cw = fmaxf(b1[2], b2[2]) - fminf(b1[0], b2[0]);
    const float ch = fmaxf(b1[3], b2[3]) - fminf(b1[1], b2[1]);
    const float c2 = cw * cw + ch * ch + eps;
    const float dc2[4] = {-(b1[0] < b2[0] ? 1.f : 0.f) * 2.f * cw, -(b1[1] < b2[1] ? 1.f : 0.f) * 2.f * ch,
                          (b1[2] > b2[2] ? 1.f : 0.f) * 2.f * cw, (b1[3] > b2[3] ? 1.f : 0.f) * 2.f * ch};
    const float sx = b2[0] + b2[2] - b1[0] - b1[2], sy = b2[1] + b2[3] - b1[1] - b1[3];
    const float rho2 = (sx * sx + sy * sy) * 0.25f;
    const float drho[4] = {-0.5f * sx, -0.5f * sy, -0.5f * sx, -0.5f * sy};
    const float u = w1 / h1;
    const float dv = atanf(w2 / h2) - atanf(u);
    const float kk = 0.40528473456935108578f;
    const float v = kk * dv * dv;
    const float alpha = v / (v - iou + (1.0f + eps));
    const float dat = 1.0f / (1.0f + u * u);
    const float du[4] = {-1.0f / h1, w1 / (h1 * h1), 1.0f / h1, -w1 / (h1 * h1)};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float diou = (dint[i] * uni - inter * duni[i]) / (uni * uni);
        const float dpen = (drho[i] * c2 - rho2 * dc2[i]) / (c2 * c2);
        const float dvv = 2.f * kk * dv * (-dat * du[i]);
        g[i] = diou - dpen - alpha * dvv;
    }
    return iou - (rho2 / c2 + v * alpha);
}

template <typename T> __device__ __forceinline__ void load16(const T* p, float (&v)[REG]);
template <> __device__ __forceinline__ void load16<float>(const float* p, float (&v)[REG]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p + 4 * i);
        v[4 * i] = t[0]; v[4 * i + 1] = t[1]; v[4 * i + 2] = t[2]; v[4 * i + 3] = t[3];
    }
}
template <> __device__ __forceinline__ void load16<bf16_t>(const bf16_t* p, float (&v)[REG]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const bf16x8 t = *reinterpret_cast<const bf16x8*>(p + 8 * i);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[8 * i + j] = (float)t[j];
    }
}
template <typename T> __device__ __forceinline__ void store16(T* p, const float (&v)[REG]);
template <> __device__ __forceinline__ void store16<float>(float* p, const float (&v)[REG]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x4 t = {v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
        *reinterpret_cast<f32x4*>(p + 4 * i) = t;
    }
}
template <> __device__ __forceinline__ void store16<bf16_t>(bf16_t* p, const float (&v)[REG]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        bf16x8 t;
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = (bf16_t)v[8 * i + j];
        *reinterpret_cast<bf16x8*>(p + 8 * i) = t;
    }
}

// softmax over one side's 16 bins -> probabilities p, returns the expectation sum_j j*p_j (reference DFL decode)
__device__ __forceinline__ float softmax_expect(const float (&x)[REG], float (&p)[REG], float* lse) {
    float m = x[0];
#pragma unroll
    for (int j = 1; j < REG; ++j) m = fmaxf(m, x[j]);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < REG; ++j) {
        p[j] = __expf(x[j] - m);
        s += p[j];
    }
    const float inv = 1.0f / s;
    float e = 0.f;
#pragma unroll
    for (int j = 0; j < REG; ++j) {
        p[j] *= inv;
        e += (float)j * p[j];
    }
    *lse = m + __logf(s);
    return e;
}

// the four side distances of one anchor live in the four lanes of a quad: gather them into every lane
__device__ __forceinline__ void quad_gather(float mine, float (&d)[4]) {
    const int base = (threadIdx.x & 63) & ~3;
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = __shfl(mine, base + k, 64);
}

// ---- targets: ragged (image, class, xywh normalised) rows -> dense [B, G, 5] (class, xyxy pixels) -------------
// reference loss.py:176-191 preprocess; order inside an image is the row order (stable), empty slots are zero
__global__ void targets_kernel(const float* __restrict__ batch_idx, const float* __restrict__ cls, const float* __restrict__ xywh, int n, int B,
                               int G, float img_w, float img_h, float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * G) return;
    const int b = t / G, slot = t - b * G;
    int seen = 0, row = -1;
    for (int i = 0; i < n; ++i) {
        if ((int)batch_idx[i] == b) {
            if (seen == slot) {
                row = i;
                break;
            }
            ++seen;
        }
    }
    float o[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (row >= 0) {
        const float cx = xywh[row * 4 + 0] * img_w, cy = xywh[row * 4 + 1] * img_h, bw = xywh[row * 4 + 2] * img_w, bh = xywh[row * 4 + 3] * img_h;
        o[0] = cls[row];
        o[1] = cx - bw / 2;
        o[2] = cy - bh / 2;
        o[3] = cx + bw / 2;
        o[4] = cy + bh / 2;
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) out[(int64_t)t * 5 + k] = o[k];
}

// ---- K1: DFL decode -> predicted boxes in grid units [B, A, 4] ------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void decode_kernel(LossGeom g, float* __restrict__ pb) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t quad = t >> 2;
    const int side = (int)(t & 3);
    const bool live = quad < (int64_t)g.B * g.A;
    const int b = live ? (int)(quad / g.A) : 0, a = live ? (int)(quad - (int64_t)b * g.A) : 0;
    const Anchor an = anchor_of(g, b, a);
    float x[REG], p[REG], lse;
    load16<T>(reinterpret_cast<const T*>(pick(g.box, an.l)) + an.pix * pick(g.ldb, an.l) + side * REG, x);
    const float e = softmax_expect(x, p, &lse);
    float d[4];
    quad_gather(e, d);
    if (live && side == 0) {
        const f32x4 o = {an.ax - d[0], an.ay - d[1], an.ax + d[2], an.ay + d[3]};
        *reinterpret_cast<f32x4*>(pb + quad * 4) = o;
    }
}

// ---- inference decode: Detect._inference (head.py:103-142, non-export branch) ---------------------------------
// y[b][0:4][a] = dist2bbox(DFL(box), anchor, xywh=True) * stride ; y[b][4+c][a] = sigmoid(cls[c]).  Output is the
// reference's [B, 4+nc, A] float32 tensor (anchor index fastest).  Four lanes per anchor as in decode_kernel; the quad
// then walks the classes.  A-major stores: the 64 lanes of a wave cover 16 consecutive anchors of 4 rows each.
template <typename T>
__global__ __launch_bounds__(256) void infer_decode_kernel(LossGeom g, float* __restrict__ y) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t quad = t >> 2;
    const int side = (int)(t & 3);
    const bool live = quad < (int64_t)g.B * g.A;
    const int b = live ? (int)(quad / g.A) : 0, a = live ? (int)(quad - (int64_t)b * g.A) : 0;
    const Anchor an = anchor_of(g, b, a);
    float x[REG], p[REG], lse;
    load16<T>(reinterpret_cast<const T*>(pick(g.box, an.l)) + an.pix * pick(g.ldb, an.l) + side * REG, x);
    const float e = softmax_expect(x, p, &lse);
    float d[4];
    quad_gather(e, d);
    if (!live) return;
    float* yo = y + (int64_t)b * (4 + g.nc) * g.A + a;
    const float x1 = an.ax - d[0], y1 = an.ay - d[1], x2 = an.ax + d[2], y2 = an.ay + d[3];
    const float box[4] = {(x1 + x2) * 0.5f * an.s, (y1 + y2) * 0.5f * an.s, (x2 - x1) * an.s, (y2 - y1) * an.s};
    yo[(int64_t)side * g.A] = side == 0 ? box[0] : side == 1 ? box[1] : side == 2 ? box[2] : box[3];
    const T* cp = reinterpret_cast<const T*>(pick(g.cls, an.l)) + an.pix * pick(g.ldc, an.l);
    for (int c = side; c < g.nc; c += 4) yo[(int64_t)(4 + c) * g.A] = 1.0f / (1.0f + expf(-to_f32(cp[c])));
}

// ---- K2: alignment metric and overlaps for every (image, gt, anchor) (tal.py:118-160) -------------------------
template <typename T>
__global__ __launch_bounds__(256) void metric_kernel(LossGeom g, const float* __restrict__ targets, const float* __restrict__ pb,
                                                     float* __restrict__ ov, float* __restrict__ al, uint8_t* __restrict__ mpos, float alpha,
                                                     float beta) {
    const int b = blockIdx.y;
    const int a = blockIdx.x * 256 + threadIdx.x;
    if (a >= g.A) return;
    const Anchor an = anchor_of(g, b, a);
    const f32x4 pq = *reinterpret_cast<const f32x4*>(pb + ((int64_t)b * g.A + a) * 4);
    const float pd[4] = {pq[0] * an.s, pq[1] * an.s, pq[2] * an.s, pq[3] * an.s};
    const float px = an.ax * an.s, py = an.ay * an.s;
    const T* cp = reinterpret_cast<const T*>(pick(g.cls, an.l)) + an.pix * pick(g.ldc, an.l);
    for (int k = 0; k < g.G; ++k) {
        const float* tg = targets + ((int64_t)b * g.G + k) * 5;
        const float gt[4] = {tg[1], tg[2], tg[3], tg[4]};
        const bool valid = (gt[0] + gt[1] + gt[2] + gt[3]) > 0.f;
        const float dmin = fminf(fminf(px - gt[0], py - gt[1]), fminf(gt[2] - px, gt[3] - py));
        const bool inside = dmin > TAL_EPS;
        float o = 0.f, m = 0.f;
        if (valid && inside) {
            int lab = (int)tg[0];
            lab = lab < 0 ? 0 : (lab >= g.nc ? g.nc - 1 : lab);
            const float score = sigmoidf_(to_f32(cp[lab]));
            o = fmaxf(ciou_f(gt, pd), 0.f);
            // score^alpha * overlap^beta (reference alpha 0.5, beta 6.0): powf keeps other exponents exact too
            m = powf(score, alpha) * powf(o, beta);
        }
        const int64_t idx = ((int64_t)b * g.G + k) * g.A + a;
        ov[idx] = o;
        al[idx] = m;
        mpos[idx] = 0;
    }
}

struct Best {
    float v;
    int i;
};
__device__ __forceinline__ Best better(Best x, Best y) { return (y.v > x.v || (y.v == x.v && y.i < x.i)) ? y : x; }
__device__ __forceinline__ Best block_best(Best mine, Best* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Best other{__shfl_xor(mine.v, o, 64), __shfl_xor(mine.i, o, 64)};
        mine = better(mine, other);
    }
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[wave] = mine;
    __syncthreads();
    Best r = sh[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = better(r, sh[w]);
    return r;
}

// ---- K3: top-k anchors per gt by alignment metric (tal.py:198-229); one workgroup per (image, gt) --------------
__global__ __launch_bounds__(256) void topk_kernel(LossGeom g, const float* __restrict__ targets, const float* __restrict__ al,
                                                   uint8_t* __restrict__ mpos, int topk) {
    __shared__ Best sh[4];
    const int bg = blockIdx.x;
    const int b = bg / g.G;
    const float* tg = targets + (int64_t)bg * 5;
    const float gt[4] = {tg[1], tg[2], tg[3], tg[4]};
    if (!((gt[0] + gt[1] + gt[2] + gt[3]) > 0.f)) return;  // padding row: selects nothing (uniform across the block)
    const float* row = al + (int64_t)bg * g.A;
    Best prev{__builtin_inff(), -1};
    // the gt's metric row is read ONCE into registers when it fits (A <= 256 * TOPK_NV: 8400 anchors at 640x640 are 33 values per
    // thread); the k selection rounds then only scan registers.  (Ten passes over the row in L2 took 71 us of the 0.18 ms loss.)
    constexpr int TOPK_NV = 40;
    const bool inreg = g.A <= 256 * TOPK_NV;
    float vals[TOPK_NV];
    if (inreg) {
#pragma unroll
        for (int i = 0; i < TOPK_NV; ++i) {
            const int a = threadIdx.x + 256 * i;
            vals[i] = a < g.A ? row[a] : -__builtin_inff();  // (never beats the initial candidate of value -1)
        }
    }
    for (int r = 0; r < topk && r < g.A; ++r) {
        Best mine{-1.0f, 0x7fffffff};
        if (inreg) {
#pragma unroll
            for (int i = 0; i < TOPK_NV; ++i) {
                const int a = threadIdx.x + 256 * i;
                const float v = vals[i];
                const bool cand = v < prev.v || (v == prev.v && a > prev.i);
                if (cand) mine = better(mine, Best{v, a});
            }
        } else {
            for (int a = threadIdx.x; a < g.A; a += 256) {
                const float v = row[a];
                // candidates: strictly after the previous pick in (value desc, index asc) order
                const bool cand = v < prev.v || (v == prev.v && a > prev.i);
                if (cand) mine = better(mine, Best{v, a});
            }
        }
        const Best win = block_best(mine, sh);
        if (win.i == 0x7fffffff) break;
        if (threadIdx.x == 0) {
            const Anchor an = anchor_of(g, b, win.i);
            const float px = an.ax * an.s, py = an.ay * an.s;
            const float dmin = fminf(fminf(px - gt[0], py - gt[1]), fminf(gt[2] - px, gt[3] - py));
            if (dmin > TAL_EPS) mpos[(int64_t)bg * g.A + win.i] = 1;
        }
        prev = win;
    }
}

// ---- K4: one gt per anchor; an anchor claimed by several gts goes to the highest overlap (tal.py:305-327) -----
__global__ __launch_bounds__(256) void assign_kernel(LossGeom g, const float* __restrict__ ov, const uint8_t* __restrict__ mpos,
                                                     int* __restrict__ gidx) {
    const int b = blockIdx.y;
    const int a = blockIdx.x * 256 + threadIdx.x;
    if (a >= g.A) return;
    int cnt = 0, first = -1, best = 0;
    float bo = -1.0f;
    for (int k = 0; k < g.G; ++k) {
        const int64_t idx = ((int64_t)b * g.G + k) * g.A + a;
        if (mpos[idx]) {
            ++cnt;
            if (first < 0) first = k;
        }
        const float o = ov[idx];
        if (o > bo) {
            bo = o;
            best = k;
        }
    }
    gidx[(int64_t)b * g.A + a] = cnt > 1 ? best : first;
}

// ---- K5: per gt, the largest metric and overlap among the anchors assigned to it (tal.py:96-100) ---------------
__global__ __launch_bounds__(256) void posmax_kernel(LossGeom g, const float* __restrict__ ov, const float* __restrict__ al,
                                                     const int* __restrict__ gidx, float* __restrict__ pa, float* __restrict__ po) {
    __shared__ float sh[2][4];
    const int bg = blockIdx.x;
    const int b = bg / g.G, k = bg - b * g.G;
    float ma = 0.f, mo = 0.f;
    for (int a = threadIdx.x; a < g.A; a += 256) {
        if (gidx[(int64_t)b * g.A + a] == k) {
            ma = fmaxf(ma, al[(int64_t)bg * g.A + a]);
            mo = fmaxf(mo, ov[(int64_t)bg * g.A + a]);
        }
    }
    ma = wave_max(ma);
    mo = wave_max(mo);
    if ((threadIdx.x & 63) == 0) {
        sh[0][threadIdx.x >> 6] = ma;
        sh[1][threadIdx.x >> 6] = mo;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        pa[bg] = fmaxf(fmaxf(sh[0][0], sh[0][1]), fmaxf(sh[0][2], sh[0][3]));
        po[bg] = fmaxf(fmaxf(sh[1][0], sh[1][1]), fmaxf(sh[1][2], sh[1][3]));
    }
}

// ---- K6: per anchor target box (grid units), weight (= sum of target scores) and label; partial sums of weight --
__global__ __launch_bounds__(256) void finalize_kernel(LossGeom g, const float* __restrict__ targets, const float* __restrict__ al,
                                                       const int* __restrict__ gidx, const float* __restrict__ pa, const float* __restrict__ po,
                                                       float* __restrict__ tgt, float* __restrict__ wgt, int* __restrict__ lab,
                                                       float* __restrict__ tss_part) {
    __shared__ float sh[4];
    const int b = blockIdx.y;
    const int a = blockIdx.x * 256 + threadIdx.x;
    float w = 0.f;
    if (a < g.A) {
        const int64_t ia = (int64_t)b * g.A + a;
        const int k = gidx[ia];
        f32x4 tb = {0.f, 0.f, 0.f, 0.f};
        int lb = -1;
        if (k >= 0) {
            const int64_t bg = (int64_t)b * g.G + k;
            const float* tg = targets + bg * 5;
            const Anchor an = anchor_of(g, b, a);
            w = al[bg * g.A + a] * po[bg] / (pa[bg] + TAL_EPS);
            lb = (int)tg[0];
            lb = lb < 0 ? 0 : (lb >= g.nc ? g.nc - 1 : lb);
            tb = f32x4{tg[1] / an.s, tg[2] / an.s, tg[3] / an.s, tg[4] / an.s};
        }
        *reinterpret_cast<f32x4*>(tgt + ia * 4) = tb;
        wgt[ia] = w;
        lab[ia] = lb;
    }
    w = wave_sum(w);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) tss_part[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// ---- K7: loss sums (GRAD = false) or gradients with respect to the maps (GRAD = true) --------------------------
// four lanes per anchor, one per box side; classes are strided over the four lanes
template <typename T, bool GRAD>
__global__ __launch_bounds__(256) void loss_kernel(LossGeom g, const float* __restrict__ tgt, const float* __restrict__ wgt,
                                                   const int* __restrict__ lab, const float* __restrict__ scal, const float* __restrict__ gout,
                                                   const float* __restrict__ gscale, float* __restrict__ part) {
    __shared__ float sh[3][4];
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t quad = t >> 2;
    const int side = (int)(t & 3);
    const bool live = quad < (int64_t)g.B * g.A;
    const int b = live ? (int)(quad / g.A) : 0, a = live ? (int)(quad - (int64_t)b * g.A) : 0;
    const Anchor an = anchor_of(g, b, a);
    const int64_t ia = (int64_t)b * g.A + a;
    float x[REG], p[REG], lse;
    load16<T>(reinterpret_cast<const T*>(pick(g.box, an.l)) + an.pix * pick(g.ldb, an.l) + side * REG, x);
    const float e = softmax_expect(x, p, &lse);
    float d[4];
    quad_gather(e, d);
    const float w = wgt[ia];
    const int lb = lab[ia];
    const f32x4 tq = *reinterpret_cast<const f32x4*>(tgt + ia * 4);
    const float tb[4] = {tq[0], tq[1], tq[2], tq[3]};
    const float pbx[4] = {an.ax - d[0], an.ay - d[1], an.ax + d[2], an.ay + d[3]};
    float s_box = 0.f, s_cls = 0.f, s_dfl = 0.f;
    // DFL target of this side (loss.py:101-104, 75-83)
    const float raw_t = side == 0 ? an.ax - tb[0] : side == 1 ? an.ay - tb[1] : side == 2 ? tb[2] - an.ax : tb[3] - an.ay;
    const float tt = fminf(fmaxf(raw_t, 0.f), (float)(REG - 1) - 0.01f);
    const int tl = (int)tt;
    const float wl = (float)(tl + 1) - tt, wr = 1.0f - wl;
    float inv_t = 0.f, g_box = 0.f, g_cls = 0.f, g_dfl = 0.f;
    if (GRAD) {
        inv_t = 1.0f / fmaxf(scal[0], 1.0f);
        g_box = gout[0] * (gscale ? gscale[0] : 1.0f) * inv_t;
        g_cls = gout[1] * (gscale ? gscale[1] : 1.0f) * inv_t;
        g_dfl = gout[2] * (gscale ? gscale[2] : 1.0f) * inv_t;
    }
    if (lb >= 0) {  // foreground anchor
        if (!GRAD) {
            if (side == 0) s_box = (1.0f - ciou_f(pbx, tb)) * w;
            float xl = 0.f, xr = 0.f;
#pragma unroll
            for (int j = 0; j < REG; ++j) {
                xl = j == tl ? x[j] : xl;
                xr = j == tl + 1 ? x[j] : xr;
            }
            s_dfl = ((lse - xl) * wl + (lse - xr) * wr) * 0.25f * w;
        } else {
            float gc[4];
            (void)ciou_grad(pbx, tb, gc);
            // d loss_box / d dist_k: x1 = ax - d0, y1 = ay - d1, x2 = ax + d2, y2 = ay + d3 ; loss = (1 - ciou) * w
            const float gd = (side < 2 ? gc[side] : -gc[side]) * w * g_box;
            const float kd = 0.25f * w * g_dfl;
            float o[REG];
#pragma unroll
            for (int j = 0; j < REG; ++j) {
                const float hot = (j == tl ? wl : 0.f) + (j == tl + 1 ? wr : 0.f);
                o[j] = gd * p[j] * ((float)j - e) + kd * (p[j] - hot);
            }
            if (live) store16<T>(reinterpret_cast<T*>(pick(g.dbox, an.l)) + an.pix * pick(g.lddb, an.l) + side * REG, o);
        }
    } else if (GRAD && live) {
        float o[REG];
#pragma unroll
        for (int j = 0; j < REG; ++j) o[j] = 0.f;
        store16<T>(reinterpret_cast<T*>(pick(g.dbox, an.l)) + an.pix * pick(g.lddb, an.l) + side * REG, o);
    }
    // classification: BCE with logits against target score (label == c) * weight  (loss.py:233)
    if (live) {
        const T* cp = reinterpret_cast<const T*>(pick(g.cls, an.l)) + an.pix * pick(g.ldc, an.l);
        for (int c = side; c < g.nc; c += 4) {
            const float xv = to_f32(cp[c]);
            const float tv = c == lb ? w : 0.f;
            if (!GRAD) {
                s_cls += fmaxf(xv, 0.f) - xv * tv + log1pf(__expf(-fabsf(xv)));
            } else {
                T* dp = reinterpret_cast<T*>(pick(g.dcls, an.l)) + an.pix * pick(g.lddc, an.l);
                dp[c] = from_f32<T>((sigmoidf_(xv) - tv) * g_cls);
            }
        }
        if (GRAD) {  // zero padding channels of a padded gradient buffer (the consumer's weight-gradient GEMM reads them)
            T* dp = reinterpret_cast<T*>(pick(g.dcls, an.l)) + an.pix * pick(g.lddc, an.l);
            for (int c = g.nc + side; c < g.dcw; c += 4) dp[c] = from_f32<T>(0.f);
        }
    }
    if (!GRAD) {
        if (!live) s_box = s_cls = s_dfl = 0.f;
        s_box = wave_sum(s_box);
        s_cls = wave_sum(s_cls);
        s_dfl = wave_sum(s_dfl);
        if ((threadIdx.x & 63) == 0) {
            sh[0][threadIdx.x >> 6] = s_box;
            sh[1][threadIdx.x >> 6] = s_cls;
            sh[2][threadIdx.x >> 6] = s_dfl;
        }
        __syncthreads();
        if (threadIdx.x < 3) part[(int64_t)blockIdx.x * 3 + threadIdx.x] = (sh[threadIdx.x][0] + sh[threadIdx.x][1]) + (sh[threadIdx.x][2] + sh[threadIdx.x][3]);
    }
}

// ---- K8: fixed-order final sums: scal[0] = sum of target scores, loss[k] = sum_k / max(scal[0], 1) ------------
// out_scale (optional, [6]): loss[k] = raw_k * out_scale[k] and loss[3 + k] = raw_k * out_scale[3 + k] - the criterion's two results
// (loss * gains * batch for backward, loss * gains for logging: reference loss.py:250-255) without elementwise launches behind this one
__global__ __launch_bounds__(256) void loss_final_kernel(const float* __restrict__ tss_part, int n_tss, const float* __restrict__ part, int n_part,
                                                         float* __restrict__ scal, float* __restrict__ loss, const float* __restrict__ out_scale) {
    __shared__ double sh[4][256];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < n_tss; i += 256) acc[3] += (double)tss_part[i];
    for (int i = threadIdx.x; i < n_part; i += 256) {
        acc[0] += (double)part[(int64_t)i * 3 + 0];
        acc[1] += (double)part[(int64_t)i * 3 + 1];
        acc[2] += (double)part[(int64_t)i * 3 + 2];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) sh[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
#pragma unroll
            for (int k = 0; k < 4; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double tss = sh[3][0];
        const double den = tss > 1.0 ? tss : 1.0;
        scal[0] = (float)tss;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float raw = (float)(sh[k][0] / den);
            if (out_scale) {
                loss[k] = raw * out_scale[k];
                loss[3 + k] = raw * out_scale[3 + k];
            } else {
                loss[k] = raw;
            }
        }
    }
}

struct Carve {
    char* p;
    size_t used;
    template <typename U> U* take(size_t n) {
        U* r = reinterpret_cast<U*>(p + used);
        used += (n * sizeof(U) + 255) / 256 * 256;
        return r;
    }
};

struct StateView {
    float *tgt, *wgt, *scal;
    int* lab;
};
static StateView carve_state(void* state, int64_t BA, size_t* bytes) {
    Carve c{reinterpret_cast<char*>(state), 0};
    StateView v;
    v.tgt = c.take<float>((size_t)BA * 4);
    v.wgt = c.take<float>((size_t)BA);
    v.lab = c.take<int>((size_t)BA);
    v.scal = c.take<float>(4);
    *bytes = c.used;
    return v;
}

static int fill_geom(LossGeom& g, int32_t nl, const ymi_tensor* box, const ymi_tensor* cls, const float* strides, int64_t G, const char* what) {
    YMI_CHECK_ARG(nl >= 1 && nl <= LOSS_MAXL && box && cls && strides, "%s: 1..%d levels", what, LOSS_MAXL);
    g = LossGeom{};
    g.nl = nl;
    int off = 0;
    for (int l = 0; l < nl; ++l) {
        YMI_CHECK_ARG(ymi_tensor_ok(&box[l]) && ymi_tensor_ok(&cls[l]), "%s: level %d tensors", what, l);
        YMI_CHECK_ARG(box[l].c == 4 * REG, "%s: box maps must have 4*16 channels (reg_max 16)", what);
        YMI_CHECK_ARG(box[l].n == box[0].n && cls[l].n == box[0].n && cls[l].h == box[l].h && cls[l].w == box[l].w && cls[l].c == cls[0].c &&
                          cls[l].dtype == box[0].dtype && box[l].dtype == box[0].dtype,
                      "%s: level %d shapes / dtypes", what, l);
        const int es = (int)ymi_esize(box[l].dtype);
        YMI_CHECK_ARG(((uintptr_t)box[l].data % 16) == 0 && (box[l].ld * es) % 16 == 0, "%s: box map alignment", what);
        g.box[l] = box[l].data;
        g.cls[l] = cls[l].data;
        g.ldb[l] = box[l].ld;
        g.ldc[l] = cls[l].ld;
        g.H[l] = (int)box[l].h;
        g.W[l] = (int)box[l].w;
        g.off[l] = off;
        g.stride[l] = strides[l];
        off += (int)(box[l].h * box[l].w);
    }
    for (int l = nl; l <= LOSS_MAXL; ++l) g.off[l] = off;
    g.B = (int)box[0].n;
    g.A = off;
    g.G = (int)G;
    g.nc = (int)cls[0].c;
    YMI_CHECK_ARG(G >= 1 && (int64_t)g.B * g.G * g.A < (1ll << 31), "%s: max_boxes >= 1 and B*G*A < 2^31", what);
    return YMI_OK;
}
}  // namespace

extern "C" int ymi_detect_targets(const float* batch_idx, const float* cls, const float* bboxes_xywhn, int64_t n, int64_t batch, int64_t max_boxes,
                                  float img_w, float img_h, float* out, void* stream) {
    YMI_CHECK_ARG(out && batch > 0 && max_boxes > 0 && n >= 0 && (n == 0 || (batch_idx && cls && bboxes_xywhn)), "detect_targets: args");
    const int total = (int)(batch * max_boxes);
    hipLaunchKernelGGL(targets_kernel, dim3((total + 127) / 128), dim3(128), 0, (hipStream_t)stream, batch_idx, cls, bboxes_xywhn, (int)n, (int)batch,
                       (int)max_boxes, img_w, img_h, out);
    YMI_CHECK_LAUNCH("detect_targets");
    return YMI_OK;
}

extern "C" int ymi_detect_loss_sizes(int64_t batch, int64_t anchors, int64_t max_boxes, size_t* state_bytes, size_t* workspace_bytes) {
    YMI_CHECK_ARG(batch > 0 && anchors > 0 && max_boxes > 0 && state_bytes && workspace_bytes, "detect_loss_sizes: args");
    const int64_t BA = batch * anchors, BGA = BA * max_boxes, BG = batch * max_boxes;
    (void)carve_state(nullptr, BA, state_bytes);
    Carve c{nullptr, 0};
    c.take<float>((size_t)BA * 4);      // predicted boxes
    c.take<float>((size_t)BGA);         // overlaps
    c.take<float>((size_t)BGA);         // alignment metric
    c.take<uint8_t>((size_t)BGA);       // positive mask
    c.take<int>((size_t)BA);            // gt index per anchor
    c.take<float>((size_t)BG);          // per-gt max metric
    c.take<float>((size_t)BG);          // per-gt max overlap
    c.take<float>((size_t)(BA / 256 + batch + 1));      // weight partial sums
    c.take<float>((size_t)(BA * 4 / 256 + 2) * 3);      // loss partial sums
    *workspace_bytes = c.used;
    return YMI_OK;
}

extern "C" int ymi_detect_loss_fwd(int32_t nl, const ymi_tensor* box_maps, const ymi_tensor* cls_maps, const float* strides, const float* targets,
                                   int64_t max_boxes, int32_t topk, float alpha, float beta, const float* out_scale, float* loss_out, void* state,
                                   size_t state_bytes, void* workspace, size_t workspace_bytes, void* stream) {
    LossGeom g;
    int rc = fill_geom(g, nl, box_maps, cls_maps, strides, max_boxes, "detect_loss_fwd");
    if (rc) return rc;
    YMI_CHECK_ARG(targets && loss_out && state && workspace && topk >= 1, "detect_loss_fwd: null argument");
    size_t need_s = 0, need_w = 0;
    ymi_detect_loss_sizes(g.B, g.A, g.G, &need_s, &need_w);
    if (state_bytes < need_s || workspace_bytes < need_w) {
        ymi_set_error("detect_loss_fwd: state %zu < %zu or workspace %zu < %zu bytes", state_bytes, need_s, workspace_bytes, need_w);
        return YMI_EWORKSPACE;
    }
    const int64_t BA = (int64_t)g.B * g.A, BGA = BA * g.G, BG = (int64_t)g.B * g.G;
    size_t sb;
    StateView st = carve_state(state, BA, &sb);
    Carve c{reinterpret_cast<char*>(workspace), 0};
    float* pb = c.take<float>((size_t)BA * 4);
    float* ov = c.take<float>((size_t)BGA);
    float* al = c.take<float>((size_t)BGA);
    uint8_t* mpos = c.take<uint8_t>((size_t)BGA);
    int* gidx = c.take<int>((size_t)BA);
    float* pa = c.take<float>((size_t)BG);
    float* po = c.take<float>((size_t)BG);
    float* tss_part = c.take<float>((size_t)(BA / 256 + g.B + 1));
    float* part = c.take<float>((size_t)(BA * 4 / 256 + 2) * 3);
    hipStream_t s = (hipStream_t)stream;
    const bool bf = box_maps[0].dtype == YMI_BF16;
    const unsigned qblocks = (unsigned)((BA * 4 + 255) / 256);
    const dim3 agrid((unsigned)((g.A + 255) / 256), (unsigned)g.B);
    if (bf) hipLaunchKernelGGL(decode_kernel<bf16_t>, dim3(qblocks), dim3(256), 0, s, g, pb);
    else hipLaunchKernelGGL(decode_kernel<float>, dim3(qblocks), dim3(256), 0, s, g, pb);
    if (bf) hipLaunchKernelGGL(metric_kernel<bf16_t>, agrid, dim3(256), 0, s, g, targets, pb, ov, al, mpos, alpha, beta);
    else hipLaunchKernelGGL(metric_kernel<float>, agrid, dim3(256), 0, s, g, targets, pb, ov, al, mpos, alpha, beta);
    hipLaunchKernelGGL(topk_kernel, dim3((unsigned)BG), dim3(256), 0, s, g, targets, al, mpos, (int)topk);
    hipLaunchKernelGGL(assign_kernel, agrid, dim3(256), 0, s, g, ov, mpos, gidx);
    hipLaunchKernelGGL(posmax_kernel, dim3((unsigned)BG), dim3(256), 0, s, g, ov, al, gidx, pa, po);
    hipLaunchKernelGGL(finalize_kernel, agrid, dim3(256), 0, s, g, targets, al, gidx, pa, po, st.tgt, st.wgt, st.lab, tss_part);
    if (bf) hipLaunchKernelGGL((loss_kernel<bf16_t, false>), dim3(qblocks), dim3(256), 0, s, g, st.tgt, st.wgt, st.lab, st.scal, nullptr, nullptr, part);
    else hipLaunchKernelGGL((loss_kernel<float, false>), dim3(qblocks), dim3(256), 0, s, g, st.tgt, st.wgt, st.lab, st.scal, nullptr, nullptr, part);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, s, tss_part, (int)(agrid.x * agrid.y), part, (int)qblocks, st.scal, loss_out, out_scale);
    YMI_CHECK_LAUNCH("detect_loss_fwd");
    return YMI_OK;
}

extern "C" int ymi_detect_loss_bwd(int32_t nl, const ymi_tensor* box_maps, const ymi_tensor* cls_maps, const float* strides, const void* state,
                                   size_t state_bytes, const float* grad_loss, const float* grad_scale, const ymi_tensor* dbox_maps, const ymi_tensor* dcls_maps,
                                   void* stream) {
    LossGeom g;
    int rc = fill_geom(g, nl, box_maps, cls_maps, strides, 1, "detect_loss_bwd");
    if (rc) return rc;
    YMI_CHECK_ARG(state && grad_loss && dbox_maps && dcls_maps, "detect_loss_bwd: null argument");
    const int64_t BA = (int64_t)g.B * g.A;
    size_t sb;
    StateView st = carve_state(const_cast<void*>(state), BA, &sb);
    YMI_CHECK_ARG(state_bytes >= sb, "detect_loss_bwd: state too small");
    for (int l = 0; l < nl; ++l) {
        const ymi_tensor &dc = dcls_maps[l], &cm = cls_maps[l];
        YMI_CHECK_ARG(ymi_tensor_ok(&dbox_maps[l]) && ymi_tensor_ok(&dc) && ymi_same_shape(&dbox_maps[l], &box_maps[l]) && dc.n == cm.n && dc.h == cm.h &&
                          dc.w == cm.w && dc.c >= cm.c && dc.c == dcls_maps[0].c && dbox_maps[l].dtype == box_maps[l].dtype && dc.dtype == cm.dtype,
                      "detect_loss_bwd: gradient map %d", l);
        const int es = (int)ymi_esize(dbox_maps[l].dtype);
        YMI_CHECK_ARG(((uintptr_t)dbox_maps[l].data % 16) == 0 && (dbox_maps[l].ld * es) % 16 == 0, "detect_loss_bwd: gradient map alignment");
        g.dbox[l] = dbox_maps[l].data;
        g.dcls[l] = dcls_maps[l].data;
        g.dcw = (int)dcls_maps[l].c;
        g.lddb[l] = dbox_maps[l].ld;
        g.lddc[l] = dcls_maps[l].ld;
    }
    const unsigned qblocks = (unsigned)((BA * 4 + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    if (box_maps[0].dtype == YMI_BF16)
        hipLaunchKernelGGL((loss_kernel<bf16_t, true>), dim3(qblocks), dim3(256), 0, s, g, st.tgt, st.wgt, st.lab, st.scal, grad_loss, grad_scale, nullptr);
    else
        hipLaunchKernelGGL((loss_kernel<float, true>), dim3(qblocks), dim3(256), 0, s, g, st.tgt, st.wgt, st.lab, st.scal, grad_loss, grad_scale, nullptr);
    YMI_CHECK_LAUNCH("detect_loss_bwd");
    return YMI_OK;
}

extern "C" int ymi_detect_decode(int32_t nl, const ymi_tensor* box_maps, const ymi_tensor* cls_maps, const float* strides, float* y, void* stream) {
    LossGeom g;
    int rc = fill_geom(g, nl, box_maps, cls_maps, strides, 1, "detect_decode");
    if (rc) return rc;
    YMI_CHECK_ARG(y, "detect_decode: null output");
    const int64_t BA = (int64_t)g.B * g.A;
    const unsigned qblocks = (unsigned)((BA * 4 + 255) / 256);
    if (box_maps[0].dtype == YMI_BF16) hipLaunchKernelGGL(infer_decode_kernel<bf16_t>, dim3(qblocks), dim3(256), 0, (hipStream_t)stream, g, y);
    else hipLaunchKernelGGL(infer_decode_kernel<float>, dim3(qblocks), dim3(256), 0, (hipStream_t)stream, g, y);
    YMI_CHECK_LAUNCH("detect_decode");
    return YMI_OK;
}
