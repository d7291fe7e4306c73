// SwinBlock's second half as ONE kernel per direction:  out = x + fc2(gelu(fc1(LayerNorm2(x))))   (reference nn/modules/swin_block.py:31-35,53)
//
// The unfused path (igemm.hip: ymi_swin_mlp_fwd / _bwd_data) runs fc1 and fc2 as two token GEMMs that stream the [T, 4C] hidden matrix
// (115 MB at the model's 56,448 x 1024) through HBM: written twice, read once in the forward, read and written again in the backward -
// 87 + 54 us forward and 107 + 54 us of data gradients per block against 18 us bounds (profiles/r04_per_launch_bounds.txt).  Here the
// hidden activations of a token never leave the registers of the wave that owns the token:
//
//  * a workgroup = 128 tokens, a wave = 32 tokens (256 threads, two workgroups per CU).  Waves split ROWS only, so nothing but the weights
//    is shared: W1 / W2 stream through a two-stage LDS ring in chunks of 32 hidden units (16 KB + 16 KB per chunk, filled by LDS-DMA),
//    one barrier per chunk.
//  * LayerNorm-2 is the prologue: a lane holds half a token row (16 x 16 bytes), the statistics need one cross-half exchange, and the
//    normalised row IS the B operand of fc1 (v_mfma_f32_32x32x16_bf16, operands swapped: weights are A, tokens are B, so the 32 x 32
//    result holds a token per lane and 16 hidden units in its registers).
//  * bias + exact-erf GELU act on those registers; rounded to bf16 they are - with no lane movement and no LDS - the B operand of fc2
//    (cdna_hip_programming.md, "An accumulator tile as the next MFMA's operand"): the k order inside a 16-deep step is permuted
//    (element j of lane half h is hidden unit 16 s + 8 (j >> 2) + 4 h + (j & 3)), so W2 is packed with the same permutation.
//  * fc2 accumulates the [32 tokens x 256 channels] output tile in 128 accumulator registers over all 32 chunks; the epilogue adds bias
//    and the skip (the block's input, re-read from L2) before the single bf16 rounding and leaves through LDS in whole 512-byte rows.
//  * training: the normalised tokens u (weight gradient of fc1), the LayerNorm statistics and the bf16 pre-activations are stored once;
//    the pre-activations in the register order of the kernel (a private layout the backward kernel reads back with 512-byte contiguous
//    wave accesses).  The backward kernel recomputes gelu / gelu' from them, forms d_pre = (d_out W2) * gelu'(pre) and d_u = d_pre W1 in
//    the same one-pass form, and writes post = gelu(pre) and d_pre row-major for the two weight-gradient GEMMs.
//
// Arithmetic and rounding points are those of the unfused path (u, pre, post, out stored / consumed as bf16; f32 accumulation), so the
// bf16 parity bounds of tests/test_gpu_bf16_matched.py are unchanged.  bf16, C = 256, hidden a multiple of 32; other shapes keep the unfused path.
#include "common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int MLP_C = 256;         // channels
constexpr int MLP_BM = 128;        // tokens per workgroup (4 waves x 32)
constexpr int MLP_HC = 32;         // hidden units per chunk
constexpr int MLP_STAGE = 32768;   // bytes per ring stage: [32][512] + [256][64]

// position p (0..31) of a chunk's permuted hidden order -> hidden unit of the chunk (see the header)
__host__ __device__ __forceinline__ int mlp_unit_of_pos(int p) {
    const int s = p >> 4, h = (p >> 3) & 1, j = p & 7;
    return 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
}

// ---- weight operands ---------------------------------------------------------------------------------------------------------------
// w1 [hidden][C], w2 [C][hidden] (float32, the nn.Linear layouts) -> four bf16 images of hidden * C elements each:
//   [0] w1p  [hidden][C]            fc1 forward:   A rows = hidden units, k = channel
//   [1] w2q  [hidden/32][C][32]     fc2 forward:   A rows = channels, k = the chunk's hidden units in permuted order
//   [2] w2t  [hidden][C]            d_post = d_out W2:  A rows = hidden units, k = channel  (w2 transposed)
//   [3] w1tq [hidden/32][C][32]     d_u = d_pre W1:     A rows = channels, k = permuted hidden units   (w1 transposed, chunked)
__global__ __launch_bounds__(256) void swin_mlp_pack_kernel(const float* __restrict__ w1, const float* __restrict__ w2, int C, int hidden, bf16_t* __restrict__ dst) {
    const int64_t n = (int64_t)C * hidden;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        // plain images: i = hid * C + c
        const int hid = (int)(i / C), c = (int)(i % C);
        dst[i] = (bf16_t)w1[i];
        dst[2 * n + i] = (bf16_t)w2[(int64_t)c * hidden + hid];
        // chunked images: i = (jc * C + cc) * 32 + p
        const int p = (int)(i & 31);
        const int cc = (int)((i >> 5) % C), jc = (int)((i >> 5) / C);
        const int hu = jc * 32 + mlp_unit_of_pos(p);
        dst[n + i] = (bf16_t)w2[(int64_t)cc * hidden + hu];
        dst[3 * n + i] = (bf16_t)w1[(int64_t)hu * C + cc];
    }
}

struct MlpFwdArgs {
    const bf16_t* x;  // block input tokens [T][ldx]: LayerNorm-2's input and the skip
    int64_t ldx;
    const float* gamma;
    const float* beta;
    float eps;
    const bf16_t* w1p;
    const bf16_t* w2q;
    const float* b1;
    const float* b2;
    bf16_t* u;  // [T][ldu] normalised tokens (training) or nullptr
    int64_t ldu;
    float* mean;
    float* rstd;
    bf16_t* pre;  // private layout (training) or nullptr
    bf16_t* out;
    int64_t ldo;
    int T, hidden;
};

__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    const bf16x2 v = {(bf16_t)a, (bf16_t)b};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf16_lo(uint32_t v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t v) { return __builtin_bit_cast(float, v & 0xffff0000u); }

// gelu(x) = max(x, 0) - |x| * (0.5 * (1 - erf(|x| / sqrt 2))), the complementary term by Abramowitz-Stegun 7.1.26 (common.h: erf_as; the
// same six-term form, with the halving folded into the coefficients): 15 issue slots + two transcendentals per element
__device__ __forceinline__ float gelu_fast(float x, float* half_erfc = nullptr, float* e_out = nullptr) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, ax, 1.0f));
    const float e = __builtin_amdgcn_exp2f(ax * ax * (-0.5f * 1.44269504088896340736f));
    float p = 0.5f * 1.061405429f;
    p = fmaf(p, t, -0.5f * 1.453152027f);
    p = fmaf(p, t, 0.5f * 1.421413741f);
    p = fmaf(p, t, -0.5f * 0.284496736f);
    p = fmaf(p, t, 0.5f * 0.254829592f);
    const float w = p * t * e;  // 0.5 * erfc(|x| / sqrt 2)
    if (half_erfc) *half_erfc = w;
    if (e_out) *e_out = e;
    return fmaf(-ax, w, fmaxf(x, 0.0f));
}

// issue the LDS-DMA pieces of hidden chunk jc into ring stage `stage` (8 per thread): images [32 rows][512 B] (chunk swizzle ^ (row & 15))
// and [256 rows][64 B] (chunk swizzle ^ ((row >> 2) & 3)); the destination is lane-linear, the swizzle lives in the source address
__device__ __forceinline__ void mlp_issue(const bf16_t* rows_img, const bf16_t* cols_img, int jc, char* stage, int tid, int wave) {
    const bf16_t* a = rows_img + (size_t)jc * 32 * MLP_C;
    const bf16_t* b = cols_img + (size_t)jc * MLP_C * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = i * 256 + tid, row = q >> 5, slot = q & 31, kc = slot ^ (row & 15);
        __builtin_amdgcn_global_load_lds((gptr_t)(a + row * MLP_C + kc * 8), (lptr_t)(stage + (i * 256 + wave * 64) * 16), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = i * 256 + tid, row = q >> 2, slot = q & 3, kc = slot ^ ((row >> 2) & 3);
        __builtin_amdgcn_global_load_lds((gptr_t)(b + row * 32 + kc * 8), (lptr_t)(stage + 16384 + (i * 256 + wave * 64) * 16), 16, 0, 0);
    }
}

// the output tile of a wave ([32 tokens][256 channels] in the 32x32 accumulator layout) -> `dst` rows as whole 512-byte lines, through the
// wave's own 16 KB of LDS: v[ct][r] + bias (+ the skip, gathered in the accumulator layout) is rounded once, dropped as 8-byte pieces into
// a [32][512 B] image (16-byte chunks swizzled by the row), read back row-contiguous and stored 1 KB per wave instruction
template <bool SKIP>
__device__ __forceinline__ void mlp_store_tile(f32x16 (&acc)[8], const float* bias, const bf16_t* skip, int64_t ldskip, bf16_t* dst, int64_t ldd, int row0, int T,
                                               char* img, int lane) {
    const int px = lane & 31, h = lane >> 5;
    const int row = row0 + px;
    const bool rok = row < T;
    const int rowc = rok ? row : T - 1;
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ch = 32 * ct + 8 * q + 4 * h;
            float v[4];
            const f32x4 bb = bias ? *reinterpret_cast<const f32x4*>(bias + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[ct][4 * q + r] + bb[r];
            if (SKIP) {
                const u32x2 s = *reinterpret_cast<const u32x2*>(skip + (int64_t)rowc * ldskip + ch);
                v[0] += bf16_lo(s[0]); v[1] += bf16_hi(s[0]); v[2] += bf16_lo(s[1]); v[3] += bf16_hi(s[1]);
            }
            u32x2 o;
            o[0] = pack_bf16x2(v[0], v[1]);
            o[1] = pack_bf16x2(v[2], v[3]);
            *reinterpret_cast<u32x2*>(img + px * 512 + 16 * ((4 * ct + q) ^ px) + 8 * h) = o;
        }
    }
    // (a wave's own LDS writes are visible to its own later reads in program order: no barrier)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = 2 * i + h, cl = px;  // this lane: row r of the wave's tile, 16-byte chunk cl
        const u32x4 val = *reinterpret_cast<const u32x4*>(img + r * 512 + 16 * (cl ^ r));
        if (row0 + r < T) *reinterpret_cast<u32x4*>(dst + (int64_t)(row0 + r) * ldd + cl * 8) = val;
    }
}

template <bool TRAIN>
__global__ __launch_bounds__(256, 2) void swin_mlp_fwd_kernel(MlpFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][32 KB] | b1 [hidden] floats
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 31, h = lane >> 5;
    const int row0 = blockIdx.x * MLP_BM + wave * 32;
    const int row = row0 + px;
    const bool rok = row < a.T;
    const int rowc = rok ? row : a.T - 1;
    float* b1s = reinterpret_cast<float*>(smem + 2 * MLP_STAGE);
    const int nch = a.hidden / MLP_HC;

    mlp_issue(a.w1p, a.w2q, 0, smem, tid, wave);
    for (int i = tid; i < a.hidden; i += 256) b1s[i] = a.b1[i];

    // ---- LayerNorm-2 (swin_block.py:53 norm2): lane (px, h) holds channels 16 s + 8 h + 0..7 of token row0 + px, s = 0..15 ----------
    bf16x8 uf[16];
    {
        const bf16_t* xr = a.x + (int64_t)rowc * a.ldx + 8 * h;
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) uf[s] = *reinterpret_cast<const bf16x8*>(xr + 16 * s);
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int i = 0; i < 8; ++i) sum += (float)uf[s][i];
        sum += __shfl_xor(sum, 32, 64);
        const float mu = sum * (1.0f / MLP_C);
        float q = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float d = (float)uf[s][i] - mu;
                q += d * d;
            }
        q += __shfl_xor(q, 32, 64);
        const float rs = rsqrtf(q * (1.0f / MLP_C) + a.eps);
        if (TRAIN && rok && h == 0) {
            a.mean[row] = mu;
            a.rstd[row] = rs;
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int c = 16 * s + 8 * h;
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(a.gamma + c), g1 = *reinterpret_cast<const f32x4*>(a.gamma + c + 4);
            const f32x4 e0 = *reinterpret_cast<const f32x4*>(a.beta + c), e1 = *reinterpret_cast<const f32x4*>(a.beta + c + 4);
            bf16x8 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o[i] = (bf16_t)(((float)uf[s][i] - mu) * rs * g0[i] + e0[i]);
                o[4 + i] = (bf16_t)(((float)uf[s][4 + i] - mu) * rs * g1[i] + e1[i]);
            }
            uf[s] = o;
            if (TRAIN && rok) *reinterpret_cast<bf16x8*>(a.u + (int64_t)row * a.ldu + c) = o;
        }
    }

    f32x16 acc[8];
#pragma unroll
    for (int ct = 0; ct < 8; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;

    const uint32_t x1 = (uint32_t)(16 * (h ^ (px & 15)));                      // fc1 image: chunk (2 s + h) ^ (row & 15), row = px
    const uint32_t k0 = (uint32_t)(16 * (h ^ ((px >> 2) & 3))), k1 = k0 ^ 32u;  // fc2 image: chunk (2 s + h) ^ ((row >> 2) & 3), row = 32 ct + px
    bf16_t* prew = TRAIN ? a.pre + ((((size_t)blockIdx.x * nch) * 4 + wave) * 4 * 64 + lane) * 4 : nullptr;

    for (int jc = 0; jc < nch; ++jc) {
        // chunk jc has landed (this wave's pieces: vmcnt; everybody's: the barrier), and every wave is done with the other stage
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        char* st = smem + (jc & 1) * MLP_STAGE;
        if (jc + 1 < nch) mlp_issue(a.w1p, a.w2q, jc + 1, smem + ((jc + 1) & 1) * MLP_STAGE, tid, wave);
        // ---- fc1: d1[hidden unit 8 (r >> 2) + 4 h + (r & 3)][token px] over K = 256 channels
        f32x16 d1;
#pragma unroll
        for (int r = 0; r < 16; ++r) d1[r] = 0.f;
        const char* w1a = st + px * 512;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(w1a + ((uint32_t)(32 * s) ^ x1));
            d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, uf[s], d1, 0, 0, 0);
        }
        // ---- bias, bf16 rounding of the pre-activation (saved), exact-erf GELU, bf16 again: the B operand of fc2
        uint32_t hf[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(b1s + jc * 32 + 8 * q + 4 * h);
            const uint32_t p01 = pack_bf16x2(d1[4 * q + 0] + bb[0], d1[4 * q + 1] + bb[1]);
            const uint32_t p23 = pack_bf16x2(d1[4 * q + 2] + bb[2], d1[4 * q + 3] + bb[3]);
            if (TRAIN) {
                u32x2 pv;
                pv[0] = p01;
                pv[1] = p23;
                *reinterpret_cast<u32x2*>(prew + ((size_t)jc * 4 * 4 + q) * 64 * 4) = pv;
            }
            hf[2 * q + 0] = pack_bf16x2(gelu_fast(bf16_lo(p01)), gelu_fast(bf16_hi(p01)));
            hf[2 * q + 1] = pack_bf16x2(gelu_fast(bf16_lo(p23)), gelu_fast(bf16_hi(p23)));
        }
        // ---- fc2: acc[ct][channel 32 ct + 8 (r >> 2) + 4 h + (r & 3)][token px] += over this chunk's 32 hidden units (two 16-deep steps)
        const char* w2a = st + 16384 + px * 64;
        bf16x8 hv[2];
        {
            u32x4 t0 = {hf[0], hf[1], hf[2], hf[3]}, t1 = {hf[4], hf[5], hf[6], hf[7]};
            hv[0] = __builtin_bit_cast(bf16x8, t0);
            hv[1] = __builtin_bit_cast(bf16x8, t1);
        }
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) {
            const bf16x8 wa = *reinterpret_cast<const bf16x8*>(w2a + ct * 2048 + k0);
            const bf16x8 wb = *reinterpret_cast<const bf16x8*>(w2a + ct * 2048 + k1);
            acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, hv[0], acc[ct], 0, 0, 0);
            acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb, hv[1], acc[ct], 0, 0, 0);
        }
    }
    __syncthreads();  // every wave has finished reading the ring: it becomes the output staging image
    mlp_store_tile<true>(acc, a.b2, a.x, a.ldx, a.out, a.ldo, row0, a.T, smem + wave * 16384, lane);
}

// ---- backward, data path ----------------------------------------------------------------------------------------------------------
// d_post = d_out W2 (K = 256), d_pre = bf16(d_post) * gelu'(pre) (the unfused epilogue's arithmetic, igemm.hip), d_u += d_pre W1 chunk by
// chunk: the same register hand-over as the forward (d_post's 32 x 32 tile -> the B operand of the second product).  post = gelu(pre)
// (bit for bit what the forward fed to fc2: the same expression on the same stored bf16 pre-activation) and d_pre leave row-major for
// the two weight-gradient GEMMs.
struct MlpBwdArgs {
    const bf16_t* dout;
    int64_t lddo;
    const bf16_t* w2t;
    const bf16_t* w1tq;
    const bf16_t* pre;  // private layout
    bf16_t* post;
    int64_t ldpost;
    bf16_t* dpre;
    int64_t lddpre;
    bf16_t* du;
    int64_t lddu;
    int T, hidden;
};

__global__ __launch_bounds__(256, 2) void swin_mlp_bwd_kernel(MlpBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][32 KB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 31, h = lane >> 5;
    const int row0 = blockIdx.x * MLP_BM + wave * 32;
    const int row = row0 + px;
    const bool rok = row < a.T;
    const int rowc = rok ? row : a.T - 1;
    const int nch = a.hidden / MLP_HC;

    mlp_issue(a.w2t, a.w1tq, 0, smem, tid, wave);
    bf16x8 df[16];
    {
        const bf16_t* xr = a.dout + (int64_t)rowc * a.lddo + 8 * h;
#pragma unroll
        for (int s = 0; s < 16; ++s) df[s] = *reinterpret_cast<const bf16x8*>(xr + 16 * s);
    }
    f32x16 acc[8];
#pragma unroll
    for (int ct = 0; ct < 8; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
    const uint32_t x1 = (uint32_t)(16 * (h ^ (px & 15)));
    const uint32_t k0 = (uint32_t)(16 * (h ^ ((px >> 2) & 3))), k1 = k0 ^ 32u;
    const bf16_t* prer = a.pre + ((((size_t)blockIdx.x * nch) * 4 + wave) * 4 * 64 + lane) * 4;
    bf16_t* postw = a.post + (int64_t)rowc * a.ldpost + 4 * h;
    bf16_t* dprew = a.dpre + (int64_t)rowc * a.lddpre + 4 * h;

    for (int jc = 0; jc < nch; ++jc) {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        char* st = smem + (jc & 1) * MLP_STAGE;
        if (jc + 1 < nch) mlp_issue(a.w2t, a.w1tq, jc + 1, smem + ((jc + 1) & 1) * MLP_STAGE, tid, wave);
        u32x2 pv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) pv[q] = *reinterpret_cast<const u32x2*>(prer + ((size_t)jc * 4 * 4 + q) * 64 * 4);
        f32x16 d1;
#pragma unroll
        for (int r = 0; r < 16; ++r) d1[r] = 0.f;
        const char* w1a = st + px * 512;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(w1a + ((uint32_t)(32 * s) ^ x1));
            d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, df[s], d1, 0, 0, 0);
        }
        uint32_t hf[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float po[4], dp[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = (r & 1) ? bf16_hi(pv[q][r >> 1]) : bf16_lo(pv[q][r >> 1]);
                float w, e;
                po[r] = gelu_fast(x, &w, &e);
                const float phi = x >= 0.f ? 1.0f - w : w;
                const float grad = fmaf(x * 0.39894228040143267794f, e, phi);
                dp[r] = to_f32(from_f32<bf16_t>(d1[4 * q + r])) * grad;
            }
            u32x2 o, d;
            o[0] = pack_bf16x2(po[0], po[1]);
            o[1] = pack_bf16x2(po[2], po[3]);
            d[0] = pack_bf16x2(dp[0], dp[1]);
            d[1] = pack_bf16x2(dp[2], dp[3]);
            if (rok) {
                *reinterpret_cast<u32x2*>(postw + jc * 32 + 8 * q) = o;
                *reinterpret_cast<u32x2*>(dprew + jc * 32 + 8 * q) = d;
            }
            hf[2 * q + 0] = d[0];
            hf[2 * q + 1] = d[1];
        }
        const char* w2a = st + 16384 + px * 64;
        bf16x8 hv[2];
        {
            u32x4 t0 = {hf[0], hf[1], hf[2], hf[3]}, t1 = {hf[4], hf[5], hf[6], hf[7]};
            hv[0] = __builtin_bit_cast(bf16x8, t0);
            hv[1] = __builtin_bit_cast(bf16x8, t1);
        }
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) {
            const bf16x8 wa = *reinterpret_cast<const bf16x8*>(w2a + ct * 2048 + k0);
            const bf16x8 wb = *reinterpret_cast<const bf16x8*>(w2a + ct * 2048 + k1);
            acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, hv[0], acc[ct], 0, 0, 0);
            acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb, hv[1], acc[ct], 0, 0, 0);
        }
    }
    __syncthreads();
    mlp_store_tile<false>(acc, nullptr, nullptr, 0, a.du, a.lddu, row0, a.T, smem + wave * 16384, lane);
}

// ---- host -------------------------------------------------------------------------------------------------------------------------
extern "C" int ymi_swin_ln_mlp_supported(int64_t c, int64_t hidden, int32_t dtype) {
    return dtype == YMI_BF16 && c == MLP_C && hidden % MLP_HC == 0 && hidden >= MLP_HC && hidden <= 8192;
}
// elements of the packed weight images (bfloat16) and of the private pre-activation buffer for `tokens` tokens
extern "C" int64_t ymi_swin_ln_mlp_pack_elems(int64_t c, int64_t hidden) { return 4 * c * hidden; }
extern "C" int64_t ymi_swin_ln_mlp_pre_elems(int64_t tokens, int64_t hidden) { return (tokens + MLP_BM - 1) / MLP_BM * MLP_BM * hidden; }

extern "C" int ymi_swin_ln_mlp_pack(const float* w1, const float* w2, int64_t c, int64_t hidden, void* packed, void* stream) {
    YMI_CHECK_ARG(w1 && w2 && packed && ymi_swin_ln_mlp_supported(c, hidden, YMI_BF16), "swin_ln_mlp_pack: C = %d, hidden a multiple of %d", MLP_C, MLP_HC);
    YMI_CHECK_ARG(((uintptr_t)packed & 15) == 0, "swin_ln_mlp_pack: 16-byte alignment");
    const int64_t n = c * hidden;
    hipLaunchKernelGGL(swin_mlp_pack_kernel, dim3((unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024)), dim3(256), 0, (hipStream_t)stream, w1, w2, (int)c, (int)hidden,
                       (bf16_t*)packed);
    YMI_CHECK_LAUNCH("swin_ln_mlp_pack");
    return YMI_OK;
}

extern "C" int ymi_swin_ln_mlp_fwd(const ymi_tensor* x, const float* gamma, const float* beta, float eps, const void* packed, const float* b1, const float* b2,
                                   int64_t hidden, const ymi_tensor* u, float* mean, float* rstd, void* pre, const ymi_tensor* out, void* stream) {
    YMI_CHECK_ARG(ymi_tensor_ok(x) && ymi_tensor_ok(out) && gamma && beta && packed && b1 && b2, "swin_ln_mlp_fwd: args");
    YMI_CHECK_ARG(ymi_swin_ln_mlp_supported(x->c, hidden, x->dtype) && out->dtype == x->dtype && out->c == x->c && ymi_pixels(out) == ymi_pixels(x),
                  "swin_ln_mlp_fwd: bfloat16 tokens of %d channels, hidden a multiple of %d", MLP_C, MLP_HC);
    const bool train = u != nullptr;
    if (train) YMI_CHECK_ARG(ymi_tensor_ok(u) && u->dtype == x->dtype && u->c == x->c && ymi_pixels(u) == ymi_pixels(x) && mean && rstd && pre, "swin_ln_mlp_fwd: saved tensors");
    YMI_CHECK_ARG(x->ld % 8 == 0 && out->ld % 8 == 0 && (!train || u->ld % 8 == 0) && ((((uintptr_t)x->data) | ((uintptr_t)out->data) | ((uintptr_t)packed) |
                  ((uintptr_t)gamma) | ((uintptr_t)beta) | ((uintptr_t)b1) | ((uintptr_t)b2) | (train ? ((uintptr_t)u->data | (uintptr_t)pre) : 0)) & 15) == 0,
                  "swin_ln_mlp_fwd: 16-byte alignment");
    const int64_t T = ymi_pixels(x);
    YMI_CHECK_ARG(T < (1ll << 31) && T * x->ld < (1ll << 31), "swin_ln_mlp_fwd: too large");
    const int64_t n = (int64_t)MLP_C * hidden;
    MlpFwdArgs a{};
    a.x = (const bf16_t*)x->data; a.ldx = x->ld;
    a.gamma = gamma; a.beta = beta; a.eps = eps;
    a.w1p = (const bf16_t*)packed; a.w2q = a.w1p + n;
    a.b1 = b1; a.b2 = b2;
    a.u = train ? (bf16_t*)u->data : nullptr; a.ldu = train ? u->ld : 0;
    a.mean = mean; a.rstd = rstd; a.pre = (bf16_t*)pre;
    a.out = (bf16_t*)out->data; a.ldo = out->ld;
    a.T = (int)T; a.hidden = (int)hidden;
    const size_t lds = 2 * MLP_STAGE + (size_t)hidden * sizeof(float);
    const dim3 grid((unsigned)((T + MLP_BM - 1) / MLP_BM));
    if (train) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(swin_mlp_fwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(swin_mlp_fwd_kernel<true>, grid, dim3(256), lds, (hipStream_t)stream, a);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(swin_mlp_fwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(swin_mlp_fwd_kernel<false>, grid, dim3(256), lds, (hipStream_t)stream, a);
    }
    YMI_CHECK_LAUNCH("swin_ln_mlp_fwd");
    return YMI_OK;
}

extern "C" int ymi_swin_ln_mlp_bwd_data(const ymi_tensor* dout, const void* packed, const void* pre, int64_t hidden, const ymi_tensor* post, const ymi_tensor* dpre,
                                        const ymi_tensor* du, void* stream) {
    YMI_CHECK_ARG(ymi_tensor_ok(dout) && ymi_tensor_ok(post) && ymi_tensor_ok(dpre) && ymi_tensor_ok(du) && packed && pre, "swin_ln_mlp_bwd_data: args");
    const int64_t T = ymi_pixels(dout);
    YMI_CHECK_ARG(ymi_swin_ln_mlp_supported(dout->c, hidden, dout->dtype) && post->dtype == dout->dtype && dpre->dtype == dout->dtype && du->dtype == dout->dtype &&
                      post->c == hidden && dpre->c == hidden && du->c == dout->c && ymi_pixels(post) == T && ymi_pixels(dpre) == T && ymi_pixels(du) == T,
                  "swin_ln_mlp_bwd_data: shapes");
    YMI_CHECK_ARG(dout->ld % 8 == 0 && du->ld % 8 == 0 && post->ld % 4 == 0 && dpre->ld % 4 == 0 &&
                      ((((uintptr_t)dout->data) | ((uintptr_t)du->data) | ((uintptr_t)packed) | ((uintptr_t)pre)) & 15) == 0 &&
                      ((((uintptr_t)post->data) | ((uintptr_t)dpre->data)) & 7) == 0,
                  "swin_ln_mlp_bwd_data: alignment");
    YMI_CHECK_ARG(T < (1ll << 31) && T * post->ld < (1ll << 31), "swin_ln_mlp_bwd_data: too large");
    const int64_t n = (int64_t)MLP_C * hidden;
    MlpBwdArgs a{};
    a.dout = (const bf16_t*)dout->data; a.lddo = dout->ld;
    a.w2t = (const bf16_t*)packed + 2 * n; a.w1tq = (const bf16_t*)packed + 3 * n;
    a.pre = (const bf16_t*)pre;
    a.post = (bf16_t*)post->data; a.ldpost = post->ld;
    a.dpre = (bf16_t*)dpre->data; a.lddpre = dpre->ld;
    a.du = (bf16_t*)du->data; a.lddu = du->ld;
    a.T = (int)T; a.hidden = (int)hidden;
    const size_t lds = 2 * MLP_STAGE;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(swin_mlp_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(swin_mlp_bwd_kernel, dim3((unsigned)((T + MLP_BM - 1) / MLP_BM)), dim3(256), lds, (hipStream_t)stream, a);
    YMI_CHECK_LAUNCH("swin_ln_mlp_bwd_data");
    return YMI_OK;
}
