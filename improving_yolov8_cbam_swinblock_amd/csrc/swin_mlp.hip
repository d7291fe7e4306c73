// SwinBlock's second half as ONE kernel per direction:  out = x + fc2(gelu(fc1(LayerNorm2(x))))   (reference nn/modules/swin_block.py:31-35,53)
//
// The unfused path (igemm.hip: ymi_swin_mlp_fwd / _bwd_data) runs fc1 and fc2 as two token GEMMs that stream the [T, 4C] hidden matrix
// (115 MB at the model's 56,448 x 1024) through HBM: written twice, read once in the forward, read and written again in the backward -
// 87 + 54 us forward and 107 + 54 us of data gradients per block against 18 us bounds (profiles/r04_per_launch_bounds.txt).  Here the
// hidden activations of a token never leave the registers of the wave that owns the token:
//
//  * a workgroup = 256 tokens, a wave = 32 tokens (512 threads, one workgroup per CU).  Waves split ROWS only, so nothing but the weights
//    is shared: W1 / W2 stream through a three-stage LDS ring in chunks of 32 hidden units (16 KB + 16 KB per chunk, filled by LDS-DMA);
//    the two halves of the workgroup run one phase apart (matrix phase beside vector phase on every SIMD: see the schedule note below).
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
#include <type_traits>

#include "common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N-1>{})
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

constexpr int MLP_C = 256;         // channels
constexpr int MLP_BM = 256;        // tokens per workgroup (8 waves x 32)
constexpr int MLP_HC = 32;         // hidden units per chunk
constexpr int MLP_STAGE = 32768;   // bytes per ring stage: two 16 KB fragment-major images
constexpr int MLP_LDS_TILE = 8 * 16384;  // the ring (3 stages) lives in the first 96 KB of the 128 KB the output staging needs

// Which hidden unit sits where is this file's choice (it fixes the row order of the fc1 weight image): the unit in row rho of a chunk's 32 x 32
// result tile is mlp_unit_of_row(rho), chosen so that the 16 results a lane holds (rows 8 q + 4 h + r of the MFMA's C/D layout) are 16
// CONSECUTIVE hidden units, 16 h + 4 q + r: the bias is four 16-byte loads, the saved pre-activations / post / d_pre leave as 16-byte stores,
// and the second product's k-step s takes from lane half h the units 16 h + 8 s + 0..7 - 16 contiguous bytes of the natural weight row.
__host__ __device__ __forceinline__ int mlp_unit_of_row(int rho) { return 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3); }
// position p (0..31) of the second product's k order (k-step s = p >> 4, lane half h = (p >> 3) & 1, element j = p & 7) -> hidden unit of the chunk
__host__ __device__ __forceinline__ int mlp_unit_of_pos(int p) { return 16 * ((p >> 3) & 1) + 8 * (p >> 4) + (p & 7); }

// ---- weight operands ---------------------------------------------------------------------------------------------------------------
// w1 [hidden][C], w2 [C][hidden] (float32, the nn.Linear layouts) -> four bf16 images of hidden * C elements each, all FRAGMENT-MAJOR: a
// chunk of 32 hidden units is 16 KB laid out as the MFMA A fragments the kernels read - [k-step or tile][lane 0..63][8 elements] - so the
// global image, the LDS image and the lane order coincide: LDS-DMA copies it linearly, a wave's ds_read_b128 of one fragment is 1 KB
// contiguous (conflict-free, no swizzle), and every LDS address is one per-lane base plus an immediate.
//   [0] w1f  rows form  [hidden/32][16 k-steps][64 lanes][8]:  A[row = lane & 31 (hidden unit mlp_unit_of_row(row))][k = 16 s + 8 (lane >> 5) + j (channel)] = w1
//   [1] w2f  cols form  [hidden/32][8 tiles][2 steps][64][8]:  A[row = 32 ct + (lane & 31) (channel)][k = position 16 s + 8 (lane >> 5) + j] = w2,
//                                                              position p <-> hidden unit mlp_unit_of_pos(p) of the chunk
//   [2] w2tf rows form of w2 transposed (d_post = d_out W2):   A[hidden unit][channel] = w2[channel][hidden unit]
//   [3] w1tf cols form of w1 transposed (d_u = d_pre W1):      A[channel][position] = w1[hidden unit of the position][channel]
__global__ __launch_bounds__(256) void swin_mlp_pack_kernel(const float* __restrict__ w1, const float* __restrict__ w2, int C, int hidden, bf16_t* __restrict__ dst) {
    const int64_t n = (int64_t)C * hidden;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
        const int64_t blk = i >> 9;  // 512-element fragment blocks
        {   // rows form: blk = jc * 16 + s
            const int s = (int)(blk & 15), jc = (int)(blk >> 4);
            const int hu = jc * 32 + mlp_unit_of_row(lane & 31), c = 16 * s + 8 * (lane >> 5) + j;
            dst[i] = (bf16_t)w1[(int64_t)hu * C + c];
            dst[2 * n + i] = (bf16_t)w2[(int64_t)c * hidden + hu];
        }
        {   // cols form: blk = (jc * 8 + ct) * 2 + s
            const int s = (int)(blk & 1), ct = (int)((blk >> 1) & 7), jc = (int)(blk >> 4);
            const int c = 32 * ct + (lane & 31), hu = jc * 32 + mlp_unit_of_pos(16 * s + 8 * (lane >> 5) + j);
            dst[n + i] = (bf16_t)w2[(int64_t)c * hidden + hu];
            dst[3 * n + i] = (bf16_t)w1[(int64_t)hu * C + c];
        }
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
// same six-term form with the halving folded into the coefficients and the argument scalings into the constants): 11 issue slots + two
// transcendentals per element - the vector phases of these kernels are bound by exactly this count.
__device__ __forceinline__ float gelu_fast(float x, float* half_erfc = nullptr, float* e_out = nullptr) {
#if (YMI_MLP_ABL & 1)
    if (half_erfc) *half_erfc = 0.25f;
    if (e_out) *e_out = 0.5f;
    return x;
#endif
    const float ax = fabsf(x);                                                                // (a source modifier, no instruction)
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, ax, 1.0f));
    const float s = ax * 0.84932180028801904272f;                                              // sqrt(0.5 log2 e) |x|
    const float e = __builtin_amdgcn_exp2f(-(s * s));                                          // e^(-x^2 / 2)
    float p = 0.5f * 1.061405429f;
    p = fmaf(p, t, -0.5f * 1.453152027f);
    p = fmaf(p, t, 0.5f * 1.421413741f);
    p = fmaf(p, t, -0.5f * 0.284496736f);
    p = fmaf(p, t, 0.5f * 0.254829592f);
    const float w = p * t * e;  // 0.5 * erfc(|x| / sqrt 2)
    if (half_erfc) *half_erfc = w;
    if (e_out) *e_out = e;
    float relu;
    asm("v_max_f32 %0, 0, %1" : "=v"(relu) : "v"(x));  // (fmaxf would first canonicalise x with a second v_max)
    return fmaf(-ax, w, relu);
}

// issue the LDS-DMA pieces of hidden chunk jc into ring stage `stage` (4 per thread): two 16 KB fragment-major images, copied linearly
__device__ __forceinline__ void mlp_issue(const bf16_t* rows_img, const bf16_t* cols_img, int jc, char* stage, int tid, int wave) {
#if (YMI_MLP_ABL & 8)
    return;
#endif
    const bf16_t* a = rows_img + (size_t)jc * 32 * MLP_C + tid * 8;
    const bf16_t* b = cols_img + (size_t)jc * 32 * MLP_C + tid * 8;
#pragma unroll
    for (int i = 0; i < 2; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(a + i * 4096), (lptr_t)(stage + (i * 512 + wave * 64) * 16), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(b + i * 4096), (lptr_t)(stage + 16384 + (i * 512 + wave * 64) * 16), 16, 0, 0);
}

// The two products of a chunk with the fragment reads kept AHEAD of the MFMAs.  hipcc places an LDS read right before its use, waits with
// lgkmcnt(0) and - at 250 live registers - recycles ONE fragment buffer: an exposed LDS round trip (~130 cycles) per 32-cycle MFMA.  So the
// reads and their counted waits are written out (as igemm.hip's K step): four fragment buffers rotate, three reads stay in flight behind
// every MFMA.  (Other LDS operations the compiler interleaves only make a counted wait stricter: LDS operations return in order.)
// diagnostic builds (-DYMI_MLP_ABL=mask, results wrong by design; tools/probes/r5_mlp_ablate.sh): bit 1 no GELU arithmetic, 2 no MFMAs, 4 no fragment
// reads, 8 no weight copies, 16 no pre-activation stores, 32 in-kernel phase stamps, 64 no s_setprio around the matrix phase
#ifndef YMI_MLP_ABL
#define YMI_MLP_ABL 0
#endif
#if (YMI_MLP_ABL & 4)
#define MLP_RD(F, ADDR, OFF) asm volatile("" : "+v"(F) : "v"(ADDR), "n"(OFF))
#else
#define MLP_RD(F, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(F) : "v"(ADDR), "n"(OFF))
#endif
#if (YMI_MLP_ABL & 2)
#define MLP_MFMA(A, B, C) (C)
#else
#define MLP_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0)
#endif
#define MLP_WAIT(N)                                                    \
    do {                                                               \
        __builtin_amdgcn_sched_barrier(0);                             \
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");     \
    } while (0)
__device__ __forceinline__ uint32_t lds_u32(const void* p) { return (uint32_t)(uintptr_t)(lptr_t)p; }
// d += A(rows image: 16 k-steps of 1 KB from `addr`) . b[0..15]
__device__ __forceinline__ void mlp_rows_product(uint32_t addr, const bf16x8 (&b)[16], f32x16& d) {
    bf16x8 f[4] = {};
    MLP_RD(f[0], addr, 0);
    MLP_RD(f[1], addr, 1024);
    MLP_RD(f[2], addr, 2048);
    MLP_RD(f[3], addr, 3072);
    static_for<0, 16>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        const uint32_t ad = addr;  // (an odr-use outside the asm operand: clang does not capture a variable it only sees there)
        MLP_WAIT((15 - s) < 3 ? (15 - s) : 3);
        asm volatile("" : "+v"(f[s & 3]));
        d = MLP_MFMA(f[s & 3], b[s], d);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (s + 4 < 16) MLP_RD(f[s & 3], ad, (s + 4) * 1024);
    });
}
// acc[ct] += A(cols image: 8 tiles x 2 steps of 1 KB from `addr`) . hv[0..1]
__device__ __forceinline__ void mlp_cols_product(uint32_t addr, const bf16x8 (&hv)[2], f32x16 (&acc)[8]) {
    bf16x8 f[4] = {};
    MLP_RD(f[0], addr, 0);
    MLP_RD(f[1], addr, 1024);
    MLP_RD(f[2], addr, 2048);
    MLP_RD(f[3], addr, 3072);
    static_for<0, 16>([&](auto ic) {
        constexpr int i = decltype(ic)::value;  // i = 2 ct + s
        const uint32_t ad = addr;
        MLP_WAIT((15 - i) < 3 ? (15 - i) : 3);
        asm volatile("" : "+v"(f[i & 3]));
        acc[i >> 1] = MLP_MFMA(f[i & 3], hv[i & 1], acc[i >> 1]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (i + 4 < 16) MLP_RD(f[i & 3], ad, (i + 4) * 1024);
    });
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

// Schedule (both kernels).  A workgroup is 8 waves = 256 tokens, one workgroup per CU; waves w and w + 4 share a SIMD.  A wave's chunk
// is a MATRIX phase (fc2 of the previous chunk + fc1 of this one: 32 MFMAs, fragment reads) followed by a VECTOR phase (bias / GELU /
// roundings / stores: ~270 VALU instructions), and the two halves of the workgroup (waves 0-3, waves 4-7) run ONE PHASE APART, every
// phase ending at a workgroup barrier: each SIMD always holds one wave in its matrix phase beside one in its vector phase - the
// arrangement MI355X_MICROARCH.md "Two waves per SIMD" describes - instead of two waves that drift into the same phase (the first form
// of this kernel, two independent 4-wave workgroups per CU, measured 24 % MFMA-busy with 37 % of the wave cycles issue-stalled:
// profiles/r05_swin_mlp_fused.txt).  Global phase t: half 0 runs matrix phases at even t, half 1 at odd t.  Weight chunk c (32 KB) is
// copied into ring stage c % 3 at the start of phase 2c - 2 (its previous tenant, chunk c - 3, was last read in phase 2c - 3), every
// wave retires its pieces before the barrier that ends phase 2c - 1, and the first read is in phase 2c.
__device__ __forceinline__ void mlp_barrier() { asm volatile("s_barrier" ::: "memory"); }
#if (YMI_MLP_ABL & 32)
// diagnostic build: s_memtime stamps of workgroup 100, waves 0 and 4, chunks 8..11, into the (otherwise unused) `mean` array of an evaluation-mode call:
// [wave half][chunk - 8][point 0..5] = matrix phase start | after the fc2 product | after the fc1 product | past the barrier | vector work done | past the barrier
#define MLP_STAMP(P)                                                                                                              \
    do {                                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                                        \
        asm volatile("" : "+v"(hv[0]), "+v"(hv[1]));                                                                              \
        if (!TRAIN && a.mean && blockIdx.x == 100 && (wave & 3) == 0 && lane == 0 && jc >= 8 && jc < 12)                         \
            reinterpret_cast<unsigned long long*>(a.mean)[(half * 4 + (jc - 8)) * 6 + (P)] = __builtin_amdgcn_s_memtime();           \
        __builtin_amdgcn_sched_barrier(0);                                                                                        \
    } while (0)
#else
#define MLP_STAMP(P) do { } while (0)
#endif

__global__ __launch_bounds__(512, 2) void swin_mlp_fwd_kernel(MlpFwdArgs a) {
    const bool TRAIN = a.u != nullptr;  // (kernel argument: uniform) save u, the LayerNorm statistics and the pre-activations
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [3 stages][32 KB] (the output staging image afterwards: 8 x 16 KB) | b1 [hidden] floats
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave >> 2;  // (waves w and w + 4 share a SIMD: pairing by wave & 1 or (wave >> 1) & 1 measured 25 % slower)
    const int px = lane & 31, h = lane >> 5;
    const int row0 = blockIdx.x * MLP_BM + wave * 32;
    const int row = row0 + px;
    const bool rok = row < a.T;
    const int rowc = rok ? row : a.T - 1;
    float* b1s = reinterpret_cast<float*>(smem + MLP_LDS_TILE);
    const int nch = a.hidden / MLP_HC;

    mlp_issue(a.w1p, a.w2q, 0, smem, tid, wave);
    if (nch > 1) mlp_issue(a.w1p, a.w2q, 1, smem + MLP_STAGE, tid, wave);
    for (int i = tid; i < a.hidden; i += 512) b1s[i] = a.b1[i];

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

    // saved pre-activations, private layout [tile][chunk][wave][g = 0, 1][lane][8]: units 16 h + 8 g + 0..7 of the chunk for token px
    bf16_t* prew = TRAIN ? a.pre + ((((size_t)blockIdx.x * nch) * 8 + wave) * 2 * 64 + lane) * 8 : nullptr;
    bf16x8 hv[2];
    hv[0] = hv[1] = uf[0];  // (defined values; never used before the first vector phase writes them)

    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // chunks 0 and 1, the bias vector
    mlp_barrier();
    const uint32_t lds0 = lds_u32(smem) + lane * 16;
    if (half == 1) mlp_barrier();  // the second half idles one phase
    // iteration jc = this wave's matrix phase (global phase t = 2 jc + half) and vector phase (t + 1) of chunk jc
    for (int jc = 0; jc <= nch; ++jc) {
        // ---- matrix phase: fc2 of chunk jc - 1 (acc[ct][channel 32 ct + 8 (r >> 2) + 4 h + (r & 3)][token px] += over its 32 hidden units), then
        //      fc1 of chunk jc (d1[hidden unit 8 (r >> 2) + 4 h + (r & 3)][token px] over K = 256 channels, starting at the bias)
        MLP_STAMP(0);
        if (half == 0 && jc >= 1 && jc + 1 < nch) mlp_issue(a.w1p, a.w2q, jc + 1, smem + ((jc + 1) % 3) * MLP_STAGE, tid, wave);  // (even global phase)
        if (!(YMI_MLP_ABL & 64)) __builtin_amdgcn_s_setprio(1);
        if (jc > 0) mlp_cols_product(lds0 + ((jc - 1) % 3) * MLP_STAGE + 16384, hv, acc);
        MLP_STAMP(1);
        f32x16 d1;
        {
            const int jb = jc < nch ? jc : nch - 1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(b1s + jb * 32 + 16 * h + 4 * q);
#pragma unroll
                for (int r = 0; r < 4; ++r) d1[4 * q + r] = bb[r];
            }
        }
        if (jc < nch) mlp_rows_product(lds0 + (jc % 3) * MLP_STAGE, uf, d1);
        if (!(YMI_MLP_ABL & 64)) __builtin_amdgcn_s_setprio(0);
        MLP_STAMP(2);
        // (odd global phase for the second half) the pieces issued one phase ago have landed: only the two pre-activation stores are younger
        if (half == 1) {
            if (TRAIN && !(YMI_MLP_ABL & 16)) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        mlp_barrier();
        MLP_STAMP(3);
        if (jc == nch) break;
        // ---- vector phase: bf16 rounding of the pre-activation (saved), exact-erf GELU, bf16 again: the B operand of fc2
        if (half == 1 && jc + 2 < nch) mlp_issue(a.w1p, a.w2q, jc + 2, smem + ((jc + 2) % 3) * MLP_STAGE, tid, wave);  // (even global phase)
        uint32_t hf[8], pf[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            pf[i] = pack_bf16x2(d1[2 * i], d1[2 * i + 1]);
            hf[i] = pack_bf16x2(gelu_fast(bf16_lo(pf[i])), gelu_fast(bf16_hi(pf[i])));
        }
        if (TRAIN && !(YMI_MLP_ABL & 16)) {
            *reinterpret_cast<u32x4*>(prew + ((size_t)jc * 8 * 2 + 0) * 64 * 8) = u32x4{pf[0], pf[1], pf[2], pf[3]};
            *reinterpret_cast<u32x4*>(prew + ((size_t)jc * 8 * 2 + 1) * 64 * 8) = u32x4{pf[4], pf[5], pf[6], pf[7]};
        }
        u32x4 t0 = {hf[0], hf[1], hf[2], hf[3]}, t1 = {hf[4], hf[5], hf[6], hf[7]};
        hv[0] = __builtin_bit_cast(bf16x8, t0);
        hv[1] = __builtin_bit_cast(bf16x8, t1);
        MLP_STAMP(4);
        if (half == 0) {  // (odd global phase for the first half)
            if (TRAIN && !(YMI_MLP_ABL & 16)) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        mlp_barrier();
        MLP_STAMP(5);
    }
    if (half == 0) mlp_barrier();  // the first half waits out the second half's last phase
    // (the last barrier: every wave has finished reading the ring, which becomes the output staging image)
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

__global__ __launch_bounds__(512, 2) void swin_mlp_bwd_kernel(MlpBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [3 stages][32 KB]; the d_u staging image afterwards (8 x 16 KB)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave >> 2;
    const int px = lane & 31, h = lane >> 5;
    const int row0 = blockIdx.x * MLP_BM + wave * 32;
    const int row = row0 + px;
    const int rowc = row < a.T ? row : a.T - 1;
    const int nch = a.hidden / MLP_HC;

    mlp_issue(a.w2t, a.w1tq, 0, smem, tid, wave);
    if (nch > 1) mlp_issue(a.w2t, a.w1tq, 1, smem + MLP_STAGE, tid, wave);
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
    // addresses as a wave-uniform base (scalar registers) + a 32-bit per-lane offset: the kernel has no vector registers to spare for pointers
    const char* pre_base = reinterpret_cast<const char*>(a.pre + (((size_t)blockIdx.x * nch) * 8 + wave) * 2 * 64 * 8);
    const uint32_t pre_off = (uint32_t)lane * 16u;
    // post / d_pre rows: ALWAYS stored (rows beyond T land in the padding the caller provides), so that every wave issues the same number
    // of vector-memory operations per phase: the counted vmcnt below relies on it.  Both are dense [rows][hidden]: one offset serves both
    const uint32_t row_off = (uint32_t)(((int64_t)row * a.hidden + 16 * h) * 2);
    char* post_base = reinterpret_cast<char*>(a.post);
    char* dpre_base = reinterpret_cast<char*>(a.dpre);
    bf16x8 hv[2];
    hv[0] = hv[1] = df[0];

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    mlp_barrier();
    const uint32_t lds0 = lds_u32(smem) + lane * 16;
    if (half == 1) mlp_barrier();
    for (int jc = 0; jc <= nch; ++jc) {
        // ---- matrix phase: d_u += d_pre(jc - 1) W1 chunk, then d_post(jc) = d_out W2 chunk; the stored pre-activations of chunk jc are
        //      requested first (two 16-byte loads per lane, used in the vector phase that follows)
        if (half == 0 && jc >= 1 && jc + 1 < nch) mlp_issue(a.w2t, a.w1tq, jc + 1, smem + ((jc + 1) % 3) * MLP_STAGE, tid, wave);
        u32x4 pv[2];
        {
            const int jl = jc < nch ? jc : nch - 1;  // (always two loads: see the counted wait)
#pragma unroll
            for (int g = 0; g < 2; ++g) pv[g] = *reinterpret_cast<const u32x4*>(pre_base + ((size_t)jl * 8 * 2 + g) * 64 * 16 + pre_off);
        }
        __builtin_amdgcn_s_setprio(1);
        if (jc > 0) mlp_cols_product(lds0 + ((jc - 1) % 3) * MLP_STAGE + 16384, hv, acc);
        f32x16 d1;
#pragma unroll
        for (int r = 0; r < 16; ++r) d1[r] = 0.f;
        if (jc < nch) mlp_rows_product(lds0 + (jc % 3) * MLP_STAGE, df, d1);
        __builtin_amdgcn_s_setprio(0);
        // (odd global phase for the second half) the pieces issued one phase ago have landed: younger are only four stores (its vector phase) and
        // these two loads; the first half below: two loads (its matrix phase) and four stores
        if (half == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        mlp_barrier();
        if (jc == nch) break;
        // ---- vector phase: post = gelu(pre), d_pre = bf16(d_post) * gelu'(pre), both stored row-major; d_pre is the B operand of the d_u product
        if (half == 1 && jc + 2 < nch) mlp_issue(a.w2t, a.w1tq, jc + 2, smem + ((jc + 2) % 3) * MLP_STAGE, tid, wave);
        uint32_t hf[8], of[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {  // registers 2 i, 2 i + 1: units 16 h + 2 i, + 1
            float po[2], dp[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const float x = r ? bf16_hi(pv[i >> 2][i & 3]) : bf16_lo(pv[i >> 2][i & 3]);
                float w, e;
                po[r] = gelu_fast(x, &w, &e);
                const float phi = x >= 0.f ? 1.0f - w : w;
                const float grad = fmaf(x * 0.39894228040143267794f, e, phi);
                dp[r] = to_f32(from_f32<bf16_t>(d1[2 * i + r])) * grad;
            }
            of[i] = pack_bf16x2(po[0], po[1]);
            hf[i] = pack_bf16x2(dp[0], dp[1]);
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            *reinterpret_cast<u32x4*>(post_base + (size_t)(jc * 64 + 16 * g) + row_off) = u32x4{of[4 * g], of[4 * g + 1], of[4 * g + 2], of[4 * g + 3]};
            *reinterpret_cast<u32x4*>(dpre_base + (size_t)(jc * 64 + 16 * g) + row_off) = u32x4{hf[4 * g], hf[4 * g + 1], hf[4 * g + 2], hf[4 * g + 3]};
        }
        u32x4 t0 = {hf[0], hf[1], hf[2], hf[3]}, t1 = {hf[4], hf[5], hf[6], hf[7]};
        hv[0] = __builtin_bit_cast(bf16x8, t0);
        hv[1] = __builtin_bit_cast(bf16x8, t1);
        if (half == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        mlp_barrier();
    }
    if (half == 0) mlp_barrier();
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
    const size_t lds = MLP_LDS_TILE + (size_t)hidden * sizeof(float);
    const dim3 grid((unsigned)((T + MLP_BM - 1) / MLP_BM));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(swin_mlp_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(swin_mlp_fwd_kernel, grid, dim3(512), lds, (hipStream_t)stream, a);
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
                      ((((uintptr_t)post->data) | ((uintptr_t)dpre->data)) & 15) == 0,
                  "swin_ln_mlp_bwd_data: alignment");
    YMI_CHECK_ARG(T < (1ll << 31) && (T + MLP_BM) * post->ld < (1ll << 31), "swin_ln_mlp_bwd_data: too large");
    YMI_CHECK_ARG(post->ld == hidden && dpre->ld == hidden, "swin_ln_mlp_bwd_data: post / dpre are dense [T][hidden] views of buffers of ymi_swin_ln_mlp_pre_elems elements");
    const int64_t n = (int64_t)MLP_C * hidden;
    MlpBwdArgs a{};
    a.dout = (const bf16_t*)dout->data; a.lddo = dout->ld;
    a.w2t = (const bf16_t*)packed + 2 * n; a.w1tq = (const bf16_t*)packed + 3 * n;
    a.pre = (const bf16_t*)pre;
    a.post = (bf16_t*)post->data; a.ldpost = post->ld;
    a.dpre = (bf16_t*)dpre->data; a.lddpre = dpre->ld;
    a.du = (bf16_t*)du->data; a.lddu = du->ld;
    a.T = (int)T; a.hidden = (int)hidden;
    const size_t lds = MLP_LDS_TILE;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(swin_mlp_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(swin_mlp_bwd_kernel, dim3((unsigned)((T + MLP_BM - 1) / MLP_BM)), dim3(512), lds, (hipStream_t)stream, a);
    YMI_CHECK_LAUNCH("swin_ln_mlp_bwd_data");
    return YMI_OK;
}
