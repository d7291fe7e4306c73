// Implicit-GEMM NHWC convolution on MFMA for gfx950.
//
//   Y[pixel m][channel n] = sum_k A[m][k] * Wp[n][k],   k = (tap, ci)
//
// A is the im2col view of the NHWC input, never materialised: row m is an output pixel, the K axis
// walks a TAP TABLE (dh,dw per tap) outer and the input channels inner, so every 16-byte chunk of a
// row is one contiguous piece of one input pixel (or zeros when the tap falls outside the image).
// Forward convolution, stride-1 data-gradient and the four output-parity classes of a stride-2
// data-gradient are all the same kernel with different tap tables / output strides; with one tap it
// is a plain token GEMM (Swin linears).
//
// MI355X mapping
//  * 256-thread workgroups (4 waves), block tile BM pixels x BN channels, K step = 64 bytes per row
//    (32 bf16 / 16 f32): both operand tiles are [rows][64 B] images in LDS.
//  * tiles are filled with global_load_lds_dwordx4 (16 B per lane straight into LDS, no VGPR
//    staging).  The LDS destination is lane-linear, so the bank-conflict swizzle
//    chunk' = chunk ^ ((-(row>>2))&3) is applied to the per-lane SOURCE address and again on the read
//    (rows r and r+4 share banks: 4 rows x 64 B = one 256-B bank row; this map gives every 16-lane group
//    of ds_read_b128 sixteen distinct 16-byte slots).
//  * NS-stage LDS ring: LDS-DMA for K step kt+NS-1 is issued while step kt is multiplied; steps are retired
//    with a COUNTED s_waitcnt vmcnt(N) + one raw s_barrier per K step, so NS-2 steps of loads stay in flight
//    across barriers (the kernel was latency-bound with a single prefetch stage: 3000 cycles per K step).
//  * when Cin is a multiple of the K step (every layer but the first) the tap is wave-uniform and the
//    im2col pixel address / bounds test is computed once per tap, not per K step.
//  * MFMA operands are swapped (A = weights, B = activations): the 16x16 accumulator then holds 4
//    consecutive CHANNELS of one pixel per lane, which is a contiguous 8/16-byte NHWC store.
//  * bf16: v_mfma_f32_16x16x32_bf16; f32 parity mode: v_mfma_f32_16x16x4_f32 (exact f32 FMA chain).
//  * epilogue variants: (a) raw output + deterministic per-block BatchNorm partial sums
//    (no atomics), (b) scale/bias/activation/residual.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

#ifdef YMI_STAMPS
// diagnostic build only: per-wave s_memtime stamps of one workgroup's K steps 2..9 (phases marked in the loops below), kept
// in a spare 4 KB of LDS during the loop and copied out at the end.  extern "C" ymi_debug_stamp_buffer sets the target.
__device__ unsigned long long* g_stamp_buf = nullptr;
extern "C" int ymi_debug_stamp_buffer(void* p) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#define YMI_STAMP_DECL                                                                                     \
    const bool stamp_on = g_stamp_buf && blockIdx.x == gridDim.x / 2 && lane == 0;                         \
    unsigned long long* stamp_lds = reinterpret_cast<unsigned long long*>(smem + stamp_off) + wave_all * 64; \
    int stamp_i = 0;                                                                                        \
    const unsigned long long stamp_mt0 = stamp_on ? __builtin_amdgcn_s_memtime() : 0ull, stamp_rt0 = stamp_on ? __builtin_amdgcn_s_memrealtime() : 0ull;
#define YMI_STAMP(kt)                                                                   \
    do {                                                                                \
        if (stamp_on && (kt) >= 2 && (kt) < 10 && stamp_i < 64) stamp_lds[stamp_i++] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#define YMI_STAMP_DUMP                                                                  \
    do {                                                                                \
        if (stamp_on) {                                                                 \
            for (int q = 0; q < 64; ++q) g_stamp_buf[wave_all * 64 + q] = q < stamp_i ? stamp_lds[q] : 0ull; \
            /* clock calibration: shader cycles and 100 MHz ticks over the whole K loop */ \
            g_stamp_buf[8 * 64 + wave_all * 2 + 0] = __builtin_amdgcn_s_memtime() - stamp_mt0; \
            g_stamp_buf[8 * 64 + wave_all * 2 + 1] = __builtin_amdgcn_s_memrealtime() - stamp_rt0; \
        }                                                                               \
    } while (0)
#define YMI_STAMP_MARK(i)                                                               \
    do {                                                                                \
        if (stamp_on) g_stamp_buf[8 * 64 + 16 + wave_all * 8 + (i)] = __builtin_amdgcn_s_memtime() - stamp_mt0; \
    } while (0)
#else
#define YMI_STAMP_MARK(i) do { } while (0)
#define YMI_STAMP_DECL
#define YMI_STAMP(kt) do { } while (0)
#define YMI_STAMP_DUMP do { } while (0)
#endif

struct IgemmArgs {
    const void* x;
    const void* w;
    void* y;
    const void* res;
    const void* res2;  // second addend of the epilogue (gradient sums of tensors with several consumers)
    void* y2;          // optional second output: act2(value stored to y)  (Swin MLP: pre-activation and GELU of it from one GEMM)
    const void* mul;   // optional multiplier: y = value * act'(mul) with mul_act's derivative (GELU backward inside fc2's data gradient)
    int64_t ldy2, ldmul;
    int act2, mul_act;
    const float* scale;
    const float* bias;
    float* partials;
    long long* stat_acc;   // optional: the statistics as fixed-point atomic sums [4 replicas][2][pstride] instead of per-block rows (see the epilogue)
    float stat_scale;      // ... in steps of 1 / stat_scale (a power of two chosen from the pixel count: ymi_stat_fixed_point_shift)
    int pstride, poff;     // statistics rows: [M block][2][pstride] floats, this problem's channels at column poff (several problems of one
                           // BatchNorm group - the two branches of a Detect level - fill one row array side by side); 0, 0: [2][Cout]
    const void* zero;
    int64_t ldx, ldy, ldres, ldres2, ktot;
    int M, H, W, Ho, Wo, Hy, Wy;
    int s_in, s_out, oh_off, ow_off;
    int Cout, cpt, ntaps, KC;
    uint64_t tap_dh, tap_dw;
    int act, vec_store, vec16;
    uint32_t wo_mul, wo_shr, ho_mul, ho_shr;  // fast division by Wo / Ho
    int nmb, nnb, mpx;                        // M blocks, N blocks, most M blocks any XCD owns (launch geometry, set by the launcher)
    int span;                                 // rows of M an XCD owns: ymi_xcd_span(M) (common.h, XCD ownership of the pixel axis)
};

// Several problems in one launch.
//  * interleaved (hetero == 0; the four output-parity classes of a stride-2 data gradient: same N blocks, same rows read): consecutive
//    ids of an XCD walk the classes of one M block back to back, so the dY rows they all read enter that XCD's L2 once;
//  * one after another (hetero == 1; independent convolutions of any shapes that share only the tile form and the channel-chunk
//    geometry - the same stage of Detect's three levels, reference head.py:66-74): an XCD's id sequence runs through problem 0's tiles,
//    then problem 1's, ...; each problem keeps its own XCD ownership of ITS pixel order.  Small problems (the 20 x 20 level: 100 tiles)
//    fill the partial last round of the large ones instead of paying a launch of their own.
constexpr int IGEMM_MAX_PROBLEMS = 8;
struct IgemmMulti {
    IgemmArgs c[IGEMM_MAX_PROBLEMS];
    int ncls;
    int hetero;
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// n / d for n < 2^31 with a host-computed magic (d == 1 <=> mul == 0)
__device__ __forceinline__ int fast_div(int n, uint32_t mul, uint32_t shr, int d) {
    (void)d;
    return mul ? (int)(__umulhi((uint32_t)n, mul) >> shr) : n;
}

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N-1>{})
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

#ifndef YMI_IGEMM_ABL  // diagnostic builds (results wrong by design): bit 1 no LDS-DMA pieces inside the K loop, 2 no MFMAs, 4 no fragment reads,
#define YMI_IGEMM_ABL 0  // 8 no global stores in the epilogue, 16 no epilogue at all, 32 A pieces of one tap in three only
#endif
template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    // one K step = CPR 16-byte chunks per row = CPR/4 16x16x32 MFMAs per tile pair.
    // swizzle: 64-B rows  -> chunk ^ ((-(row>>2))&3)   (rows r, r+4 share banks)
    //          128-B rows -> chunk ^ ((row>>1)&7)      (rows r, r+2 share banks); both are conflict-free for the
    //          4x16-lane groups of ds_read_b128 when a fragment's 16 rows start at a multiple of 16.
    template <int TM, int TN, int CPR>
    static __device__ __forceinline__ void step(const char* As, const char* Bs, int a_row0, int b_row0, int lane, f32x4 (&acc)[TN][TM]) {
        constexpr int ROWB = CPR * 16;
        const int l15 = lane & 15, l4 = lane >> 4;
        const int sw = CPR == 4 ? ((-(l15 >> 2)) & 3) : ((l15 >> 1) & 7);
        // Every fragment of the K step is requested before the first MFMA, and each 32-deep sub-step waits only for ITS
        // reads (counted lgkmcnt): the second sub-step's reads travel while the first multiplies.  hipcc schedules LDS
        // reads next to their uses and always waits with lgkmcnt(0) here (one exposed LDS round trip per 4-8 MFMAs; PMC,
        // profiles/r02_pmc_igemm.txt: waves parked 39 % of their cycles), so the reads and waits are written out.
        constexpr int KS = CPR / 4;
        static_assert(KS <= 2, "one or two 32-deep sub-steps");
        bf16x8 wf[KS][TN], xf[KS][TM];
        const uint32_t bbase = (uint32_t)(uintptr_t)(lptr_t)(Bs + (b_row0 + l15) * ROWB);
        const uint32_t abase = (uint32_t)(uintptr_t)(lptr_t)(As + (a_row0 + l15) * ROWB);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const uint32_t coff = (uint32_t)(((4 * ks + l4) ^ sw) << 4);
            if (YMI_IGEMM_ABL & 4) {
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) wf[ks][tn] = bf16x8{};
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) xf[ks][tm] = bf16x8{};
                continue;
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[ks][tn]) : "v"(bbase + coff), "n"(tn * 16 * ROWB));
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xf[ks][tm]) : "v"(abase + coff), "n"(tm * 16 * ROWB));
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            // wait for sub-step ks: the (KS-1-ks)*(TN+TM) younger reads may still be in flight.  The fragments are tied
            // to the wait as in/out operands so that no MFMA of this sub-step is scheduled above it.
            __builtin_amdgcn_sched_barrier(0);  // the MFMAs of the previous sub-step stay above this wait
            if (ks + 1 < KS) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(TN + TM) : "memory");  // (TN + TM <= 8: fits the 4-bit counter)
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) asm volatile("" : "+v"(wf[ks][tn]));
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) asm volatile("" : "+v"(xf[ks][tm]));
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
                    if (!(YMI_IGEMM_ABL & 2)) acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][tn], xf[ks][tm], acc[tn][tm], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // The two halves of a K step for the ping-pong kernel: fragment reads into registers (memory phase) ...
    template <int TM, int TN, int CPR>
    static __device__ __forceinline__ void read_frags(const char* As, const char* Bs, int a_row0, int b_row0, int lane, bf16x8 (&wf)[CPR / 4][TN],
                                                      bf16x8 (&xf)[CPR / 4][TM]) {
        constexpr int ROWB = CPR * 16;
        constexpr int KS = CPR / 4;
        const int l15 = lane & 15, l4 = lane >> 4;
        const int sw = CPR == 4 ? ((-(l15 >> 2)) & 3) : ((l15 >> 1) & 7);
        const uint32_t bbase = (uint32_t)(uintptr_t)(lptr_t)(Bs + (b_row0 + l15) * ROWB);
        const uint32_t abase = (uint32_t)(uintptr_t)(lptr_t)(As + (a_row0 + l15) * ROWB);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const uint32_t coff = (uint32_t)(((4 * ks + l4) ^ sw) << 4);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[ks][tn]) : "v"(bbase + coff), "n"(tn * 16 * ROWB));
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xf[ks][tm]) : "v"(abase + coff), "n"(tm * 16 * ROWB));
        }
    }
    // ... and the MFMAs on those registers (compute phase)
    template <int TM, int TN, int CPR>
    static __device__ __forceinline__ void mma_frags(bf16x8 (&wf)[CPR / 4][TN], bf16x8 (&xf)[CPR / 4][TM], f32x4 (&acc)[TN][TM]) {
        constexpr int KS = CPR / 4;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) asm volatile("" : "+v"(wf[ks][tn]));
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) asm volatile("" : "+v"(xf[ks][tm]));
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][tn], xf[ks][tm], acc[tn][tm], 0, 0, 0);
        }
    }
};
template <> struct Mma<float> {
    // one K step = 16 floats per row = four 16x16x4 f32 MFMAs per tile pair (64-byte rows only)
    template <int TM, int TN, int CPR>
    static __device__ __forceinline__ void step(const char* As, const char* Bs, int a_row0, int b_row0, int lane, f32x4 (&acc)[TN][TM]) {
        const int l15 = lane & 15, l4 = lane >> 4;
        const int sw = (-(l15 >> 2)) & 3;  // g(q) = (-q)&3: conflict-free for the 4x16-lane groups of ds_read_b128
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int coff = ((ks ^ sw) << 4) + (l4 << 2);
            float wf[TN], xf[TM];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) wf[tn] = *reinterpret_cast<const float*>(Bs + (b_row0 + tn * 16 + l15) * 64 + coff);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) xf[tm] = *reinterpret_cast<const float*>(As + (a_row0 + tm * 16 + l15) * 64 + coff);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[tn], xf[tm], acc[tn][tm], 0, 0, 0);
        }
    }
};


// sum of a value over the 16 lanes of its DPP row, result in every lane: xor-1 and xor-2 quad permutes, then the half-row and
// row mirrors (each lane already holds its quad's / half-row's total, so the mirrored partner supplies the other one)
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float v) {
    v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);  // row_half_mirror
    v = dpp_add<0x140>(v);  // row_mirror
    return v;
}

// ---- epilogue shared by the GEMM kernels -------------------------------------------------------------------------
// The tile leaves through LDS: lanes drop their 4-channel groups into a [pixel][channel] image, then the workgroup
// stores it as 16-byte chunks along C, so every store instruction writes whole 128-byte lines (per-lane 8-byte stores
// to 16 different rows cost 2-3x the time of the same bytes stored this way).  STATS: raw output + deterministic
// per-block BatchNorm partial sums; otherwise scale / bias / activation / up to two addends.
template <typename T, int BM, int BN, int WM, int WN, bool STATS, int NT>
__device__ __forceinline__ void igemm_epilogue(const IgemmArgs& a, f32x4 (&acc)[BN / WN / 16][BM / WM / 16], char* smem, int m0, int n0, int mb, int wm,
                                               int wn, int lane, int tid_all
#ifdef YMI_STAMPS
                                               , bool stamp_on = false, int wave_all = 0, unsigned long long stamp_mt0 = 0
#endif
                                               ) {
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    const int l15 = lane & 15, l4 = lane >> 4;
    T* yg = reinterpret_cast<T*>(a.y);
    constexpr int ES = (int)sizeof(T);
    constexpr int CROW = BN * ES + 16;  // padded LDS row of the output image
    char* Cimg = smem;
    float* red = reinterpret_cast<float*>(smem + BM * CROW);  // [WM][2][BN] (STATS)
    const T* rg = reinterpret_cast<const T*>(a.res);    // (no __restrict__: an addend may be the output buffer itself, read before it is written)
    const T* rg2 = reinterpret_cast<const T*>(a.res2);

    // output pixel (row of y, and of the epilogue addends) that GEMM row m produces
    auto out_pixel = [&](int m) -> int64_t {
        if (a.s_out == 1 && a.Hy == a.Ho && a.Wy == a.Wo) return (int64_t)m;
        const int t = fast_div(m, a.wo_mul, a.wo_shr, a.Wo);
        const int wo = m - t * a.Wo;
        const int n = fast_div(t, a.ho_mul, a.ho_shr, a.Ho);
        const int ho = t - n * a.Ho;
        return ((int64_t)n * a.Hy + ho * a.s_out + a.oh_off) * a.Wy + wo * a.s_out + a.ow_off;
    };
    auto out_offset = [&](int m) -> int64_t { return out_pixel(m) * a.ldy; };

    // Code size matters here: this block is unrolled TN x TM times around register-indexed accumulators, and every workgroup runs
    // it once.  With the activation (erff), the scalar residual fall-backs and the unaligned stores inlined per tile it was 11,500
    // instructions - in-kernel stamps (profiles/r02_igemm_phase_stamps.txt) showed 10-11 k cycles for this phase, 20 % of a
    // workgroup's life on an 18-step layer and 30-45 % on 1x1 layers (instruction fetch, not arithmetic).  So the per-tile code
    // only scales, adds the aligned addends and drops the tile into LDS; activation, unaligned addends and unaligned stores
    // work on the LDS image in run-time loops below.
#ifndef YMI_EPI_PRIO
#define YMI_EPI_PRIO 2
#endif
    // the co-resident workgroup is in its K loop: its MFMAs hold the SIMD's vector issue half of the time and, being older, win
    // the arbitration - raise this wave's priority for its ~500 VALU instructions so that LDS and the wave slots are freed sooner
    if (YMI_EPI_PRIO) __builtin_amdgcn_s_setprio(YMI_EPI_PRIO);
    const int act = STATS ? (int)YMI_ACT_NONE : a.act;
    const bool res1 = !STATS && rg && a.vec_store && (a.Cout & 3) == 0 && act == YMI_ACT_NONE;  // addends joined per tile (f32, before the one rounding)
    const bool res2nd = !STATS && rg && !res1;                              // ... or after the activation, from the LDS image
    // activation-gradient multiplier (Swin fc2's data gradient: dpre = (dout W2) * gelu'(pre)) applied per register tile too: the loads of
    // all 16 tiles are hoisted together by the compiler.  In the store loop below each of a thread's 8 chunks fetched its multiplier
    // and waited for it - eight exposed memory round trips per tile, on top of a 256-deep K loop (fc2 data gradient 109 -> 104 us)
    const T* mulq = STATS ? nullptr : reinterpret_cast<const T*>(a.mul);
    const bool mul1 = !STATS && mulq && !res2nd && a.vec_store && (a.Cout & 3) == 0 && (a.ldmul & 3) == 0 && act == YMI_ACT_NONE;
    {
    int64_t rpx[TM];
    if (res1 || mul1) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int m = m0 + (wm * TM + tm) * 16 + l15;
            rpx[tm] = m < a.M ? out_pixel(m) : -1;
        }
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int chl = (wn * TN + tn) * 16 + 4 * l4;  // channel within the block tile
        const int ch = n0 + chl;
        float sc[4], bi[4];
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (!STATS) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool cok = ch + r < a.Cout;
                sc[r] = (a.scale && cok) ? a.scale[ch + r] : 1.0f;
                bi[r] = (a.bias && cok) ? a.bias[ch + r] : 0.0f;
            }
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int row = (wm * TM + tm) * 16 + l15;
            float v[4];
            if constexpr (STATS) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = to_f32(from_f32<T>(acc[tn][tm][r]));  // statistics of what is stored
                    s1[r] += v[r];
                    s2[r] += v[r] * v[r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[tn][tm][r] * sc[r] + bi[r];
                if (res1 && rpx[tm] >= 0 && ch + 3 < a.Cout) {
                    float rr[4];
                    Pack<T, 4>::load(rg + rpx[tm] * a.ldres + ch, rr);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += rr[r];
                    if (rg2) {
                        Pack<T, 4>::load(rg2 + rpx[tm] * a.ldres2 + ch, rr);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += rr[r];
                    }
                }
                if (mul1 && rpx[tm] >= 0 && ch + 3 < a.Cout) {
                    float mm[4];
                    Pack<T, 4>::load(mulq + rpx[tm] * a.ldmul + ch, mm);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = to_f32(from_f32<T>(v[r])) * act_grad_rt(mm[r], a.mul_act);  // (the product of the STORED value, as before)
                }
            }
            Pack<T, 4>::store(reinterpret_cast<T*>(Cimg + row * CROW) + chl, v);
        }
        if constexpr (STATS) {
            // sum over the 16 lanes of a DPP row (= the 16 pixels of the tile) with 4 DPP adds per value; __shfl_xor compiles to
            // ds_bpermute_b32 - 128 LDS-pipe round trips per wave that queue behind the other workgroup's fragment reads
            // (stamps: 11 k cycles for this block, profiles/r02_igemm_phase_stamps.txt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s1[r] = row16_sum(s1[r]);
                s2[r] = row16_sum(s2[r]);
            }
            if (l15 == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    red[(wm * 2 + 0) * BN + chl + r] = s1[r];
                    red[(wm * 2 + 1) * BN + chl + r] = s2[r];
                }
            }
        }
    }
    }
    YMI_STAMP_MARK(3);  // accumulators converted and dropped into the LDS image
    __syncthreads();
    YMI_STAMP_MARK(4);  // past the barrier
    if constexpr (STATS) {
        if (tid_all < 2 * BN) {
            const int which = tid_all / BN, chl = tid_all % BN;
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < WM; ++q) sum += red[(q * 2 + which) * BN + chl];
            const int ch = n0 + chl;
            if (ch < a.Cout) {
                if (a.stat_acc) {
                    // Round 5: the per-block sums leave as 64-bit FIXED-POINT atomic adds (steps of 2^-shift) into one of four replica rows, and the
                    // consumer - the BatchNorm affine pass, a kernel boundary later - sums the replicas and finalizes in its prologue: the
                    // separate finalize launch (and, from 1024 row blocks up, the row pre-reduction before it: 65 launches of ~5 us a step) is
                    // gone.  Integer addition is exact and order-free, so the statistics stay bit-for-bit reproducible whatever order the
                    // tiles finish in; a float atomic would not be.  shift = 37 - ceil(log2(pixel count)) (common.h): sums of squares up to
                    // count * 2^24 (an r.m.s. of 4096) fit with two bits to spare, and the step is 2^-15 at 3.3 M pixels, 2^-31 at 64 - after
                    // the division by the count below 1e-11 of the variance in either case, far under any BatchNorm eps.
                    const long long q = __float2ll_rn(sum * a.stat_scale);
                    __hip_atomic_fetch_add(a.stat_acc + ((int64_t)((mb & 3) * 2 + which)) * a.pstride + a.poff + ch, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    a.partials[((int64_t)mb * 2 + which) * a.pstride + a.poff + ch] = sum;
                }
            }
        }
    }
    if (a.vec16) {
        constexpr int CPW = BN * ES / 16;  // 16-byte chunks per output row
        constexpr int EPC = 16 / ES;       // elements per chunk
        const T* mulp = (STATS || mul1) ? nullptr : reinterpret_cast<const T*>(a.mul);
        const bool post = act != YMI_ACT_NONE || res2nd || mulp;  // (workgroup-uniform) something left to do on the stored values
        if (!post) {
#pragma unroll 4
            for (int idx = tid_all; idx < BM * CPW; idx += NT) {  // every wave of the workgroup stores
                const int row = idx / CPW, cc = idx % CPW;
                const int m = m0 + row, ch = n0 + cc * EPC;
                if (m < a.M && ch < a.Cout) {
                    const u32x4 val = *reinterpret_cast<const u32x4*>(Cimg + row * CROW + cc * 16);
                    if (YMI_IGEMM_ABL & 8) asm volatile("" :: "v"(val));
                    else *reinterpret_cast<u32x4*>(yg + out_offset(m) + ch) = val;
                }
            }
        } else {  // fused inference convolutions (SiLU, then the shortcut), element-aligned addends, activation-gradient multiplier
#pragma unroll 1
            for (int idx = tid_all; idx < BM * CPW; idx += NT) {
                const int row = idx / CPW, cc = idx % CPW;
                const int m = m0 + row, ch0 = n0 + cc * EPC;
                if (m < a.M && ch0 < a.Cout) {
                    u32x4 val = *reinterpret_cast<const u32x4*>(Cimg + row * CROW + cc * 16);
                    T* vp = reinterpret_cast<T*>(&val);
                    const int64_t px = (res2nd || mulp) ? out_pixel(m) : 0;
#pragma unroll
                    for (int h = 0; h < EPC / 4; ++h) {
                        const int ch = ch0 + 4 * h;
                        float v[4], rr[4];
                        Pack<T, 4>::load(vp + 4 * h, v);
                        if (act != YMI_ACT_NONE) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = apply_act_rt(v[r], act);
                        }
                        if (res2nd) {
                            if (a.vec_store) {
                                Pack<T, 4>::load(rg + px * a.ldres + ch, rr);
#pragma unroll
                                for (int r = 0; r < 4; ++r) v[r] += rr[r];
                                if (rg2) {
                                    Pack<T, 4>::load(rg2 + px * a.ldres2 + ch, rr);
#pragma unroll
                                    for (int r = 0; r < 4; ++r) v[r] += rr[r];
                                }
                            } else {
#pragma unroll
                                for (int r = 0; r < 4; ++r) v[r] += to_f32(rg[px * a.ldres + ch + r]) + (rg2 ? to_f32(rg2[px * a.ldres2 + ch + r]) : 0.f);
                            }
                        }
                        if (mulp) {  // (host: 4-element-aligned)
                            Pack<T, 4>::load(mulp + px * a.ldmul + ch, rr);
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] *= act_grad_rt(rr[r], a.mul_act);
                        }
                        Pack<T, 4>::store(vp + 4 * h, v);
                    }
                    *reinterpret_cast<u32x4*>(yg + out_offset(m) + ch0) = val;
                }
            }
        }
        if (!STATS && a.y2) {  // second output: the activation of what was just stored (read back from the LDS image, so both outputs
                               // see the same rounded value - the arithmetic of a separate activation kernel reading the first output)
            T* y2g = reinterpret_cast<T*>(a.y2);
#pragma unroll 1
            for (int idx = tid_all; idx < BM * CPW; idx += NT) {
                const int row = idx / CPW, cc = idx % CPW;
                const int m = m0 + row, ch0 = n0 + cc * EPC;
                if (m < a.M && ch0 < a.Cout) {
                    u32x4 val = *reinterpret_cast<const u32x4*>(Cimg + row * CROW + cc * 16);
                    T* vp = reinterpret_cast<T*>(&val);
#pragma unroll
                    for (int h = 0; h < EPC / 4; ++h) {
                        float v[4];
                        Pack<T, 4>::load(vp + 4 * h, v);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = apply_act_rt(v[r], a.act2);
                        Pack<T, 4>::store(vp + 4 * h, v);
                    }
                    *reinterpret_cast<u32x4*>(y2g + out_pixel(m) * a.ldy2 + ch0) = val;
                }
            }
        }
    } else {
        // unaligned / odd channel counts (Detect's class maps at small nc, first-layer data gradients): one element per lane and trip
        for (int idx = tid_all; idx < BM * BN; idx += NT) {
            const int row = idx / BN, col = idx % BN;
            const int m = m0 + row, ch = n0 + col;
            if (m < a.M && ch < a.Cout) {
                float v = apply_act_rt(to_f32(reinterpret_cast<const T*>(Cimg + row * CROW)[col]), act);
                if (res2nd) {
                    const int64_t px = out_pixel(m);
                    v += to_f32(rg[px * a.ldres + ch]) + (rg2 ? to_f32(rg2[px * a.ldres2 + ch]) : 0.f);
                }
                yg[out_offset(m) + ch] = from_f32<T>(v);
            }
        }
    }
    YMI_STAMP_MARK(5);  // stores issued
}

// The kernel has two forms.
// Default (NTHR = 256): four waves in a WM x WN grid, an NS-stage LDS ring, one barrier per K step (see the file header).
// PP, ping-pong (NTHR = 512, 256-row tile): waves 0-3 own rows 0-127, waves 4-7 rows 128-255, and the two halves run half a K step
// apart.  A half's step is a MEMORY phase (fragment reads of step k into registers, its share of the LDS-DMA pieces of step k+2,
// waits) followed by a COMPUTE phase (32 MFMAs on registers only), every phase ends at a workgroup barrier, so each SIMD always
// holds one wave in its memory phase beside one in its compute phase (the arrangement MI355X_MICROARCH.md, Two waves per SIMD,
// describes) instead of two waves in the same phase.  Three LDS stages: the pieces of step k+2 overwrite the stage of step k-1,
// which both halves finished reading at least one phase earlier.
// (Round-2 variants that measured slower - wave specialisation, a lockstep 256x128 tile, ring depths 1 / 3, pieces interleaved
// with the MFMAs, LDS-resident 3x3 input rows in linear pixel order, BatchNorm-backward sums in the data-gradient epilogue -
// were removed in round 3; their tables are profiles/r02_conv_bench_*.txt and profiles/r02_bn_bwd_fuse.txt, the code is in git.
// Round 3 measured two more, parity-green and removed again:
//  * the LDS-resident 3x3 kernel rebuilt in a padded-linear pixel order (one zero column per image row, one zero row per image: a
//    tap is a constant row shift, no masks; image double-buffered over 32-channel chunks; 2.2x fewer operand bytes): no longer
//    bound by bytes, yet 0-9 % on the 80x80 layers and slower wherever 256-row tiles leave fewer workgroups than CUs
//    (profiles/r03_conv_bench_dconv3.txt);
//  * a 200-row tile stride for this ping-pong form (this model's maps have 25 * 2^k pixels, so power-of-two tiles give 400 / 800 /
//    1,600 workgroups for 512 slots; 200-row tiles give 256 / 512 / 1,024): 3-7 % on some 80x80 layers in isolation
//    (profiles/r03_conv_bench_stride200.txt), 13.98 against 13.97 ms on the whole step.
// Both K loops sit at ~58 % MFMA-busy; what separates a layer from that figure is its grid against 512 slots (tools/quant_probe.sh,
// profiles/r03_tile_count_probe.txt: 400 tiles of 128x128 take as long as 512) and the prologue / epilogue of a 1.5-round grid.)
template <typename T, int BM, int BN, int WM, int WN, int NS, int CPR, bool FAST, bool STATS, int NTHR = 256, bool PP = false>
__global__ __launch_bounds__(NTHR) void igemm_kernel(IgemmMulti P) {
    constexpr int CH = ElemTraits<T>::CH;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int ROWB = CPR * 16;                    // bytes per LDS row = K step per row (64: 32 bf16 / 16 f32; 128: 64 bf16)
    constexpr int LT = NTHR;                          // threads that issue loads (all of them)
    constexpr int RPI = LT / CPR, RPW = 64 / CPR;     // rows filled per block-wide / per wave load instruction
    constexpr int NA = (BM * CPR + LT - 1) / LT, NB = (BN * CPR + LT - 1) / LT;
    constexpr int STAGE = (BM + BN) * ROWB;
    static_assert(CPR == 4 || (CPR == 8 && FAST), "128-byte rows need tap-uniform K steps");
    static_assert(WM * WN == NTHR / 64, "one wave tile per wave");
    static_assert((BM * CPR) % LT == 0, "every loading wave issues all A loads");
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef YMI_STAMPS
    constexpr int stamp_off = NS * STAGE;  // the launcher adds 4 KB behind the ring in this build
#endif

    constexpr int NT = NTHR;
    const int tid_all = threadIdx.x, lane = tid_all & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid_all >> 6);  // provably wave-uniform: LDS-DMA bases go to M0 without a waterfall loop
    const int tid = tid_all, wave = wave_all;
    const int wm = wave / WN, wn = wave % WN;
    YMI_STAMP_DECL
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs (id & 7), each with its own L2.  Every XCD gets a
    // CONTIGUOUS range of M blocks (neighbouring pixel tiles share 3x3 halo rows) and walks the N blocks of one M block
    // back to back, so the A tile an M block gathers is fetched into that XCD's L2 once and reused by all its N blocks
    // (with N blocks on grid.y they ran a whole grid apart and A came back from MALL/HBM once per N block).
    // The range is the XCD's EIGHTH of the pixel order (round 4): the kernel that produced the rows (a BatchNorm pass, another
    // GEMM's epilogue) wrote that eighth from this XCD too, so they are in this L2, not in another one's.
    // The 1-D grid is padded to 8 * mpx * nnb ids; ids that fall outside the XCD's range leave before any barrier.
    const int orig = blockIdx.x, xcd = orig & 7, seq0 = orig >> 3;
    int cls = 0, seq = seq0;  // block-uniform
    if (P.hetero) {
        // problems one after another: skip the tiles this XCD owns of problems 0 .. cls-1 (scalar arithmetic on kernel arguments)
#pragma unroll 1
        for (; cls < P.ncls; ++cls) {
            const int sp = P.c[cls].span, nmbc = P.c[cls].nmb;
            const int f = (xcd * sp + BM - 1) / BM;
            int l = ((xcd + 1) * sp + BM - 1) / BM;
            l = l < nmbc ? l : nmbc;
            const int cnt = (l > f ? l - f : 0) * P.c[cls].nnb;
            if (seq < cnt) break;
            seq -= cnt;
        }
        if (cls >= P.ncls) return;  // grid padding
    } else if (P.ncls > 1) {
        cls = seq0 % P.ncls;
        seq = seq0 / P.ncls;
    }
    const IgemmArgs a = P.c[cls];
    const int nb = seq % a.nnb, ml = seq / a.nnb;
    // this XCD owns the M blocks whose first row lies in its span of the pixel order (the rule every streaming kernel follows, common.h)
    const int first = (xcd * a.span + BM - 1) / BM;
    int last = ((xcd + 1) * a.span + BM - 1) / BM;
    last = last < a.nmb ? last : a.nmb;
    const int mb = first + ml;
    if (mb >= last) return;
    const int m0 = mb * BM, n0 = nb * BN;
    const T* __restrict__ xg = reinterpret_cast<const T*>(a.x);
    const T* __restrict__ wg = reinterpret_cast<const T*>(a.w);
    const T* zero = reinterpret_cast<const T*>(a.zero);

    // ---- warm this XCD's L2 with the weight rows of this N block ---------------------------------------------------------------------
    // Inside the training step a layer's packed weights are NOT in L2 (they were written at the start of the step), and every workgroup
    // walks K in the same order: each K step's weight lines miss once per XCD with everybody waiting on that one fill - K / 64 serialized
    // memory latencies per launch (det.cv2[2].0, K = 4608: 21 us with warm caches, 52 us in the step and with evicted caches in
    // tools/conv_bench.py --cold - 33-35 us in the step with this warm-up; the data gradients, K <= 1152, barely notice).  So the first workgroups of an XCD touch every 128-byte
    // line of their N block's rows up front, all requests in flight at once - as 4-byte LDS-DMA loads into the first bytes of the ring:
    // no register receives the data (a VGPR destination made the register allocator wait for the loads at once, and an inline-asm load
    // is simply wrong: the compiler reuses the register before the data lands).  A wave's loads complete in order - ONLY a wave's own: the
    // bytes a wave aims at are the first ones its own first A-row load of stage 0 writes (wave * 64 chunks of 16 bytes), so that load,
    // issued later by the same wave, overwrites them before anything reads them; the counted waits of the K loop are unaffected.
    {
        const int rows_valid = (a.Cout - n0 < BN) ? a.Cout - n0 : BN;
        const uint32_t lines = (uint32_t)(((int64_t)rows_valid * a.ktot * (int64_t)sizeof(T) + 127) >> 7);
        const uint32_t pw = (lines + 4 * NT - 1) / (4 * NT);  // workgroups needed at 4 lines per thread
        if ((uint32_t)ml < pw) {  // (workgroup-uniform)
            const char* wbase = reinterpret_cast<const char*>(wg + (int64_t)n0 * a.ktot);
            const uint32_t last = lines - 1, st = pw * NT;
            uint32_t li = (uint32_t)ml * NT + (uint32_t)tid_all;
#pragma unroll
            for (int q = 0; q < 4; ++q) {  // (lines beyond the last one re-touch it: every lane of a wave issues, as LDS-DMA requires)
                __builtin_amdgcn_global_load_lds((gptr_t)(wbase + (size_t)(li < last ? li : last) * 128), (lptr_t)(smem + wave * 64 * 16), 4, 0, 0);
                li += st;
            }
        }
    }

    // ---- per-thread load descriptors -----------------------------------------------------------
    // source chunk of the K step this thread fetches: LDS position (row, tid % CPR) holds chunk pos ^ f(row)
    const int c = CPR == 4 ? ((tid & 3) ^ ((-(tid >> 4)) & 3)) : ((tid & 7) ^ ((tid >> 4) & 7));
    int a_nH[NA], a_h[NA], a_w[NA];
    bool a_ok[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int r = tid / CPR + RPI * i;
        const int m = m0 + r;
        a_ok[i] = (m < a.M);
        const int mm = a_ok[i] ? m : 0;
        const int t = fast_div(mm, a.wo_mul, a.wo_shr, a.Wo);
        const int wo = mm - t * a.Wo;
        const int n = fast_div(t, a.ho_mul, a.ho_shr, a.Ho);
        const int ho = t - n * a.Ho;
        a_nH[i] = n * a.H;
        a_h[i] = ho * a.s_in;
        a_w[i] = wo * a.s_in;
    }
    const T* b_ptr[NB];
    bool b_ok[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int rn = tid / CPR + RPI * j;
        const int n = n0 + rn;
        b_ok[j] = (n < a.Cout) && (rn < BN);
        b_ptr[j] = wg + (int64_t)(b_ok[j] ? n : 0) * a.ktot;
    }

    // K iterator.  FAST (Cin % K-step == 0): the tap is wave-uniform, `cic` is this thread's chunk inside the
    // tap, and the per-row pixel address / bounds test is refreshed once per tap.  Otherwise (first layer,
    // Cin = 8): every chunk of a K step may belong to a different tap and is decoded per step.
    int tap = FAST ? 0 : c / a.cpt;
    int cic = FAST ? c : c - tap * a.cpt;
    const int adv_tap = 4 / a.cpt, adv_c = 4 - adv_tap * a.cpt;
    // FAST state: one running pointer + one per-step increment per row.  Rows whose tap falls outside the image
    // (or outside M) point at the zero page with increment 0, so the K loop has no selects and no flags.
    const T* a_ptr[NA];
    int a_inc[NA];
    const int steps_per_tap = a.cpt / CPR;  // scalar
    int tap_s = 0, left = steps_per_tap;     // scalar (kernel arguments and loop counters only)
    // per-row pointer of the tap-(0,0) pixel; a tap only adds the scalar (dh*W + dw)*ldx and re-tests the bounds
    const T* a_center[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) a_center[i] = xg + ((int64_t)(a_nH[i] + a_h[i]) * a.W + a_w[i]) * a.ldx + c * CH;
    auto setup_tap = [&](int tp) {
        const int dh = (int)((a.tap_dh >> (4 * tp)) & 15) - 8;
        const int dw = (int)((a.tap_dw >> (4 * tp)) & 15) - 8;
        const int64_t toff = ((int64_t)dh * a.W + dw) * a.ldx;  // scalar
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const bool in = a_ok[i] && (unsigned)(a_h[i] + dh) < (unsigned)a.H && (unsigned)(a_w[i] + dw) < (unsigned)a.W;
            a_ptr[i] = in ? a_center[i] + toff : zero;
            a_inc[i] = in ? CPR * CH : 0;
        }
    };
    int b_inc[NB];
    if (FAST) {
        setup_tap(0);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            b_ptr[j] = b_ok[j] ? b_ptr[j] + c * CH : zero;
            b_inc[j] = b_ok[j] ? CPR * CH : 0;
        }
    }

    auto issue = [&](int s) {
        char* As = smem + s * STAGE;
        char* Bs = As + BM * ROWB;
        if constexpr (FAST) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                __builtin_amdgcn_global_load_lds((gptr_t)a_ptr[i], (lptr_t)(As + (i * LT + wave * 64) * 16), 16, 0, 0);
                a_ptr[i] += a_inc[i];
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if ((wave * RPW + RPI * j) < BN) {  // wave-uniform
                    __builtin_amdgcn_global_load_lds((gptr_t)b_ptr[j], (lptr_t)(Bs + (j * LT + wave * 64) * 16), 16, 0, 0);
                    b_ptr[j] += b_inc[j];
                }
            }
            if (--left == 0) {  // next tap (scalar branch)
                left = steps_per_tap;
                if (++tap_s < a.ntaps) setup_tap(tap_s);
            }
        } else {
            const bool kvalid = tap < a.ntaps;
            const int dh = (int)((a.tap_dh >> (4 * (tap & 15))) & 15) - 8;
            const int dw = (int)((a.tap_dw >> (4 * (tap & 15))) & 15) - 8;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int hi = a_h[i] + dh, wi = a_w[i] + dw;
                const bool ok = a_ok[i] && kvalid && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
                const T* src = ok ? xg + ((int64_t)(a_nH[i] + hi) * a.W + wi) * a.ldx + cic * CH : zero;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(As + (i * LT + wave * 64) * 16), 16, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if ((wave * RPW + RPI * j) < BN) {  // wave-uniform
                    const bool ok = b_ok[j] && kvalid;
                    const T* src = ok ? b_ptr[j] + (int64_t)(tap * a.cpt + cic) * CH : zero;
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Bs + (j * LT + wave * 64) * 16), 16, 0, 0);
                }
            }
            cic += adv_c;
            tap += adv_tap;
            if (cic >= a.cpt) {
                cic -= a.cpt;
                ++tap;
            }
        }
    };

    YMI_STAMP_MARK(0);  // prologue (address set-up) done
    f32x4 acc[TN][TM];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) acc[tn][tm] = f32x4{0.f, 0.f, 0.f, 0.f};

    // loads this wave issues per K step (vmcnt counts LDS-DMA operations per wave, in order)
    constexpr int LPT_FULL = NA + NB;
    constexpr int LPT_AONLY = NA;  // waves beyond the B tile's rows (BN < 64) issue no B loads
    const bool b_wave = (BN >= RPI) || (wave * RPW < BN);
    const int nkt = (a.KC + CPR - 1) / CPR;
    if constexpr (PP) {
        static_assert(NTHR == 512 && NS == 3 && FAST && std::is_same<T, bf16_t>::value && WM == 4, "ping-pong form: 512 threads, 3 stages, bf16");
        const int half = wave_all >> 2;  // 0: rows 0..BM/2-1 (starts first), 1: the other rows, half a step behind
        bf16x8 wf[CPR / 4][TN], xf[CPR / 4][TM];
#ifdef YMI_PP_STAGGER  // diagnostic: the workgroup in the upper wave slots of its SIMDs starts YMI_PP_STAGGER * 64 cycles late
        if (__builtin_amdgcn_s_getreg(4 | (3 << 11)) & 2) {
#ifdef YMI_PP_STAGGER_REP  // ... times 8,128 cycles: offsets of a fraction of a workgroup's lifetime (epilogue against K loop)
#pragma unroll 1
            for (int q = 0; q < YMI_PP_STAGGER_REP; ++q) __builtin_amdgcn_s_sleep(127);
#else
            __builtin_amdgcn_s_sleep(YMI_PP_STAGGER);
#endif
        }
#endif
        issue(0);
        if (nkt > 1) issue(1);
        // step 0 (and only it) has landed when the pieces of step 1 may still be outstanding
        if (nkt > 1) {
            if (b_wave) wait_vmcnt_barrier<LPT_FULL>();
            else wait_vmcnt_barrier<LPT_AONLY>();
        } else {
            wait_vmcnt_barrier<0>();
        }
        if (half == 1) asm volatile("s_barrier" ::: "memory");  // the second half idles one phase
        // Phase p (p = 0, 1, ...) is the memory phase of step p/2 for half 0 (p even) and of step (p-1)/2 for half 1 (p odd).
        // Step k+1 is read in phases 2k+2 (half 0) and 2k+3 (half 1), so ALL its pieces must be in LDS when phase 2k+1 ends:
        // half 0 waits for its share at the end of its compute phase of step k, half 1 at the end of its memory phase of
        // step k - each then has only its pieces of step k+2 outstanding.  Those pieces go to the stage of step k-1, last read
        // in phase 2k-1, and are issued in phases 2k / 2k+1.
        // The pieces of step k+2 are split between the two phases of step k (stamps, profiles/r02_igemm_phase_stamps.txt: with all
        // six in the memory phase it lasted ~1000 cycles against ~500 of MFMAs): the A rows go out in the memory phase, the
        // weight rows between the MFMAs, and the pointer / tap bookkeeping follows the last MFMA, outside the memory phase.
#ifndef YMI_PP_ALLMEM
#define YMI_PP_ALLMEM 0
#endif
#ifndef YMI_PP_ADV_MEM
#define YMI_PP_ADV_MEM 0
#endif
#ifndef YMI_PP_PRIO
#define YMI_PP_PRIO 0
#endif
        constexpr int NMEM = YMI_PP_ALLMEM ? NA + NB : NA;  // pieces issued in the memory phase (pieces are numbered A rows first)
        constexpr int NPC = NA + NB;       // pieces per wave and step
        constexpr int NM = (CPR / 4) * TN * TM;
        auto load_piece = [&](int s, auto pc) {  // piece p of the step whose stage is s, WITHOUT advancing the pointers
            constexpr int p = decltype(pc)::value;
            char* Ad = smem + s * STAGE;
            char* Bd = Ad + BM * ROWB;
            if (YMI_IGEMM_ABL & 1) return;
            if constexpr (p < NA) {
                if ((YMI_IGEMM_ABL & 32) && (tap_s % 3) != 1) return;  // emulates an input tile shared by the three taps of a kernel row (upper bound)
            }
            if constexpr (p < NA) __builtin_amdgcn_global_load_lds((gptr_t)a_ptr[p], (lptr_t)(Ad + (p * LT + wave * 64) * 16), 16, 0, 0);
            else __builtin_amdgcn_global_load_lds((gptr_t)b_ptr[p - NA], (lptr_t)(Bd + ((p - NA) * LT + wave * 64) * 16), 16, 0, 0);
        };
        auto advance = [&]() {  // what issue() does after its loads
#pragma unroll
            for (int i = 0; i < NA; ++i) a_ptr[i] += a_inc[i];
#pragma unroll
            for (int jj = 0; jj < NB; ++jj) b_ptr[jj] += b_inc[jj];
            if (--left == 0) {
                left = steps_per_tap;
                if (++tap_s < a.ntaps) setup_tap(tap_s);
            }
        };
        static_assert(BN >= RPI * NB, "every wave issues every weight piece");
        for (int kt = 0; kt < nkt; ++kt) {
            const bool more = kt + 2 < nkt;
            const int sn = (kt + 2) % NS;
            // ---- memory phase of step kt
            YMI_STAMP(kt);  // 0: phase start
            const char* As = smem + (kt % NS) * STAGE;
            if (!(YMI_IGEMM_ABL & 4) || kt == 0) Mma<T>::template read_frags<TM, TN, CPR>(As, As + BM * ROWB, wm * TM * 16, wn * TN * 16, lane, wf, xf);
#if YMI_PP_ADV_MEM
            if (kt > 0 && kt + 1 < nkt) advance();  // the bookkeeping of step kt-1's pieces, behind the reads instead of behind the MFMAs
#endif
            if (more) static_for<0, NMEM>([&](auto pc) { load_piece(sn, pc); });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            YMI_STAMP(kt);  // 1: fragments in registers, A pieces issued
            if (half == 1) {
                if (more) wait_vmcnt_barrier<NMEM>();  // everything older than this phase's pieces: all of step kt+1
                else wait_vmcnt_barrier<0>();
            } else {
                asm volatile("s_barrier" ::: "memory");
            }
            // ---- compute phase of step kt
            YMI_STAMP(kt);  // 2: past the barrier that ends the memory phase
            __builtin_amdgcn_sched_barrier(0);
#if YMI_PP_PRIO
            __builtin_amdgcn_s_setprio(1);
#endif
            static_for<0, CPR / 4>([&](auto ksc) {
                constexpr int ks = decltype(ksc)::value;
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) asm volatile("" : "+v"(wf[ks][tn]));
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) asm volatile("" : "+v"(xf[ks][tm]));
                static_for<0, TN>([&](auto tnc) {
                    constexpr int tn = decltype(tnc)::value;
                    static_for<0, TM>([&](auto tmc) {
                        constexpr int tm = decltype(tmc)::value;
                        constexpr int q = (ks * TN + tn) * TM + tm;
                        if (!(YMI_IGEMM_ABL & 2)) acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][tn], xf[ks][tm], acc[tn][tm], 0, 0, 0);
                        static_for<NMEM, NPC>([&](auto pc) {  // weight piece j after MFMA 4 + 6 j
                            constexpr int pp = decltype(pc)::value;
                            if constexpr (q == 4 + 6 * (pp - NMEM) && q < NM) {
                                __builtin_amdgcn_sched_barrier(0);
                                if (more) load_piece(sn, pc);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        });
                    });
                });
            });
            __builtin_amdgcn_sched_barrier(0);
#if YMI_PP_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
#if !YMI_PP_ADV_MEM
            if (more) advance();
#endif
            YMI_STAMP(kt);  // 3: MFMAs issued, pointers advanced
            if (half == 0) {
                if (more) wait_vmcnt_barrier<NPC>();  // everything older than step kt+2's pieces
                else wait_vmcnt_barrier<0>();
            } else {
                asm volatile("s_barrier" ::: "memory");
            }
        }
        if (half == 0) asm volatile("s_barrier" ::: "memory");  // the first half waits out the second half's last phase
    } else {
        static_assert(NS >= 2, "the ring needs a stage to fill while another is multiplied");
#pragma unroll
        for (int s = 0; s < NS - 1; ++s)
            if (s < nkt) issue(s);
        for (int kt = 0; kt < nkt; ++kt) {
            YMI_STAMP(kt);  // 0: step start
            // K step kt has landed when at most the loads of the NS-2 younger steps are outstanding
            if (kt + NS - 2 < nkt) {
                if (b_wave) wait_vmcnt_barrier<LPT_FULL * (NS - 2)>();
                else wait_vmcnt_barrier<LPT_AONLY * (NS - 2)>();
            } else {
                wait_vmcnt_barrier<0>();  // pipeline tail: fewer steps in flight than the count assumes
            }
            YMI_STAMP(kt);  // 1: past the wait + barrier
            // every wave has passed the barrier => nobody still reads the buffer of step kt-1: refill it
            if (kt + NS - 1 < nkt && !(YMI_IGEMM_ABL & 1)) issue((kt + NS - 1) % NS);
            YMI_STAMP(kt);  // 2: pieces issued
            const char* As = smem + (kt % NS) * STAGE;
            Mma<T>::template step<TM, TN, CPR>(As, As + BM * ROWB, wm * TM * 16, wn * TN * 16, lane, acc);
            YMI_STAMP(kt);  // 3: fragments read, MFMAs issued
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // epilogue reuses LDS
    YMI_STAMP_DUMP;
    YMI_STAMP_MARK(1);  // K loop done

    // ---- epilogue -----------------------------------------------------------------------------
    // The tile leaves through LDS: lanes drop their 4-channel groups into a [pixel][channel] image, then the
    // workgroup stores it as 16-byte chunks along C, so every store instruction writes whole 128-byte lines
    // (per-lane 8-byte stores to 16 different rows cost 2-3x the time of the same bytes stored this way).
#ifdef YMI_STAMPS
    igemm_epilogue<T, BM, BN, WM, WN, STATS, NT>(a, acc, smem, m0, n0, mb, wm, wn, lane, tid_all, stamp_on, wave_all, stamp_mt0);
#else
    if (YMI_IGEMM_ABL & 16) {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) asm volatile("" :: "v"(acc[tn][tm]));
    } else
    igemm_epilogue<T, BM, BN, WM, WN, STATS, NT>(a, acc, smem, m0, n0, mb, wm, wn, lane, tid_all);
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    YMI_STAMP_MARK(2);  // epilogue done, stores retired
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct TileChoice {
    int bm, bn;
    bool pp = false;  // ping-pong form of the 256x128 tile
};

// Largest tile that still yields ~1.5 workgroups per CU (measured on the 40x40 / 20x20 layers of the model: 400 tiles of
// 128x128 beat 800 of 64x128 by 25-30 %, and below that 128x64, then 64x64, win); short-K GEMMs (K <= 384) are
// prologue/epilogue-dominated and run best as 128x64 (three resident workgroups per CU).
static TileChoice choose_tile(int64_t M, int64_t cout, int64_t ktot, bool bf16) {
    TileChoice t;
    auto blocks = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((cout + bn - 1) / bn); };
    const int64_t enough = 400;
    if (cout <= 32) {
        t.bm = 128; t.bn = 32;
    } else if (cout <= 64) {
        t.bn = 64;
        t.bm = blocks(128, 64) >= enough ? 128 : 64;
    } else if (blocks(128, 128) >= enough) {
        t.bm = 128;
        t.bn = ktot <= 384 ? 64 : 128;  // 1x1 convs up to 384 input channels: 128x64 measured 5-30 % faster, forward and backward (512: slower)
    } else if (blocks(128, 64) >= enough) {
        t.bm = 128; t.bn = 64;
    } else {
        t.bm = 64; t.bn = 64;
    }
    // ping-pong form of the 256x128 tile (the two 128-row halves half a K step apart), where it leaves >= `pp_env` workgroups.
    // Default from 300 workgroups (measured, profiles/r02_conv_bench_pp64.txt: 64-byte rows - three 24 KB stages, TWO resident
    // workgroups per CU - win 5-19 % on every layer that yields >= 400 such tiles and lose 15-25 % at 200; step 14.71 -> 14.38 ms).
    constexpr int pp_env = 300;
    if (bf16 && cout >= 128 && ktot % 32 == 0 && blocks(256, 128) >= pp_env) {
        t.bm = 256; t.bn = 128; t.pp = true;
    }
    const int fbm = ymi_opt(OPT_IGEMM_TILE_BM), fbn = ymi_opt(OPT_IGEMM_TILE_BN);  // force a tile (tools/conv_bench.py sweeps)
    if (fbm > 0 && fbn > 0 && (fbn <= 32 ? cout <= 32 : true) && (fbm < 256 || (bf16 && ktot % 32 == 0))) {  // (the ping-pong tile: bf16, whole 32-deep steps)
        t.bm = fbm;
        t.bn = fbn;
        t.pp = fbm == 256;
    }
    return t;
}

// Row width of the LDS operand images: 128-byte rows (K step = 64 bf16) whenever Cin allows it: every LDS-DMA
// instruction then touches 8 full 128-byte cache lines instead of 16 half lines, and there is one barrier per
// 32 MFMAs per wave instead of per 16.
template <typename T, bool STATS>
static int launch_igemm_t(const IgemmArgs* arr, int ncls, TileChoice t, hipStream_t stream, bool hetero = false) {
    IgemmMulti P{};
    P.ncls = ncls;
    P.hetero = hetero ? 1 : 0;
    int64_t per_xcd[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // hetero: tiles each XCD runs, over all problems
    int mpx = 0;
    for (int i = 0; i < ncls; ++i) {
        P.c[i] = arr[i];
        P.c[i].nmb = (arr[i].M + t.bm - 1) / t.bm;
        P.c[i].nnb = (arr[i].Cout + t.bn - 1) / t.bn;
        const int64_t span = ymi_xcd_span(arr[i].M);
        YMI_CHECK_ARG(8 * span < (1ll << 31), "igemm: M too large");
        P.c[i].span = (int)span;
        for (int x = 0; x < 8; ++x) {  // as the kernel counts them
            const int64_t first = (x * span + t.bm - 1) / t.bm;
            int64_t last = ((x + 1) * span + t.bm - 1) / t.bm;
            if (last > P.c[i].nmb) last = P.c[i].nmb;
            if (last - first > mpx) mpx = (int)(last - first);
            if (last > first) per_xcd[x] += (last - first) * P.c[i].nnb;
        }
    }
    for (int i = 0; i < ncls; ++i) P.c[i].mpx = mpx;  // interleaved form: one id decode for all classes (same Cout => same nnb)
    const IgemmArgs& a = P.c[0];
    dim3 grid((unsigned)(8 * mpx * a.nnb * ncls));
    if (hetero) {
        int64_t mx = 0;
        for (int x = 0; x < 8; ++x) mx = per_xcd[x] > mx ? per_xcd[x] : mx;
        grid = dim3((unsigned)(8 * mx));
    }
    // the K-step form every problem of the launch can run under (8-chunk steps need cpt % 8, whole-step taps cpt % 4; else the general form)
    bool fast = true, wide = std::is_same<T, bf16_t>::value;
    for (int i = 0; i < ncls; ++i) {
        fast = fast && (P.c[i].cpt % 4) == 0;
        wide = wide && (P.c[i].cpt % 8) == 0;
    }
    size_t lds = (size_t)2 * (t.bm + t.bn) * (wide ? 128 : 64);
    const size_t epi = (size_t)t.bm * (t.bn * sizeof(T) + 16) + (STATS ? 4 * 2 * t.bn * sizeof(float) : 0);
    if (epi > lds) lds = epi;
    unsigned nthreads = 256;
#ifdef YMI_STAMPS
#define YMI_STAMP_LDS 4096
#else
#define YMI_STAMP_LDS 0
#endif
#define YMI_LAUNCH1(KERNEL)                                                                                          \
    do {                                                                                                             \
        lds += YMI_STAMP_LDS;                                                                                        \
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(KERNEL, grid, dim3(nthreads), lds, stream, P);                                             \
    } while (0)
#define YMI_LAUNCH(BM, BN, WM, WN)                                                              \
    do {                                                                                        \
        if constexpr (std::is_same<T, bf16_t>::value) {                                         \
            if (wide) { YMI_LAUNCH1((igemm_kernel<T, BM, BN, WM, WN, 2, 8, true, STATS>)); break; } \
        }                                                                                       \
        if (fast) YMI_LAUNCH1((igemm_kernel<T, BM, BN, WM, WN, 2, 4, true, STATS>));            \
        else YMI_LAUNCH1((igemm_kernel<T, BM, BN, WM, WN, 2, 4, false, STATS>));                \
    } while (0)
    if (t.bm == 256 && t.bn == 128) {
        if constexpr (std::is_same<T, bf16_t>::value) {
            YMI_CHECK_ARG(t.pp && fast, "igemm: the 256x128 ping-pong tile needs input channels that are a multiple of 32");
            nthreads = 512;
            lds = (size_t)3 * (256 + 128) * 64;  // 64-byte rows (32-deep K steps): 72 KB, two workgroups per CU (128-byte rows: one, slower)
            if (epi > lds) lds = epi;
            YMI_LAUNCH1((igemm_kernel<T, 256, 128, 4, 2, 3, 4, true, STATS, 512, true>));
        } else {
            ymi_set_error("igemm: the 256x128 tile is bf16 only");
            return YMI_EINVAL;
        }
    } else if (t.bm == 128 && t.bn == 128) YMI_LAUNCH(128, 128, 2, 2);
    else if (t.bm == 128 && t.bn == 64) YMI_LAUNCH(128, 64, 2, 2);
    else if (t.bm == 128 && t.bn == 32) YMI_LAUNCH(128, 32, 4, 1);
    else if (t.bm == 64 && t.bn == 128) YMI_LAUNCH(64, 128, 2, 2);
    else if (t.bm == 64 && t.bn == 64) YMI_LAUNCH(64, 64, 2, 2);
    else {
        ymi_set_error("igemm: no tile %dx%d", t.bm, t.bn);
        return YMI_EINVAL;
    }
#undef YMI_LAUNCH
#undef YMI_LAUNCH1
    YMI_CHECK_LAUNCH("igemm");
    return YMI_OK;
}


bool ymi_prof_enabled();
int ymi_prof_start(hipStream_t stream, int family, double flop, double bytes, double peak_tflops);
void ymi_prof_stop(hipStream_t stream, int idx);

// ncls problems in one launch.  hetero == false: same dtype, Cout, channel geometry (the interleaved form: parity classes of a stride-2
// data gradient).  hetero == true: any shapes, one after another; they agree on dtype, and the launch runs in the K-step
// form ALL of them allow (8-chunk steps only when every problem's input width is a multiple of 64 bf16 channels).  host_blocks: ncls entries (statistics rows per problem).
static int launch_igemm_n(const IgemmArgs* arr, int ncls, int dtype, bool stats, int* host_blocks, hipStream_t stream, bool hetero = false) {
    YMI_CHECK_ARG(ncls >= 1 && ncls <= IGEMM_MAX_PROBLEMS, "igemm: %d problems in one launch (at most %d)", ncls, IGEMM_MAX_PROBLEMS);
    int64_t mmax = 0, msum = 0;
    int64_t kmax = 0;
    int cmax = 0, cmin = 1 << 30;
    for (int i = 0; i < ncls; ++i) {
        mmax = arr[i].M > mmax ? arr[i].M : mmax;
        msum += arr[i].M;
        kmax = arr[i].ktot > kmax ? arr[i].ktot : kmax;
        cmax = arr[i].Cout > cmax ? arr[i].Cout : cmax;
        cmin = arr[i].Cout < cmin ? arr[i].Cout : cmin;
    }
    // one tile form for the launch: chosen for the widest problem over all the rows (a narrower problem pads its N block)
    TileChoice t = hetero ? choose_tile(msum, cmax, kmax, dtype == YMI_BF16) : choose_tile(mmax * ncls, arr[0].Cout, kmax, dtype == YMI_BF16);
    if (hetero && t.bn > 64 && cmin <= 64 && cmax > 64 && !t.pp) t.bn = 64;  // (mixed widths: 64-column tiles waste nothing on the narrow ones)
    if (host_blocks)
        for (int i = 0; i < (hetero ? ncls : 1); ++i) host_blocks[i] = (arr[i].M + t.bm - 1) / t.bm;
    int prof = -1;
    if (ymi_prof_enabled()) {
        const double es = dtype == YMI_BF16 ? 2.0 : 4.0;
        double flop = 0.0, bytes = 0.0;
        for (int i = 0; i < ncls; ++i) {
            const IgemmArgs& a = arr[i];
            flop += 2.0 * (double)a.M * (double)a.Cout * (double)a.ktot;
            bytes += (double)a.ktot * a.Cout * es + (double)a.M * a.Cout * es * (a.res ? 2.0 : 1.0);
        }
        for (int i = 0; i < (hetero ? ncls : 1); ++i) {
            const IgemmArgs& a0 = arr[i];
            bytes += (double)a0.M / ((double)a0.Ho * a0.Wo) * (double)a0.H * a0.W * a0.cpt * 16.0;  // the whole input map, read once
        }
        prof = ymi_prof_start(stream, 0, flop, bytes, dtype == YMI_BF16 ? 2500.0 : 157.3);
    }
    int rc;
    if (dtype == YMI_BF16) rc = stats ? launch_igemm_t<bf16_t, true>(arr, ncls, t, stream, hetero) : launch_igemm_t<bf16_t, false>(arr, ncls, t, stream, hetero);
    else rc = stats ? launch_igemm_t<float, true>(arr, ncls, t, stream, hetero) : launch_igemm_t<float, false>(arr, ncls, t, stream, hetero);
    ymi_prof_stop(stream, prof);
    return rc;
}


int ymi_launch_igemm(const IgemmArgs& a, int dtype, bool stats, int* host_blocks, hipStream_t stream) {
    return launch_igemm_n(&a, 1, dtype, stats, host_blocks, stream);
}

static void find_divisor(int d, uint32_t* mul, uint32_t* shr) {
    if (d <= 1) { *mul = 0; *shr = 0; return; }
    int lg = 0;
    while ((1 << lg) < d) ++lg;
    const int p = 31 + lg;
    *mul = (uint32_t)(((1ull << p) + (uint64_t)d - 1) / (uint64_t)d);
    *shr = (uint32_t)(p - 32);
}
static void finish_args(IgemmArgs& a, const ymi_tensor* y, const ymi_tensor* residual) {
    find_divisor(a.Wo, &a.wo_mul, &a.wo_shr);
    find_divisor(a.Ho, &a.ho_mul, &a.ho_shr);
    const size_t es = ymi_esize(y->dtype);
    const int epc = (int)(16 / es);
    a.vec16 = a.vec_store && (y->ld % epc == 0) && (((uintptr_t)y->data) % 16 == 0) && (a.Cout % epc == 0);
    (void)residual;
}

static void pack_taps(const int* dh, const int* dw, int n, uint64_t* pdh, uint64_t* pdw) {
    uint64_t a = 0, b = 0;
    for (int i = 0; i < n; ++i) {
        a |= (uint64_t)((dh[i] + 8) & 15) << (4 * i);
        b |= (uint64_t)((dw[i] + 8) & 15) << (4 * i);
    }
    *pdh = a;
    *pdw = b;
}

extern "C" int64_t ymi_conv2d_stat_blocks(int64_t m_rows, int64_t cout) {
    (void)cout;
    return (m_rows + 63) / 64 + 64;  // smallest BM any tile choice uses, + the 64 staging rows ymi_bn_finalize may append
}

static int conv_fwd_args(const ymi_tensor* x, const void* w_packed, int64_t cout, int64_t kh, int64_t kw, int64_t stride,
                         const float* scale, const float* bias, int32_t act, const ymi_tensor* residual, const ymi_tensor* y,
                         const ymi_tensor* y2, int32_t act2, float* stat_partials, IgemmArgs* out);
static int conv_fwd_impl(const ymi_tensor* x, const void* w_packed, int64_t cout, int64_t kh, int64_t kw, int64_t stride,
                         const float* scale, const float* bias, int32_t act, const ymi_tensor* residual, const ymi_tensor* y,
                         const ymi_tensor* y2, int32_t act2, float* stat_partials, int64_t* host_stat_blocks, void* stream) {
    IgemmArgs a{};
    int rc = conv_fwd_args(x, w_packed, cout, kh, kw, stride, scale, bias, act, residual, y, y2, act2, stat_partials, &a);
    if (rc) return rc;
    int blocks = 0;
    rc = ymi_launch_igemm(a, x->dtype, stat_partials != nullptr, &blocks, (hipStream_t)stream);
    if (host_stat_blocks) *host_stat_blocks = blocks;
    return rc;
}
static int conv_fwd_args(const ymi_tensor* x, const void* w_packed, int64_t cout, int64_t kh, int64_t kw, int64_t stride,
                         const float* scale, const float* bias, int32_t act, const ymi_tensor* residual, const ymi_tensor* y,
                         const ymi_tensor* y2, int32_t act2, float* stat_partials, IgemmArgs* out) {
    YMI_CHECK_ARG(ymi_tensor_ok(x) && ymi_tensor_ok(y) && w_packed, "conv2d_fwd: bad tensor");
    if (y2) {
        YMI_CHECK_ARG(ymi_tensor_ok(y2) && ymi_same_shape(y2, y) && y2->dtype == y->dtype && !stat_partials, "conv2d_fwd: second output");
        const int epc = (int)(16 / ymi_esize(y->dtype));
        YMI_CHECK_ARG(y2->ld % epc == 0 && ((uintptr_t)y2->data & 15) == 0 && y->ld % epc == 0 && ((uintptr_t)y->data & 15) == 0 && cout % epc == 0,
                      "conv2d_fwd: a second output needs 16-byte-aligned rows");
    }
    YMI_CHECK_ARG(x->dtype == y->dtype, "conv2d_fwd: dtype mismatch");
    const int ch = x->dtype == YMI_BF16 ? 8 : 4;
    YMI_CHECK_ARG(x->c % ch == 0 && x->ld % ch == 0, "conv2d_fwd: input channels (%lld, ld %lld) must be a multiple of %d",
                  (long long)x->c, (long long)x->ld, ch);
    YMI_CHECK_ARG(((uintptr_t)x->data & 15) == 0 && ((uintptr_t)w_packed & 15) == 0, "conv2d_fwd: 16-byte alignment");
    YMI_CHECK_ARG(kh == kw && (kh == 1 || kh == 3) && (stride == 1 || stride == 2), "conv2d_fwd: k in {1,3}, stride in {1,2}");
    const int64_t pad = kh / 2;
    const int64_t ho = (x->h + 2 * pad - kh) / stride + 1, wo = (x->w + 2 * pad - kw) / stride + 1;
    YMI_CHECK_ARG(y->n == x->n && y->h == ho && y->w == wo && y->c == cout, "conv2d_fwd: output shape");
    YMI_CHECK_ARG(x->n * ho * wo < (1ll << 31) && ymi_pixels(x) * x->ld < (1ll << 31) && ymi_pixels(y) * y->ld < (1ll << 31),
                  "conv2d_fwd: tensor too large for 32-bit indexing");
    if (residual) YMI_CHECK_ARG(ymi_tensor_ok(residual) && ymi_same_shape(residual, y) && residual->dtype == y->dtype, "conv2d_fwd: residual");
    YMI_CHECK_ARG(!(stat_partials && (scale || bias || residual || act != YMI_ACT_NONE)), "conv2d_fwd: statistics mode stores the raw output");

    IgemmArgs a{};
    a.x = x->data; a.w = w_packed; a.y = y->data; a.res = residual ? residual->data : nullptr;
    a.scale = scale; a.bias = bias; a.partials = stat_partials; a.zero = ymi_zero_page();
    a.pstride = (int)cout; a.poff = 0;
    a.ldx = x->ld; a.ldy = y->ld; a.ldres = residual ? residual->ld : 0;
    a.M = (int)(x->n * ho * wo); a.H = (int)x->h; a.W = (int)x->w; a.Ho = (int)ho; a.Wo = (int)wo; a.Hy = (int)ho; a.Wy = (int)wo;
    a.s_in = (int)stride; a.s_out = 1; a.oh_off = 0; a.ow_off = 0;
    a.Cout = (int)cout; a.cpt = (int)(x->c / ch); a.ntaps = (int)(kh * kw); a.KC = a.ntaps * a.cpt; a.ktot = (int64_t)a.KC * ch;
    int dh[9], dw[9];
    for (int i = 0; i < kh; ++i)
        for (int j = 0; j < kw; ++j) { dh[i * kw + j] = i - (int)pad; dw[i * kw + j] = j - (int)pad; }
    pack_taps(dh, dw, a.ntaps, &a.tap_dh, &a.tap_dw);
    a.act = act;
    a.y2 = y2 ? y2->data : nullptr; a.ldy2 = y2 ? y2->ld : 0; a.act2 = act2;
    const int g = 4;
    a.vec_store = (y->ld % g == 0) && (((uintptr_t)y->data) % (g * ymi_esize(y->dtype)) == 0) &&
                  (!residual || (residual->ld % g == 0 && ((uintptr_t)residual->data) % (g * ymi_esize(y->dtype)) == 0));
    finish_args(a, y, residual);
    *out = a;
    return YMI_OK;
}

// Several independent convolutions in ONE launch (problems one after another: IgemmMulti, hetero form) - the same stage of Detect's three
// levels (reference nn/modules/head.py:66-74 runs them in a Python loop), whose 40 x 40 and 20 x 20 levels are too small to fill the chip
// on their own.  All problems: the same dtype, statistics mode for all or none (callers group problems of like input-width class: a mixed launch runs in the slowest form).
extern "C" int ymi_conv2d_fwd_multi(const ymi_conv_problem* problems, int32_t n, void* stream) {
    YMI_CHECK_ARG(problems && n >= 1 && n <= IGEMM_MAX_PROBLEMS, "conv2d_fwd_multi: 1..%d problems", IGEMM_MAX_PROBLEMS);
    IgemmArgs arr[IGEMM_MAX_PROBLEMS];
    int blocks[IGEMM_MAX_PROBLEMS];
    const bool stats = problems[0].stat_partials != nullptr;
    for (int i = 0; i < n; ++i) {
        const ymi_conv_problem& p = problems[i];
        YMI_CHECK_ARG(p.x && p.y && p.x->dtype == problems[0].x->dtype && (p.stat_partials != nullptr) == stats, "conv2d_fwd_multi: problem %d: dtype / statistics mode differ", i);
        int rc = conv_fwd_args(p.x, p.w_packed, p.cout, p.kh, p.kw, p.stride, p.scale, p.bias, p.act, p.residual, p.y, nullptr, YMI_ACT_NONE, p.stat_partials, &arr[i]);
        if (rc) return rc;
        if (p.stat_stride > 0) {
            YMI_CHECK_ARG(p.stat_offset >= 0 && p.stat_offset + p.cout <= p.stat_stride, "conv2d_fwd_multi: problem %d: statistics columns", i);
            arr[i].pstride = (int)p.stat_stride;
            arr[i].poff = (int)p.stat_offset;
        }
    }
    int rc = launch_igemm_n(arr, n, problems[0].x->dtype, stats, blocks, (hipStream_t)stream, true);
    if (rc) return rc;
    for (int i = 0; i < n; ++i) const_cast<ymi_conv_problem*>(problems)[i].stat_blocks = blocks[i];
    return YMI_OK;
}

// raw convolution output + BatchNorm statistics as fixed-point atomic sums (stat_acc: [4][2][cout] int64, ZERO on entry); used by
// ymi_conv2d_bn_silu_fwd_acc (elementwise.hip)
int ymi_conv2d_fwd_statacc(const ymi_tensor* x, const void* w_packed, int64_t cout, int64_t kh, int64_t kw, int64_t stride, const ymi_tensor* y,
                           long long* stat_acc, void* stream) {
    IgemmArgs a{};
    float dummy = 0.f;  // (statistics mode is selected by a non-null row pointer; it is never written when stat_acc is set)
    int rc = conv_fwd_args(x, w_packed, cout, kh, kw, stride, nullptr, nullptr, YMI_ACT_NONE, nullptr, y, nullptr, YMI_ACT_NONE, &dummy, &a);
    if (rc) return rc;
    a.partials = nullptr;
    a.stat_acc = stat_acc;
    a.stat_scale = (float)ldexp(1.0, ymi_stat_fixed_point_shift(ymi_pixels(y)));
    return ymi_launch_igemm(a, x->dtype, true, nullptr, (hipStream_t)stream);
}

extern "C" int ymi_conv2d_fwd(const ymi_tensor* x, const void* w_packed, int64_t cout, int64_t kh, int64_t kw, int64_t stride,
                              const float* scale, const float* bias, int32_t act, const ymi_tensor* residual, const ymi_tensor* y,
                              float* stat_partials, int64_t* host_stat_blocks, void* stream) {
    return conv_fwd_impl(x, w_packed, cout, kh, kw, stride, scale, bias, act, residual, y, nullptr, YMI_ACT_NONE, stat_partials, host_stat_blocks, stream);
}

// dx = sum over taps of dy (x) w : stride 1 -> one launch; stride 2 -> one launch per output parity class.
extern "C" int64_t ymi_conv_dgrad_pack_elems(int64_t o, int64_t i, int64_t kh, int64_t kw, int64_t stride) {
    (void)stride;
    return o * i * kh * kw;  // the classes partition the taps
}

extern "C" int ymi_conv2d_bwd_data(const ymi_tensor* dy, const void* w_dgrad_packed, int64_t cin, int64_t kh, int64_t kw,
                                   int64_t stride, const ymi_tensor* dx, void* stream) {
    return ymi_conv2d_bwd_data_add(dy, w_dgrad_packed, cin, kh, kw, stride, nullptr, nullptr, dx, stream);
}

static int dgrad_impl(const ymi_tensor* dy, const void* w_dgrad_packed, int64_t cin, int64_t kh, int64_t kw, int64_t stride,
                      const ymi_tensor* add1, const ymi_tensor* add2, const ymi_tensor* mul, int32_t mul_act, const ymi_tensor* dx, void* stream);

extern "C" int ymi_conv2d_bwd_data_add(const ymi_tensor* dy, const void* w_dgrad_packed, int64_t cin, int64_t kh, int64_t kw,
                                       int64_t stride, const ymi_tensor* add1, const ymi_tensor* add2, const ymi_tensor* dx, void* stream) {
    return dgrad_impl(dy, w_dgrad_packed, cin, kh, kw, stride, add1, add2, nullptr, YMI_ACT_NONE, dx, stream);
}

static int dgrad_impl(const ymi_tensor* dy, const void* w_dgrad_packed, int64_t cin, int64_t kh, int64_t kw, int64_t stride,
                      const ymi_tensor* add1, const ymi_tensor* add2, const ymi_tensor* mul, int32_t mul_act, const ymi_tensor* dx, void* stream) {
    YMI_CHECK_ARG(ymi_tensor_ok(dy) && ymi_tensor_ok(dx) && w_dgrad_packed, "conv2d_bwd_data: bad tensor");
    if (mul) {
        const int epc = (int)(16 / ymi_esize(dx->dtype));
        YMI_CHECK_ARG(ymi_tensor_ok(mul) && ymi_same_shape(mul, dx) && mul->dtype == dx->dtype && mul->ld % 4 == 0 &&
                          ((uintptr_t)mul->data) % (4 * ymi_esize(dx->dtype)) == 0 && dx->ld % epc == 0 && ((uintptr_t)dx->data & 15) == 0 && cin % epc == 0,
                      "conv2d_bwd_data: the multiplier needs the output's shape and 16-byte-aligned output rows");
    }
    if (add1 || add2) {
        YMI_CHECK_ARG(add1 && ymi_tensor_ok(add1) && ymi_same_shape(add1, dx) && add1->dtype == dx->dtype, "conv2d_bwd_data_add: first addend");
        YMI_CHECK_ARG(!add2 || (ymi_tensor_ok(add2) && ymi_same_shape(add2, dx) && add2->dtype == dx->dtype), "conv2d_bwd_data_add: second addend");
    }
    YMI_CHECK_ARG(dy->dtype == dx->dtype, "conv2d_bwd_data: dtype mismatch");
    const int ch = dy->dtype == YMI_BF16 ? 8 : 4;
    YMI_CHECK_ARG(dy->c % ch == 0 && dy->ld % ch == 0, "conv2d_bwd_data: dy channels must be a multiple of %d", ch);
    YMI_CHECK_ARG(kh == kw && (kh == 1 || kh == 3) && (stride == 1 || stride == 2), "conv2d_bwd_data: k in {1,3}, stride in {1,2}");
    const int64_t pad = kh / 2;
    YMI_CHECK_ARG(dx->c == cin && dx->n == dy->n && dy->h == (dx->h + 2 * pad - kh) / stride + 1 && dy->w == (dx->w + 2 * pad - kw) / stride + 1,
                  "conv2d_bwd_data: shapes");
    YMI_CHECK_ARG(ymi_pixels(dx) * dx->ld < (1ll << 31) && ymi_pixels(dy) * dy->ld < (1ll << 31), "conv2d_bwd_data: too large");
    const size_t es = ymi_esize(dy->dtype);
    const char* wbase = reinterpret_cast<const char*>(w_dgrad_packed);
    int64_t woff = 0;  // elements
    const int nclass = stride == 1 ? 1 : 4;
    IgemmArgs classes[4];
    int nlaunch = 0;
    for (int cls = 0; cls < nclass; ++cls) {
        const int ph = stride == 1 ? 0 : cls / 2, pw = stride == 1 ? 0 : cls % 2;
        int dh[9], dw[9], nt = 0;
        for (int i = 0; i < kh; ++i)
            for (int j = 0; j < kw; ++j) {
                const int nh = ph + (int)pad - i, nw = pw + (int)pad - j;
                if (nh % (int)stride != 0 || nw % (int)stride != 0) continue;
                dh[nt] = nh / (int)stride; dw[nt] = nw / (int)stride; ++nt;
            }
        const int64_t ho = (dx->h - ph + stride - 1) / stride, wo = (dx->w - pw + stride - 1) / stride;
        if (nt > 0 && ho > 0 && wo > 0) {
            IgemmArgs a{};
            a.x = dy->data; a.w = wbase + woff * es; a.y = dx->data; a.zero = ymi_zero_page();
            a.ldx = dy->ld; a.ldy = dx->ld;
            a.res = add1 ? add1->data : nullptr; a.ldres = add1 ? add1->ld : 0;
            a.res2 = add2 ? add2->data : nullptr; a.ldres2 = add2 ? add2->ld : 0;
            a.M = (int)(dx->n * ho * wo); a.H = (int)dy->h; a.W = (int)dy->w; a.Ho = (int)ho; a.Wo = (int)wo; a.Hy = (int)dx->h; a.Wy = (int)dx->w;
            a.s_in = 1; a.s_out = (int)stride; a.oh_off = ph; a.ow_off = pw;
            a.Cout = (int)cin; a.cpt = (int)(dy->c / ch); a.ntaps = nt; a.KC = nt * a.cpt; a.ktot = (int64_t)a.KC * ch;
            pack_taps(dh, dw, nt, &a.tap_dh, &a.tap_dw);
            a.act = YMI_ACT_NONE;
            a.mul = mul ? mul->data : nullptr; a.ldmul = mul ? mul->ld : 0; a.mul_act = mul_act;
            a.vec_store = (dx->ld % 4 == 0) && (((uintptr_t)dx->data) % (4 * es) == 0) &&
                          (!add1 || (add1->ld % 4 == 0 && ((uintptr_t)add1->data) % (4 * es) == 0)) &&
                          (!add2 || (add2->ld % 4 == 0 && ((uintptr_t)add2->data) % (4 * es) == 0));
            finish_args(a, dx, nullptr);
            classes[nlaunch++] = a;
        } else if (ho > 0 && wo > 0) {
            // a parity class no tap reaches (k = 1, stride 2: three of the four classes): its pixels receive no gradient.  They are left
            // as the caller prepared them - the contract of this case: dx pre-filled with zeros, addends applied by the caller
            if (add1 || add2) {
                ymi_set_error("conv2d_bwd_data: epilogue addends are not available when a parity class has no taps (k=1 stride=2): add them separately");
                return YMI_EINVAL;
            }
        }
        woff += (int64_t)nt * dy->c * cin;
    }
    // the parity classes of a stride-2 data gradient as ONE multi-problem launch (measured: pays from 64 output channels up; the
    // 32-channel layer 1 is 6 % faster class by class)
    if (cin >= 64 && nlaunch > 1) return launch_igemm_n(classes, nlaunch, dy->dtype, false, nullptr, (hipStream_t)stream);
    for (int i = 0; i < nlaunch; ++i) {
        int rc = ymi_launch_igemm(classes[i], dy->dtype, false, nullptr, (hipStream_t)stream);
        if (rc) return rc;
    }
    return YMI_OK;
}

// Several independent stride-1 data gradients in one launch (ymi_conv2d_fwd_multi's counterpart; each problem may carry its epilogue addends).
static int dgrad_args_s1(const ymi_tensor* dy, const void* w_dgrad_packed, int64_t cin, int64_t k, const ymi_tensor* add1, const ymi_tensor* add2, const ymi_tensor* dx, IgemmArgs* out) {
    YMI_CHECK_ARG(ymi_tensor_ok(dy) && ymi_tensor_ok(dx) && w_dgrad_packed, "conv2d_bwd_data_multi: bad tensor");
    if (add1 || add2) {
        YMI_CHECK_ARG(add1 && ymi_tensor_ok(add1) && ymi_same_shape(add1, dx) && add1->dtype == dx->dtype, "conv2d_bwd_data_multi: first addend");
        YMI_CHECK_ARG(!add2 || (ymi_tensor_ok(add2) && ymi_same_shape(add2, dx) && add2->dtype == dx->dtype), "conv2d_bwd_data_multi: second addend");
    }
    YMI_CHECK_ARG(dy->dtype == dx->dtype, "conv2d_bwd_data_multi: dtype mismatch");
    const int ch = dy->dtype == YMI_BF16 ? 8 : 4;
    YMI_CHECK_ARG(dy->c % ch == 0 && dy->ld % ch == 0 && (k == 1 || k == 3), "conv2d_bwd_data_multi: dy channels in whole chunks, k in {1, 3}");
    YMI_CHECK_ARG(dx->c == cin && dx->n == dy->n && dy->h == dx->h && dy->w == dx->w, "conv2d_bwd_data_multi: shapes (stride 1)");
    YMI_CHECK_ARG(ymi_pixels(dx) * dx->ld < (1ll << 31) && ymi_pixels(dy) * dy->ld < (1ll << 31), "conv2d_bwd_data_multi: too large");
    const size_t es = ymi_esize(dy->dtype);
    const int pad = (int)k / 2;
    int dh[9], dw[9], nt = 0;
    for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j) { dh[nt] = pad - i; dw[nt] = pad - j; ++nt; }
    IgemmArgs a{};
    a.x = dy->data; a.w = w_dgrad_packed; a.y = dx->data; a.zero = ymi_zero_page();
    a.ldx = dy->ld; a.ldy = dx->ld;
    a.res = add1 ? add1->data : nullptr; a.ldres = add1 ? add1->ld : 0;
    a.res2 = add2 ? add2->data : nullptr; a.ldres2 = add2 ? add2->ld : 0;
    a.M = (int)ymi_pixels(dx); a.H = (int)dy->h; a.W = (int)dy->w; a.Ho = (int)dx->h; a.Wo = (int)dx->w; a.Hy = (int)dx->h; a.Wy = (int)dx->w;
    a.s_in = 1; a.s_out = 1; a.oh_off = 0; a.ow_off = 0;
    a.Cout = (int)cin; a.cpt = (int)(dy->c / ch); a.ntaps = nt; a.KC = nt * a.cpt; a.ktot = (int64_t)a.KC * ch;
    pack_taps(dh, dw, nt, &a.tap_dh, &a.tap_dw);
    a.act = YMI_ACT_NONE;
    a.vec_store = (dx->ld % 4 == 0) && (((uintptr_t)dx->data) % (4 * es) == 0) && (!add1 || (add1->ld % 4 == 0 && ((uintptr_t)add1->data) % (4 * es) == 0)) &&
                  (!add2 || (add2->ld % 4 == 0 && ((uintptr_t)add2->data) % (4 * es) == 0));
    finish_args(a, dx, nullptr);
    *out = a;
    return YMI_OK;
}
extern "C" int ymi_conv2d_bwd_data_multi(const ymi_dgrad_problem* problems, int32_t n, void* stream) {
    YMI_CHECK_ARG(problems && n >= 1 && n <= IGEMM_MAX_PROBLEMS, "conv2d_bwd_data_multi: 1..%d problems", IGEMM_MAX_PROBLEMS);
    IgemmArgs arr[IGEMM_MAX_PROBLEMS];
    for (int i = 0; i < n; ++i) {
        const ymi_dgrad_problem& p = problems[i];
        YMI_CHECK_ARG(p.dy && p.dx && p.dy->dtype == problems[0].dy->dtype, "conv2d_bwd_data_multi: problem %d", i);
        int rc = dgrad_args_s1(p.dy, p.w_dgrad_packed, p.cin, p.k, p.add1, p.add2, p.dx, &arr[i]);
        if (rc) return rc;
    }
    return launch_igemm_n(arr, n, problems[0].dy->dtype, false, nullptr, (hipStream_t)stream, true);
}

// ---- SwinBlock MLP (swin_block.py:33,53: Linear(C, 4C) -> GELU -> Linear(4C, C), + the skip) ------------------------------------
// forward: two GEMM launches.  fc1's epilogue stores the pre-activation (saved for backward) AND its exact-erf GELU, so the
// [T, 4C] matrix is written twice and never read back by an activation kernel; fc2 adds the bias and the skip in its epilogue.
extern "C" int ymi_swin_mlp_fwd(const ymi_tensor* u, const void* w1_packed, const float* b1, int64_t hidden, const void* w2_packed, const float* b2,
                                const ymi_tensor* residual, const ymi_tensor* pre, const ymi_tensor* post, const ymi_tensor* out, void* stream) {
    YMI_CHECK_ARG(ymi_tensor_ok(u) && ymi_tensor_ok(pre) && ymi_tensor_ok(post) && ymi_tensor_ok(out) && w1_packed && w2_packed, "swin_mlp_fwd: bad tensor");
    YMI_CHECK_ARG(pre->c == hidden && post->c == hidden && out->c == u->c, "swin_mlp_fwd: channels");
    int rc = conv_fwd_impl(u, w1_packed, hidden, 1, 1, 1, nullptr, b1, YMI_ACT_NONE, nullptr, pre, post, YMI_ACT_GELU, nullptr, nullptr, stream);
    if (rc) return rc;
    return conv_fwd_impl(post, w2_packed, out->c, 1, 1, 1, nullptr, b2, YMI_ACT_NONE, residual, out, nullptr, YMI_ACT_NONE, nullptr, nullptr, stream);
}

// backward, data path: d_pre = (d_out . W2) * gelu'(pre) in fc2's data-gradient epilogue (no [T, 4C] gradient of the activation
// output is ever stored), then d_u = d_pre . W1 (+ up to two addends: the gradient sums of the tensor LayerNorm-2 produced).
// Weight and bias gradients are ordinary ymi_conv2d_bwd_weight calls on (post, d_out) and (u, d_pre).
extern "C" int ymi_swin_mlp_bwd_data(const ymi_tensor* dout, const void* w2_dgrad_packed, const ymi_tensor* pre, const ymi_tensor* dpre,
                                     const void* w1_dgrad_packed, const ymi_tensor* add1, const ymi_tensor* add2, const ymi_tensor* du, void* stream) {
    YMI_CHECK_ARG(ymi_tensor_ok(dout) && ymi_tensor_ok(pre) && ymi_tensor_ok(dpre) && w2_dgrad_packed, "swin_mlp_bwd_data: bad tensor");
    int rc = dgrad_impl(dout, w2_dgrad_packed, pre->c, 1, 1, 1, nullptr, nullptr, pre, YMI_ACT_GELU, dpre, stream);
    if (rc || !du) return rc;
    YMI_CHECK_ARG(ymi_tensor_ok(du) && w1_dgrad_packed, "swin_mlp_bwd_data: bad tensor");
    return dgrad_impl(dpre, w1_dgrad_packed, du->c, 1, 1, 1, add1, add2, nullptr, YMI_ACT_NONE, du, stream);
}
