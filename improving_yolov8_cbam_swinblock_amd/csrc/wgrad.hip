// Weight gradient of the NHWC convolution as a split-K MFMA GEMM:
//
//   dW[co][(tap, ci)] = sum over output pixels m of  dY[m][co] * X[pixel(m, tap)][ci]
//
// Both operands are "K-major" in memory (the reduction index m is the slow one), so tiles are staged
// in LDS exactly as they lie in HBM ([pixel][channel] rows, filled by 16-byte LDS-DMA) and the MFMA
// fragments are read TRANSPOSED with ds_read_b64_tr_b16 (bf16) or plain ds_read_b32 (f32 parity
// mode, 16x16x4 MFMA).  The pixel axis is split across workgroups; each split writes a slab of partial
// sums, a second kernel sums the slabs in float32 in a fixed order (deterministic, no atomics) and scatters
// into the reference's OIHW layout.  Slabs are float32 in parity mode and bfloat16 in the bf16 path
// (round 3): the first-level partials of a split - a sum over >= 256 pixels held in float32 registers -
// are rounded once on their way out, the second level stays float32.  That halves the slab traffic
// (36 MB -> 18 MB written per launch at bs 32, and the batched reduce reads half as much).
#include <stdlib.h>

#include <mutex>
#include <type_traits>

#include "common.h"

int ymi_chan_reduce_final(const float* part, int blocks, int C, float* out0, float* out1, hipStream_t stream);
bool ymi_prof_enabled();
int ymi_prof_start(hipStream_t stream, int family, double flop, double bytes, double peak_tflops);
void ymi_prof_stop(hipStream_t stream, int idx);

#ifdef YMI_STAMPS
// diagnostic build only (see igemm.hip): cycles from kernel entry at three marks of one workgroup + the in-kernel clock
__device__ unsigned long long* g_wstamp_buf = nullptr;
extern "C" int ymi_debug_stamp_buffer_wgrad(void* p) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_wstamp_buf), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#define WG_MARK(i)                                                                                              \
    do {                                                                                                        \
        if (wstamp_on) g_wstamp_buf[8 * 64 + 16 + wave * 8 + (i)] = __builtin_amdgcn_s_memtime() - wstamp_mt0; \
    } while (0)
#else
#define WG_MARK(i) do { } while (0)
#endif

#ifdef YMI_STAMPS
#define WG_STAMP_LDS 4096
#else
#define WG_STAMP_LDS 0
#endif
struct WgradArgs {
    const void* x;
    const void* dy;
    void* slab;
    int slab_bf16;
    float* bias_slab;  // optional [splits][CoutP]: column sums of dY per split (the bias gradient's partials), formed by the tiles of column block 0
    const void* zero;
    int64_t ldx, ldy;
    int Mpix, H, W, Ho, Wo;
    int stride, pad, KW;
    int CoutP, Cin, NG;
    int pix_per_split;
    int nx, ny, splits, xcd_map;  // launch geometry (set by the launcher)
    int pw_shift;                 // PATCH kernels: a K step is a (32 >> pw_shift) x (1 << pw_shift) patch of output pixels (see wgrad_kernel)
    uint32_t x_bytes, y_bytes;    // ... and the operands as buffers (sizes in bytes, < 2^31)
    uint32_t wo_mul, wo_shr, ho_mul, ho_shr;  // n / Wo, n / Ho by multiply-high (wg_fast_div)
};

// n / d for 0 <= n < 2^31 as (umulhi(n, mul) + n) >> shr with mul = floor(2^32 (2^shr - d) / d) + 1, shr = ceil(log2 d)
// (Granlund-Montgomery; exact also for d = 1 and powers of two, so the K loop needs no special case and no branch)
__device__ __forceinline__ int wg_fast_div(int n, uint32_t mul, uint32_t shr) { return (int)((__umulhi((uint32_t)n, mul) + (uint32_t)n) >> shr); }

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;

constexpr int WG_BM = 64;    // rows  (co)
constexpr int WG_BN = 128;   // cols  ((tap, ci))
constexpr int WG_BK = 32;    // pixels per K step (64 was measured 10-15 % slower: half as many resident workgroups per CU)

#ifndef YMI_WGRAD_ABL  // diagnostic builds (results wrong by design): bit 1 no LDS-DMA pieces inside the K loop, 2 no MFMAs, 4 no fragment reads, 8 no pixel walk, 16 no slab stores
#define YMI_WGRAD_ABL 0
#endif
#ifndef YMI_WGRAD_NS  // LDS ring stages of the bf16 kernels
#define YMI_WGRAD_NS 2
#endif
#ifndef YMI_WGRAD_DIRECT_SLAB  // 1: bfloat16 slabs stored straight from the accumulators (the form before round 5; A/B builds)
#define YMI_WGRAD_DIRECT_SLAB 0
#endif
#ifndef YMI_WGRAD_WAVES
#define YMI_WGRAD_WAVES 5  // waves per SIMD the register allocation must allow (the kernel is latency-bound: occupancy pays)
#endif
// `buffer_load_dwordx4 ... lds`: 16 bytes per lane from base + voff + soff into LDS (lane-linear behind `dst`), zeros for lanes whose offset is
// outside [0, bytes).  The resource is rebuilt from (base, bytes) at every call - four scalar moves - because a local of the resource type in a
// kernel TEMPLATE makes the host pass drop the instantiation's launch stub without a word (ROCm 7.2 clang): the type stays inside this function.
__device__ __forceinline__ void wg_buffer_to_lds(const void* base, uint32_t bytes, lptr_t dst, uint32_t voff, uint32_t soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000), dst, 16, voff, soff, 0, 0);
#endif
}
template <typename T> struct WFrag;
// LDS bank swizzle for the transposed reads (bf16).  ds_read_b64_tr_b16 is serviced per 32-lane half: 2 groups x
// (4 rows x 4 eight-byte units); the 8 rows of a half are {r0..r0+3, r0+8..r0+11}.  With 128-byte (dY) or 256-byte
// (X) rows those rows share banks (4-way / 8-way conflict on the plain image), so the 8-byte unit index is XORed
// with a per-row value that spreads the 8 rows over the 32 unit slots of a 256-byte bank row.  The XOR never touches
// bit 0 of the unit index, so it is a permutation of 16-byte chunks and can be applied to the LDS-DMA SOURCE address.
__device__ __forceinline__ int wg_swz_y(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }        // 2 bits
__device__ __forceinline__ int wg_swz_x(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }                // 3 bits
// 64-byte rows (the dY image of the 32-channel tile): rows r and r+8 of a half share banks, rows r..r+3 do not: one bit
__device__ __forceinline__ int wg_swz_y32(int row) { return (row >> 3) & 1; }
// swizzle mode of an LDS image by its row width: 0 = 128-byte rows, 1 = 256-byte (and wider) rows, 2 = 64-byte rows
template <int SW> __device__ __forceinline__ int wg_swz(int row) { return SW == 1 ? wg_swz_x(row) : SW == 2 ? wg_swz_y32(row) : wg_swz_y(row); }
template <int BM> struct YSwz { static constexpr int SW = BM == 128 ? 1 : BM == 32 ? 2 : 0; };

template <> struct WFrag<bf16_t> {
    // a fragment = 8 k-values (pixels 8g..8g+7) for column c0+i of a [pixel][col] LDS image with `rowb` bytes per row
    // The reads are written as inline assembly (round 5).  Through the builtin the compiler sees LDS reads that may alias the LDS-DMA pieces
    // issued a few instructions earlier and puts `s_waitcnt vmcnt(0)` in front of them: every K step then waited for the pieces of the NEXT
    // step before it read its own fragments - the whole load latency exposed in every wave, every step (found in the ISA while building
    // the patch walk; it had been there since round 1).  The ring's protocol (counted wait + barrier at the top of a step, pieces only into
    // the stage nobody reads) is what orders the two; the compiler cannot know and must not add to it.
    struct Addr { uint32_t lo, hi; };
    template <int SW>
    static __device__ __forceinline__ Addr addr(const char* img, int rowb, int c0, int lane) {
        const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
        const int r_lo = 8 * g + q, r_hi = r_lo + 4;
        const int u = (c0 >> 2) + p;  // logical 8-byte unit of this lane's 4 columns
        const int u_lo = u ^ (wg_swz<SW>(r_lo) << 2);
        const int u_hi = u ^ (wg_swz<SW>(r_hi) << 2);
        return Addr{(uint32_t)(uintptr_t)(lptr_t)(img + r_lo * rowb + u_lo * 8), (uint32_t)(uintptr_t)(lptr_t)(img + r_hi * rowb + u_hi * 8)};
    }
    static __device__ __forceinline__ void read2(bf16x8& f, const Addr& ad) {  // two transposed 8-byte reads fill one 8-value fragment
        typedef __attribute__((ext_vector_type(2))) uint32_t u2;
        u2 lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(ad.lo));
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(ad.hi));
        const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
        f = __builtin_bit_cast(bf16x8, v);
    }
    template <int BM, int TR, int TC, int BNW = WG_BN, bool BIAS = false>
    static __device__ __forceinline__ void step(const char* Ys, const char* Xs, int r0, int c0, int lane, f32x4 (&acc)[TR][TC], bool bias, f32x4 (&accb)[BIAS ? TR : 1]) {
        static_assert(WG_BK == 32, "one 32-deep sub-step");
        bf16x8 af[TR], bfr[TC];
        if (YMI_WGRAD_ABL & 4) {
#pragma unroll
            for (int t = 0; t < TR; ++t) { af[t] = bf16x8{}; asm volatile("" : "+v"(af[t])); }
#pragma unroll
            for (int t = 0; t < TC; ++t) { bfr[t] = bf16x8{}; asm volatile("" : "+v"(bfr[t])); }
        } else {
            // addresses first (plain arithmetic), then every read of the step back to back: X fragments, then the dY fragments in the
            // order the MFMA rows use them
            Addr ay[TR], ax[TC];
#pragma unroll
            for (int t = 0; t < TR; ++t) ay[t] = addr<YSwz<BM>::SW>(Ys, BM * 2, r0 + t * 16, lane);  // by the dY image's row width
            // (512-byte X rows of the 256-column tile: the XOR only touches the low five bits of the 8-byte unit index, i.e. it
            // permutes units inside each 256-byte bank row exactly as for 256-byte rows)
#pragma unroll
            for (int t = 0; t < TC; ++t) ax[t] = addr<1>(Xs, BNW * 2, c0 + t * 16, lane);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < TC; ++t) read2(bfr[t], ax[t]);
#pragma unroll
            for (int t = 0; t < TR; ++t) read2(af[t], ay[t]);
        }
#pragma unroll
        for (int a = 0; a < TR; ++a) {
            // row a needs the X fragments and dY fragment a: the 2 * (TR - 1 - a) younger reads may still be in flight
            __builtin_amdgcn_sched_barrier(0);
            if (!(YMI_WGRAD_ABL & 4)) {
                if (a == 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (TR - 1)) : "memory");
                else if (a == 1) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(TR > 2 ? 2 * (TR - 2) : 0) : "memory");
                else if (a == 2) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(TR > 3 ? 2 * (TR - 3) : 0) : "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            asm volatile("" : "+v"(af[a]));
            if (a == 0) {
#pragma unroll
                for (int b = 0; b < TC; ++b) asm volatile("" : "+v"(bfr[b]));
            }
#pragma unroll
            for (int b = 0; b < TC; ++b) {
                if (YMI_WGRAD_ABL & 2) { asm volatile("" :: "v"(bfr[b]), "v"(af[a])); continue; }
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[b], af[a], acc[a][b], 0, 0, 0);  // operands swapped: see the epilogue
            }
            if (BIAS && bias) {  // (wave-uniform) column sums of dY: a row of ones against the dY fragment already in registers - one more MFMA per row
                typedef __attribute__((ext_vector_type(8))) short s16x8;
                const s16x8 one8 = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};  // 1.0 in bfloat16
                const bf16x8 ones = __builtin_bit_cast(bf16x8, one8);
                if constexpr (BIAS) accb[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[a], accb[a], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
};
template <> struct WFrag<float> {
    template <int BM, int TR, int TC, int BNW = WG_BN, bool BIAS = false>
    static __device__ __forceinline__ void step(const char* Ys, const char* Xs, int r0, int c0, int lane, f32x4 (&acc)[TR][TC], bool bias, f32x4 (&accb)[BIAS ? TR : 1]) {
        static_assert(BNW == WG_BN, "f32 parity mode uses the 128-column tile");
        const int kq = lane >> 4, i = lane & 15;
#pragma unroll
        for (int ks = 0; ks < WG_BK / 4; ++ks) {
            float af[TR], bfr[TC];
#pragma unroll
            for (int t = 0; t < TR; ++t) af[t] = *reinterpret_cast<const float*>(Ys + ((4 * ks + kq) * BM + r0 + t * 16 + i) * 4);
#pragma unroll
            for (int t = 0; t < TC; ++t) bfr[t] = *reinterpret_cast<const float*>(Xs + ((4 * ks + kq) * WG_BN + c0 + t * 16 + i) * 4);
#pragma unroll
            for (int a = 0; a < TR; ++a)
#pragma unroll
                for (int b = 0; b < TC; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(bfr[b], af[a], acc[a][b], 0, 0, 0);
            if (BIAS && bias) {
#pragma unroll
                for (int a = 0; a < (BIAS ? TR : 1); ++a) accb[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, af[a], accb[a], 0, 0, 0);
            }
        }
    }
};

// BM = output channels per workgroup tile: 64, or 128 for layers with >= 128 output channels (16 instead of 8 MFMAs per
// wave and K step against the same address arithmetic: the K loop is instruction-issue-bound, not MFMA-bound), or 32 for the
// layers with <= 32 output channels (the 320x320 / 160x160 maps: millions of pixels against a 32 x 72..288 weight matrix; the
// 64-row tile spent half its MFMAs on padding rows)
// (a 128x256 tile - 4 or 8 waves - measured 1.17-1.66x slower, profiles/r02_conv_bench_wgrad256.txt; removed in round 3)
// BIAS: the instantiation that also forms the bias gradient's partials (its own code object: the 4 * TR accumulator registers and the branch
// cost the bias-free launches 3-4 % when they were a run-time option of one kernel)
// the final pass of a BatchNorm backward as a rider (common.h: YmiBnRider): workgroup w sums the partial rows of channels [32 w, 32 w + 32) - 8 row
// slices x 32 channels per workgroup, eight loads in flight per lane, combined in double in a fixed order - and writes dbeta / dgamma and the apply
// pass's coefficients (reduce_bwd.hip: chan_reduce_final_kernel is the same pass as its own launch)
__device__ __forceinline__ void wgrad_rider_final(const YmiBnRider& r, int w, char* smem) {
    // BIT FOR BIT chan_reduce_final_kernel's arithmetic (reduce_bwd.hip: 32 row slices per channel, four float chains each, combined in double in
    // slice order), so a step gives the same gradients with the final passes riding or not: a thread plays four of the 32 slices
    double* red = reinterpret_cast<double*>(smem);  // [2][32][33]
    const int cl = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int c = w * 32 + cl;
    if (w * 32 >= r.C) return;  // (padding workgroups of the rider block: uniform)
#pragma unroll
    for (int es = 0; es < 4; ++es) {
        const int slice = grp * 4 + es;
        double s0 = 0.0, s1 = 0.0;
        if (c < r.C) {
            float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f, c0 = 0.f, c1 = 0.f, d0 = 0.f, d1 = 0.f;
            int b = slice;
            for (; b + 96 < r.blocks; b += 128) {
                a0 += r.part[((int64_t)b * 2 + 0) * r.C + c];
                a1 += r.part[((int64_t)b * 2 + 1) * r.C + c];
                b0 += r.part[((int64_t)(b + 32) * 2 + 0) * r.C + c];
                b1 += r.part[((int64_t)(b + 32) * 2 + 1) * r.C + c];
                c0 += r.part[((int64_t)(b + 64) * 2 + 0) * r.C + c];
                c1 += r.part[((int64_t)(b + 64) * 2 + 1) * r.C + c];
                d0 += r.part[((int64_t)(b + 96) * 2 + 0) * r.C + c];
                d1 += r.part[((int64_t)(b + 96) * 2 + 1) * r.C + c];
            }
            for (; b < r.blocks; b += 32) {
                a0 += r.part[((int64_t)b * 2 + 0) * r.C + c];
                a1 += r.part[((int64_t)b * 2 + 1) * r.C + c];
            }
            s0 = ((double)a0 + (double)b0) + ((double)c0 + (double)d0);
            s1 = ((double)a1 + (double)b1) + ((double)c1 + (double)d1);
        }
        red[(0 * 32 + slice) * 33 + cl] = s0;
        red[(1 * 32 + slice) * 33 + cl] = s1;
    }
    __syncthreads();
    if (grp == 0 && c < r.C) {
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            a0 += red[(0 * 32 + q) * 33 + cl];
            a1 += red[(1 * 32 + q) * 33 + cl];
        }
        if (r.out0) r.out0[c] = (float)a0;
        if (r.out1) r.out1[c] = (float)a1;
        if (r.coef) {
            const bool second = r.split > 0 && c >= r.split;
            const float* gp = second ? r.gamma2 : r.gamma;
            const float* bp = second ? r.beta2 : r.beta;
            const int pc = second ? c - r.split : c;
            const float ga = gp ? gp[pc] : 1.0f, be = bp ? bp[pc] : 0.0f;
            const float p0 = r.inv[c], p1 = -r.mean[c] * p0;
            const float k1 = (float)a0 * r.inv_count, k2 = (float)a1 * r.inv_count;
            r.coef[0 * r.C + c] = p0 * ga;
            r.coef[1 * r.C + c] = p1 * ga + be;
            r.coef[2 * r.C + c] = ga * p0;
            r.coef[3 * r.C + c] = ga * p0 * p0 * k2;
            r.coef[4 * r.C + c] = ga * p0 * (k1 + p1 * k2);
        }
    }
}

// PATCH (round 5): the K axis runs over the output pixels in PATCH order instead of raster order - step k is a ph x pw patch (ph * pw = 32, pw a power
// of two dividing Wo, ph dividing Ho) whose origin (n, ho0, wo0) is SCALAR state.  A lane's row of the step is a fixed (pr, pc) inside the patch, so its
// source offset is a per-thread constant plus a scalar: the pieces become `buffer_load_dwordx4 ... lds` with the constant in the vector offset and the
// patch origin in the scalar offset, padding taps are lanes whose vector offset is pushed out of range (the buffer unit writes zeros to LDS for them:
// tools/probes/bar/buffer_lds_probe.hip), and the per-lane (ho, wo) walk with its wraps - as many issue slots as the MFMAs (tools/probes/
// r5_wgrad_ablate.sh) - is five vector instructions per X piece and none per dY piece.  Any fixed order of the pixel sum is as good as raster order;
// maps that do not tile into such patches (20 x 20) keep the raster walk.
template <typename T, int NS, int BM, bool BIAS = false, bool PATCH = false>
__global__ __launch_bounds__(256, (BM == 128 ? 3 : YMI_WGRAD_WAVES)) void wgrad_kernel(WgradArgs a, YmiBnRider rider) {
    constexpr int BNW = WG_BN, NT = 256;
    constexpr int CH = ElemTraits<T>::CH;
    constexpr int ES = (int)sizeof(T);
    constexpr int YCW = BM * ES / 16, XCW = BNW * ES / 16;            // 16-byte chunks per tile row
    constexpr int YT = WG_BK * YCW < NT ? WG_BK * YCW : NT;           // threads that cover one dY load of the workgroup (32-channel tile: 128 -
                                                                      // the other two waves fetch the SAME chunks to the SAME LDS bytes, so every
                                                                      // wave issues the same number of loads and the counted waits stay uniform)
    constexpr int NY = (WG_BK * YCW + NT - 1) / NT, NX = WG_BK * XCW / NT;  // chunks per thread per K step
    constexpr int WCOLS = BNW / 64, WROWS = (NT / 64) / WCOLS;        // wave grid: every wave owns a (BM / WROWS) x 64 tile
    constexpr int YBYTES = WG_BK * BM * ES, XBYTES = WG_BK * BNW * ES;
    constexpr int STAGE = YBYTES + XBYTES;
    static_assert(NY >= 1 && NX >= 1, "tile too small");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WCOLS, wc = wave % WCOLS;
#ifdef YMI_STAMPS
    const bool wstamp_on = g_wstamp_buf && blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0;
    const unsigned long long wstamp_mt0 = wstamp_on ? __builtin_amdgcn_s_memtime() : 0ull, wstamp_rt0 = wstamp_on ? __builtin_amdgcn_s_memrealtime() : 0ull;
#endif
    // XCD-aware order (xcd_map): workgroup ids are dealt round-robin to the 8 XCDs; all tiles of one pixel split read the
    // same dY / X rows (the nine taps are the same pixels, shifted), so a split's tiles are given to ONE XCD, back to
    // back, and its rows are fetched into that L2 once.  An XCD gets CONSECUTIVE splits, z = xcd * ceil(splits / 8) + i: its
    // eighth of the pixel order (common.h, XCD ownership of the pixel axis) - the BatchNorm apply pass wrote those dY rows from
    // this XCD a launch ago (round 3 dealt z = 8 i + xcd: every split's rows came from the other seven L2s).
    int bx, by, bz;
    if (rider.nwg && (int)blockIdx.x < rider.nwg) {  // (only in the 1-D XCD-mapped grid; workgroup-uniform)
        wgrad_rider_final(rider, (int)blockIdx.x, smem);
        return;
    }
    if (a.xcd_map) {
        const int id = (int)blockIdx.x - rider.nwg, xcd = id & 7, seq = id >> 3, nxy = a.nx * a.ny;
        const int zi = seq / nxy, t = seq - zi * nxy;
        const int spx = (a.splits + 7) >> 3;
        bz = xcd * spx + zi;
        if (zi >= spx || bz >= a.splits) return;  // padding ids leave before any barrier
        by = t / a.nx;
        bx = t - by * a.nx;
    } else {
        bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
    }
    const int j0 = bx * BNW, co0 = by * BM;
    const int m_begin = bz * a.pix_per_split;
    const int m_end = min(a.Mpix, m_begin + a.pix_per_split);
    const T* __restrict__ xg = reinterpret_cast<const T*>(a.x);
    const T* __restrict__ yg = reinterpret_cast<const T*>(a.dy);
    const T* zero = reinterpret_cast<const T*>(a.zero);

    // dY loader: thread -> (row, chunk) ; chunk column fixed per thread.  bf16: LDS position (row, chunk') holds
    // source chunk chunk' ^ (swizzle(row) << 1); the row bits the swizzle uses are the same for all of a thread's rows.
    constexpr bool SWZ = std::is_same<T, bf16_t>::value;
    const int ytid = tid % YT;
    const int ycc = SWZ ? ((ytid % YCW) ^ (wg_swz<YSwz<BM>::SW>(ytid / YCW) << 1)) : (ytid % YCW);
    const bool y_cok = co0 + ycc * CH < a.CoutP;
    // X loader: column chunk fixed per thread -> fixed tap / input-channel offset
    static_assert(NT / XCW >= 16 || !std::is_same<T, bf16_t>::value, "bf16: a block-wide load covers >= 16 rows, so a thread's rows share the row bits the swizzle uses");
    const int xrow0 = tid / XCW;
    const int xcc = SWZ ? ((tid % XCW) ^ (wg_swz_x(xrow0) << 1)) : (tid % XCW);
    const int j = j0 + xcc * CH;
    const bool x_cok = j < a.NG;
    const int tap = x_cok ? j / a.Cin : 0;
    const int ci = x_cok ? j - tap * a.Cin : 0;
    const int dh = tap / a.KW - a.pad, dw = tap % a.KW - a.pad;
    const int ldx32 = (int)a.ldx, ldy32 = (int)a.ldy;
    // Address state of this thread's rows, advanced by WG_BK pixels per K step with adds and compares only (the pixel ->
    // (n, ho, wo) decomposition by two multiply-highs and five multiplies per piece and step cost ~160 cycles per piece: the
    // issue phase was half of a K step, stamps in profiles/r02_igemm_phase_stamps.txt section 6).  issue() is called for
    // consecutive steps m_begin, m_begin + WG_BK, ...: it uses the state and then moves it one step on.
    //   X row: ws = wo * stride, hs = ho * stride (input coordinates of the tap-(0,0) pixel), xo = element offset of that pixel
    //          + this thread's tap shift and channel; a step adds (q, r) = divmod(WG_BK, Wo) rows / columns, then wraps.
    //   Y row: element offset m * ldy + channel.
    const int s_ = a.stride;
    const int q_ = WG_BK / a.Wo, r_ = WG_BK - q_ * a.Wo;                   // scalar
    const uint32_t d_step = (uint32_t)((q_ * s_ * a.W + r_ * s_) * ldx32);   // offset change of (ho += q, wo += r)
    const uint32_t d_wwrap = (uint32_t)((s_ * a.W - a.Wo * s_) * ldx32);     // ... of (wo -= Wo, ho += 1)
    const uint32_t d_hwrap = (uint32_t)((a.H * a.W - a.Ho * s_ * a.W) * ldx32);  // ... of (ho -= Ho, n += 1)
    const int ws_lim = a.Wo * s_, hs_lim = a.Ho * s_;
    int x_ws[NX], x_hs[NX], x_row[NX];
    uint32_t x_off[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int row = tid / XCW + i * (NT / XCW);
        const int m = m_begin + row;
        const int t = wg_fast_div(m, a.wo_mul, a.wo_shr);
        const int wo = m - t * a.Wo;
        const int n = wg_fast_div(t, a.ho_mul, a.ho_shr);
        const int ho = t - n * a.Ho;
        x_row[i] = row;
        x_ws[i] = wo * s_;
        x_hs[i] = ho * s_;
        x_off[i] = (uint32_t)(((n * a.H + ho * s_ + dh) * a.W + wo * s_ + dw) * ldx32 + ci);
    }
    int y_row[NY];
    uint32_t y_off[NY];
#pragma unroll
    for (int i = 0; i < NY; ++i) {
        y_row[i] = ytid / YCW + i * (NT / YCW);
        y_off[i] = (uint32_t)((m_begin + y_row[i]) * ldy32 + co0 + ycc * CH);
    }
    const uint32_t y_step = (uint32_t)(WG_BK * ldy32);
    // ---- PATCH: per-thread constants and the scalar patch walk
    constexpr uint32_t OOR = 0x80000000u;  // a vector offset beyond any buffer of < 2^31 bytes (vector + scalar offset stays below 2^32)
    const int pw = 1 << a.pw_shift, ph = WG_BK >> a.pw_shift;
    const uint32_t xbias = PATCH ? (uint32_t)((a.W + 1) * ldx32 * ES) : 0u;  // keeps the (dh, dw) = (-1, -1) offsets non-negative
    const char* xbase = reinterpret_cast<const char*>(xg) - xbias;
    uint32_t pvx[NX], pvy[NY];
    int pch[NX], pcw[NX];
    int p_wo = 0, p_ho = 0;            // scalar: the next patch's origin
    uint32_t p_sx = 0, p_sy = 0;       // scalar byte offsets of that origin in X (tap (0, 0), channel 0) and dY
    if constexpr (PATCH) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int row = tid / XCW + i * (NT / XCW);
            const int pr = row >> a.pw_shift, pc = row & (pw - 1);
            pch[i] = pr * s_ + dh;
            pcw[i] = pc * s_ + dw;
            pvx[i] = x_cok ? (uint32_t)(((pch[i] * a.W + pcw[i]) * ldx32 + ci) * ES) + xbias : OOR;
        }
#pragma unroll
        for (int i = 0; i < NY; ++i) {
            const int row = ytid / YCW + i * (NT / YCW);
            const int pr = row >> a.pw_shift, pc = row & (pw - 1);
            pvy[i] = y_cok ? (uint32_t)(((pr * a.Wo + pc) * ldy32 + co0 + ycc * CH) * ES) : OOR;
        }
        const int pidx = __builtin_amdgcn_readfirstlane(m_begin / WG_BK);  // first patch of this split
        const int npw = a.Wo >> a.pw_shift, nph = a.Ho / ph;
        const int t = pidx / npw, wp = pidx - t * npw;
        const int n = t / nph, hp = t - n * nph;
        p_wo = wp * pw;
        p_ho = hp * ph;
        p_sx = (uint32_t)(((n * a.H + p_ho * s_) * a.W + p_wo * s_) * ldx32 * ES);
        p_sy = (uint32_t)(((n * a.Ho + p_ho) * a.Wo + p_wo) * ldy32 * ES);
    }
    auto issue = [&](int s, int mk) {
        char* Ys = smem + s * STAGE;
        char* Xs = Ys + YBYTES;
        const bool live = !(YMI_WGRAD_ABL & 1) || mk < m_begin + (NS - 1) * WG_BK;
        if constexpr (PATCH) {
            const uint32_t sx = __builtin_amdgcn_readfirstlane(p_sx), sy = __builtin_amdgcn_readfirstlane(p_sy);  // (scalar by construction; said so)
#pragma unroll
            for (int i = 0; i < NY; ++i)
                if (live) wg_buffer_to_lds(yg, a.y_bytes, (lptr_t)(Ys + (i * NT + (wave * 64) % YT) * 16), pvy[i], sy);
            const int hs0 = __builtin_amdgcn_readfirstlane(p_ho * s_), ws0 = __builtin_amdgcn_readfirstlane(p_wo * s_);
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const bool ok = (unsigned)(hs0 + pch[i]) < (unsigned)a.H && (unsigned)(ws0 + pcw[i]) < (unsigned)a.W;
                if (live) wg_buffer_to_lds(xbase, a.x_bytes + xbias, (lptr_t)(Xs + (i * NT + wave * 64) * 16), ok ? pvx[i] : OOR, sx);
            }
            // one patch on (scalar)
            p_wo += pw;
            p_sx += (uint32_t)(pw * s_ * ldx32 * ES);
            p_sy += (uint32_t)(pw * ldy32 * ES);
            if (p_wo == a.Wo) {
                p_wo = 0;
                p_ho += ph;
                p_sx += (uint32_t)((ph * s_ * a.W - a.Wo * s_) * ldx32 * ES);
                p_sy += (uint32_t)((ph * a.Wo - a.Wo) * ldy32 * ES);
                if (p_ho == a.Ho) {
                    p_ho = 0;
                    p_sx += (uint32_t)((a.H * a.W - a.Ho * s_ * a.W) * ldx32 * ES);
                }
            }
            return;
        }
        const int left = m_end - mk;  // scalar: rows below it are inside this split
#pragma unroll
        for (int i = 0; i < NY; ++i) {
            const bool ok = y_cok && y_row[i] < left;
            const T* src = ok ? yg + y_off[i] : zero;
            if (live) __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Ys + (i * NT + (wave * 64) % YT) * 16), 16, 0, 0);
            else asm volatile("" :: "v"(src));
            y_off[i] += y_step;
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const bool ok = x_cok && x_row[i] < left && (unsigned)(x_hs[i] + dh) < (unsigned)a.H && (unsigned)(x_ws[i] + dw) < (unsigned)a.W;
            const T* src = ok ? xg + x_off[i] : zero;
            if (live) __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Xs + (i * NT + wave * 64) * 16), 16, 0, 0);
            else asm volatile("" :: "v"(src));
            if (YMI_WGRAD_ABL & 8) continue;
            // one K step on
            x_ws[i] += r_ * s_;
            x_hs[i] += q_ * s_;
            x_off[i] += d_step;
            if (x_ws[i] >= ws_lim) {
                x_ws[i] -= ws_lim;
                x_hs[i] += s_;
                x_off[i] += d_wwrap;
            }
            while (x_hs[i] >= hs_lim) {  // at most once unless the map has fewer rows than a K step covers
                x_hs[i] -= hs_lim;
                x_off[i] += d_hwrap;
            }
        }
    };

    constexpr int TR = BM / WROWS / 16, TC = 4;  // per wave: BM / WROWS rows x 64 cols
    f32x4 acc[TR][TC];
#pragma unroll
    for (int r = 0; r < TR; ++r)
#pragma unroll
        for (int c = 0; c < TC; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    // bias gradient (column sums of dY): formed by the waves of column block 0 that own the first 64 columns - every (row block, split) once
    const bool do_bias = BIAS && bx == 0 && wc == 0;  // (wave-uniform)
    f32x4 accb[BIAS ? TR : 1];
#pragma unroll
    for (int r = 0; r < (BIAS ? TR : 1); ++r) accb[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    WG_MARK(0);  // prologue done
    // NS-stage LDS ring, one raw barrier per K step, loads of NS-2 younger steps stay in flight (see igemm.hip)
    const int nk = (m_end - m_begin + WG_BK - 1) / WG_BK;
    constexpr int LPT = NY + NX;
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nk) issue(s, m_begin + s * WG_BK);
#ifdef YMI_STAMPS
    int wstamp_i = 0;
    // stamps go to a spare 2 KB of LDS behind the ring (a global store per stamp would sit in vmcnt and distort the counted waits)
    unsigned long long* wstamp_lds = reinterpret_cast<unsigned long long*>(smem + NS * STAGE) + wave * 64;
#define WG_STEP_STAMP() do { if (wstamp_on && kt >= 2 && kt < 10 && wstamp_i < 64) wstamp_lds[wstamp_i++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WG_STEP_STAMP() do { } while (0)
#endif
    for (int kt = 0; kt < nk; ++kt) {
        WG_STEP_STAMP();  // 0: step start
        if (kt + NS - 2 < nk) wait_vmcnt_barrier<LPT * (NS - 2)>();
        else wait_vmcnt_barrier<0>();
        WG_STEP_STAMP();  // 1: past wait + barrier
        if (kt + NS - 1 < nk) issue((kt + NS - 1) % NS, m_begin + (kt + NS - 1) * WG_BK);
        WG_STEP_STAMP();  // 2: pieces issued
        const char* Ys = smem + (kt % NS) * STAGE;
        if constexpr (BIAS) WFrag<T>::template step<BM, TR, TC, BNW, true>(Ys, Ys + YBYTES, wr * (BM / WROWS), wc * 64, lane, acc, do_bias, accb);
        else WFrag<T>::template step<BM, TR, TC, BNW, false>(Ys, Ys + YBYTES, wr * (BM / WROWS), wc * 64, lane, acc, false, accb);
        WG_STEP_STAMP();  // 3: fragments read, MFMAs issued
    }

#ifdef YMI_STAMPS
    if (wstamp_on) for (int q = 0; q < 64; ++q) g_wstamp_buf[wave * 64 + q] = q < wstamp_i ? wstamp_lds[q] : 0ull;
#endif
    WG_MARK(1);  // K loop done
#ifdef YMI_STAMPS
    if (wstamp_on) g_wstamp_buf[8 * 64 + 16 + wave * 8 + 6] = nk;
#endif
    const int64_t slab_off = (int64_t)bz * a.CoutP * a.NG;
    // The MFMA operands are swapped (A = the X fragment, B = the dY fragment), so a lane's four accumulator values are four
    // CONSECUTIVE (tap, ci) columns of one output channel: one 16-byte store instead of four 4-byte stores to four rows
    // (stores are issue-bound on this chip: 8 / 16 instructions per lane instead of 32 / 64).  NG is a multiple of 4.
    const int l15 = lane & 15, l4 = lane >> 4;
    if (BIAS && do_bias && l4 == 0) {  // every row of the ones product holds the same sums: lanes 0-15 write their output channel's
#pragma unroll
        for (int r = 0; r < (BIAS ? TR : 1); ++r) {
            const int co = co0 + wr * (BM / WROWS) + r * 16 + l15;
            if (co < a.CoutP) a.bias_slab[(int64_t)bz * a.CoutP + co] = accb[r][0];
        }
    }
    if (std::is_same<T, bf16_t>::value && a.slab_bf16 && !(YMI_WGRAD_ABL & 16) && !YMI_WGRAD_DIRECT_SLAB) {
        // bfloat16 slabs leave through LDS (round 5): written straight from the accumulators, a store instruction covers 16 rows x 32 bytes -
        // sixteen partial lines - and those stores cost 8 % of the family's time (0.7 ms per step if nothing hid them: tools/probes/
        // r5_wgrad_ablate.sh).  Each wave drops 32 rows x 64 columns of its tile into a private padded image and stores it back as whole
        // 128-byte row segments, eight rows per instruction.  Same values, same rounding.
        __syncthreads();  // the other waves may still read the last K step's stage
        constexpr int SROW = 144;                 // padded row: 16 rows hit 16 distinct bank groups
        constexpr int RP = TR < 2 ? TR : 2;       // 16-row tiles per pass
        char* st = smem + wave * (RP * 16 * SROW);
        bf16_t* slab = reinterpret_cast<bf16_t*>(a.slab) + slab_off;
        const int srow = lane >> 3, sch = lane & 7;
        const int col = j0 + wc * 64 + sch * 8;
#pragma unroll
        for (int r0 = 0; r0 < TR; r0 += RP) {
#pragma unroll
            for (int rr = 0; rr < RP; ++rr)
#pragma unroll
                for (int c = 0; c < TC; ++c) {
                    const bf16x4 v = {(bf16_t)acc[r0 + rr][c][0], (bf16_t)acc[r0 + rr][c][1], (bf16_t)acc[r0 + rr][c][2], (bf16_t)acc[r0 + rr][c][3]};
                    *reinterpret_cast<bf16x4*>(st + (rr * 16 + l15) * SROW + (c * 16 + 4 * l4) * 2) = v;
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int k = 0; k < RP * 2; ++k) {
                const int row = k * 8 + srow;
                const uint4 u = *reinterpret_cast<const uint4*>(st + row * SROW + sch * 16);
                const int co = co0 + wr * (BM / WROWS) + r0 * 16 + row;
                if (co < a.CoutP && col < a.NG) *reinterpret_cast<uint4*>(slab + (int64_t)co * a.NG + col) = u;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    } else if (std::is_same<T, bf16_t>::value && !(YMI_WGRAD_ABL & 16) && !YMI_WGRAD_DIRECT_SLAB) {
        // float32 slabs of the bf16 path (fewer than 16 splits): the same, 16 rows x 64 columns per pass, 256-byte row segments, four rows per instruction
        __syncthreads();
        constexpr int SROW = 272;
        char* st = smem + wave * (16 * SROW);
        float* slab = reinterpret_cast<float*>(a.slab) + slab_off;
        const int srow = lane >> 4, sch = lane & 15;
        const int col = j0 + wc * 64 + sch * 4;
#pragma unroll
        for (int r = 0; r < TR; ++r) {
#pragma unroll
            for (int c = 0; c < TC; ++c) *reinterpret_cast<f32x4*>(st + l15 * SROW + (c * 16 + 4 * l4) * 4) = acc[r][c];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = k * 4 + srow;
                const f32x4 u = *reinterpret_cast<const f32x4*>(st + row * SROW + sch * 16);
                const int co = co0 + wr * (BM / WROWS) + r * 16 + row;
                if (co < a.CoutP && col < a.NG) *reinterpret_cast<f32x4*>(slab + (int64_t)co * a.NG + col) = u;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    } else
#pragma unroll
    for (int r = 0; r < TR; ++r) {
        const int co = co0 + wr * (BM / WROWS) + r * 16 + l15;
#pragma unroll
        for (int c = 0; c < TC; ++c) {
            const int col = j0 + wc * 64 + c * 16 + 4 * l4;
            if (YMI_WGRAD_ABL & 16) { asm volatile("" :: "v"(acc[r][c])); continue; }
            if (co < a.CoutP && col < a.NG) {
                const int64_t e = slab_off + (int64_t)co * a.NG + col;
                if (std::is_same<T, bf16_t>::value && a.slab_bf16) {
                    const bf16x4 v = {(bf16_t)acc[r][c][0], (bf16_t)acc[r][c][1], (bf16_t)acc[r][c][2], (bf16_t)acc[r][c][3]};
                    *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(a.slab) + e) = v;
                } else {
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.slab) + e) = acc[r][c];
                }
            }
        }
    }
    WG_MARK(5);  // stores issued
#ifdef YMI_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WG_MARK(2);  // stores retired
    if (wstamp_on) {
        g_wstamp_buf[8 * 64 + wave * 2 + 0] = __builtin_amdgcn_s_memtime() - wstamp_mt0;
        g_wstamp_buf[8 * 64 + wave * 2 + 1] = __builtin_amdgcn_s_memrealtime() - wstamp_rt0;
    }
#endif
}

// a slab element as float (slabs are float32, or bfloat16 in the bf16 path)
template <bool BF> __device__ __forceinline__ float slab_at(const void* slab, int64_t i) {
    if constexpr (BF) return (float)reinterpret_cast<const bf16_t*>(slab)[i];
    else return reinterpret_cast<const float*>(slab)[i];
}
template <bool BF> __device__ __forceinline__ f32x4 slab4_at(const void* slab, int64_t i) {  // i a multiple of 4
    if constexpr (BF) {
        const bf16x4 v = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(slab) + i);
        return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    } else {
        return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(slab) + i);
    }
}

// Per-layer slab reduction (ymi_conv2d_bwd_weight, the non-deferred entry point), deterministic: one thread per output element,
// coalesced along (tap, ci); four independent partial sums keep the split loads in flight, combined in a fixed order.
// dw[co][ci][kh][kw] = sum_s slab[s][co][tap*Cin + ci]  (scatter into OIHW).
template <bool BF>
__global__ void wgrad_reduce_kernel(const void* __restrict__ slab, int splits, int CoutP, int NG, int Cin, int cout_real, int cin_real, int ntaps,
                                    float* __restrict__ dw) {
    // 32-bit index arithmetic (a weight tensor has < 2^31 elements)
    const uint32_t total = (uint32_t)cout_real * (uint32_t)ntaps * (uint32_t)cin_real;
    const int64_t sstride = (int64_t)CoutP * NG;
    for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const uint32_t t = idx / (uint32_t)cin_real;
        const uint32_t ci = idx - t * (uint32_t)cin_real;
        const uint32_t co = t / (uint32_t)ntaps;
        const uint32_t tap = t - co * (uint32_t)ntaps;
        const int64_t e = (int64_t)co * NG + tap * Cin + ci;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int k = 0;
        for (; k + 3 < splits; k += 4) {
            s0 += slab_at<BF>(slab, e + (int64_t)k * sstride);
            s1 += slab_at<BF>(slab, e + (int64_t)(k + 1) * sstride);
            s2 += slab_at<BF>(slab, e + (int64_t)(k + 2) * sstride);
            s3 += slab_at<BF>(slab, e + (int64_t)(k + 3) * sstride);
        }
        for (; k < splits; ++k) s0 += slab_at<BF>(slab, e + (int64_t)k * sstride);
        dw[((int64_t)co * cin_real + ci) * ntaps + tap] = (s0 + s1) + (s2 + s3);
    }
}

// All pending slab reductions of a backward pass in ONE launch (ymi_wgrad_reduce_batch): a 256-thread workgroup finds its
// tensor by binary search over the table's first_block column, then sums `lanes` interleaved split chains per output
// element in a fixed order (deterministic) and scatters into OIHW.  By the end of the backward pass the slabs have left the
// caches: this kernel streams them from HBM, 4 elements per lane and load, four split chains in flight per lane.
// dbias[co] = sum over the splits of the bias partials [split][coutp], four chains in flight, fixed order
__device__ __forceinline__ void bias_slab_sum(const float* __restrict__ bs, int splits, int coutp, float* __restrict__ db, int tid, int nthr) {
    for (int co = tid; co < coutp; co += nthr) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int k = 0;
        for (; k + 3 < splits; k += 4) {
            s0 += bs[(int64_t)k * coutp + co];
            s1 += bs[(int64_t)(k + 1) * coutp + co];
            s2 += bs[(int64_t)(k + 2) * coutp + co];
            s3 += bs[(int64_t)(k + 3) * coutp + co];
        }
        for (; k < splits; ++k) s0 += bs[(int64_t)k * coutp + co];
        db[co] = (s0 + s1) + (s2 + s3);
    }
}
__global__ void wgrad_bias_reduce_kernel(const float* __restrict__ bs, int splits, int coutp, float* __restrict__ db) {
    bias_slab_sum(bs, splits, coutp, db, (int)(blockIdx.x * blockDim.x + threadIdx.x), (int)(gridDim.x * blockDim.x));
}
template <bool BF> __device__ __forceinline__ void reduce_batch_body(const ymi_wgrad_pending& e, f32x4* red) {
    // the record's first workgroup also sums the bias partials (a few KB)
    if (e.bias_slab && (int)blockIdx.x == e.first_block) bias_slab_sum(e.bias_slab, e.splits, (int)(e.elems / e.ng), e.dbias, (int)threadIdx.x, 256);
    // a lane owns EPT consecutive elements: 8 (one 16-byte load per split) of a bfloat16 slab, 4 of a float32 slab
    constexpr int EPT = BF ? 8 : 4, Q = EPT / 4;
    const int SL = e.lanes, OUTS = 256 / SL;              // OUTS groups of EPT consecutive elements per workgroup
    const int ol = threadIdx.x % OUTS, lane = threadIdx.x / OUTS;
    const int64_t el = ((int64_t)((int)blockIdx.x - e.first_block) * OUTS + ol) * EPT;  // first element of this lane's group
    f32x4 s0[Q], s1[Q], s2[Q], s3[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) s0[q] = s1[q] = s2[q] = s3[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto add_split = [&](f32x4 (&acc)[Q], int k) {
        const int64_t off = el + (int64_t)k * e.elems;
        if constexpr (BF) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(e.slab) + off);
            acc[0] += f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
            acc[1] += f32x4{(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
        } else {
            acc[0] += *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(e.slab) + off);
        }
    };
    if (el < e.elems) {  // elems is a multiple of 8
        int k = lane;
        for (; k + 3 * SL < e.splits; k += 4 * SL) {  // four split chains in flight per lane
            add_split(s0, k);
            add_split(s1, k + SL);
            add_split(s2, k + 2 * SL);
            add_split(s3, k + 3 * SL);
        }
        for (; k < e.splits; k += SL) add_split(s0, k);
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) red[(q * SL + lane) * OUTS + ol] = (s0[q] + s1[q]) + (s2[q] + s3[q]);
    __syncthreads();
    if (lane == 0 && el < e.elems) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            for (int c = 0; c < SL; ++c) s += red[(q * SL + c) * OUTS + ol];
            const uint32_t eu = (uint32_t)el + 4u * q;  // < 2^31; 4 elements share (co, tap): ng and cin are multiples of 4
            const int co = (int)(eu / (uint32_t)e.ng), col = (int)(eu - (uint32_t)co * (uint32_t)e.ng);
            const int tap = (int)((uint32_t)col / (uint32_t)e.cin), ci = col - tap * e.cin;
            if (co < e.cout_real) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (ci + r < e.cin_real) e.dw[((int64_t)co * e.cin_real + ci + r) * e.ntaps + tap] = s[r];
            }
        }
    }
}
__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const ymi_wgrad_pending* __restrict__ tab, int n) {
    __shared__ f32x4 red[512];
    int lo = 0, hi = n - 1;  // last entry whose first_block <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].first_block <= (int)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const ymi_wgrad_pending e = tab[lo];
    if (e.slab_bf16) reduce_batch_body<true>(e, red);  // (workgroup-uniform)
    else reduce_batch_body<false>(e, red);
}

// magic numbers for wg_fast_div
static void wg_find_divisor(int d, uint32_t* mul, uint32_t* shr) {
    int lg = 0;
    while ((1 << lg) < d) ++lg;
    *mul = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << lg) - (uint64_t)d)) / (uint64_t)d + 1);
    *shr = (uint32_t)lg;
}

struct WgradPlan {
    int splits, pix_per_split;
    size_t slab_bytes;
};
// row-tile choice.  Measured (round 1): with the same workgroup target as the 64-row tile the 128-row tile is 5-45 % SLOWER (the
// pixel axis is split twice as often, doubling the slab traffic); with HALF the workgroups (same split count, 16 MFMAs
// per wave and K step) it is 7-12 % faster on every layer with >= 128 output channels.
static int wgrad_bm(int64_t coutp, bool bf16) {
    if (!bf16) return 64;
    if (coutp <= 32) return 32;
    return (coutp >= 128 && coutp % 128 == 0) || coutp >= 256 ? 128 : 64;
}
static WgradPlan wgrad_plan(int64_t mpix, int64_t coutp, int64_t ng, int bm, size_t slab_esize) {
    const int64_t tiles = ((ng + WG_BN - 1) / WG_BN) * ((coutp + bm - 1) / bm);
    // workgroups to aim for (tuning knobs; the in-situ optimum differs from the isolated one: see profiles/r03_wgrad_targets.txt).  Round 4,
    // swept inside the step again (profiles/r04_wgrad_targets_sweep.txt): ONE RESIDENT ROUND - 5 workgroups per CU for the 64-row tile
    // (its launch bounds), 3 for the 128-row tile - is 0.09 ms/step faster than 1024 / 640; a fourth 128-row workgroup per CU costs 0.3 ms
    const int target64 = ymi_opt(OPT_WGRAD_BLOCKS);
    const int target128 = ymi_opt(OPT_WGRAD_BLOCKS128);
    const int target = bm == 128 ? target128 : target64;
    int64_t s = (target + tiles - 1) / tiles;
    const int64_t smax = (mpix + 255) / 256;
    if (s > smax) s = smax;
    if (s < 1) s = 1;
    if (s >= 8) s = (s + 7) / 8 * 8 <= smax ? (s + 7) / 8 * 8 : s / 8 * 8;  // splits are dealt to the 8 XCDs (see the kernel): keep them balanced
    int64_t pps = (mpix + s - 1) / s;
    pps = (pps + WG_BK - 1) / WG_BK * WG_BK;
    s = (mpix + pps - 1) / pps;
    WgradPlan p;
    p.splits = (int)s;
    p.pix_per_split = (int)pps;
    p.slab_bytes = ((size_t)s * coutp * ng * slab_esize + 255) / 256 * 256;
    return p;
}

static int64_t pad_to(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

extern "C" size_t ymi_conv2d_bwd_weight_workspace(int64_t m_rows, int64_t cout, int64_t cin, int64_t kh, int64_t kw) {
    // upper bound over both dtypes' channel padding (8), plus room for the bias-gradient partials
    const int64_t coutp = pad_to(cout, 8), cinp = pad_to(cin, 8);
    WgradPlan p = wgrad_plan(m_rows, coutp, kh * kw * cinp, 64, 4);  // (float32 slabs: the larger of the two dtypes' needs)
    const WgradPlan p128 = wgrad_plan(m_rows, coutp, kh * kw * cinp, 128, 4);  // the 128-row tile splits the pixel axis further
    if (p128.slab_bytes > p.slab_bytes) p = p128;
    return p.slab_bytes + (size_t)(2048 * 2 + 1) * coutp * sizeof(float) + 256;
}

static int wgrad_impl(const ymi_tensor* x, const ymi_tensor* dy, int64_t cout_real, int64_t cin_real, int64_t kh, int64_t kw, int64_t stride,
                      float* dw_oihw, float* dbias, void* workspace, size_t workspace_bytes, ymi_wgrad_pending* pending, void* stream);

// ---- a launch, possibly held back (common.h: YmiBnRider) -------------------------------------------------------------------------------
struct WgradLaunch {
    WgradArgs a;
    dim3 grid;
    int bm;
    bool bf16;
    hipStream_t s;
    double flop, bytes;
};
static void wgrad_launch(const WgradLaunch& h, const YmiBnRider* rider) {
    YmiBnRider r{};
    dim3 grid = h.grid;
    if (rider && h.a.xcd_map) {  // rider workgroups at the front of the 1-D grid
        r = *rider;
        r.nwg = ((rider->C + 31) / 32 + 7) / 8 * 8;
        grid.x += (unsigned)r.nwg;
    }
    const WgradArgs& a = h.a;
    hipStream_t s = h.s;
    int prof = -1;
    if (ymi_prof_enabled()) prof = ymi_prof_start(s, 1, h.flop, h.bytes, h.bf16 ? 2500.0 : 157.3);
#define YMI_WG_LAUNCH(T, BMV, LDS)                                                                                   \
    do {                                                                                                            \
        if (a.pw_shift >= 0) {                                                                                      \
            if (a.bias_slab) hipLaunchKernelGGL((wgrad_kernel<T, YMI_WGRAD_NS, BMV, true, true>), grid, dim3(256), (LDS), s, a, r);  \
            else hipLaunchKernelGGL((wgrad_kernel<T, YMI_WGRAD_NS, BMV, false, true>), grid, dim3(256), (LDS), s, a, r);        \
        } else if (a.bias_slab) hipLaunchKernelGGL((wgrad_kernel<T, YMI_WGRAD_NS, BMV, true>), grid, dim3(256), (LDS), s, a, r);  \
        else hipLaunchKernelGGL((wgrad_kernel<T, YMI_WGRAD_NS, BMV, false>), grid, dim3(256), (LDS), s, a, r);                  \
    } while (0)
    if (h.bf16) {
        // two LDS stages (deeper rings measured equal: same bytes in flight per CU)
        if (h.bm == 128) YMI_WG_LAUNCH(bf16_t, 128, (size_t)YMI_WGRAD_NS * (WG_BK * (128 + WG_BN) * 2) + WG_STAMP_LDS);
        else if (h.bm == 32) YMI_WG_LAUNCH(bf16_t, 32, (size_t)YMI_WGRAD_NS * (WG_BK * (32 + WG_BN) * 2) + WG_STAMP_LDS);
        else YMI_WG_LAUNCH(bf16_t, 64, (size_t)YMI_WGRAD_NS * (WG_BK * (64 + WG_BN) * 2) + WG_STAMP_LDS);
    } else {
        const size_t lds = 2 * (size_t)(WG_BK * (WG_BM + WG_BN) * 4);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<float, 2, 64, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<float, 2, 64, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        // (parity mode keeps the raster walk: pw_shift is -1 for float32 operands)
        if (a.bias_slab) hipLaunchKernelGGL((wgrad_kernel<float, 2, 64, true>), grid, dim3(256), lds, s, a, r);
        else hipLaunchKernelGGL((wgrad_kernel<float, 2, 64, false>), grid, dim3(256), lds, s, a, r);
    }
#undef YMI_WG_LAUNCH
    ymi_prof_stop(s, prof);
}
static std::mutex g_hold_mu;
static bool g_hold_on = false, g_held_valid = false;
static WgradLaunch g_held;
static void wgrad_flush_held_locked() {
    if (g_held_valid) {
        g_held_valid = false;
        wgrad_launch(g_held, nullptr);
    }
}
bool ymi_wgrad_issue_held(const YmiBnRider* rider, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_hold_mu);
    if (!g_held_valid || g_held.s != stream || !g_held.a.xcd_map) return false;
    g_held_valid = false;
    wgrad_launch(g_held, rider);
    return true;
}
// mode 1: hold deferred weight-gradient launches for riders; 0: stop holding (a held launch is issued as it is); 2: forget a held launch (after a
// backward pass that raised: its operands are gone), the mode stays
extern "C" int ymi_wgrad_hold(int32_t mode) {
    std::lock_guard<std::mutex> lk(g_hold_mu);
    if (mode == 2) {
        g_held_valid = false;
        return YMI_OK;
    }
    YMI_CHECK_ARG(mode == 0 || mode == 1, "wgrad_hold: mode 0, 1 or 2");
    if (mode == 0) wgrad_flush_held_locked();
    g_hold_on = mode == 1;
    return hipGetLastError() == hipSuccess ? YMI_OK : YMI_ELAUNCH;
}

extern "C" int ymi_conv2d_bwd_weight(const ymi_tensor* x, const ymi_tensor* dy, int64_t cout_real, int64_t cin_real, int64_t kh, int64_t kw,
                                     int64_t stride, float* dw_oihw, float* dbias, void* workspace, size_t workspace_bytes, void* stream) {
    return wgrad_impl(x, dy, cout_real, cin_real, kh, kw, stride, dw_oihw, dbias, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int ymi_conv2d_bwd_weight_deferred(const ymi_tensor* x, const ymi_tensor* dy, int64_t cout_real, int64_t cin_real, int64_t kh, int64_t kw,
                                              int64_t stride, float* dw_oihw, float* dbias, void* workspace, size_t workspace_bytes,
                                              ymi_wgrad_pending* pending, void* stream) {
    YMI_CHECK_ARG(pending, "conv2d_bwd_weight_deferred: null pending record");
    return wgrad_impl(x, dy, cout_real, cin_real, kh, kw, stride, dw_oihw, dbias, workspace, workspace_bytes, pending, stream);
}

// the records travel to the device table in kernel ARGUMENTS (by value): no host staging buffer, so the launches are
// graph-capturable (a captured host-to-device copy would re-read its host buffer at replay time)
constexpr int WG_TABLE_CHUNK = 48;  // 48 x 80 B = 3.75 KB of kernel arguments
static_assert(sizeof(ymi_wgrad_pending) == 80, "ymi_wgrad_pending layout (mirrored in _lib.py)");
struct WgradTableChunk {
    ymi_wgrad_pending e[WG_TABLE_CHUNK];
};
__global__ void wgrad_table_write_kernel(WgradTableChunk c, int n, ymi_wgrad_pending* __restrict__ table) {
    const int i = threadIdx.x;
    if (i < n) table[i] = c.e[i];
}

extern "C" int ymi_wgrad_reduce_batch(const ymi_wgrad_pending* host_records, int32_t n, ymi_wgrad_pending* device_table, void* stream) {
    YMI_CHECK_ARG(host_records && device_table && n > 0, "wgrad_reduce_batch: args");
    hipStream_t s = (hipStream_t)stream;
    {   // a launch still held back for a rider (ymi_wgrad_hold): its slabs are summed below
        std::lock_guard<std::mutex> lk(g_hold_mu);
        wgrad_flush_held_locked();
    }
    int64_t total = 0;
    for (int base = 0; base < n; base += WG_TABLE_CHUNK) {
        WgradTableChunk c{};
        const int m = n - base < WG_TABLE_CHUNK ? n - base : WG_TABLE_CHUNK;
        for (int i = 0; i < m; ++i) {
            c.e[i] = host_records[base + i];
            YMI_CHECK_ARG(c.e[i].slab && c.e[i].dw && c.e[i].blocks > 0 && (c.e[i].lanes == 4 || c.e[i].lanes == 8 || c.e[i].lanes == 16 || c.e[i].lanes == 32),
                          "wgrad_reduce_batch: record %d", base + i);
            c.e[i].first_block = (int32_t)total;
            total += c.e[i].blocks;
        }
        hipLaunchKernelGGL(wgrad_table_write_kernel, dim3(1), dim3(64), 0, s, c, m, device_table + base);
    }
    YMI_CHECK_ARG(total < (1ll << 31), "wgrad_reduce_batch: too many workgroups");
    hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3((unsigned)total), dim3(256), 0, s, (const ymi_wgrad_pending*)device_table, (int)n);
    YMI_CHECK_LAUNCH("wgrad_reduce_batch");
    return YMI_OK;
}

static int wgrad_impl(const ymi_tensor* x, const ymi_tensor* dy, int64_t cout_real, int64_t cin_real, int64_t kh, int64_t kw, int64_t stride,
                      float* dw_oihw, float* dbias, void* workspace, size_t workspace_bytes, ymi_wgrad_pending* pending, void* stream) {
    YMI_CHECK_ARG(ymi_tensor_ok(x) && ymi_tensor_ok(dy) && dw_oihw && workspace, "conv2d_bwd_weight: bad argument");
    YMI_CHECK_ARG(x->dtype == dy->dtype, "conv2d_bwd_weight: dtype mismatch");
    const int ch = x->dtype == YMI_BF16 ? 8 : 4;
    YMI_CHECK_ARG(x->c % ch == 0 && x->ld % ch == 0 && dy->c % ch == 0 && dy->ld % ch == 0, "conv2d_bwd_weight: channels must be multiples of %d", ch);
    YMI_CHECK_ARG(kh == kw && (kh == 1 || kh == 3) && (stride == 1 || stride == 2), "conv2d_bwd_weight: k in {1,3}, stride in {1,2}");
    const int64_t pad = kh / 2;
    YMI_CHECK_ARG(dy->n == x->n && dy->h == (x->h + 2 * pad - kh) / stride + 1 && dy->w == (x->w + 2 * pad - kw) / stride + 1, "conv2d_bwd_weight: shapes");
    YMI_CHECK_ARG(cout_real <= dy->c && cin_real <= x->c, "conv2d_bwd_weight: real channel counts");
    YMI_CHECK_ARG(ymi_pixels(x) * x->ld < (1ll << 31) && ymi_pixels(dy) * dy->ld < (1ll << 31), "conv2d_bwd_weight: too large");
    const int64_t mpix = ymi_pixels(dy);
    const int64_t ng = kh * kw * x->c;
    const bool bf16 = x->dtype == YMI_BF16;
    const int bm = wgrad_bm(dy->c, bf16), bn = WG_BN;
    WgradPlan p = wgrad_plan(mpix, dy->c, ng, bm, bf16 ? 2 : 4);
    // bfloat16 first-level slabs only where there are many of them: a handful of rounded partials, each covering a large part of the pixel
    // sum, put 1.7e-3 on dW (the 20 x 20 maps at small batch, tests/test_gpu_bf16_matched.py); from 16 splits up the roundings average
    // out (<= 2e-3 asserted at the bs-32 shapes), and launches with few splits write little slab traffic to save anyway
    const bool slab_bf16 = bf16 && p.splits >= 16;
    if (bf16 && !slab_bf16) p = wgrad_plan(mpix, dy->c, ng, bm, 4);
    size_t need = p.slab_bytes + (dbias ? (size_t)(2048 * 2 + 1) * dy->c * sizeof(float) : 0);
    if (workspace_bytes < need) {
        ymi_set_error("conv2d_bwd_weight: workspace %zu < %zu bytes", workspace_bytes, need);
        return YMI_EWORKSPACE;
    }
    WgradArgs a{};
    a.x = x->data; a.dy = dy->data; a.slab = workspace; a.slab_bf16 = slab_bf16 ? 1 : 0; a.zero = ymi_zero_page();
    // bias gradient: per-split column sums of dY come out of the GEMM itself (a ones row against the dY fragments), behind the slabs
    a.bias_slab = dbias ? reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + p.slab_bytes) : nullptr;
    YMI_CHECK_ARG(!dbias || (size_t)p.splits * dy->c <= (size_t)(2048 * 2 + 1) * dy->c, "conv2d_bwd_weight: too many splits for the bias partials");
    a.ldx = x->ld; a.ldy = dy->ld;
    a.Mpix = (int)mpix; a.H = (int)x->h; a.W = (int)x->w; a.Ho = (int)dy->h; a.Wo = (int)dy->w;
    a.stride = (int)stride; a.pad = (int)pad; a.KW = (int)kw;
    a.CoutP = (int)dy->c; a.Cin = (int)x->c; a.NG = (int)ng; a.pix_per_split = p.pix_per_split;
    wg_find_divisor(a.Wo, &a.wo_mul, &a.wo_shr);
    wg_find_divisor(a.Ho, &a.ho_mul, &a.ho_shr);
    a.nx = (int)((ng + bn - 1) / bn); a.ny = (int)((dy->c + bm - 1) / bm); a.splits = p.splits;
    // patch order of the pixel sum (see wgrad_kernel): the widest power-of-two patch width that divides Wo with a height that divides Ho
    a.pw_shift = -1;
    {
        const int64_t xb = ymi_pixels(x) * x->ld * ymi_esize(x->dtype), yb = ymi_pixels(dy) * dy->ld * ymi_esize(dy->dtype);
        const int64_t margin = (x->w + 2) * x->ld * (int64_t)ymi_esize(x->dtype);
        if (bf16 && ymi_opt(OPT_WGRAD_PATCH) && xb + margin < (1ll << 31) && yb < (1ll << 31) && mpix % WG_BK == 0) {
            for (int sh = 5; sh >= 0; --sh) {
                const int pw = 1 << sh, ph = WG_BK >> sh;
                if (dy->w % pw == 0 && dy->h % ph == 0) {
                    a.pw_shift = sh;
                    break;
                }
            }
            a.x_bytes = (uint32_t)xb;
            a.y_bytes = (uint32_t)yb;
        }
    }
    a.xcd_map = p.splits >= 8 ? 1 : 0;  // (all tiles of a pixel split on one XCD: 275 -> 105 MB of HBM reads per launch, round 1)
    dim3 grid((unsigned)a.nx, (unsigned)a.ny, (unsigned)p.splits);
    if (a.xcd_map) grid = dim3((unsigned)(8 * ((p.splits + 7) / 8) * a.nx * a.ny), 1, 1);
    hipStream_t s = (hipStream_t)stream;
    const double es = (double)ymi_esize(x->dtype);
    WgradLaunch h{a, grid, bm, bf16, s, 2.0 * (double)mpix * (double)dy->c * (double)ng,
                  ((double)ymi_pixels(x) * x->c + (double)mpix * dy->c) * es + (double)dy->c * ng * 4.0};
    {
        std::lock_guard<std::mutex> lk(g_hold_mu);
        wgrad_flush_held_locked();  // (at most one launch is held)
        if (pending && g_hold_on && a.xcd_map) {
            // held back: the next BatchNorm backward of this stream issues it with its final pass riding along (or the next deferred call / the
            // batched slab sum / ymi_wgrad_hold(0) as it is).  The caller keeps operands and workspace alive until the slab sum anyway.
            g_held = h;
            g_held_valid = true;
        } else {
            wgrad_launch(h, nullptr);
        }
    }
    YMI_CHECK_LAUNCH("wgrad");
    const int64_t elems = (int64_t)a.CoutP * a.NG;
    if (pending) {  // the slab sum is left to ymi_wgrad_reduce_batch (one launch for every layer of the backward pass)
        const int lanes = p.splits > 128 ? 32 : p.splits > 32 ? 16 : p.splits > 8 ? 8 : 4;
        const int ept = a.slab_bf16 ? 8 : 4;  // elements per lane of the batched sum (one 16-byte load per split)
        *pending = ymi_wgrad_pending{a.slab, dw_oihw, elems, p.splits, a.NG, a.Cin, (int32_t)cout_real, (int32_t)cin_real, (int32_t)(kh * kw), lanes, 0,
                                     (int32_t)((elems / ept + 256 / lanes - 1) / (256 / lanes)), a.slab_bf16, a.bias_slab, dbias};
    } else {
        const int64_t total = cout_real * kh * kw * cin_real;
        int64_t gb = (total + 255) / 256;
        if (gb > 2048) gb = 2048;
        if (slab_bf16) hipLaunchKernelGGL(wgrad_reduce_kernel<true>, dim3((unsigned)gb), dim3(256), 0, s, (const void*)a.slab, p.splits, a.CoutP, a.NG, a.Cin,
                                     (int)cout_real, (int)cin_real, (int)(kh * kw), dw_oihw);
        else hipLaunchKernelGGL(wgrad_reduce_kernel<false>, dim3((unsigned)gb), dim3(256), 0, s, (const void*)a.slab, p.splits, a.CoutP, a.NG, a.Cin,
                                (int)cout_real, (int)cin_real, (int)(kh * kw), dw_oihw);
    }
    YMI_CHECK_LAUNCH("wgrad_reduce");
    if (dbias && !pending) {
        // bias gradient: the sum of the per-split column sums, written straight into the caller's buffer of dy->c floats (the real channels come first)
        hipLaunchKernelGGL(wgrad_bias_reduce_kernel, dim3((unsigned)((dy->c + 255) / 256)), dim3(256), 0, s, (const float*)a.bias_slab, p.splits, (int)dy->c, dbias);
        YMI_CHECK_LAUNCH("wgrad_bias_reduce");
    }
    return YMI_OK;
}
