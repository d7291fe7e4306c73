// Shared device/host helpers for libyolo_mi355.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ymi.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define YMI_WAVE 64

// ---- error plumbing (host) -------------------------------------------------------------------
void ymi_set_error(const char* fmt, ...);
#define YMI_CHECK_ARG(cond, ...)          \
    do {                                  \
        if (!(cond)) {                    \
            ymi_set_error(__VA_ARGS__);   \
            return YMI_EINVAL;            \
        }                                 \
    } while (0)
#define YMI_CHECK_LAUNCH(what)                                                   \
    do {                                                                         \
        hipError_t e_ = hipGetLastError();                                       \
        if (e_ != hipSuccess) {                                                  \
            ymi_set_error("%s: launch failed: %s", what, hipGetErrorString(e_)); \
            return YMI_ELAUNCH;                                                  \
        }                                                                        \
    } while (0)

static inline bool ymi_tensor_ok(const ymi_tensor* t) {
    return t && t->data && t->n > 0 && t->h > 0 && t->w > 0 && t->c > 0 && t->ld >= t->c && (t->dtype == YMI_F32 || t->dtype == YMI_BF16);
}
static inline int64_t ymi_pixels(const ymi_tensor* t) { return t->n * t->h * t->w; }
// binary point of the fixed-point BatchNorm statistics (igemm.hip statistics epilogue -> elementwise.hip BnFin): 37 - ceil(log2(count)), in [8, 40]
static inline int ymi_stat_fixed_point_shift(int64_t count) {
    int lg = 0;
    while (((int64_t)1 << lg) < count) ++lg;
    const int sh = 37 - lg;
    return sh < 8 ? 8 : sh > 40 ? 40 : sh;
}
static inline size_t ymi_esize(int dtype) { return dtype == YMI_BF16 ? 2 : 4; }
static inline bool ymi_same_shape(const ymi_tensor* a, const ymi_tensor* b) {
    return a->n == b->n && a->h == b->h && a->w == b->w && a->c == b->c;
}
// Development options (runtime.hip): named integers behind ymi_set_option / ymi_get_option - the "before" arm of a measured change, grid
// sizes for in-step sweeps, forced code paths for tests.  Each starts at its default, or at the value of the environment variable
// YMI_<NAME> when the process starts with one (tools/*.sh sweeps); production code never sets any.
enum YmiOpt {
    OPT_EW_PPT,           // streaming BatchNorm passes: pixels per thread aimed for (64)
    OPT_EW_CAP,           // ... and their workgroup cap (512)
    OPT_RED_CAP,          // BatchNorm backward reduce: workgroup cap (512)
    OPT_XCD_SHIFT,        // streaming passes work on XCD (x + k) % 8's pixels: the anti-affine arrangement (0)
    OPT_ATTN_TILED,       // 1: tiled window attention for every window size (tests) (0)
    OPT_WGRAD_BLOCKS,     // weight gradient: split-K workgroup target, 64-row tiles (1280)
    OPT_WGRAD_BLOCKS128,  // ... 128-row tiles (768)
    OPT_IGEMM_TILE_BM,    // force a GEMM tile (tools/conv_bench.py sweeps); 0: choose_tile
    OPT_IGEMM_TILE_BN,
    OPT_BN_TAIL,          // 1: BatchNorm backward's final pass inside the reduce kernel (last-arriver hand-off: measured slower,
                          //    profiles/r05_bn_tail_ab.txt); 0 (default): its own launch
    OPT_WGRAD_PATCH,      // 1 (default): the weight gradient sums pixels in patch order with scalar addressing where the map tiles; 0: raster walk
    OPT_COUNT
};
int ymi_opt(int id);
static inline int ew_ppt() { return ymi_opt(OPT_EW_PPT); }
static inline int ew_cap() { return ymi_opt(OPT_EW_CAP); }
// a 16-byte block of zeros in device memory (source of out-of-bounds im2col taps)
const void* ymi_zero_page();

// ---- element access helpers (device) ----------------------------------------------------------
template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
    static constexpr int CH = 4;  // elements per 16-byte chunk
    static constexpr int DT = YMI_F32;
};
template <> struct ElemTraits<bf16_t> {
    static constexpr int CH = 8;
    static constexpr int DT = YMI_BF16;
};

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }  // v_cvt_pk_bf16_f32: RNE, NaN-safe

// load / store a group of G consecutive elements as floats (vector access when aligned by construction)
template <typename T, int G> struct Pack;
template <> struct Pack<float, 4> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
        f32x4 t = *reinterpret_cast<const f32x4*>(p);
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
        f32x4 t = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(p) = t;
    }
};
template <> struct Pack<bf16_t, 4> {
    static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[4]) {
        bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
        v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[4]) {
        bf16x4 t = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        *reinterpret_cast<bf16x4*>(p) = t;
    }
};
template <> struct Pack<bf16_t, 8> {
    static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
        bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
        bf16x8 t;
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = (bf16_t)v[i];
        *reinterpret_cast<bf16x8*>(p) = t;
    }
};

// e^x as one v_exp_f32 (2^(x*log2 e)): ~1 ulp, results below 2^-126 flush to zero; no range-reduction code
// four consecutive elements as they lie in memory (8 bytes of bfloat16, 16 bytes of float32) and their conversion
template <typename T> struct Raw4;
template <> struct Raw4<bf16_t> {
    typedef bf16x4 type;
    static __device__ __forceinline__ void to_f32(const bf16x4& r, float (&v)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (float)r[i];
    }
};
template <> struct Raw4<float> {
    typedef f32x4 type;
    static __device__ __forceinline__ void to_f32(const f32x4& r, float (&v)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = r[i];
    }
};

// Workgroups are dealt round-robin to the 8 XCDs (flat id & 7), each with its own L2.  Work units that read neighbouring bytes
// (the 16-byte channel chunks of one 128-byte line, the heads of one token row) should therefore NOT sit on consecutive flat
// ids: every XCD would fetch the whole line.  xcd_unit() turns the flat workgroup id into a work-unit index such that
// consecutive UNITS run on one XCD, back to back (a permutation of [0, total) when total % 8 == 0, the identity otherwise).
// ---- XCD ownership of the pixel axis (round 4) -----------------------------------------------------------------------------------
// A consumer finds its producer's bytes in its own XCD's L2 - 3.7-6x faster for a latency-shaped reader such as a GEMM's operand
// ring than from another XCD's L2 / the Infinity Cache (profiles/r04_xcd_affinity_probe.txt) - when both kernels give an XCD the
// same part of the tensor.  ONE rule for every kernel that walks pixels (or tokens): the XCD a workgroup runs on (flat id & 7:
// workgroups are dealt round-robin, measured stable from launch to launch) owns the pixels [x * span, (x + 1) * span) of the
// batch-major pixel order, span = ymi_xcd_span(P).  The rule is a FRACTION of the pixel order, so it lines up across resolutions
// (a stride-2 consumer's eighth of the output reads the same eighth of its input), and at batch sizes that are multiples of 8 it is
// whole images.  Placement is a speed matter only: nothing depends on it for correctness.
__host__ __device__ __forceinline__ int64_t ymi_xcd_span(int64_t P) { return (((P + 63) >> 6) + 7) >> 3 << 6; }
// the pixel range [lo, hi) of this workgroup's XCD, the workgroup's index among that XCD's workgroups and their number
// (the launch grid must be a multiple of 8 workgroups)
struct XcdRange {
    int64_t lo, hi;
    int bi, nbx;
};
int64_t ymi_xcd_span_arg(int64_t P);  // host: ymi_xcd_span(P), with the diagnostic shift of YMI_XCD_SHIFT in bits 56..58
__device__ __forceinline__ XcdRange xcd_range(int64_t P, int64_t span_arg) {
    const int id = blockIdx.x, x = ((id & 7) + (int)(span_arg >> 56)) & 7;
    const int64_t span = span_arg & ((1ll << 56) - 1);
    XcdRange r;
    r.lo = x * span;
    r.hi = r.lo + span < P ? r.lo + span : P;
    if (r.lo > P) r.lo = P;
    r.bi = id >> 3;
    r.nbx = gridDim.x >> 3;
    return r;
}

__device__ __forceinline__ int xcd_unit(int flat, int total) { return (total & 7) ? flat : (flat & 7) * (total >> 3) + (flat >> 3); }
__device__ __forceinline__ int flat_block_id() { return blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); }

__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
// 1 / (1 + e^-x) with v_exp_f32 + v_rcp_f32 (each ~1 ulp) instead of the IEEE division sequence
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + fast_exp(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x * sigmoidf_(x); }
__device__ __forceinline__ float silu_grad_f(float x) {
    float s = sigmoidf_(x);
    return s * (1.0f + x * (1.0f - s));
}
// erf(z) by Abramowitz & Stegun 7.1.26: 1 - (a1 t + .. + a5 t^5) e^(-z^2), t = 1 / (1 + p |z|); absolute error <= 1.5e-7 - float32 noise
// next to the 1 the GELU adds it to - in 6 multiply-adds, v_rcp_f32 and v_exp_f32.  libm's erff is ~45 instructions with two branches
// that a wave executes both of; in the Swin MLP's GEMM epilogues (32768 activations per 256 x 128 tile) that was as long as the K loop.
// e2 (optional): receives e^(-z^2), which GELU's derivative needs as well.
__device__ __forceinline__ float erf_as(float z, float* e2 = nullptr) {
    const float az = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * az);
    const float e = fast_exp(-az * az);
    float p = 1.061405429f;
    p = p * t - 1.453152027f;
    p = p * t + 1.421413741f;
    p = p * t - 0.284496736f;
    p = p * t + 0.254829592f;
    if (e2) *e2 = e;
    return copysignf(1.0f - p * t * e, z);
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float c = 0.39894228040143267794f;  // 1/sqrt(2 pi)
    float e;                                   // e^(-x^2 / 2)
    const float er = erf_as(x * 0.70710678118654752440f, &e);
    return 0.5f * (1.0f + er) + x * c * e;
}
template <int ACT> __device__ __forceinline__ float apply_act(float x) {
    if (ACT == YMI_ACT_SILU) return silu_f(x);
    if (ACT == YMI_ACT_GELU) return gelu_f(x);
    return x;
}
__device__ __forceinline__ float apply_act_rt(float x, int act) {
    return act == YMI_ACT_SILU ? silu_f(x) : act == YMI_ACT_GELU ? gelu_f(x) : x;
}
template <int ACT> __device__ __forceinline__ float act_grad(float x) {
    if (ACT == YMI_ACT_SILU) return silu_grad_f(x);
    if (ACT == YMI_ACT_GELU) return gelu_grad_f(x);
    return 1.0f;
}
__device__ __forceinline__ float act_grad_rt(float x, int act) {
    return act == YMI_ACT_SILU ? silu_grad_f(x) : act == YMI_ACT_GELU ? gelu_grad_f(x) : 1.0f;
}

// ---- in-launch hand-offs between workgroups (round 5) ----------------------------------------------------------------------------
// The per-XCD L2s are not coherent and a CU's L1 is never refreshed by another CU's stores, so a workgroup that consumes another
// workgroup's bytes INSIDE a launch needs them written through and read past its L1.  The form used here (MI355X_MICROARCH.md,
// "Valid forms", first table row; cdna_hip_programming.md Guideline 16 R1): every handed-off byte is stored with an agent-scope relaxed
// atomic store (global_store ... sc1: write-through, no release fence), every storing wave drains with s_waitcnt vmcnt(0), the workgroup
// meets at a barrier, ONE lane takes a ticket with an agent-scope relaxed fetch-add, and the workgroup whose add returned the last
// ticket reads the bytes with agent-scope relaxed atomic loads (global_load ... sc1: past L1).  No __threadfence(), no buffer_wbl2, no
// buffer_inv anywhere: round 4's form of the same idea (every thread of every workgroup fencing twice) cost 130 us per launch.
// Results never depend on who arrives last: the combining workgroup sums fixed rows in fixed order.
__device__ __forceinline__ void st_wt(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_wt2(float* p, float a, float b) {  // 8-byte aligned pair
    const uint64_t v = (uint64_t)__builtin_bit_cast(uint32_t, a) | ((uint64_t)__builtin_bit_cast(uint32_t, b) << 32);
    __hip_atomic_store(reinterpret_cast<uint64_t*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_wt(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_wt(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_wt(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ld_wt2(const float* p, float& a, float& b) {
    const uint64_t v = __hip_atomic_load(reinterpret_cast<const uint64_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a = __builtin_bit_cast(float, (uint32_t)v);
    b = __builtin_bit_cast(float, (uint32_t)(v >> 32));
}
// Ticket: call from EVERY thread of the workgroup after the write-through stores.  `flag` is one LDS word nothing else uses until the
// second barrier.  Returns (workgroup-uniform) whether this workgroup drew ticket `last`; that workgroup also resets the counter, so
// counters are zero between launches (they start zero: __device__ storage, runtime.hip).
__device__ __forceinline__ bool ticket_is_last(unsigned* counter, unsigned last, int* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // every workgroup has drawn
        *flag = t == last;
    }
    __syncthreads();
    return *flag != 0;
}
// ---- a BatchNorm-backward final pass riding in a weight-gradient launch (round 5; wgrad.hip, reduce_bwd.hip) --------------------------
// The final pass of a BatchNorm backward (sum <= 512 partial rows per channel, derive the apply pass's coefficients) is a 1-16 workgroup
// kernel at a dependent-launch latency (~5.4 us, 57 times a step), and it cannot move into its producer or its consumer (a hand-off level
// costs ~2.5 us, profiles/r05_bn_tail_ab.txt).  But the weight-gradient GEMM of the PREVIOUS layer of the backward pass is independent of it
// and sits right beside it in the stream: with ymi_wgrad_hold(1) a deferred weight-gradient launch is held back until the next BatchNorm
// backward has issued its reduce pass and then launched with that layer's final pass as extra workgroups at the front of its grid.
struct YmiBnRider {
    const float* part;  // [blocks][2][C] partial rows of the reduce pass
    int blocks, C;
    int nwg;            // rider workgroups at the front of the grid (a multiple of 8: the XCD dealing of the rest is unchanged); 0: none
    float* out0;        // dbeta
    float* out1;        // dgamma
    const float* gamma; const float* beta; const float* mean; const float* inv;
    const float* gamma2; const float* beta2; int split;  // second parameter set of a convolution pair (channels >= split); split == 0: one
    float inv_count;
    float* coef;        // [5][C]
};
// issue the held weight-gradient launch of `stream` with `rider` in it -> true; false when nothing is held there (the caller launches its final pass itself)
bool ymi_wgrad_issue_held(const YmiBnRider* rider, hipStream_t stream);

// a slot of 64 ticket counters for one launch (host; round-robin over 1024 slots, zero between launches)
unsigned* ymi_ticket_slot();

// retire all but the N youngest vector-memory operations of this wave, then meet the workgroup: bytes written
// to LDS by LDS-DMA (global_load_lds) are readable by other waves only after BOTH (counted wait, then barrier).
template <int N> __device__ __forceinline__ void wait_vmcnt_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// wave-level reductions (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
