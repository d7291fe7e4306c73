// Optimizer step of the training path as three multi-tensor launches over ALL parameters:
//   reference engine/trainer.py:614-622  optimizer_step: clip_grad_norm_(max_norm 10) -> SGD(nesterov) step -> EMA update
//   reference engine/trainer.py:788-849  build_optimizer: three parameter groups (biases / decayed weights / norm weights)
//   reference utils/torch_utils.py:657-673 ModelEMA.update: v = d*v + (1-d)*model value, d = decay*(1 - exp(-updates/tau))
//
//  1. ymi_opt_sumsq   : per-chunk sums of squares of every gradient           (reads g once)
//  2. (same call)     : fixed-order final sum -> total norm -> clip coefficient, update counter += 1, EMA decay
//  3. ymi_opt_update  : g*clip (+ wd*p) -> momentum buffer -> nesterov -> p -= lr*g -> ema  (one pass: p, g, buf, ema)
//     or, hyper[14] = 1 / 2, torch.optim.AdamW / Adam (the reference's 'AdamW' / 'Adam' branches, trainer.py:829-830, which
//     'auto' picks for short runs, :812): decoupled decay p *= 1 - lr*wd (Adam: g += wd*p), m = lerp(m, g, 1-b1),
//     v = b2*v + (1-b2)*g*g, p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps); t counted on the device by pass 2.
//     hyper[14] = 3 .. 6: the remaining branches of build_optimizer (trainer.py:827-832) with torch's default hyper-parameters -
//     Adamax (exp_inf = max(b2*exp_inf, |g| + eps)), NAdam (Nesterov momentum schedule mu_t, its running product kept in the state),
//     RAdam (variance rectification once rho_t > 5), RMSprop(momentum) - each one straight-line body of the same pass.
//
// HBM-bound: 28 B per parameter (p, g, buf, ema read; p, buf, ema written; Adam: 36 B with the second moment).  No torch.stack / foreach chains: every
// operand is addressed through a device table built once (parameters, momentum and EMA buffers never move); the gradient
// tensors, whose addresses change from step to step in eager mode, travel by value in the kernel arguments, so the
// launches are graph-capturable without any host staging buffer.
// Hyper-parameters (per-group lr / weight decay, momentum, max norm, EMA decay and tau) live in a small device array the
// host rewrites only when a scheduler changes them: a replayed HIP graph follows the schedule.
#include <math.h>

#include "common.h"

namespace {
constexpr int OPT_CHUNK = 4096;  // elements per workgroup (256 threads x 4 float4)
constexpr int OPT_GRADS = YMI_OPT_MAX_GRADS;

struct GradPtrs {
    const float* g[OPT_GRADS];
};

// hyper layout (floats): [0..2] lr of group 0..2, [3..5] weight decay of group 0..2, [6] momentum, [7] max_norm,
//                        [8] ema decay, [9] ema tau, [10] nesterov (0/1), [11] grad scale (1/world),
//                        [12] beta2 (RMSprop: alpha), [13] eps, [14] rule (0 SGD, 1 AdamW, 2 Adam, 3 Adamax, 4 NAdam, 5 RAdam, 6 RMSprop),
//                        [15] 1 - beta2, [16] 1 - beta1 (the host's double differences, as torch passes them), [17] NAdam's
//                        momentum_decay, [18] / [19] float32 tails of beta1 / momentum_decay (NAdam)                                  ([6] is beta1 for the Adam family)
// state layout: float clip, float total_norm, float ema_d, float one_minus_d, int64 updates, int64 steps (optimizer steps
//               taken: Adam's t), float 1/(1-b1^t), float sqrt(1-b2^t), float c0..c3 (per-step scalars of rules 4 / 5), double mu_product (NAdam)
struct OptState {
    float clip, norm, ema_d, ema_1md;
    long long updates;
    long long steps;
    float inv_bc1, bc2_sqrt;
    float c0, c1, c2, c3;
    double mu_product;
};
static_assert(sizeof(OptState) == 64, "OptState is the 64-byte state buffer engine/optim.py allocates");

__global__ __launch_bounds__(256) void opt_sumsq_kernel(const ymi_opt_entry* __restrict__ tab, const int2* __restrict__ chunks, int first_tensor, GradPtrs gp,
                                                        const float* __restrict__ hyper, float* __restrict__ part) {
    const int2 ch = chunks[blockIdx.x];
    const ymi_opt_entry e = tab[ch.x];
    const float* g = gp.g[ch.x - first_tensor];
    float acc = 0.f;
    if (g) {
        const float gs = hyper[11];
        const int64_t base = (int64_t)ch.y * OPT_CHUNK;
        const int64_t end = base + OPT_CHUNK < e.numel ? base + OPT_CHUNK : e.numel;
        if ((((uintptr_t)g) & 15) == 0) {
            for (int64_t i = base + threadIdx.x * 4; i < end; i += 1024) {
                if (i + 3 < end) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(g + i);
                    acc += (v[0] * gs) * (v[0] * gs) + (v[1] * gs) * (v[1] * gs) + (v[2] * gs) * (v[2] * gs) + (v[3] * gs) * (v[3] * gs);
                } else {
                    for (int64_t j = i; j < end; ++j) acc += (g[j] * gs) * (g[j] * gs);
                }
            }
        } else {
            for (int64_t i = base + threadIdx.x; i < end; i += 256) acc += (g[i] * gs) * (g[i] * gs);
        }
    }
    __shared__ float sh[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void opt_finalize_kernel(const float* __restrict__ part, int n, const float* __restrict__ hyper, OptState* __restrict__ st) {
    __shared__ double sh[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += (double)part[i];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(sh[0]);
        // torch.nn.utils.clip_grad_norm_: clip_coef = max_norm / (total_norm + 1e-6), clamped to 1
        const float coef = hyper[7] / (norm + 1e-6f);
        st->norm = norm;
        st->clip = hyper[7] > 0.f ? (coef < 1.0f ? coef : 1.0f) : 1.0f;
        const long long u = st->updates + 1;
        st->updates = u;
        // ModelEMA.decay(updates) in double, as the reference's Python float arithmetic
        const double d = (double)hyper[8] * (1.0 - exp(-(double)u / (double)hyper[9]));
        st->ema_d = (float)d;
        st->ema_1md = (float)(1.0 - d);
        // torch.optim.Adam: bias_correction1 = 1 - beta1 ** step, bias_correction2_sqrt = (1 - beta2 ** step) ** 0.5 (Python floats)
        const long long t = st->steps + 1;
        st->steps = t;
        if (hyper[14] != 0.f) {
            // (the betas are rebuilt from their float32 COMPLEMENTS: 1 - 0.999f is off by 1.3e-5 relative, 1 - float(0.001) by 5e-8)
            const double b1 = 1.0 - (double)hyper[16], b2 = 1.0 - (double)hyper[15];
            const double bc1 = 1.0 - pow(b1, (double)t), bc2 = 1.0 - pow(b2, (double)t);
            st->inv_bc1 = (float)(1.0 / bc1);
            st->bc2_sqrt = (float)sqrt(bc2);
            const int rule = (int)hyper[14];
            if (rule == 4) {
                // torch.optim.NAdam: mu_t = beta1 * (1 - 0.5 * 0.96 ** (t * momentum_decay)), mu_product *= mu_t (Python floats)
                // beta1 and momentum_decay to double precision (float32 head + tail: hyper[6] + hyper[18], hyper[17] + hyper[19]) - mu_product is
                // a running product that torch rounds to float32 every step, and a 1e-9 error of beta1 flips one of those roundings within ten steps
                const double b1x = (double)hyper[6] + (double)hyper[18], md = (double)hyper[17] + (double)hyper[19];
                const double mu = b1x * (1.0 - 0.5 * pow(0.96, (double)t * md)), mu_next = b1x * (1.0 - 0.5 * pow(0.96, (double)(t + 1) * md));
                // (torch keeps mu_product as a float32 state tensor and multiplies it in place: one float32 rounding per step)
                const float mpf = (t == 1 ? 1.0f : (float)st->mu_product) * (float)mu;
                const double mp = (double)mpf;
                st->mu_product = mp;
                st->c0 = (float)((1.0 - mu) / (1.0 - mp));            // weight of grad / denom
                st->c1 = (float)(mu_next / (1.0 - mp * mu_next));     // weight of exp_avg / denom
                st->c2 = (float)bc2;                                  // denom = sqrt(exp_avg_sq / bias_correction2) + eps
            } else if (rule == 5) {
                // torch.optim.RAdam: rho_t = rho_inf - 2 t beta2^t / (1 - beta2^t); rectified step once rho_t > 5
                const double rho_inf = 2.0 / (1.0 - b2) - 1.0;
                const double rho_t = rho_inf - 2.0 * (double)t * pow(b2, (double)t) / bc2;
                if (rho_t > 5.0) {
                    st->c0 = (float)sqrt((rho_t - 4.0) * (rho_t - 2.0) * rho_inf / ((rho_inf - 4.0) * (rho_inf - 2.0) * rho_t));
                    st->c1 = 1.0f;
                } else {
                    st->c0 = 1.0f;
                    st->c1 = 0.0f;
                }
            }
        }
    }
}

// MODE 0: parameters with gradients (update + EMA);  MODE 1: EMA only (buffers, frozen parameters)
// RULE 0: SGD-momentum; 1: AdamW; 2: Adam; 3: Adamax; 4: NAdam; 5: RAdam; 6: RMSprop (a compile-time parameter: one straight-line body per rule)
template <int MODE, int RULE>
__global__ __launch_bounds__(256) void opt_update_kernel(const ymi_opt_entry* __restrict__ tab, const int2* __restrict__ chunks, int first_tensor, GradPtrs gp,
                                                         const float* __restrict__ hyper, const OptState* __restrict__ st) {
    const int2 ch = chunks[blockIdx.x];
    const ymi_opt_entry e = tab[ch.x];
    const float* g = MODE == 0 ? gp.g[ch.x - first_tensor] : nullptr;
    float* p = e.param;
    float* buf = e.momentum;
    float* sec = e.second;  // Adam: exp_avg_sq
    float* ema = e.ema;
    const int64_t base = (int64_t)ch.y * OPT_CHUNK;
    const int64_t end = base + OPT_CHUNK < e.numel ? base + OPT_CHUNK : e.numel;
    const float d = st->ema_d, omd = st->ema_1md;
    float lr = 0.f, wd = 0.f, mom = 0.f, gscale = 0.f, b2 = 0.f, eps = 0.f, step_size = 0.f, bc2s = 1.f, omb1 = 0.f, omb2 = 0.f;
    float c0 = 0.f, c1 = 0.f, c2 = 1.f, ibc1 = 1.f;
    bool nest = false;
    if (MODE == 0 && g) {
        const int grp = e.group;
        lr = hyper[grp];
        wd = hyper[3 + grp];
        mom = hyper[6];
        nest = hyper[10] != 0.f;
        gscale = st->clip * hyper[11];
        if (RULE) {
            b2 = hyper[12];
            eps = hyper[13];
            omb2 = hyper[15];
            omb1 = hyper[16];
            step_size = lr * st->inv_bc1;
            bc2s = st->bc2_sqrt;
            ibc1 = st->inv_bc1;
            c0 = st->c0; c1 = st->c1; c2 = st->c2;
        }
    }
    constexpr bool adam = RULE != 0;
    const float decay_mul = 1.f - lr * wd;
    auto one = [&](float pv, float gv, float bv, float sv, float ev, float& po, float& bo, float& so, float& eo) {
        so = sv;
        if (MODE == 0 && RULE >= 3 && g) {
            float gr = gv * gscale;
            if (wd != 0.f) gr = gr + wd * pv;                // grad.add(param, alpha=weight_decay)
            if (RULE == 3) {                                 // torch.optim.Adamax
                bo = bv + (gr - bv) * omb1;                  // exp_avg.lerp_(grad, 1 - beta1)
                so = fmaxf(sv * b2, fabsf(gr) + eps);        // exp_inf = max(exp_inf * beta2, |grad| + eps)
                po = pv - step_size * (bo / so);             // param.addcdiv_(exp_avg, exp_inf, value=-lr / bias_correction1)
            } else if (RULE == 4) {                          // torch.optim.NAdam
                bo = bv + (gr - bv) * omb1;
                so = sv * b2 + omb2 * gr * gr;
                const float denom = sqrtf(so / c2) + eps;    // exp_avg_sq.div(bias_correction2).sqrt().add_(eps)
                const float p1 = pv - (lr * c0) * (gr / denom);   // param.addcdiv_(grad, denom, value=-lr (1 - mu) / (1 - mu_product))
                po = p1 - (lr * c1) * (bo / denom);               // param.addcdiv_(exp_avg, denom, value=-lr mu_next / (1 - mu_product mu_next))
            } else if (RULE == 5) {                          // torch.optim.RAdam
                bo = bv + (gr - bv) * omb1;
                so = sv * b2 + omb2 * gr * gr;
                const float bce = bo * ibc1;                 // exp_avg / bias_correction1
                if (c1 != 0.f) po = pv - ((bce * lr) * (bc2s / (sqrtf(so) + eps))) * c0;  // bias_corrected_exp_avg * lr * adaptive * rect
                else po = pv - bce * lr;
            } else {                                         // torch.optim.RMSprop(momentum > 0), alpha = b2
                so = sv * b2 + omb2 * gr * gr;               // square_avg.mul_(alpha).addcmul_(grad, grad, value=1 - alpha)
                const float avg = sqrtf(so) + eps;
                bo = bv * mom + gr / avg;                    // buf.mul_(momentum).addcdiv_(grad, avg)
                po = pv - lr * bo;
            }
        } else if (MODE == 0 && adam && g) {
            float gr = gv * gscale;
            float pw = pv;
            if (RULE == 1) pw = pv * decay_mul;             // AdamW: param.mul_(1 - lr * weight_decay)
            else if (wd != 0.f) gr = gr + wd * pv;          // Adam: grad.add(param, alpha=weight_decay)
            bo = bv + (gr - bv) * omb1;                     // exp_avg.lerp_(grad, 1 - beta1)
            so = sv * b2 + omb2 * gr * gr;                  // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
            const float denom = sqrtf(so) / bc2s + eps;     // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
            po = pw - step_size * (bo / denom);             // param.addcdiv_(exp_avg, denom, value=-lr / bias_correction1)
        } else if (MODE == 0 && g) {
            float gr = gv * gscale;
            if (wd != 0.f) gr = gr + wd * pv;           // grad.add(param, alpha=weight_decay)
            bo = bv * mom + gr;                         // buf.mul_(momentum).add_(grad)   (first step: buf = 0 -> grad)
            gr = nest ? gr + mom * bo : bo;             // grad.add(buf, alpha=momentum)
            po = pv - lr * gr;                          // param.add_(grad, alpha=-lr)
        } else {
            po = pv;
            bo = bv;
        }
        eo = ev * d + omd * po;                         // v *= d; v += (1 - d) * model value
    };
    const bool vec = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)buf) | ((uintptr_t)sec) | ((uintptr_t)ema)) & 15) == 0;
    const bool upd = MODE == 0 && g;
    if (vec) {
        for (int64_t i = base + threadIdx.x * 4; i < end; i += 1024) {
            if (i + 3 < end) {
                const f32x4 pv = *reinterpret_cast<const f32x4*>(p + i);
                const f32x4 gv = upd ? *reinterpret_cast<const f32x4*>(g + i) : f32x4{0.f, 0.f, 0.f, 0.f};
                const f32x4 bv = (upd && buf) ? *reinterpret_cast<const f32x4*>(buf + i) : f32x4{0.f, 0.f, 0.f, 0.f};
                const f32x4 sv = (upd && adam) ? *reinterpret_cast<const f32x4*>(sec + i) : f32x4{0.f, 0.f, 0.f, 0.f};
                const f32x4 ev = ema ? *reinterpret_cast<const f32x4*>(ema + i) : f32x4{0.f, 0.f, 0.f, 0.f};
                float pa[4], ba[4], sa[4], ea[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) one(pv[r], gv[r], bv[r], sv[r], ev[r], pa[r], ba[r], sa[r], ea[r]);
                const f32x4 po = {pa[0], pa[1], pa[2], pa[3]}, bo = {ba[0], ba[1], ba[2], ba[3]}, so = {sa[0], sa[1], sa[2], sa[3]},
                            eo = {ea[0], ea[1], ea[2], ea[3]};
                if (upd) {
                    *reinterpret_cast<f32x4*>(p + i) = po;
                    if (buf) *reinterpret_cast<f32x4*>(buf + i) = bo;
                    if (adam) *reinterpret_cast<f32x4*>(sec + i) = so;
                }
                if (ema) *reinterpret_cast<f32x4*>(ema + i) = eo;
            } else {
                for (int64_t j = i; j < end; ++j) {
                    float po, bo, so, eo;
                    one(p[j], upd ? g[j] : 0.f, (upd && buf) ? buf[j] : 0.f, (upd && adam) ? sec[j] : 0.f, ema ? ema[j] : 0.f, po, bo, so, eo);
                    if (upd) {
                        p[j] = po;
                        if (buf) buf[j] = bo;
                        if (adam) sec[j] = so;
                    }
                    if (ema) ema[j] = eo;
                }
            }
        }
    } else {
        for (int64_t j = base + threadIdx.x; j < end; j += 256) {
            float po, bo, so, eo;
            one(p[j], upd ? g[j] : 0.f, (upd && buf) ? buf[j] : 0.f, (upd && adam) ? sec[j] : 0.f, ema ? ema[j] : 0.f, po, bo, so, eo);
            if (upd) {
                p[j] = po;
                if (buf) buf[j] = bo;
                if (adam) sec[j] = so;
            }
            if (ema) ema[j] = eo;
        }
    }
}
}  // namespace

extern "C" int64_t ymi_opt_chunk_elems(void) { return OPT_CHUNK; }

static int opt_check(const ymi_opt_entry* table, const int32_t* chunk_map, int32_t first_tensor, int32_t n_tensors, int64_t n_chunks, const float* hyper,
                     void* state, const char* what) {
    YMI_CHECK_ARG(table && chunk_map && hyper && state, "%s: null argument", what);
    YMI_CHECK_ARG(first_tensor >= 0 && n_tensors >= 1 && n_tensors <= OPT_GRADS, "%s: 1..%d tensors per launch", what, OPT_GRADS);
    YMI_CHECK_ARG(n_chunks >= 1 && n_chunks < (1ll << 31), "%s: chunk count", what);
    return YMI_OK;
}

extern "C" int ymi_opt_grad_norm(const ymi_opt_entry* table, const int32_t* chunk_map, int32_t first_tensor, int32_t n_tensors, int64_t n_chunks,
                                 const float* const* host_grads, const float* hyper, float* partials, int64_t partials_offset, int64_t partials_total,
                                 void* state, int32_t finalize, void* stream) {
    int rc = opt_check(table, chunk_map, first_tensor, n_tensors, n_chunks, hyper, state, "opt_grad_norm");
    if (rc) return rc;
    YMI_CHECK_ARG(host_grads && partials && partials_offset >= 0 && partials_offset + n_chunks <= partials_total, "opt_grad_norm: partials range");
    GradPtrs gp{};
    for (int i = 0; i < n_tensors; ++i) gp.g[i] = host_grads[i];
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(opt_sumsq_kernel, dim3((unsigned)n_chunks), dim3(256), 0, s, table, reinterpret_cast<const int2*>(chunk_map), first_tensor, gp, hyper,
                       partials + partials_offset);
    if (finalize) hipLaunchKernelGGL(opt_finalize_kernel, dim3(1), dim3(256), 0, s, partials, (int)partials_total, hyper, reinterpret_cast<OptState*>(state));
    YMI_CHECK_LAUNCH("opt_grad_norm");
    return YMI_OK;
}

extern "C" int ymi_opt_update(const ymi_opt_entry* table, const int32_t* chunk_map, int32_t first_tensor, int32_t n_tensors, int64_t n_chunks,
                              const float* const* host_grads, const float* hyper, const void* state, int32_t rule, void* stream) {
    int rc = opt_check(table, chunk_map, first_tensor, n_tensors, n_chunks, hyper, const_cast<void*>(state), "opt_update");
    if (rc) return rc;
    YMI_CHECK_ARG(rule >= 0 && rule <= 6, "opt_update: rule 0 (SGD), 1 (AdamW), 2 (Adam), 3 (Adamax), 4 (NAdam), 5 (RAdam) or 6 (RMSprop)");
    GradPtrs gp{};
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)n_chunks), block(256);
    const int2* cm = reinterpret_cast<const int2*>(chunk_map);
    const OptState* st = reinterpret_cast<const OptState*>(state);
    if (host_grads) {
        for (int i = 0; i < n_tensors; ++i) gp.g[i] = host_grads[i];
        if (rule == 0) hipLaunchKernelGGL((opt_update_kernel<0, 0>), grid, block, 0, s, table, cm, first_tensor, gp, hyper, st);
        else if (rule == 1) hipLaunchKernelGGL((opt_update_kernel<0, 1>), grid, block, 0, s, table, cm, first_tensor, gp, hyper, st);
        else if (rule == 2) hipLaunchKernelGGL((opt_update_kernel<0, 2>), grid, block, 0, s, table, cm, first_tensor, gp, hyper, st);
        else if (rule == 3) hipLaunchKernelGGL((opt_update_kernel<0, 3>), grid, block, 0, s, table, cm, first_tensor, gp, hyper, st);
        else if (rule == 4) hipLaunchKernelGGL((opt_update_kernel<0, 4>), grid, block, 0, s, table, cm, first_tensor, gp, hyper, st);
        else if (rule == 5) hipLaunchKernelGGL((opt_update_kernel<0, 5>), grid, block, 0, s, table, cm, first_tensor, gp, hyper, st);
        else hipLaunchKernelGGL((opt_update_kernel<0, 6>), grid, block, 0, s, table, cm, first_tensor, gp, hyper, st);
    } else {
        hipLaunchKernelGGL((opt_update_kernel<1, 0>), grid, block, 0, s, table, cm, first_tensor, gp, hyper, st);
    }
    YMI_CHECK_LAUNCH("opt_update");
    return YMI_OK;
}
