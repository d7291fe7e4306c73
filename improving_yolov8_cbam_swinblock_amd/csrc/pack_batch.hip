// One launch packs EVERY conv / linear weight of a model into its kernel operand layouts (forward [O][tap][Ipad]
// and data-gradient [I][tap][Opad] per stride-parity class), driven by a descriptor table in device memory.
// Replaces ~160 tiny per-layer pack launches per training step.
#include "common.h"

struct PackDesc {  // mirrors ymi_pack_desc
    const float* src;
    void* dst_fwd;
    void* dst_dgrad;
    int32_t o, i, kh, kw, ipad, opad, stride, pad_;
};

template <typename T>
__global__ __launch_bounds__(256) void pack_batch_kernel(const PackDesc* __restrict__ descs, const int32_t* __restrict__ block_start, int count) {
    // binary search: tensor t with block_start[t] <= blockIdx.x < block_start[t+1]
    int lo = 0, hi = count;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (block_start[mid] <= (int)blockIdx.x) lo = mid;
        else hi = mid;
    }
    const PackDesc d = descs[lo];
    const int64_t base = ((int64_t)blockIdx.x - block_start[lo]) * 1024;
    const int taps = d.kh * d.kw;
    const int64_t nfwd = d.dst_fwd ? (int64_t)d.o * taps * d.ipad : 0;
    const int64_t ndg = d.dst_dgrad ? (int64_t)d.i * taps * d.opad : 0;
    const int pad = d.kh / 2;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t idx = base + u * 256 + threadIdx.x;
        if (idx < nfwd) {  // dst[o][kh][kw][ip] <- src[o][i][kh][kw]
            const int ip = (int)(idx % d.ipad);
            int64_t t = idx / d.ipad;
            const int tp = (int)(t % taps);
            const int o = (int)(t / taps);
            const float v = ip < d.i ? d.src[((int64_t)o * d.i + ip) * taps + tp] : 0.0f;
            reinterpret_cast<T*>(d.dst_fwd)[idx] = from_f32<T>(v);
        }
        if (idx < ndg) {  // class blocks back to back: dst[ci][t_in_class][op] <- src[o][ci][kh_t][kw_t]
            int64_t rem = idx;
            const int nclass = d.stride == 1 ? 1 : 4;
            for (int cls = 0; cls < nclass; ++cls) {
                const int ph = d.stride == 1 ? 0 : cls >> 1, pw = d.stride == 1 ? 0 : cls & 1;
                int nt = 0;
                for (int a = 0; a < d.kh; ++a)
                    for (int b = 0; b < d.kw; ++b)
                        if ((ph + pad - a) % d.stride == 0 && (pw + pad - b) % d.stride == 0) ++nt;
                const int64_t sz = (int64_t)d.i * nt * d.opad;
                if (rem < sz) {
                    const int o = (int)(rem % d.opad);
                    int64_t t = rem / d.opad;
                    const int tq = (int)(t % nt);
                    const int ci = (int)(t / nt);
                    int seen = 0, ka = 0, kb = 0;
                    for (int a = 0; a < d.kh; ++a)
                        for (int b = 0; b < d.kw; ++b)
                            if ((ph + pad - a) % d.stride == 0 && (pw + pad - b) % d.stride == 0) {
                                if (seen == tq) { ka = a; kb = b; }
                                ++seen;
                            }
                    const float v = o < d.o ? d.src[(((int64_t)o * d.i + ci) * d.kh + ka) * d.kw + kb] : 0.0f;
                    reinterpret_cast<T*>(d.dst_dgrad)[idx] = from_f32<T>(v);
                    break;
                }
                rem -= sz;
            }
        }
    }
}

extern "C" int ymi_pack_conv_weights_batch(const void* descs_device, const int32_t* block_start_device, int32_t count, int32_t total_blocks,
                                           int32_t dtype, void* stream) {
    YMI_CHECK_ARG(descs_device && block_start_device && count > 0 && total_blocks > 0, "pack_conv_weights_batch: args");
    static_assert(sizeof(PackDesc) == 56, "ymi_pack_desc layout");
    if (dtype == YMI_BF16)
        hipLaunchKernelGGL(pack_batch_kernel<bf16_t>, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const PackDesc*>(descs_device), block_start_device, (int)count);
    else
        hipLaunchKernelGGL(pack_batch_kernel<float>, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const PackDesc*>(descs_device), block_start_device, (int)count);
    YMI_CHECK_LAUNCH("pack_conv_weights_batch");
    return YMI_OK;
}
