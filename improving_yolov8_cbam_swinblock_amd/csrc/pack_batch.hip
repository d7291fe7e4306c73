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
    // 32-bit index arithmetic throughout (the host checks every operand has < 2^31 elements): 64-bit integer
    // division costs an order of magnitude more instructions and this kernel is nothing but index arithmetic
    const uint32_t base = (uint32_t)((int)blockIdx.x - block_start[lo]) * 1024u;
    const uint32_t taps = (uint32_t)(d.kh * d.kw);
    const uint32_t nfwd = d.dst_fwd ? (uint32_t)d.o * taps * (uint32_t)d.ipad : 0u;
    const uint32_t ndg = d.dst_dgrad ? (uint32_t)d.i * taps * (uint32_t)d.opad : 0u;
    const int pad = d.kh / 2;
    const int smask = d.stride - 1;  // stride is 1 or 2: (v % stride == 0) <=> ((v & smask) == 0), without a division
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t idx = base + u * 256 + threadIdx.x;
        if (idx < nfwd) {  // dst[o][kh][kw][ip] <- src[o][i][kh][kw]
            const uint32_t t = idx / (uint32_t)d.ipad;
            const uint32_t ip = idx - t * (uint32_t)d.ipad;
            const uint32_t o = t / taps;
            const uint32_t tp = t - o * taps;
            const float v = ip < (uint32_t)d.i ? d.src[(o * (uint32_t)d.i + ip) * taps + tp] : 0.0f;
            reinterpret_cast<T*>(d.dst_fwd)[idx] = from_f32<T>(v);
        }
        if (idx < ndg) {  // class blocks back to back: dst[ci][t_in_class][op] <- src[o][ci][kh_t][kw_t]
            uint32_t rem = idx;
            const int nclass = d.stride == 1 ? 1 : 4;
            for (int cls = 0; cls < nclass; ++cls) {
                const int ph = d.stride == 1 ? 0 : cls >> 1, pw = d.stride == 1 ? 0 : cls & 1;
                int nt = 0;
                for (int a = 0; a < d.kh; ++a)
                    for (int b = 0; b < d.kw; ++b)
                        if (((ph + pad - a) & smask) == 0 && ((pw + pad - b) & smask) == 0) ++nt;
                const uint32_t sz = (uint32_t)d.i * (uint32_t)nt * (uint32_t)d.opad;
                if (rem < sz) {
                    const uint32_t t = rem / (uint32_t)d.opad;
                    const int o = (int)(rem - t * (uint32_t)d.opad);
                    const uint32_t ci = t / (uint32_t)nt;
                    const int tq = (int)(t - ci * (uint32_t)nt);
                    int seen = 0, ka = 0, kb = 0;
                    for (int a = 0; a < d.kh; ++a)
                        for (int b = 0; b < d.kw; ++b)
                            if (((ph + pad - a) & smask) == 0 && ((pw + pad - b) & smask) == 0) {
                                if (seen == tq) { ka = a; kb = b; }
                                ++seen;
                            }
                    const float v = o < d.o ? d.src[(((uint32_t)o * (uint32_t)d.i + ci) * (uint32_t)d.kh + (uint32_t)ka) * (uint32_t)d.kw + (uint32_t)kb] : 0.0f;
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
