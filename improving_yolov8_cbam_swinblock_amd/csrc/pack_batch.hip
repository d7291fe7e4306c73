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

// One thread per (output channel, padded input channel) pair for the forward operand and per (input channel, padded
// output channel) pair for the data-gradient operand; the thread walks the kh*kw taps, which lie CONTIGUOUSLY in the
// OIHW source (one 36-byte run for a 3x3 kernel), so every source byte is fetched once, and for a fixed tap
// consecutive lanes write consecutive destination elements.  A tensor owns ceil(max(O*ipad, I*opad) / 256) workgroups.
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
    const uint32_t idx = (uint32_t)((int)blockIdx.x - block_start[lo]) * 256u + threadIdx.x;
    const uint32_t taps = (uint32_t)(d.kh * d.kw), I = (uint32_t)d.i, O = (uint32_t)d.o;
    if (d.dst_fwd && idx < O * (uint32_t)d.ipad) {  // dst[o][tap][ip] <- src[o][i][tap]
        const uint32_t o = idx / (uint32_t)d.ipad, ip = idx - o * (uint32_t)d.ipad;
        const float* src = d.src + ((uint64_t)o * I + ip) * taps;
        T* dst = reinterpret_cast<T*>(d.dst_fwd) + (uint64_t)o * taps * d.ipad + ip;
        for (uint32_t t = 0; t < taps; ++t) dst[(uint64_t)t * d.ipad] = from_f32<T>(ip < I ? src[t] : 0.0f);
    }
    if (d.dst_dgrad && idx < I * (uint32_t)d.opad) {  // class blocks back to back: dst[ci][tap in class][op] <- src[o][ci][kh][kw]
        const uint32_t ci = idx / (uint32_t)d.opad, o = idx - ci * (uint32_t)d.opad;
        const float* src = d.src + ((uint64_t)o * I + ci) * taps;
        T* dst = reinterpret_cast<T*>(d.dst_dgrad);
        const int pad = d.kh / 2, smask = d.stride - 1;  // stride 1 or 2: (v % stride == 0) <=> ((v & smask) == 0)
        const int nclass = d.stride == 1 ? 1 : 4;
        uint64_t class_off = 0;
        for (int cls = 0; cls < nclass; ++cls) {
            const int ph = d.stride == 1 ? 0 : cls >> 1, pw = d.stride == 1 ? 0 : cls & 1;
            int nt = 0;
            for (int a = 0; a < d.kh; ++a)
                for (int b = 0; b < d.kw; ++b)
                    if (((ph + pad - a) & smask) == 0 && ((pw + pad - b) & smask) == 0) ++nt;
            int tq = 0;
            for (int a = 0; a < d.kh; ++a)
                for (int b = 0; b < d.kw; ++b)
                    if (((ph + pad - a) & smask) == 0 && ((pw + pad - b) & smask) == 0) {
                        const float v = o < O ? src[a * d.kw + b] : 0.0f;
                        dst[class_off + ((uint64_t)ci * nt + tq) * d.opad + o] = from_f32<T>(v);
                        ++tq;
                    }
            class_off += (uint64_t)I * nt * d.opad;
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
