// One launch packs EVERY conv / linear weight of a model into its kernel operand layouts (forward [O][tap][Ipad]
// and data-gradient [I][tap][Opad] per stride-parity class), driven by a descriptor table in device memory.
// Replaces ~160 tiny per-layer pack launches per training step.
#include "common.h"

struct PackDesc {  // mirrors ymi_pack_desc
    const float* src;
    void* dst_fwd;
    void* dst_dgrad;
    int32_t o, i, kh, kw, ipad, opad, stride;
    int32_t ostride;  // row length of the data-gradient operand (0: opad).  Larger than opad when several weights share one operand:
    int32_t o_off;    // ... this weight's first column in it (Detect's sibling convolutions, head.py:71-72, run as ONE convolution)
    int32_t pad_;
};

// Both operands are re-orderings of the OIHW source ([O][I][taps]): forward [O][tap][Ipad], data gradient [I][tap in class][Opad] - the
// second a transpose.  A workgroup owns a tile of 32 output x 32 input channels: it reads the tile ONCE with lanes along the input channel
// (runs of 32 * taps floats: 1152 B for a 3x3 kernel; one thread per (ci, o) pair with lanes along o read 36-byte runs 4.6 KB apart, and
// the source was read once per operand: 81 -> 42 us), keeps it in LDS as [tap][ci][o], and writes the forward operand with lanes
// along ci and the data-gradient operand with lanes along o (64-byte runs both).  Channels beyond the real counts are written as zeros.
// A tensor owns ceil(max(O, opad) / 32) * ceil(max(I, ipad) / 32) workgroups.
constexpr int PK_T = 32;  // tile edge

template <typename T>
__global__ __launch_bounds__(256) void pack_batch_kernel(const PackDesc* __restrict__ descs, const int32_t* __restrict__ block_start, int count) {
    __shared__ T tile[9 * PK_T * (PK_T + 2)];  // [tap][ci][o (+2: both transposed access patterns spread over the banks)]
    // binary search: tensor t with block_start[t] <= blockIdx.x < block_start[t+1]
    int lo = 0, hi = count;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (block_start[mid] <= (int)blockIdx.x) lo = mid;
        else hi = mid;
    }
    const PackDesc d = descs[lo];
    const uint32_t blk = (uint32_t)((int)blockIdx.x - block_start[lo]);
    const uint32_t taps = (uint32_t)(d.kh * d.kw), I = (uint32_t)d.i, O = (uint32_t)d.o;
    if (taps > 9) return;  // the LDS tile holds nine taps: a larger kernel would write past it (the host-side callers only build k <= 3; workgroup-uniform)
    const uint32_t omax = d.dst_dgrad && (uint32_t)d.opad > O ? (uint32_t)d.opad : O;  // (the launcher's tile count uses the padded input extent likewise)
    const uint32_t tiles_o = (omax + PK_T - 1) / PK_T;
    const uint32_t o0 = (blk % tiles_o) * PK_T, c0 = (blk / tiles_o) * PK_T;
    const uint32_t lane32 = threadIdx.x % PK_T, grp = threadIdx.x / PK_T;  // 8 groups of 32 lanes
    // read: lanes along ci
    for (uint32_t ol = grp; ol < PK_T; ol += 256 / PK_T) {
        const uint32_t o = o0 + ol, ci = c0 + lane32;
        const bool ok = o < O && ci < I;
        const float* src = d.src + ((uint64_t)o * I + ci) * taps;
        for (uint32_t t = 0; t < taps; ++t) tile[(t * PK_T + lane32) * (PK_T + 2) + ol] = from_f32<T>(ok ? src[t] : 0.0f);
    }
    __syncthreads();
    if (d.dst_fwd) {  // dst[o][tap][ip], lanes along ip
        T* dst = reinterpret_cast<T*>(d.dst_fwd);
        const uint32_t ip = c0 + lane32;
        if (ip < (uint32_t)d.ipad)
            for (uint32_t ol = grp; ol < PK_T && o0 + ol < O; ol += 256 / PK_T)
                for (uint32_t t = 0; t < taps; ++t) dst[((uint64_t)(o0 + ol) * taps + t) * d.ipad + ip] = tile[(t * PK_T + lane32) * (PK_T + 2) + ol];
    }
    if (d.dst_dgrad) {  // class blocks back to back: dst[ci][tap in class][op], lanes along op
        T* dst = reinterpret_cast<T*>(d.dst_dgrad);
        const int pad = d.kh / 2, smask = d.stride - 1;  // stride 1 or 2: (v % stride == 0) <=> ((v & smask) == 0)
        const int nclass = d.stride == 1 ? 1 : 4;
        const uint32_t o = o0 + lane32;
        const uint32_t ostride = d.ostride ? (uint32_t)d.ostride : (uint32_t)d.opad;
        uint64_t class_off = (uint64_t)d.o_off;
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
                        if (o < (uint32_t)d.opad)
                            for (uint32_t cl = grp; cl < PK_T && c0 + cl < I; cl += 256 / PK_T)
                                dst[class_off + ((uint64_t)(c0 + cl) * nt + tq) * ostride + o] = tile[((a * d.kw + b) * PK_T + cl) * (PK_T + 2) + lane32];
                        ++tq;
                    }
            class_off += (uint64_t)I * nt * ostride;
        }
    }
}

extern "C" int ymi_pack_conv_weights_batch(const void* descs_device, const int32_t* block_start_device, int32_t count, int32_t total_blocks,
                                           int32_t dtype, void* stream) {
    YMI_CHECK_ARG(descs_device && block_start_device && count > 0 && total_blocks > 0, "pack_conv_weights_batch: args");  // (kh * kw <= 9: the LDS tile)
    static_assert(sizeof(PackDesc) == 64 && sizeof(PackDesc) == sizeof(ymi_pack_desc), "ymi_pack_desc layout");
    if (dtype == YMI_BF16)
        hipLaunchKernelGGL(pack_batch_kernel<bf16_t>, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const PackDesc*>(descs_device), block_start_device, (int)count);
    else
        hipLaunchKernelGGL(pack_batch_kernel<float>, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const PackDesc*>(descs_device), block_start_device, (int)count);
    YMI_CHECK_LAUNCH("pack_conv_weights_batch");
    return YMI_OK;
}
