// Does a consumer kernel find its producer's output in its XCD's L2 when both give an XCD the same byte range?
// W writes a buffer, R reads it back; block b (dealt to XCD b % 8) owns chunk (b % 8) * (nb / 8) + b / 8 in the "affine" form,
// the next XCD's range in the "shifted" form.  Time of R by HIP events, median over rounds.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/probes/xcd_affinity_probe.bin tools/probes/xcd_affinity_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int chunk_of(int b, int nb, int shift) {
    const int x = ((b & 7) + shift) & 7;
    return x * (nb >> 3) + (b >> 3);
}

__global__ __launch_bounds__(256) void wkernel(u32x4* buf, size_t units_per_chunk, int shift, unsigned seed) {
    const int c = chunk_of(blockIdx.x, gridDim.x, shift);
    u32x4* p = buf + (size_t)c * units_per_chunk;
    for (size_t i = threadIdx.x; i < units_per_chunk; i += 256) p[i] = u32x4{seed + (unsigned)i, seed, (unsigned)c, 1u};
}

__global__ __launch_bounds__(256) void rkernel(const u32x4* buf, size_t units_per_chunk, int shift, unsigned* out, unsigned* xcc) {
    const int c = chunk_of(blockIdx.x, gridDim.x, shift);
    const u32x4* p = buf + (size_t)c * units_per_chunk;
    unsigned acc = 0;
    for (size_t i = threadIdx.x; i < units_per_chunk; i += 256 * 4) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = i + u * 256 < units_per_chunk ? p[i + u * 256] : u32x4{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u].x + v[u].w;
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
    if (xcc && threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[blockIdx.x] = id & 15;
    }
}

// latency-sensitive reader, shaped like a GEMM's operand ring: 2 workgroups per CU, each thread keeps only DEPTH 16-byte loads in flight
template <int DEPTH>
__global__ __launch_bounds__(256) void rlat_kernel(const u32x4* buf, size_t units_per_chunk, int shift, unsigned* out) {
    const int c = chunk_of(blockIdx.x, gridDim.x, shift);
    const u32x4* p = buf + (size_t)c * units_per_chunk;
    unsigned acc = 0;
    for (size_t i = threadIdx.x; i + (DEPTH - 1) * 256 < units_per_chunk; i += 256 * DEPTH) {
        u32x4 v[DEPTH];
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) v[u] = p[i + u * 256];
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) acc += v[u].x + v[u].w;
        asm volatile("" : "+v"(acc));  // the next round's addresses do not depend on it, but the loads are not hoisted across
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

// the same reader through LDS-DMA (global_load_lds, 16 B per lane), as the GEMM kernels stage their operands
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
template <int DEPTH>
__global__ __launch_bounds__(256) void rdma_kernel(const u32x4* buf, size_t units_per_chunk, int shift, unsigned* out) {
    __shared__ u32x4 ring[DEPTH * 256];
    const int c = chunk_of(blockIdx.x, gridDim.x, shift);
    const u32x4* p = buf + (size_t)c * units_per_chunk;
    const int wave = threadIdx.x >> 6;
    unsigned acc = 0;
    for (size_t i = threadIdx.x; i + (DEPTH - 1) * 256 < units_per_chunk; i += 256 * DEPTH) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) __builtin_amdgcn_global_load_lds((gptr_t)(p + i + u * 256), (lptr_t)(ring + u * 256 + wave * 64), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) acc += ring[u * 256 + threadIdx.x].x;
        asm volatile("" : "+v"(acc));
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

int main() {
    const int nb = 2048;
    unsigned *out, *xcc;
    hipMalloc(&out, nb * 4);
    hipMalloc(&xcc, nb * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    // is block -> XCD stable between launches?
    {
        u32x4* buf;
        hipMalloc(&buf, 1 << 20);
        std::vector<unsigned> a(nb), b(nb);
        rkernel<<<nb, 256>>>(buf, 1, 0, out, xcc);
        hipMemcpy(a.data(), xcc, nb * 4, hipMemcpyDeviceToHost);
        wkernel<<<nb, 256>>>(buf, 1, 0, 1);
        rkernel<<<nb, 256>>>(buf, 1, 0, out, xcc);
        hipMemcpy(b.data(), xcc, nb * 4, hipMemcpyDeviceToHost);
        int same = 0, rr = 0;
        for (int i = 0; i < nb; ++i) {
            same += a[i] == b[i];
            rr += ((a[i] - a[0]) & 7) == (unsigned)(i & 7);
        }
        printf("block->XCC: %d / %d equal between two launches; %d / %d follow (xcc0 + b) %% 8; xcc of block 0: %u then %u\n", same, nb, rr, nb, a[0], b[0]);
        hipFree(buf);
    }
    printf("%8s %12s %12s %12s %12s   (us of the READ kernel, median of 15; W -> R back to back)\n", "MB", "affine", "shifted", "aff GB/s", "shf GB/s");
    for (size_t mb : {2, 4, 8, 13, 16, 26, 32, 52, 64, 105, 128, 256}) {
        const size_t bytes = mb << 20, upc = bytes / 16 / nb;
        u32x4* buf;
        hipMalloc(&buf, bytes);
        double med[2];
        for (int form = 0; form < 2; ++form) {
            std::vector<float> t;
            for (int it = 0; it < 18; ++it) {
                wkernel<<<nb, 256>>>(buf, upc, 0, it);
                hipEventRecord(e0);
                rkernel<<<nb, 256>>>(buf, upc, form, out, nullptr);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (it >= 3) t.push_back(ms * 1e3f);
            }
            std::sort(t.begin(), t.end());
            med[form] = t[t.size() / 2];
        }
        printf("%8zu %12.1f %12.1f %12.0f %12.0f\n", mb, med[0], med[1], bytes / med[0] * 1e-3, bytes / med[1] * 1e-3);
        hipFree(buf);
    }
    // a third kernel in between (another 64 MB streamed): does the affinity survive one unrelated pass?
    printf("with an unrelated 64 MB write between W and R:\n");
    u32x4* other;
    hipMalloc(&other, 64 << 20);
    for (size_t mb : {8, 13, 26, 52}) {
        const size_t bytes = mb << 20, upc = bytes / 16 / nb;
        u32x4* buf;
        hipMalloc(&buf, bytes);
        double med[2];
        for (int form = 0; form < 2; ++form) {
            std::vector<float> t;
            for (int it = 0; it < 18; ++it) {
                wkernel<<<nb, 256>>>(buf, upc, 0, it);
                wkernel<<<nb, 256>>>(other, (64 << 20) / 16 / nb, 3, it);
                hipEventRecord(e0);
                rkernel<<<nb, 256>>>(buf, upc, form, out, nullptr);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (it >= 3) t.push_back(ms * 1e3f);
            }
            std::sort(t.begin(), t.end());
            med[form] = t[t.size() / 2];
        }
        printf("%8zu %12.1f %12.1f %12.0f %12.0f\n", mb, med[0], med[1], bytes / med[0] * 1e-3, bytes / med[1] * 1e-3);
        hipFree(buf);
    }
    printf("latency-shaped reader (512 workgroups, 2 x 16 B in flight per thread), W by 2048 blocks with chunk = 4 consecutive reader... same eighths:\n");
    printf("%8s %12s %12s %12s   (us, median of 15)\n", "MB", "affine", "shifted", "repeat(warm)");
    for (size_t mb : {4, 8, 13, 16, 26, 32, 52}) {
        const size_t bytes = mb << 20;
        const int nbr = 512;
        const size_t upc_w = bytes / 16 / nb, upc_r = bytes / 16 / nbr;
        u32x4* buf;
        hipMalloc(&buf, bytes);
        double med[3];
        for (int form = 0; form < 3; ++form) {
            std::vector<float> t;
            for (int it = 0; it < 18; ++it) {
                if (form < 2) wkernel<<<nb, 256>>>(buf, upc_w, 0, it);
                else rlat_kernel<2><<<nbr, 256>>>(buf, upc_r, 0, out);
                hipEventRecord(e0);
                rlat_kernel<2><<<nbr, 256>>>(buf, upc_r, form == 1 ? 1 : 0, out);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (it >= 3) t.push_back(ms * 1e3f);
            }
            std::sort(t.begin(), t.end());
            med[form] = t[t.size() / 2];
        }
        printf("%8zu %12.1f %12.1f %12.1f\n", mb, med[0], med[1], med[2]);
        hipFree(buf);
    }
    printf("chains of 40 (W, R) pairs timed as a whole, per pair (us): what a replayed graph pays; reader = latency-shaped, 512 workgroups\n");
    printf("%8s %12s %12s %12s %12s\n", "MB", "affine", "shifted", "W only", "R warm only");
    for (size_t mb : {2, 4, 8, 13, 16, 26, 32, 52}) {
        const size_t bytes = mb << 20;
        const int nbr = 512, K = 40;
        const size_t upc_w = bytes / 16 / nb, upc_r = bytes / 16 / nbr;
        u32x4* buf;
        hipMalloc(&buf, bytes);
        double med[4];
        for (int form = 0; form < 4; ++form) {
            std::vector<float> t;
            for (int it = 0; it < 9; ++it) {
                hipEventRecord(e0);
                for (int k = 0; k < K; ++k) {
                    if (form != 3) wkernel<<<nb, 256>>>(buf, upc_w, 0, it);
                    if (form != 2) rlat_kernel<2><<<nbr, 256>>>(buf, upc_r, form == 1 ? 1 : 0, out);
                }
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (it >= 2) t.push_back(ms * 1e3f / K);
            }
            std::sort(t.begin(), t.end());
            med[form] = t[t.size() / 2];
        }
        printf("%8zu %12.2f %12.2f %12.2f %12.2f\n", mb, med[0], med[1], med[2], med[3]);
        hipFree(buf);
    }
    printf("chains with the LDS-DMA reader (per pair, us):\n");
    printf("%8s %12s %12s\n", "MB", "affine", "shifted");
    for (size_t mb : {4, 8, 13, 26, 52}) {
        const size_t bytes = mb << 20;
        const int nbr = 512, K = 40;
        const size_t upc_w = bytes / 16 / nb, upc_r = bytes / 16 / nbr;
        u32x4* buf;
        hipMalloc(&buf, bytes);
        double med[2];
        for (int form = 0; form < 2; ++form) {
            std::vector<float> t;
            for (int it = 0; it < 9; ++it) {
                hipEventRecord(e0);
                for (int k = 0; k < K; ++k) {
                    wkernel<<<nb, 256>>>(buf, upc_w, 0, it);
                    rdma_kernel<2><<<nbr, 256>>>(buf, upc_r, form, out);
                }
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (it >= 2) t.push_back(ms * 1e3f / K);
            }
            std::sort(t.begin(), t.end());
            med[form] = t[t.size() / 2];
        }
        printf("%8zu %12.2f %12.2f\n", mb, med[0], med[1]);
        hipFree(buf);
    }
    printf("the same chains captured into a hipGraph and replayed (per pair, us):\n");
    printf("%8s %12s %12s\n", "MB", "affine", "shifted");
    hipStream_t st;
    hipStreamCreate(&st);
    for (size_t mb : {4, 8, 13, 26}) {
        const size_t bytes = mb << 20;
        const int nbr = 512, K = 40;
        const size_t upc_w = bytes / 16 / nb, upc_r = bytes / 16 / nbr;
        u32x4* buf;
        hipMalloc(&buf, bytes);
        double med[2];
        for (int form = 0; form < 2; ++form) {
            hipGraph_t g;
            hipGraphExec_t ge;
            hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
            for (int k = 0; k < K; ++k) {
                wkernel<<<nb, 256, 0, st>>>(buf, upc_w, 0, k);
                rlat_kernel<2><<<nbr, 256, 0, st>>>(buf, upc_r, form, out);
            }
            hipStreamEndCapture(st, &g);
            hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            std::vector<float> t;
            for (int it = 0; it < 9; ++it) {
                hipEventRecord(e0, st);
                hipGraphLaunch(ge, st);
                hipEventRecord(e1, st);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (it >= 2) t.push_back(ms * 1e3f / K);
            }
            std::sort(t.begin(), t.end());
            med[form] = t[t.size() / 2];
            hipGraphExecDestroy(ge);
            hipGraphDestroy(g);
        }
        printf("%8zu %12.2f %12.2f\n", mb, med[0], med[1]);
        hipFree(buf);
    }
    return 0;
}
