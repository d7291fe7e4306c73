// How many LDS-DMA pieces per 16 MFMAs can a CU sustain beside the MFMAs?  Every wave: P x global_load_lds (16 B per lane, 1 KB per instruction),
// counted vmcnt (the pieces just issued stay in flight), s_barrier, M MFMAs 16x16x32 on registers.  Source: a buffer that fits the L2s (8 MB) or
// does not (1 GB).  Reports cycles per iteration, MFMA-pipe share, and bytes per clock per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
template <int P, int M>
__global__ void k(unsigned long long* out, int iters, const char* src, size_t mask, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 acc[4] = {};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(blockIdx.x + i); }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    // a wave's pieces: consecutive 1 KB blocks, the workgroups spread over the buffer
    size_t off = ((size_t)blockIdx.x * 977 + wave * 131) * 4096 + lane * 16;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        char* dst = smem + ((i & 1) * nw + wave) * (P > 0 ? P : 1) * 1024;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            __builtin_amdgcn_global_load_lds((gptr_t)(src + (off & mask)), (lptr_t)(dst + p * 1024), 16, 0, 0);
            off += 1024 * 37;
        }
        if (P == 0) asm volatile("s_barrier" ::: "memory");
        else if (P == 1) asm volatile("s_waitcnt vmcnt(1)\n\ts_barrier" ::: "memory");
        else if (P == 2) asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
        else if (P == 3) asm volatile("s_waitcnt vmcnt(3)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int q = 0; q < M; ++q) acc[q & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[q & 3], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] == 12345.f) sink[0] = 1.f;
}
static char* g_src;
template <int P, int M>
void run(int threads, int wg_per_cu, size_t bytes) {
    const int cus = 256, iters = 1500, grid = cus * wg_per_cu;
    unsigned long long* d; float* s;
    (void)hipMalloc(&d, grid * 8); (void)hipMalloc(&s, 4);
    const size_t lds = (160 * 1024 / wg_per_cu) - 2048;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<P, M>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<P, M>), dim3(grid), dim3(threads), lds, 0, d, iters, g_src, bytes - 1, s);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid);
    (void)hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (auto x : h) sum += (double)x;
    const double cyc = sum / grid / iters;
    const int waves_per_simd = threads / 64 * wg_per_cu / 4;
    const double mfma = 16.0 * M * waves_per_simd;           // pipe cycles needed per iteration and SIMD (16 per 16x16x32)
    const double bpc = (double)P * 1024 * (threads / 64) * wg_per_cu / cyc;
    printf("pieces %d  MFMAs %2d  threads %4d  wg/CU %d  source %5zu MB : %7.1f cycles/iter   MFMA share %5.1f %%   %5.1f B/clk/CU\n", P, M, threads, wg_per_cu,
           bytes >> 20, cyc, 100.0 * mfma / cyc, bpc);
    (void)hipFree(d); (void)hipFree(s);
}
int main() {
    const size_t big = (size_t)1 << 30;
    (void)hipMalloc(&g_src, big);
    (void)hipMemset(g_src, 1, big);
    for (size_t bytes : {(size_t)8 << 20, big}) {
        for (int n : {2, 3, 4}) {
            run<0, 16>(256, n, bytes);
            run<1, 16>(256, n, bytes);
            run<2, 16>(256, n, bytes);
            run<3, 16>(256, n, bytes);
            run<4, 16>(256, n, bytes);
            run<3, 0>(256, n, bytes);
            run<4, 32>(256, n, bytes);
        }
        run<3, 16>(512, 2, bytes);
        run<2, 16>(512, 2, bytes);
        run<4, 32>(512, 1, bytes);
        run<4, 32>(512, 2, bytes);
    }
    return 0;
}
