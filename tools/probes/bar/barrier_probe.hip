// what a workgroup barrier costs on MI355X: cycles per iteration of a loop that holds nothing but s_barrier (+ optional MFMAs / VALU),
// by workgroup size and workgroups per CU.  hipcc --offload-arch=gfx950 -O3 barrier_probe.hip -o barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int MFMAS, int VALU, int BARS>
__global__ void k(unsigned long long* out, int iters, float* sink) {
    extern __shared__ char smem[];
    f32x4 acc[4] = {};
    bf16x8 a = {}, b = {};
    float v = threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < MFMAS; ++q) acc[q & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[q & 3], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < VALU; ++q) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v));
#pragma unroll
        for (int q = 0; q < BARS; ++q) asm volatile("s_barrier" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (v == 12345.f) sink[0] = acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0];
}
template <int M, int V, int B>
void run(const char* name, int threads, int wg_per_cu, size_t lds) {
    const int cus = 256, iters = 2000, grid = cus * wg_per_cu;
    unsigned long long* d; float* s;
    hipMalloc(&d, grid * 8); hipMalloc(&s, 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<M, V, B>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<M, V, B>), dim3(grid), dim3(threads), lds, 0, d, iters, s);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (auto x : h) sum += (double)x;
    printf("%-44s threads %4d  wg/CU %d : %7.1f cycles per iteration\n", name, threads, wg_per_cu, sum / grid / iters);
    hipFree(d); hipFree(s);
}
int main() {
    // LDS sizes force the residency: 160 KB / n
    for (int th : {256, 512}) {
        for (int n : {1, 2, 3, 4}) {
            const size_t lds = (160 * 1024 / n) - 2048;
            if (th == 512 && n > 2) continue;
            run<0, 0, 1>("1 barrier", th, n, lds);
            run<0, 0, 2>("2 barriers", th, n, lds);
            run<16, 0, 0>("16 MFMA, no barrier", th, n, lds);
            run<16, 0, 1>("16 MFMA + 1 barrier", th, n, lds);
            run<16, 0, 2>("16 MFMA + 2 barriers", th, n, lds);
            run<16, 40, 1>("16 MFMA + 40 VALU + 1 barrier", th, n, lds);
            run<0, 40, 1>("40 VALU + 1 barrier", th, n, lds);
        }
    }
    return 0;
}
