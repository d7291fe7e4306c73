// does an out-of-range lane of `buffer_load_dwordx4 ... lds` write zeros to LDS (as an out-of-range buffer load returns zeros to registers)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void k(const char* src, unsigned n, int soff, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = -7.f;  // garbage
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, n, 0x00020000);
    unsigned voff = threadIdx.x * 16;
    if (threadIdx.x & 1) voff = 0xFFFFFFF0u;       // far out of range
    if ((threadIdx.x & 7) == 2) voff = n - 8;       // straddles the end: 8 valid bytes, 8 beyond
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)smem, 16, voff, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += blockDim.x) out[i] = reinterpret_cast<float*>(smem)[i];
}
int main() {
    const unsigned n = 4096;
    std::vector<float> h(n / 4 + 64);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)i;
    char* d; float* o;
    (void)hipMalloc(&d, h.size() * 4); (void)hipMalloc(&o, 1024);
    (void)hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int soff : {0, 64}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, n, soff, o);
        std::vector<float> r(256);
        (void)hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
        printf("soffset %d: lanes 0..7, four floats each:\n", soff);
        for (int l = 0; l < 8; ++l) printf("  lane %d: %g %g %g %g\n", l, r[l * 4], r[l * 4 + 1], r[l * 4 + 2], r[l * 4 + 3]);
    }
    return 0;
}
