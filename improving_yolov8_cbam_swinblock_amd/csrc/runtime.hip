// Library-level plumbing: version, thread-local error text, the device zero page.
#include <stdarg.h>
#include <string.h>

#include <stdlib.h>

#include "common.h"

static thread_local char g_err[512] = "";

void ymi_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int ymi_version(void) { return YMI_VERSION; }
extern "C" const char* ymi_last_error(void) { return g_err; }

// 256 zero bytes in device memory; __device__ globals are zero-initialised when the code object loads.
__device__ __attribute__((aligned(256))) unsigned char ymi_zero_page_storage[256];

const void* ymi_zero_page() {
    // one address per device; resolved once per process per device (no allocation, no sync afterwards)
    static thread_local int cached_dev = -1;
    static thread_local void* cached_ptr = nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    if (dev != cached_dev) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(ymi_zero_page_storage)) != hipSuccess) return nullptr;
        cached_ptr = p;
        cached_dev = dev;
    }
    return cached_ptr;
}

// Grid sizing of the streaming BatchNorm passes (affine + SiLU, backward apply), swept INSIDE the training step in round 4
// (profiles/r04_ew_grid_sweep.txt; the round-1 values 8 / 2048 came from isolated launches): one resident round of four 256-thread
// workgroups per CU, each thread walking up to 32 pixels, is 0.2 ms/step faster than eight per CU - fewer, longer workgroups amortise
// the coefficient loads and the launch ramp, and the half-empty SIMDs do not matter to passes that wait on HBM.
int ew_ppt() { static const int v = getenv("YMI_EW_PPT") ? atoi(getenv("YMI_EW_PPT")) : 32; return v; }    // pixels per thread the passes aim for
int ew_cap() { static const int v = getenv("YMI_EW_CAP") ? atoi(getenv("YMI_EW_CAP")) : 1024; return v; }  // their workgroup cap

// the span argument of the kernels that walk XCD-owned pixel ranges (common.h).  YMI_XCD_SHIFT=k (diagnostic knob, default 0) makes those
// kernels work on the range of XCD (x + k) % 8 instead of their own - the anti-affine arrangement a same-box A/B measures against.
int64_t ymi_xcd_span_arg(int64_t P) {
    static const int shift = getenv("YMI_XCD_SHIFT") ? atoi(getenv("YMI_XCD_SHIFT")) & 7 : 0;
    return ymi_xcd_span(P) | ((int64_t)shift << 56);
}
