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

// ---- development options (common.h: YmiOpt) ----------------------------------------------------------------------------------------
// Grid sizing of the streaming BatchNorm passes (ew_ppt / ew_cap), swept INSIDE the training step in round 4
// (profiles/r04_ew_grid_sweep.txt; the round-1 values 8 / 2048 came from isolated launches): one resident round of four 256-thread
// workgroups per CU, each thread walking up to 32 pixels, is 0.2 ms/step faster than eight per CU - fewer, longer workgroups amortise
// the coefficient loads and the launch ramp, and the half-empty SIMDs do not matter to passes that wait on HBM.  Round 5, swept again in the
// step after the finalize launches had moved into these passes' prologues (every thread now reads the statistics' replicas first): two
// workgroups per CU, up to 64 pixels per thread: 11.641 -> 11.607 ms/step (profiles/r05_ew_grid_sweep.txt).
struct OptEntry {
    const char* name;
    int value;
};
static OptEntry g_opts[OPT_COUNT] = {
    {"ew_ppt", 64}, {"ew_cap", 512}, {"red_cap", 512}, {"xcd_shift", 0}, {"attn_tiled", 0}, {"wgrad_blocks", 1280}, {"wgrad_blocks128", 768},
    {"igemm_tile_bm", 0}, {"igemm_tile_bn", 0}, {"bn_tail", 0}, {"wgrad_patch", 1},
};
static void opts_from_env() {
    for (int i = 0; i < OPT_COUNT; ++i) {
        char env[64] = "YMI_";
        size_t n = 4;
        for (const char* c = g_opts[i].name; *c && n + 1 < sizeof(env); ++c) env[n++] = (char)((*c >= 'a' && *c <= 'z') ? *c - 32 : *c);
        env[n] = 0;
        if (const char* v = getenv(env)) g_opts[i].value = atoi(v);
    }
    if (const char* t = getenv("YMI_IGEMM_TILE")) {  // "bm,bn"
        int bm = 0, bn = 0;
        if (sscanf(t, "%d,%d", &bm, &bn) == 2) {
            g_opts[OPT_IGEMM_TILE_BM].value = bm;
            g_opts[OPT_IGEMM_TILE_BN].value = bn;
        }
    }
}
int ymi_opt(int id) {
    static const bool init = (opts_from_env(), true);
    (void)init;
    return g_opts[id].value;
}
extern "C" int ymi_set_option(const char* name, int64_t value) {
    (void)ymi_opt(0);
    for (int i = 0; name && i < OPT_COUNT; ++i)
        if (!strcmp(name, g_opts[i].name)) {
            g_opts[i].value = (int)value;
            return YMI_OK;
        }
    ymi_set_error("set_option: unknown option '%s'", name ? name : "(null)");
    return YMI_EINVAL;
}
extern "C" int64_t ymi_get_option(const char* name) {
    (void)ymi_opt(0);
    for (int i = 0; name && i < OPT_COUNT; ++i)
        if (!strcmp(name, g_opts[i].name)) return g_opts[i].value;
    ymi_set_error("get_option: unknown option '%s'", name ? name : "(null)");
    return -1;
}

// the span argument of the kernels that walk XCD-owned pixel ranges (common.h).  Option xcd_shift = k (diagnostic, default 0) makes those
// kernels work on the range of XCD (x + k) % 8 instead of their own - the anti-affine arrangement a same-box A/B measures against.
int64_t ymi_xcd_span_arg(int64_t P) { return ymi_xcd_span(P) | ((int64_t)(ymi_opt(OPT_XCD_SHIFT) & 7) << 56); }

// Ticket counters of the in-launch hand-offs (common.h): 1024 slots of 64 counters, zero when the code object loads and zero again after
// every launch that used one (the workgroup that draws the last ticket resets it).  A launch takes the next slot; two launches share a
// slot only 1024 ticketed launches apart (a training step has < 300), so kernels that may be in flight together never do.
__device__ unsigned ymi_ticket_storage[1024 * 64];
unsigned* ymi_ticket_slot() {
    static thread_local int cached_dev = -1;
    static thread_local unsigned* base = nullptr;
    static unsigned counter = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    if (dev != cached_dev) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(ymi_ticket_storage)) != hipSuccess) return nullptr;
        base = (unsigned*)p;
        cached_dev = dev;
    }
    return base + (size_t)(__atomic_fetch_add(&counter, 1u, __ATOMIC_RELAXED) % 1024u) * 64;
}
