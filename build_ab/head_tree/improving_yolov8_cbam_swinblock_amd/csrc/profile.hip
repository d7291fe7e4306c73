// Optional per-launch timing of the MFMA GEMM kernels with HIP events recorded on the launch stream.
// Used by bench.py to measure the dominant kernel's achieved FLOP/s live over the timed region.
// Off by default: when disabled the launch path does not touch this file's state beyond one flag read.
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "common.h"

namespace {
struct Rec {
    hipEvent_t e0, e1;
    int family;
    double flop, bytes, bound_ms;
};
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
size_t g_next = 0;
}  // namespace

bool ymi_prof_enabled() { return g_on; }

// returns an index to pass to ymi_prof_stop, or -1
// flop / bytes: the launch's algorithmic work and HBM traffic (operands read once, result written once);
// peak_tflops: dense MFMA peak of the launch's dtype.  bound_ms = the launch's own roofline, max of the two.
int ymi_prof_start(hipStream_t stream, int family, double flop, double bytes, double peak_tflops) {
    if (!g_on || g_next + 2 > g_pool.size()) return -1;
    const double t_mfma = flop / (peak_tflops * 1e12) * 1e3, t_hbm = bytes / 8.0e12 * 1e3;
    Rec r{g_pool[g_next], g_pool[g_next + 1], family, flop, bytes, t_mfma > t_hbm ? t_mfma : t_hbm};
    g_next += 2;
    (void)hipEventRecord(r.e0, stream);
    g_recs.push_back(r);
    return (int)g_recs.size() - 1;
}
void ymi_prof_stop(hipStream_t stream, int idx) {
    if (idx >= 0) (void)hipEventRecord(g_recs[idx].e1, stream);
}

extern "C" int ymi_profile_begin(int64_t capacity) {
    YMI_CHECK_ARG(capacity > 0 && capacity <= (1 << 20), "profile_begin: capacity");
    while ((int64_t)g_pool.size() < 2 * capacity) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) {
            ymi_set_error("profile_begin: hipEventCreate failed");
            return YMI_ELAUNCH;
        }
        g_pool.push_back(e);
    }
    g_recs.clear();
    g_recs.reserve(capacity);
    g_next = 0;
    g_on = true;
    return YMI_OK;
}

// family 0: implicit-GEMM conv (forward, data gradient, token GEMMs); family 1: weight-gradient GEMM
extern "C" int ymi_profile_end_ex(double* ms_by_family, double* flop_by_family, int64_t* launches_by_family, double* bytes_by_family,
                                  double* bound_ms_by_family) {
    g_on = false;
    YMI_CHECK_ARG(ms_by_family && flop_by_family && launches_by_family, "profile_end: null");
    for (int f = 0; f < 2; ++f) {
        ms_by_family[f] = 0.0;
        flop_by_family[f] = 0.0;
        launches_by_family[f] = 0;
        if (bytes_by_family) bytes_by_family[f] = 0.0;
        if (bound_ms_by_family) bound_ms_by_family[f] = 0.0;
    }
    if (hipDeviceSynchronize() != hipSuccess) {
        ymi_set_error("profile_end: device synchronize failed");
        return YMI_ELAUNCH;
    }
    // development aid: YMI_PROF_DUMP=<file> appends one line per recorded launch (family, measured us, its own bound in us, GFLOP, MB)
    FILE* dump = getenv("YMI_PROF_DUMP") ? fopen(getenv("YMI_PROF_DUMP"), "a") : nullptr;
    for (const Rec& r : g_recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
        if (dump) fprintf(dump, "%d %.2f %.2f %.3f %.2f\n", r.family, ms * 1e3, r.bound_ms * 1e3, r.flop * 1e-9, r.bytes * 1e-6);
        ms_by_family[r.family] += ms;
        flop_by_family[r.family] += r.flop;
        launches_by_family[r.family] += 1;
        if (bytes_by_family) bytes_by_family[r.family] += r.bytes;
        if (bound_ms_by_family) bound_ms_by_family[r.family] += r.bound_ms;
    }
    if (dump) fclose(dump);
    g_recs.clear();
    g_next = 0;
    return YMI_OK;
}

extern "C" int ymi_profile_end(double* ms_by_family, double* flop_by_family, int64_t* launches_by_family) {
    return ymi_profile_end_ex(ms_by_family, flop_by_family, launches_by_family, nullptr, nullptr);
}
