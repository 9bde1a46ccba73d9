// Thread-local error message and version of libgat_mi355.so.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "common.h"

namespace gatmi {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Kernel-selection switches (A/B runs, parity tests of the fallback kernels): -1 = not set
// here -> the environment variable of the same name decides, read at every call.
static const char *const kOptNames[] = {"GATMI_STREAM", "GATMI_BWD2", "GATMI_KRES", "GATMI_KREG",
                                        "GATMI_KREG_TAIL", "GATMI_DW_STREAM", "GATMI_DW_TAIL", "GATMI_BWD3", "GATMI_FWD3",
                                        "GATMI_EDGE8", "GATMI_GROUPROWS",
                                        "GATMI_GEMM_V2", "GATMI_QGROUP", "GATMI_OUTGROUP",
                                        "GATMI_KREG_WIDE", "GATMI_SPLIT3", "GATMI_DW16_ROWS", "GATMI_ADST_FUSED", "GATMI_BWD_HOIST",
                                        "GATMI_KREG_DBUF", "GATMI_FUSED", "GATMI_FUSE_DBG", "GATMI_GCOMPACT", "GATMI_NARROW", "GATMI_NARROW_BLOCKS"};
constexpr int kNumOpts = sizeof(kOptNames) / sizeof(kOptNames[0]);
static int g_opt[kNumOpts];
static const bool g_opt_init = [] {
    for (int &v : g_opt) v = -1;
    return true;
}();
static int opt_index(const char *name) {
    for (int i = 0; i < kNumOpts; ++i)
        if (strcmp(name, kOptNames[i]) == 0) return i;
    return -1;
}
int option(const char *name, int dflt) {
    const int i = opt_index(name);
    if (i >= 0 && g_opt[i] >= 0) return g_opt[i];
    const char *e = getenv(name);
    if (e && e[0] >= '0' && e[0] <= '9') return atoi(e);
    return dflt;
}

struct ProfRec {
    char tag[32];
    hipEvent_t a, b;
};
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static std::mutex g_prof_mu;

void prof_begin(const char *tag, hipStream_t st) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r;
    snprintf(r.tag, sizeof(r.tag), "%s", tag);
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    (void)hipEventRecord(r.a, st);
    g_prof.push_back(r);
}
void prof_end(hipStream_t st) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof.empty()) (void)hipEventRecord(g_prof.back().b, st);
}
}  // namespace gatmi

extern "C" int gatmi_set_option(const char *name, int value) {
    const int i = name ? gatmi::opt_index(name) : -1;
    if (i < 0) {
        gatmi::set_error("gatmi_set_option: unknown option %s", name ? name : "(null)");
        return GATMI_ERR_ARG;
    }
    gatmi::g_opt[i] = value < 0 ? -1 : value;
    return GATMI_OK;
}

extern "C" int gatmi_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(gatmi::g_prof_mu);
    gatmi::g_prof_on = on != 0;
    return GATMI_OK;
}

extern "C" int gatmi_prof_read(char *tags, float *ms, int max_records) {
    std::lock_guard<std::mutex> lk(gatmi::g_prof_mu);
    int n = 0;
    for (auto &r : gatmi::g_prof) {
        float t = -1.f;
        if (hipEventSynchronize(r.b) == hipSuccess) (void)hipEventElapsedTime(&t, r.a, r.b);
        if (n < max_records && tags && ms) {
            memcpy(tags + (size_t)n * 32, r.tag, 32);
            ms[n] = t;
            ++n;
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    gatmi::g_prof.clear();
    return n;
}

extern "C" const char *gatmi_last_error(void) { return gatmi::g_err; }
extern "C" int gatmi_version(void) { return 100; }
