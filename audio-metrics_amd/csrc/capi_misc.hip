// Error reporting, version string and the scalar APA combination of the C ABI.
#include "am_common.h"
#include <stdarg.h>
#include <string.h>
#include <atomic>
#include <mutex>
#include <utility>
#include <vector>

namespace am {
static thread_local char g_err[512] = "";
char* last_error_buf() { return g_err; }
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- kernel clock ---------------------------------------------------------------------------------
namespace {
constexpr int N_CLOCKED = 4;
std::atomic<int> g_clock_on{0};
std::mutex g_clock_mu;
struct ClockRecord {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> done;   // begin/end pairs, both recorded
    hipEvent_t open = nullptr;                              // begin recorded, end pending
};
ClockRecord g_clock[N_CLOCKED];
}  // namespace

void clock_begin(int kernel, hipStream_t st) {
    if (!g_clock_on.load(std::memory_order_relaxed) || kernel < 0 || kernel >= N_CLOCKED) return;
    std::lock_guard<std::mutex> lock(g_clock_mu);
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    if (hipEventRecord(e, st) != hipSuccess) { (void)hipEventDestroy(e); return; }
    if (g_clock[kernel].open) (void)hipEventDestroy(g_clock[kernel].open);
    g_clock[kernel].open = e;
}

void clock_end(int kernel, hipStream_t st) {
    if (!g_clock_on.load(std::memory_order_relaxed) || kernel < 0 || kernel >= N_CLOCKED) return;
    std::lock_guard<std::mutex> lock(g_clock_mu);
    ClockRecord& r = g_clock[kernel];
    if (!r.open) return;
    hipEvent_t e;
    if (hipEventCreate(&e) == hipSuccess && hipEventRecord(e, st) == hipSuccess) {
        r.done.emplace_back(r.open, e);
    } else {
        (void)hipEventDestroy(r.open);
    }
    r.open = nullptr;
}
}  // namespace am

extern "C" int am_kernel_clock_enable(int on) {
    am::g_clock_on.store(on ? 1 : 0);
    return AM_OK;
}

extern "C" int am_kernel_clock_read(int kernel, int64_t* launches, double* total_ms) {
    AM_REQUIRE(kernel >= 0 && kernel < am::N_CLOCKED, AM_ERR_BAD_ARG, "am_kernel_clock_read: unknown kernel id %d", kernel);
    AM_REQUIRE(launches != nullptr && total_ms != nullptr, AM_ERR_BAD_ARG, "am_kernel_clock_read: null output pointer");
    std::lock_guard<std::mutex> lock(am::g_clock_mu);
    am::ClockRecord& r = am::g_clock[kernel];
    int64_t n = 0;
    double ms = 0.0;
    int rc = AM_OK;
    for (auto& pr : r.done) {
        float t = 0.f;
        if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&t, pr.first, pr.second) == hipSuccess) {
            ++n;
            ms += t;
        } else {
            am::set_error("am_kernel_clock_read: event query failed");
            rc = AM_ERR_HIP;
        }
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    r.done.clear();
    *launches = n;
    *total_ms = ms;
    return rc;
}

extern "C" const char* am_version(void) { return "audio_metrics_hip 0.1.0 (gfx950)"; }

extern "C" const char* am_last_error(void) { return am::last_error_buf(); }

extern "C" const char* am_status_string(int s) {
    switch (s) {
        case AM_OK: return "ok";
        case AM_ERR_BAD_ARG: return "bad argument";
        case AM_ERR_BAD_SHAPE: return "bad shape";
        case AM_ERR_UNSUPPORTED_K: return "unsupported nearest_k";
        case AM_ERR_WORKSPACE: return "workspace too small";
        case AM_ERR_NO_CONVERGENCE: return "Newton-Schulz did not converge";
        case AM_ERR_HIP: return "HIP runtime error";
        default: return "unknown status";
    }
}

// reference apa.py:22-32
extern "C" double am_apa_f64(double d_y_x, double d_y_xp, double d_x_xp) {
    if (d_y_x < 0) d_y_x = 0;
    if (d_y_xp < 0) d_y_xp = 0;
    if (d_x_xp < 0) d_x_xp = 0;
    const double num = d_y_xp - d_y_x;
    double den = d_x_xp;
    const double an = num < 0 ? -num : num;
    if (an > den) den = an;
    if (den <= 0) return 0.0;
    return 0.5 + num / (2 * den);
}
