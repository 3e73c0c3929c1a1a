// Error reporting, version string and the scalar APA combination of the C ABI.
#include "am_common.h"
#include <stdarg.h>
#include <string.h>
#include <atomic>
#include <mutex>
#include <utility>
#include <vector>
#include <algorithm>

namespace am {
static thread_local char g_err[512] = "";
char* last_error_buf() { return g_err; }
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- per-device kernel attributes -----------------------------------------------------------------
hipError_t ensure_dynamic_lds(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::vector<std::pair<int, const void*>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    const std::pair<int, const void*> key(dev, kernel);
    if (std::find(done.begin(), done.end(), key) != done.end()) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.push_back(key);
    return e;
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

// ---------------------------------------------------------------------------------------------------
// Kernel-distance subset indices (host arithmetic): a restatement of numpy 2.x
// `Generator(PCG64).choice(n, m, replace=False)` as the reference calls it per subset (kd.py:176,185-186), so the
// whole S x 2 table is drawn in ~1 ms instead of 200 Python-level calls (~10 ms, the largest host cost of an
// evaluate() once the PRDC kernels run in milliseconds).  Algorithm (numpy/random/_generator.pyx, distributions.c):
//   PCG64 = 128-bit LCG (multiplier 0x2360ED051FC65DA44385DF649FCCF645) with XSL-RR output, 32-bit draws buffered in
//   pairs (low word first); bounded integers by Lemire's method on 32-bit draws;
//   n > 10000 and m > n / 50:  "tail shuffle" - Fisher-Yates over arange(n) from the top down to n - m, last m kept;
//   otherwise Floyd's algorithm with an open-addressing hash set of size 2^ceil(log2(1.2 m)), then a Fisher-Yates
//   shuffle of the m results.
// The Python side (metrics/kd.py) seeds numpy's PCG64 for the state, and cross-checks this function against numpy
// once per process before trusting it.
namespace {
struct Pcg64 {
    unsigned __int128 state, inc;
    bool has32 = false;
    uint32_t buf32 = 0;
    uint64_t next64() {
        const unsigned __int128 mult = ((unsigned __int128)0x2360ED051FC65DA4ULL << 64) | 0x4385DF649FCCF645ULL;
        state = state * mult + inc;
        const uint64_t hi = (uint64_t)(state >> 64), lo = (uint64_t)state;
        const uint64_t x = hi ^ lo;
        const unsigned rot = (unsigned)(hi >> 58);
        return (x >> rot) | (x << ((-rot) & 63));
    }
    uint32_t next32() {
        if (has32) { has32 = false; return buf32; }
        const uint64_t x = next64();
        has32 = true;
        buf32 = (uint32_t)(x >> 32);
        return (uint32_t)x;
    }
    uint32_t bounded(uint32_t rng) {                  // uniform on [0, rng], rng < 0xFFFFFFFF
        if (rng == 0) return 0;
        const uint32_t excl = rng + 1;
        uint64_t m = (uint64_t)next32() * excl;
        uint32_t left = (uint32_t)m;
        if (left < excl) {
            const uint32_t thr = (0xFFFFFFFFu - rng) % excl;
            while (left < thr) {
                m = (uint64_t)next32() * excl;
                left = (uint32_t)m;
            }
        }
        return (uint32_t)(m >> 32);
    }
};

void choice_without_replacement(Pcg64& g, int64_t n, int64_t m, int64_t* out, std::vector<int64_t>& scratch) {
    if (n > 10000 && m > n / 50) {                    // tail shuffle
        scratch.resize((size_t)n);
        for (int64_t i = 0; i < n; ++i) scratch[(size_t)i] = i;
        const int64_t first = std::max<int64_t>(n - m, 1);
        for (int64_t i = n - 1; i >= first; --i) std::swap(scratch[(size_t)i], scratch[g.bounded((uint32_t)i)]);
        for (int64_t i = 0; i < m; ++i) out[i] = scratch[(size_t)(n - m + i)];
        return;
    }
    uint64_t mask = (uint64_t)(1.2 * (double)m);      // Floyd
    for (int sh = 1; sh <= 32; sh <<= 1) mask |= mask >> sh;
    scratch.assign((size_t)mask + 1, -1);
    for (int64_t j = n - m; j < n; ++j) {
        const int64_t val = (int64_t)g.bounded((uint32_t)j);
        uint64_t loc = (uint64_t)val & mask;
        while (scratch[loc] != -1 && scratch[loc] != val) loc = (loc + 1) & mask;
        if (scratch[loc] == -1) {
            scratch[loc] = val;
            out[j - n + m] = val;
        } else {
            loc = (uint64_t)j & mask;
            while (scratch[loc] != -1) loc = (loc + 1) & mask;
            scratch[loc] = j;
            out[j - n + m] = j;
        }
    }
    for (int64_t i = m - 1; i >= 1; --i) std::swap(out[i], out[g.bounded((uint32_t)i)]);
}
}  // namespace

extern "C" int am_kd_draw_indices(uint64_t state_hi, uint64_t state_lo, uint64_t inc_hi, uint64_t inc_lo, int64_t n1, int64_t n2,
                                  int S, int m, int64_t* idx1, int64_t* idx2) {
    AM_REQUIRE(idx1 != nullptr && idx2 != nullptr, AM_ERR_BAD_ARG, "am_kd_draw_indices: null output pointer");
    AM_REQUIRE(S >= 1 && m >= 1 && m <= n1 && m <= n2, AM_ERR_BAD_SHAPE, "am_kd_draw_indices: S=%d m=%d n1=%lld n2=%lld", S, m,
               (long long)n1, (long long)n2);
    AM_REQUIRE(n1 < 0xFFFFFFFFLL && n2 < 0xFFFFFFFFLL, AM_ERR_BAD_SHAPE, "am_kd_draw_indices: more than 2^32 - 1 rows");
    Pcg64 g;
    g.state = ((unsigned __int128)state_hi << 64) | state_lo;
    g.inc = ((unsigned __int128)inc_hi << 64) | inc_lo;
    std::vector<int64_t> scratch;
    for (int s = 0; s < S; ++s) {                     // per subset: set 1 first, then set 2 (kd.py:185-186)
        choice_without_replacement(g, n1, m, idx1 + (int64_t)s * m, scratch);
        choice_without_replacement(g, n2, m, idx2 + (int64_t)s * m, scratch);
    }
    return AM_OK;
}
