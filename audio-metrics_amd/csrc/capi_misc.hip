// Error reporting, version string and the scalar APA combination of the C ABI.
#include "am_common.h"
#include <stdarg.h>
#include <string.h>

namespace am {
static thread_local char g_err[512] = "";
char* last_error_buf() { return g_err; }
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace am

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
