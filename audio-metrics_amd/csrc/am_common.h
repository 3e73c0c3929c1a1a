// Shared host/device helpers for the gfx950 distribution-distance library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include "../../include/audio_metrics_hip.h"

namespace am {

// thread-local last-error text (am_last_error)
char* last_error_buf();
void set_error(const char* fmt, ...);

#define AM_HIP_TRY(expr)                                                             \
    do {                                                                             \
        hipError_t _e = (expr);                                                      \
        if (_e != hipSuccess) {                                                      \
            am::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),     \
                          __FILE__, __LINE__);                                       \
            return AM_ERR_HIP;                                                       \
        }                                                                            \
    } while (0)

#define AM_LAUNCH_CHECK()                                                            \
    do {                                                                             \
        hipError_t _e = hipGetLastError();                                           \
        if (_e != hipSuccess) {                                                      \
            am::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), \
                          __FILE__, __LINE__);                                       \
            return AM_ERR_HIP;                                                       \
        }                                                                            \
    } while (0)

#define AM_REQUIRE(cond, status, ...)                                                \
    do {                                                                             \
        if (!(cond)) {                                                               \
            am::set_error(__VA_ARGS__);                                              \
            return (status);                                                         \
        }                                                                            \
    } while (0)

// Kernels that use more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised once PER
// DEVICE (one process may drive several GPUs: AudioMetrics(device_indices=[...]) runs one thread per GPU).
hipError_t ensure_dynamic_lds(const void* kernel, int bytes);

// optional hipEvent bracket around the tile kernels (am_kernel_clock_enable / am_kernel_clock_read)
void clock_begin(int kernel, hipStream_t st);
void clock_end(int kernel, hipStream_t st);

// max(v, 0) of an EXACT squared distance, with a NaN turned into +inf.  In the reference a NaN distance - a row with a
// non-finite element - stays NaN through torch.cdist's clamp_min(0) (prdc.py:12,34) and then never counts: torch.kthvalue sorts
// NaN last, and NaN < radius is false.  +inf behaves the same way in every comparison of these kernels and, unlike NaN, is
// safe in the min / max networks of the sorted lists (list_insert: fminf / fmaxf return their non-NaN operand, a NaN would
// duplicate the list's smallest entry).  Plain fmaxf(v, 0) would put the pair at distance 0 from everybody.
__device__ __forceinline__ float clamp0(float v) { return v != v ? __builtin_inff() : fmaxf(v, 0.f); }

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Bump allocator over the caller's workspace (256-byte aligned carve-outs).
struct Carver {
    char* base;
    size_t size;
    size_t off = 0;
    Carver(void* p, size_t n) : base(static_cast<char*>(p)), size(n) {}
    template <class T>
    T* take(size_t count) {
        size_t bytes = round_up(count * sizeof(T), 256);
        T* r = (base && off + bytes <= size) ? reinterpret_cast<T*>(base + off) : nullptr;
        off += bytes;
        return r;
    }
    bool ok() const { return base != nullptr && off <= size; }
};

// ---- float64 rows through the f16 filter (round 5): pairwise.hip finds each row's candidates with the float32 path's filter
// sweep on a rounded copy and evaluates / selects them in f64; pairwise_f64.hip supplies the general f64 kernels behind a
// device flag for the calls the filter gives up on
bool knn64_filter_eligible(int64_t N, int D, int k);
size_t knn64_filter_workspace(int64_t N, int D, int k);
int knn64_filter(const double* X, int64_t N, int64_t ld, int D, int k, double* out_r, void* ws, size_t ws_bytes, hipStream_t st);
bool prdc64_filter_eligible(int64_t Nr, int64_t Nc, int D);
size_t prdc64_filter_workspace(int64_t Nr, int64_t Nc, int D);
// rt, ct: the f64 thresholds; col_count / row_any / row_cover: zeroed accumulators (int32 / unsigned); *fail_flag: the device
// flag behind which the caller launches the general f64 kernel (it runs for real only when the filter route gave up)
int prdc64_filter(const double* R, int64_t Nr, int64_t ldr, const double* C, int64_t Nc, int64_t ldc, int D, const double* rt,
                  const double* ct, int32_t* col_count, unsigned* row_any, unsigned* row_cover, const int** fail_flag, void* ws,
                  size_t ws_bytes, hipStream_t st);
size_t knn64_self_workspace(int64_t N, int k);
int knn64_self_gated(const double* X, int64_t N, int64_t ld, int D, int k, double* out_r, void* ws, size_t ws_bytes, const int* run_flag,
                     hipStream_t st);


}  // namespace am
