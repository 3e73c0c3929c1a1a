// The two 256-row f16 FILTER kernels of the PRDC path on the OPERAND-STATIONARY engine (pstat_engine.h): the bodies of
// wide_kernels.h - membership filter, symmetric k-NN sweep - with the workgroup's P block held in registers for the whole
// work item and only Q slabs streaming through LDS.  Rows of up to 512 f16 (D <= 512); wider sets keep wide_engine.h
// (pairwise_wide.hip).  Same queue protocol, same accumulator values, same results; built as its own translation unit so
// that the three PRDC objects compile in parallel.
#include "wide_kernels.h"
#include "pstat_engine.h"

namespace am {

template <int KSLABS>
struct PstatEng {
    using Lane = PLane;
    static constexpr int LDS_WORDS = pstat_lds_words(KSLABS);
    template <class TileMap, class Epi>
    static __device__ __forceinline__ void run(const float* Q, int64_t nq, int64_t ldq, const TileMap& tm, const float* P, int64_t np,
                                               int64_t ldp, int64_t prow0, int ntiles, int, float* lds, Lane& L, Epi& epi) {
        pstat_pipeline<KSLABS>(Q, nq, ldq, tm, P, np, ldp, prow0, ntiles, lds, L, epi);
    }
};

// rows of up to P64_MAX_SLABS slabs (D <= 128): 256 threads, two workgroups per CU, a wave = 64 P rows x 64-row Q units
template <int KSLABS>
struct Pstat64Eng {
    using Lane = P64Lane;
    static constexpr int LDS_WORDS = pstat64_lds_words();
    template <class TileMap, class Epi>
    static __device__ __forceinline__ void run(const float* Q, int64_t nq, int64_t ldq, const TileMap& tm, const float* P, int64_t np,
                                               int64_t ldp, int64_t prow0, int ntiles, int, float* lds, Lane& L, Epi& epi) {
        pstat64_pipeline<KSLABS>(Q, nq, ldq, tm, P, np, ldp, prow0, ntiles, lds, L, epi);
    }
};

template <int KSLABS, bool WANT_MIN>
__global__ void __launch_bounds__(PTHREADS, 1)
cross_pstat_kernel(const float* __restrict__ Rb, int64_t Nr, int64_t ldr, const float* __restrict__ rnorm,
                   const float* __restrict__ rthr, const float* __restrict__ Cb, int64_t Nc, int64_t ldc,
                   const float* __restrict__ cnorm, const float* __restrict__ cthr, int Dh, int nchunks, int grp_rows,
                   const unsigned* __restrict__ maxn, unsigned* __restrict__ rmin_approx, unsigned* __restrict__ row_any,
                   unsigned* __restrict__ row_cover, int32_t* __restrict__ col_count, uint2* __restrict__ wgq, int qcap,
                   int* __restrict__ wgq_count, uint2* __restrict__ items, uint2* __restrict__ ovq, int* __restrict__ ov_count,
                   int ovcap, int* __restrict__ fail, float fc) {
    cross_wide_body<PstatEng<KSLABS>, WANT_MIN>(Rb, Nr, ldr, rnorm, rthr, Cb, Nc, ldc, cnorm, cthr, Dh, nchunks, grp_rows, maxn, rmin_approx,
                                                row_any, row_cover, col_count, wgq, qcap, wgq_count, items, ovq, ov_count, ovcap, fail, fc);
}

template <int KSLABS, int KCAP>
__global__ void __launch_bounds__(PTHREADS, 1)
knn_pstat_kernel(const float* __restrict__ Xb, int64_t N, int64_t ldh, const float* __restrict__ xnorm, float* thr, int Dh,
                 int win_tiles, int nwin, int per_win, int k1, const unsigned* __restrict__ maxn, float* __restrict__ partial,
                 int* __restrict__ cnt, int cap, uint2* __restrict__ wgq, float* __restrict__ wgv, int qcap,
                 int* __restrict__ wgq_count, int part, int nparts, float fc, uint2* __restrict__ ovq, float* __restrict__ ovv,
                 unsigned long long* __restrict__ ovn, int ovcap, const int* __restrict__ skip, int* __restrict__ region_counter) {
    knn_wide_body<PstatEng<KSLABS>, KCAP>(Xb, N, ldh, xnorm, thr, Dh, win_tiles, nwin, per_win, k1, maxn, partial, cnt, cap, wgq, wgv, qcap,
                                          wgq_count, part, nparts, fc, ovq, ovv, ovn, ovcap, skip, region_counter);
}

template <int KSLABS, bool WANT_MIN>
__global__ void __launch_bounds__(P64_THREADS, 2)
cross_pstat64_kernel(const float* __restrict__ Rb, int64_t Nr, int64_t ldr, const float* __restrict__ rnorm,
                     const float* __restrict__ rthr, const float* __restrict__ Cb, int64_t Nc, int64_t ldc,
                     const float* __restrict__ cnorm, const float* __restrict__ cthr, int Dh, int nchunks, int grp_rows,
                     const unsigned* __restrict__ maxn, unsigned* __restrict__ rmin_approx, unsigned* __restrict__ row_any,
                     unsigned* __restrict__ row_cover, int32_t* __restrict__ col_count, uint2* __restrict__ wgq, int qcap,
                     int* __restrict__ wgq_count, uint2* __restrict__ items, uint2* __restrict__ ovq, int* __restrict__ ov_count,
                     int ovcap, int* __restrict__ fail, float fc) {
    cross_wide_body<Pstat64Eng<KSLABS>, WANT_MIN>(Rb, Nr, ldr, rnorm, rthr, Cb, Nc, ldc, cnorm, cthr, Dh, nchunks, grp_rows, maxn, rmin_approx,
                                                  row_any, row_cover, col_count, wgq, qcap, wgq_count, items, ovq, ov_count, ovcap, fail, fc);
}

template <int KSLABS, int KCAP>
__global__ void __launch_bounds__(P64_THREADS, 2)
knn_pstat64_kernel(const float* __restrict__ Xb, int64_t N, int64_t ldh, const float* __restrict__ xnorm, float* thr, int Dh,
                   int win_tiles, int nwin, int per_win, int k1, const unsigned* __restrict__ maxn, float* __restrict__ partial,
                   int* __restrict__ cnt, int cap, uint2* __restrict__ wgq, float* __restrict__ wgv, int qcap,
                   int* __restrict__ wgq_count, int part, int nparts, float fc, uint2* __restrict__ ovq, float* __restrict__ ovv,
                   unsigned long long* __restrict__ ovn, int ovcap, const int* __restrict__ skip, int* __restrict__ region_counter) {
    knn_wide_body<Pstat64Eng<KSLABS>, KCAP>(Xb, N, ldh, xnorm, thr, Dh, win_tiles, nwin, per_win, k1, maxn, partial, cnt, cap, wgq, wgv, qcap,
                                            wgq_count, part, nparts, fc, ovq, ovv, ovn, ovcap, skip, region_counter);
}

// which of the two stationary forms a row length takes: a pure function of the width (AM_PSTAT64=0 in the A/B build: the
// 512-thread form for every width)
static bool pstat64_takes(int kslabs) {
    static const int on = env_int("AM_PSTAT64", 1);
    return on != 0 && kslabs <= P64_MAX_SLABS;
}

#ifdef AM_DEV_KNOBS
// AM_WIDE_DBG -> this translation unit's copy of g_wide_dbg (1: every Q tile read from the first 2 MB, 2: epilogues skipped)
static hipError_t set_pstat_dev_symbols(hipStream_t st) {
    const int wdbg = env_int("AM_WIDE_DBG", 0);
    return hipMemcpyToSymbolAsync(HIP_SYMBOL(g_wide_dbg), &wdbg, sizeof(int), 0, hipMemcpyHostToDevice, st);
}
#endif

// rows of Dh words (Dh % 32 == 0): the stationary form holds KSLABS = Dh / 32 slabs of a row in registers
bool pstat_supported(int Dh) { return Dh % WROW == 0 && Dh / WROW >= 1 && Dh / WROW <= 8; }
bool pstat64_supported(int Dh) { return pstat_supported(Dh) && pstat64_takes(Dh / WROW); }

template <int KSLABS, bool WANT_MIN>
static int launch_cross_pstat64_t(unsigned blocks, const float* Rb, int64_t Nr, int64_t ldr, const float* rnorm, const float* rthr,
                                  const float* Cb, int64_t Nc, int64_t ldc, const float* cnorm, const float* cthr, int Dh, int nchunks,
                                  int grp_rows, const unsigned* maxn, unsigned* rmin_approx, unsigned* row_any, unsigned* row_cover,
                                  int32_t* col_count, uint2* wgq, int qcap, int* wgq_count, uint2* items, uint2* ovq, int* ov_count,
                                  int ovcap, int* fail, float fc, hipStream_t st) {
    constexpr size_t lds_bytes = wide_cross_lds_bytes<Pstat64Eng<KSLABS>>();
    AM_HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(&cross_pstat64_kernel<KSLABS, WANT_MIN>), (int)lds_bytes));
#ifdef AM_DEV_KNOBS
    AM_HIP_TRY(set_pstat_dev_symbols(st));
#endif
    hipLaunchKernelGGL((cross_pstat64_kernel<KSLABS, WANT_MIN>), dim3(blocks), dim3(P64_THREADS), lds_bytes, st, Rb, Nr, ldr, rnorm, rthr, Cb,
                       Nc, ldc, cnorm, cthr, Dh, nchunks, grp_rows, maxn, rmin_approx, row_any, row_cover, col_count, wgq, qcap,
                       wgq_count, items, ovq, ov_count, ovcap, fail, fc);
    AM_LAUNCH_CHECK();
    return AM_OK;
}

template <int KSLABS, bool WANT_MIN>
static int launch_cross_pstat_t(unsigned blocks, const float* Rb, int64_t Nr, int64_t ldr, const float* rnorm, const float* rthr,
                                const float* Cb, int64_t Nc, int64_t ldc, const float* cnorm, const float* cthr, int Dh, int nchunks,
                                int grp_rows, const unsigned* maxn, unsigned* rmin_approx, unsigned* row_any, unsigned* row_cover,
                                int32_t* col_count, uint2* wgq, int qcap, int* wgq_count, uint2* items, uint2* ovq, int* ov_count,
                                int ovcap, int* fail, float fc, hipStream_t st) {
    constexpr size_t lds_bytes = wide_cross_lds_bytes<PstatEng<KSLABS>>();
    AM_HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(&cross_pstat_kernel<KSLABS, WANT_MIN>), (int)lds_bytes));
#ifdef AM_DEV_KNOBS
    AM_HIP_TRY(set_pstat_dev_symbols(st));
#endif
    hipLaunchKernelGGL((cross_pstat_kernel<KSLABS, WANT_MIN>), dim3(blocks), dim3(PTHREADS), lds_bytes, st, Rb, Nr, ldr, rnorm, rthr, Cb, Nc,
                       ldc, cnorm, cthr, Dh, nchunks, grp_rows, maxn, rmin_approx, row_any, row_cover, col_count, wgq, qcap, wgq_count,
                       items, ovq, ov_count, ovcap, fail, fc);
    AM_LAUNCH_CHECK();
    return AM_OK;
}

int launch_cross_pstat(bool want_min, unsigned blocks, const float* Rb, int64_t Nr, int64_t ldr, const float* rnorm, const float* rthr,
                       const float* Cb, int64_t Nc, int64_t ldc, const float* cnorm, const float* cthr, int Dh, int nchunks,
                       int grp_rows, const unsigned* maxn, unsigned* rmin_approx, unsigned* row_any, unsigned* row_cover,
                       int32_t* col_count, uint2* wgq, int qcap, int* wgq_count, uint2* items, uint2* ovq, int* ov_count, int ovcap,
                       int* fail, float fc, hipStream_t st) {
#define AM_PSTAT64_CROSS(KS)                                                                                                        \
    case KS:                                                                                                                        \
        return want_min ? launch_cross_pstat64_t<KS, true>(blocks, Rb, Nr, ldr, rnorm, rthr, Cb, Nc, ldc, cnorm, cthr, Dh, nchunks, \
                                                           grp_rows, maxn, rmin_approx, row_any, row_cover, col_count, wgq, qcap,   \
                                                           wgq_count, items, ovq, ov_count, ovcap, fail, fc, st)                    \
                        : launch_cross_pstat64_t<KS, false>(blocks, Rb, Nr, ldr, rnorm, rthr, Cb, Nc, ldc, cnorm, cthr, Dh, nchunks,\
                                                            grp_rows, maxn, rmin_approx, row_any, row_cover, col_count, wgq, qcap,  \
                                                            wgq_count, items, ovq, ov_count, ovcap, fail, fc, st);
    if (pstat64_takes(Dh / WROW)) {
        switch (Dh / WROW) { AM_PSTAT64_CROSS(1) AM_PSTAT64_CROSS(2) }
    }
#undef AM_PSTAT64_CROSS
#define AM_PSTAT_CROSS(KS)                                                                                                          \
    case KS:                                                                                                                        \
        return want_min ? launch_cross_pstat_t<KS, true>(blocks, Rb, Nr, ldr, rnorm, rthr, Cb, Nc, ldc, cnorm, cthr, Dh, nchunks,   \
                                                         grp_rows, maxn, rmin_approx, row_any, row_cover, col_count, wgq, qcap,     \
                                                         wgq_count, items, ovq, ov_count, ovcap, fail, fc, st)                      \
                        : launch_cross_pstat_t<KS, false>(blocks, Rb, Nr, ldr, rnorm, rthr, Cb, Nc, ldc, cnorm, cthr, Dh, nchunks,  \
                                                          grp_rows, maxn, rmin_approx, row_any, row_cover, col_count, wgq, qcap,    \
                                                          wgq_count, items, ovq, ov_count, ovcap, fail, fc, st);
    switch (Dh / WROW) {
        AM_PSTAT_CROSS(1) AM_PSTAT_CROSS(2) AM_PSTAT_CROSS(3) AM_PSTAT_CROSS(4)
        AM_PSTAT_CROSS(5) AM_PSTAT_CROSS(6) AM_PSTAT_CROSS(7) AM_PSTAT_CROSS(8)
    }
#undef AM_PSTAT_CROSS
    set_error("the operand-stationary membership filter holds rows of up to 512 f16 (got %d words)", Dh);
    return AM_ERR_BAD_SHAPE;
}

template <int KSLABS, int KCAP>
static int launch_knn_pstat64_t(unsigned nwg, const float* Xb, int64_t N, int64_t ldh, const float* xnorm, float* thr, int Dh,
                                int win_tiles, int nwin, int per_win, int k1, const unsigned* maxn, float* partial, int* cnt, int cap,
                                uint2* wgq, float* wgv, int qcap, int* wgq_count, int part, int nparts, float fc, uint2* ovq,
                                float* ovv, unsigned long long* ovn, int ovcap, const int* skip, int* region_counter, hipStream_t st) {
    constexpr size_t lds_bytes = knn_wide_lds_bytes<Pstat64Eng<KSLABS>>();
    AM_HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(&knn_pstat64_kernel<KSLABS, KCAP>), (int)lds_bytes));
#ifdef AM_DEV_KNOBS
    AM_HIP_TRY(set_pstat_dev_symbols(st));
#endif
    hipLaunchKernelGGL((knn_pstat64_kernel<KSLABS, KCAP>), dim3(nwg), dim3(P64_THREADS), lds_bytes, st, Xb, N, ldh, xnorm, thr, Dh, win_tiles,
                       nwin, per_win, k1, maxn, partial, cnt, cap, wgq, wgv, qcap, wgq_count, part, nparts, fc, ovq, ovv, ovn, ovcap, skip,
                       region_counter);
    AM_LAUNCH_CHECK();
    return AM_OK;
}

template <int KSLABS, int KCAP>
static int launch_knn_pstat_t(unsigned nwg, const float* Xb, int64_t N, int64_t ldh, const float* xnorm, float* thr, int Dh,
                              int win_tiles, int nwin, int per_win, int k1, const unsigned* maxn, float* partial, int* cnt, int cap,
                              uint2* wgq, float* wgv, int qcap, int* wgq_count, int part, int nparts, float fc, uint2* ovq,
                              float* ovv, unsigned long long* ovn, int ovcap, const int* skip, int* region_counter, hipStream_t st) {
    constexpr size_t lds_bytes = knn_wide_lds_bytes<PstatEng<KSLABS>>();
    AM_HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(&knn_pstat_kernel<KSLABS, KCAP>), (int)lds_bytes));
#ifdef AM_DEV_KNOBS
    AM_HIP_TRY(set_pstat_dev_symbols(st));
#endif
    hipLaunchKernelGGL((knn_pstat_kernel<KSLABS, KCAP>), dim3(nwg), dim3(PTHREADS), lds_bytes, st, Xb, N, ldh, xnorm, thr, Dh, win_tiles,
                       nwin, per_win, k1, maxn, partial, cnt, cap, wgq, wgv, qcap, wgq_count, part, nparts, fc, ovq, ovv, ovn, ovcap, skip,
                       region_counter);
    AM_LAUNCH_CHECK();
    return AM_OK;
}

int launch_knn_pstat(int kcap, unsigned nwg, const float* Xb, int64_t N, int64_t ldh, const float* xnorm, float* thr, int Dh,
                     int win_tiles, int nwin, int per_win, int k1, const unsigned* maxn, float* partial, int* cnt, int cap,
                     uint2* wgq, float* wgv, int qcap, int* wgq_count, int part, int nparts, float fc, uint2* ovq, float* ovv,
                     unsigned long long* ovn, int ovcap, const int* skip, int* region_counter, hipStream_t st) {
    AM_REQUIRE(kcap == 6 || kcap == 11, AM_ERR_UNSUPPORTED_K, "the 256-row k-NN sweep holds lists of 6 or 11 entries (got %d)", kcap);
#define AM_PSTAT64_KNN(KS)                                                                                                          \
    case KS:                                                                                                                        \
        return kcap == 6 ? launch_knn_pstat64_t<KS, 6>(nwg, Xb, N, ldh, xnorm, thr, Dh, win_tiles, nwin, per_win, k1, maxn, partial,\
                                                       cnt, cap, wgq, wgv, qcap, wgq_count, part, nparts, fc, ovq, ovv, ovn, ovcap, \
                                                       skip, region_counter, st)                                                    \
                         : launch_knn_pstat64_t<KS, 11>(nwg, Xb, N, ldh, xnorm, thr, Dh, win_tiles, nwin, per_win, k1, maxn,        \
                                                        partial, cnt, cap, wgq, wgv, qcap, wgq_count, part, nparts, fc, ovq, ovv,   \
                                                        ovn, ovcap, skip, region_counter, st);
    if (pstat64_takes(Dh / WROW)) {
        switch (Dh / WROW) { AM_PSTAT64_KNN(1) AM_PSTAT64_KNN(2) }
    }
#undef AM_PSTAT64_KNN
#define AM_PSTAT_KNN(KS)                                                                                                            \
    case KS:                                                                                                                        \
        return kcap == 6 ? launch_knn_pstat_t<KS, 6>(nwg, Xb, N, ldh, xnorm, thr, Dh, win_tiles, nwin, per_win, k1, maxn, partial,  \
                                                     cnt, cap, wgq, wgv, qcap, wgq_count, part, nparts, fc, ovq, ovv, ovn, ovcap,   \
                                                     skip, region_counter, st)                                                      \
                         : launch_knn_pstat_t<KS, 11>(nwg, Xb, N, ldh, xnorm, thr, Dh, win_tiles, nwin, per_win, k1, maxn, partial, \
                                                      cnt, cap, wgq, wgv, qcap, wgq_count, part, nparts, fc, ovq, ovv, ovn, ovcap,  \
                                                      skip, region_counter, st);
    switch (Dh / WROW) {
        AM_PSTAT_KNN(1) AM_PSTAT_KNN(2) AM_PSTAT_KNN(3) AM_PSTAT_KNN(4)
        AM_PSTAT_KNN(5) AM_PSTAT_KNN(6) AM_PSTAT_KNN(7) AM_PSTAT_KNN(8)
    }
#undef AM_PSTAT_KNN
    set_error("the operand-stationary k-NN sweep holds rows of up to 512 f16 (got %d words)", Dh);
    return AM_ERR_BAD_SHAPE;
}

}  // namespace am
