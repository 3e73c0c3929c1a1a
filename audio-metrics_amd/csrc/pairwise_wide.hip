// The 256 x 256 f16 FILTER kernels of the PRDC path (wide_engine.h): the membership filter cross_wide_kernel and the
// symmetric k-NN sweep knn_wide_kernel, with their launchers.  Algorithms, error bound and queue protocol are documented
// in pairwise_fast.h, which holds the 128-row forms of the same filters and everything that runs around these kernels.
#include "wide_kernels.h"

namespace am {

#ifdef AM_DEV_KNOBS
static unsigned long long* g_trace_dev = nullptr;
// AM_WIDE_DBG -> g_wide_dbg; AM_WIDE_TRACE=1 -> a trace buffer (64 workgroups from AM_WIDE_TRACE_B0 on x 2 waves x 96 stages x 6 stamps) that
// am_wide_trace_read copies out.  Development aid of the A/B build only.
static hipError_t set_wide_dev_symbols(hipStream_t st) {
    const int wdbg = env_int("AM_WIDE_DBG", 0), b0 = env_int("AM_WIDE_TRACE_B0", 0);
    hipError_t e = hipMemcpyToSymbolAsync(HIP_SYMBOL(g_wide_dbg), &wdbg, sizeof(int), 0, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return e;
    e = hipMemcpyToSymbolAsync(HIP_SYMBOL(g_wide_trace_b0), &b0, sizeof(int), 0, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return e;
    if (env_int("AM_WIDE_TRACE", 0) && g_trace_dev == nullptr) {
        e = hipMalloc(&g_trace_dev, 64 * 2 * 96 * 6 * sizeof(unsigned long long));
        if (e != hipSuccess) return e;
    }
    if (g_trace_dev != nullptr) {
        e = hipMemsetAsync(g_trace_dev, 0, 64 * 2 * 96 * 6 * sizeof(unsigned long long), st);
        if (e != hipSuccess) return e;
    }
    return hipMemcpyToSymbolAsync(HIP_SYMBOL(g_wide_trace), &g_trace_dev, sizeof(g_trace_dev), 0, hipMemcpyHostToDevice, st);
}
}  // namespace am
extern "C" int am_wide_trace_read(unsigned long long* host, size_t count) {
    if (am::g_trace_dev == nullptr) return -1;
    return hipMemcpy(host, am::g_trace_dev, std::min<size_t>(count, 64 * 2 * 96 * 6) * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -6;
}
namespace am {
#endif

struct WideEng {
    using Lane = WLane;
    static constexpr int LDS_WORDS = WENGINE_LDS_WORDS;
    template <class TileMap, class Epi>
    static __device__ __forceinline__ void run(const float* Q, int64_t nq, int64_t ldq, const TileMap& tm, const float* P, int64_t np,
                                               int64_t ldp, int64_t prow0, int ntiles, int Dh, float* lds, Lane& L, Epi& epi) {
        wide_pipeline(Q, nq, ldq, tm, P, np, ldp, prow0, ntiles, Dh, lds, L, epi);
    }
};
constexpr size_t WIDE_CROSS_LDS_BYTES = wide_cross_lds_bytes<WideEng>();
constexpr size_t KNN_WIDE_LDS_BYTES = knn_wide_lds_bytes<WideEng>();

int64_t wide_grouped_blocks(int64_t row_blocks, int nchunks, int grp_rows) { return wide_grouped_blocks_impl(row_blocks, nchunks, grp_rows); }

template <bool WANT_MIN>
__global__ void __launch_bounds__(WTHREADS, 1)
cross_wide_kernel(const float* __restrict__ Rb, int64_t Nr, int64_t ldr, const float* __restrict__ rnorm,
                  const float* __restrict__ rthr, const float* __restrict__ Cb, int64_t Nc, int64_t ldc,
                  const float* __restrict__ cnorm, const float* __restrict__ cthr, int Dh, int nchunks, int grp_rows,
                  const unsigned* __restrict__ maxn, unsigned* __restrict__ rmin_approx, unsigned* __restrict__ row_any,
                  unsigned* __restrict__ row_cover, int32_t* __restrict__ col_count, uint2* __restrict__ wgq, int qcap,
                  int* __restrict__ wgq_count, uint2* __restrict__ items, uint2* __restrict__ ovq, int* __restrict__ ov_count,
                  int ovcap, int* __restrict__ fail, float fc) {
    cross_wide_body<WideEng, WANT_MIN>(Rb, Nr, ldr, rnorm, rthr, Cb, Nc, ldc, cnorm, cthr, Dh, nchunks, grp_rows, maxn, rmin_approx, row_any,
                                       row_cover, col_count, wgq, qcap, wgq_count, items, ovq, ov_count, ovcap, fail, fc);
}

template <int KCAP>
__global__ void __launch_bounds__(WTHREADS, 1)
knn_wide_kernel(const float* __restrict__ Xb, int64_t N, int64_t ldh, const float* __restrict__ xnorm, float* thr, int Dh,
                int win_tiles, int nwin, int per_win, int k1, const unsigned* __restrict__ maxn, float* __restrict__ partial,
                int* __restrict__ cnt, int cap, uint2* __restrict__ wgq, float* __restrict__ wgv, int qcap,
                int* __restrict__ wgq_count, int part, int nparts, float fc, uint2* __restrict__ ovq, float* __restrict__ ovv,
                unsigned long long* __restrict__ ovn, int ovcap, const int* __restrict__ skip, int* __restrict__ region_counter) {
    knn_wide_body<WideEng, KCAP>(Xb, N, ldh, xnorm, thr, Dh, win_tiles, nwin, per_win, k1, maxn, partial, cnt, cap, wgq, wgv, qcap, wgq_count,
                                 part, nparts, fc, ovq, ovv, ovn, ovcap, skip, region_counter);
}

int launch_cross_wide(bool want_min, unsigned blocks, const float* Rb, int64_t Nr, int64_t ldr, const float* rnorm, const float* rthr,
                      const float* Cb, int64_t Nc, int64_t ldc, const float* cnorm, const float* cthr, int Dh, int nchunks,
                      int grp_rows, const unsigned* maxn, unsigned* rmin_approx, unsigned* row_any, unsigned* row_cover,
                      int32_t* col_count, uint2* wgq, int qcap, int* wgq_count, uint2* items, uint2* ovq, int* ov_count, int ovcap,
                      int* fail, float fc, hipStream_t st) {
    AM_HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(&cross_wide_kernel<true>), (int)WIDE_CROSS_LDS_BYTES));
    AM_HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(&cross_wide_kernel<false>), (int)WIDE_CROSS_LDS_BYTES));
#ifdef AM_DEV_KNOBS
    AM_HIP_TRY(set_wide_dev_symbols(st));
#endif
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(WTHREADS), WIDE_CROSS_LDS_BYTES, st, Rb, Nr, ldr, rnorm, rthr, Cb, Nc, ldc,
                           cnorm, cthr, Dh, nchunks, grp_rows, maxn, rmin_approx, row_any, row_cover, col_count, wgq, qcap,
                           wgq_count, items, ovq, ov_count, ovcap, fail, fc);
    };
    if (want_min) launch(&cross_wide_kernel<true>);
    else launch(&cross_wide_kernel<false>);
    AM_LAUNCH_CHECK();
    return AM_OK;
}

struct StridedTiles {
    int64_t s0;
    int stride;
    __device__ __forceinline__ int64_t operator()(int t) const { return (s0 + t) * stride; }
};

// ---- sampled "any" pre-pass of the membership filter on the 256-row engine --------------------------------------------
// Every `stride`-th 256-row candidate tile against all reference rows: a reference row is flagged as soon as one sampled
// candidate lies inside its ball FOR CERTAIN (approximate value below T'_j - E'_j).  The main pass then carries the "any"
// direction only for the rows still without a witness.  (On the 128-row engine this pass ran at 0.8 PF: 0.80 ms.)
struct CrossSampleEpilogue {
    const float* qnorm;
    const float* qthr;
    int64_t nq;
    float fc, rnmax_c;
    float* aux;                 // LDS [2][2][256] : |c_j|^2, T'_j - E'_j of the tile
    float dsc;
    float xn[2];
    bool anyf[2];
    float aux_n, aux_hi;
    const WLane& L;
    __device__ __forceinline__ CrossSampleEpilogue(const WLane& l) : L(l) {}
    __device__ __forceinline__ void aux_issue(int, int64_t qtile) {
        if (L.tid < WTB) {
            const int64_t j = qtile * WTB + L.tid;
            const bool in = j < nq;
            aux_n = in ? qnorm[j] : INFINITY;
            aux_hi = in ? qthr[j] : -INFINITY;
        }
    }
    __device__ __forceinline__ void aux_commit(int t) {
        if (L.tid < WTB) {
            float* d = aux + (t & 1) * 2 * WTB + L.tid;
            const bool in = aux_hi > -INFINITY;
            d[0] = aux_n;
            d[WTB] = in ? aux_hi - fmaf(fc, aux_n, rnmax_c) : -INFINITY;
        }
    }
    __device__ __forceinline__ void finish(int t, int64_t, f32x16 (&acc)[4][2]) {
        const float* a = aux + (t & 1) * 2 * WTB + L.wm * 128 + L.h * 4;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            f32x4 yn[4], tl[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                yn[g4] = *reinterpret_cast<const f32x4*>(a + mt * 32 + g4 * 8);
                tl[g4] = *reinterpret_cast<const f32x4*>(a + WTB + mt * 32 + g4 * 8);
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                float mg = INFINITY;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg)
                    mg = fminf(mg, fmaf(dsc, acc[mt][nt][reg], yn[reg >> 2][reg & 3]) - tl[reg >> 2][reg & 3]);   // +inf - (-inf) = +inf past nq
                anyf[nt] = anyf[nt] || (mg + xn[nt] < 0.f);
            }
        }
    }
};

constexpr size_t CROSS_SAMPLE_LDS_BYTES = (WENGINE_LDS_WORDS + 4 * WTB) * sizeof(float);

__global__ void __launch_bounds__(WTHREADS, 1)
cross_wide_sample_kernel(const float* __restrict__ Rb, int64_t Nr, int64_t ldr, const float* __restrict__ rnorm,
                         const float* __restrict__ Cb, int64_t Nc, int64_t ldc, const float* __restrict__ cnorm,
                         const float* __restrict__ cthr, int Dh, int stride, int nchunks, const unsigned* __restrict__ maxn,
                         unsigned* __restrict__ row_any, float fc) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const WLane L;
    const int64_t T = (Nc + WTB - 1) / WTB;
    const int64_t samples = (T + stride - 1) / stride;
    const int64_t pb = blockIdx.x / nchunks;
    const int chunk = blockIdx.x % nchunks;
    const int64_t s0 = samples * chunk / nchunks, s1 = samples * (chunk + 1) / nchunks;
    if (s1 <= s0) return;
    const float gmax = fmaxf(__uint_as_float(maxn[0]), __uint_as_float(maxn[1]));
    CrossSampleEpilogue epi(L);
    epi.qnorm = cnorm;
    epi.qthr = cthr;
    epi.nq = Nc;
    epi.fc = fc;
    epi.rnmax_c = fc * gmax;
    epi.aux = lds + WENGINE_LDS_WORDS;
    epi.dsc = half_unscale(maxn[2], maxn[3]);
    int64_t row[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        row[nt] = pb * WTB + L.wn * 64 + nt * 32 + L.r;
        epi.xn[nt] = row[nt] < Nr ? rnorm[row[nt]] : INFINITY;           // past the end: every value +inf, never a witness
        epi.anyf[nt] = false;
    }
    wide_pipeline(Cb, Nc, ldc, StridedTiles{s0, stride}, Rb, Nr, ldr, pb * WTB, (int)(s1 - s0), Dh, lds, L, epi);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int other = __shfl_xor((int)epi.anyf[nt], 32);                    // unconditionally: every lane must take part
        if (L.h == 0 && row[nt] < Nr && (epi.anyf[nt] || other != 0)) atomicOr(row_any + row[nt], 1u);
    }
}

int launch_cross_wide_sample(const float* Rb, int64_t Nr, int64_t ldr, const float* rnorm, const float* Cb, int64_t Nc, int64_t ldc,
                             const float* cnorm, const float* cthr, int Dh, int stride, int nchunks, const unsigned* maxn,
                             unsigned* row_any, float fc, hipStream_t st) {
    AM_HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(&cross_wide_sample_kernel), (int)CROSS_SAMPLE_LDS_BYTES));
    const unsigned blocks = (unsigned)(ceil_div(Nr, WTB) * nchunks);
    hipLaunchKernelGGL(cross_wide_sample_kernel, dim3(blocks), dim3(WTHREADS), CROSS_SAMPLE_LDS_BYTES, st, Rb, Nr, ldr, rnorm, Cb, Nc,
                       ldc, cnorm, cthr, Dh, stride, nchunks, maxn, row_any, fc);
    AM_LAUNCH_CHECK();
    return AM_OK;
}

// ---- sampled bound pass on the 256-row engine ---------------------------------------------------------------------
// Every row needs an upper bound of its final (k+1)-th smallest value before the sweep starts.  It comes from every
// `stride`-th column tile (the row's own columns included): per accumulator tile the smallest of the sixteen approximate
// values goes into the lane's list - values of DISTINCT columns, so the (k+1)-th smallest of the merged lists bounds the
// (k+1)-th smallest over the sample, and that the row's final value, from above.  (The 128-row engine did this pass at
// 0.55 PF: 1.16 ms per set at 100k x 512.)
template <int KCAP>
struct KnnSampleEpilogue {
    const float* qnorm;
    int64_t n;
    float* aux;                 // LDS [2][WTB] : |x_j|^2 of the tile
    float dsc;
    float xn[2];
    float best[2][KCAP];        // ascending, +inf padded
    float aux_n;
    const WLane& L;
    __device__ __forceinline__ KnnSampleEpilogue(const WLane& l) : L(l) {}
    __device__ __forceinline__ void aux_issue(int, int64_t qtile) {
        if (L.tid < WTB) {
            const int64_t j = qtile * WTB + L.tid;
            aux_n = j < n ? qnorm[j] : INFINITY;
        }
    }
    __device__ __forceinline__ void aux_commit(int t) {
        if (L.tid < WTB) aux[(t & 1) * WTB + L.tid] = aux_n;
    }
    __device__ __forceinline__ void finish(int t, int64_t, f32x16 (&acc)[4][2]) {
        const float* a = aux + (t & 1) * WTB + L.wm * 128 + L.h * 4;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            f32x4 yn[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) yn[g4] = *reinterpret_cast<const f32x4*>(a + mt * 32 + g4 * 8);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                float tmin = INFINITY;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) tmin = fminf(tmin, fmaf(dsc, acc[mt][nt][reg], xn[nt] + yn[reg >> 2][reg & 3]));
                const float v = fmaxf(tmin, 0.f);
                if (__any(v < best[nt][KCAP - 1])) list_insert<KCAP>(best[nt], v);
            }
        }
    }
};


constexpr size_t KNN_SAMPLE_LDS_BYTES = (WENGINE_LDS_WORDS + 4 * WTB) * sizeof(float);

// partial[(chunk * N + i) * KCAP + s]: this chunk's smallest sampled values of row i (knn_merge_kernel combines the chunks)
template <int KCAP>
__global__ void __launch_bounds__(WTHREADS, 1)
knn_wide_sample_kernel(const float* __restrict__ Xb, int64_t N, int64_t ldh, const float* __restrict__ xnorm, int Dh, int stride,
                       int nchunks, const unsigned* __restrict__ maxn, float* __restrict__ partial, int64_t row0, int64_t nrows) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const WLane L;
    const int64_t T = (N + WTB - 1) / WTB;
    const int64_t samples = (T + stride - 1) / stride;
    const int64_t pb = blockIdx.x / nchunks;
    const int chunk = blockIdx.x % nchunks;
    const int64_t s0 = samples * chunk / nchunks, s1 = samples * (chunk + 1) / nchunks;
    KnnSampleEpilogue<KCAP> epi(L);
    epi.qnorm = xnorm;
    epi.n = N;
    epi.aux = lds + WENGINE_LDS_WORDS;
    epi.dsc = half_unscale(maxn[2], maxn[2]);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int64_t i = row0 + pb * WTB + L.wn * 64 + nt * 32 + L.r;         // rows [row0, row0 + nrows) of the set (a rank's shard)
        epi.xn[nt] = i < N ? xnorm[i] : INFINITY;
#pragma unroll
        for (int s = 0; s < KCAP; ++s) epi.best[nt][s] = INFINITY;
    }
    if (s1 > s0) wide_pipeline(Xb, N, ldh, StridedTiles{s0, stride}, Xb, N, ldh, row0 + pb * WTB, (int)(s1 - s0), Dh, lds, L, epi);
    __syncthreads();
    float* mg = lds;                                   // [256][4][KCAP]
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        float* dst = mg + ((L.wn * 64 + nt * 32 + L.r) * 4 + (L.wm * 2 + L.h)) * KCAP;
#pragma unroll
        for (int s = 0; s < KCAP; ++s) dst[s] = epi.best[nt][s];
    }
    __syncthreads();
    if (L.tid < WTB) {
        const int64_t i = pb * WTB + L.tid;                    // (relative to row0)
        if (i < nrows) {
            const float* src = mg + L.tid * 4 * KCAP;
            float m[KCAP];
#pragma unroll
            for (int s = 0; s < KCAP; ++s) m[s] = src[s];
            for (int s = KCAP; s < 4 * KCAP; ++s) list_insert<KCAP>(m, src[s]);
            float* out = partial + ((int64_t)chunk * nrows + i) * KCAP;
#pragma unroll
            for (int s = 0; s < KCAP; ++s) out[s] = m[s];
        }
    }
}

template <int KCAP>
static int launch_knn_wide_sample_t(const float* Xb, int64_t N, int64_t ldh, const float* xnorm, int Dh, int stride, int nchunks,
                                    const unsigned* maxn, float* partial, int64_t row0, int64_t nrows, hipStream_t st) {
    AM_HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(&knn_wide_sample_kernel<KCAP>), (int)KNN_SAMPLE_LDS_BYTES));
    const unsigned blocks = (unsigned)(ceil_div(nrows, WTB) * nchunks);
    hipLaunchKernelGGL(knn_wide_sample_kernel<KCAP>, dim3(blocks), dim3(WTHREADS), KNN_SAMPLE_LDS_BYTES, st, Xb, N, ldh, xnorm, Dh, stride,
                       nchunks, maxn, partial, row0, nrows);
    AM_LAUNCH_CHECK();
    return AM_OK;
}

int launch_knn_wide_sample(int kcap, const float* Xb, int64_t N, int64_t ldh, const float* xnorm, int Dh, int stride, int nchunks,
                           const unsigned* maxn, float* partial, int64_t row0, int64_t nrows, hipStream_t st) {
    if (kcap == 6) return launch_knn_wide_sample_t<6>(Xb, N, ldh, xnorm, Dh, stride, nchunks, maxn, partial, row0, nrows, st);
    AM_REQUIRE(kcap == 11, AM_ERR_UNSUPPORTED_K, "the wide sample pass holds lists of 6 or 11 entries (got %d)", kcap);
    return launch_knn_wide_sample_t<11>(Xb, N, ldh, xnorm, Dh, stride, nchunks, maxn, partial, row0, nrows, st);
}

template <int KCAP>
static int launch_knn_wide_t(unsigned nwg, const float* Xb, int64_t N, int64_t ldh, const float* xnorm, float* thr, int Dh,
                             int win_tiles, int nwin, int per_win, int k1, const unsigned* maxn, float* partial, int* cnt, int cap,
                             uint2* wgq, float* wgv, int qcap, int* wgq_count, int part, int nparts, float fc, uint2* ovq,
                             float* ovv, unsigned long long* ovn, int ovcap, const int* skip, int* region_counter, hipStream_t st) {
    AM_HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(&knn_wide_kernel<KCAP>), (int)KNN_WIDE_LDS_BYTES));
#ifdef AM_DEV_KNOBS
    AM_HIP_TRY(set_wide_dev_symbols(st));
#endif
    hipLaunchKernelGGL(knn_wide_kernel<KCAP>, dim3(nwg), dim3(WTHREADS), KNN_WIDE_LDS_BYTES, st, Xb, N, ldh, xnorm, thr, Dh, win_tiles,
                       nwin, per_win, k1, maxn, partial, cnt, cap, wgq, wgv, qcap, wgq_count, part, nparts, fc, ovq, ovv, ovn, ovcap, skip, region_counter);
    AM_LAUNCH_CHECK();
    return AM_OK;
}

int launch_knn_wide(int kcap, unsigned nwg, const float* Xb, int64_t N, int64_t ldh, const float* xnorm, float* thr, int Dh,
                    int win_tiles, int nwin, int per_win, int k1, const unsigned* maxn, float* partial, int* cnt, int cap,
                    uint2* wgq, float* wgv, int qcap, int* wgq_count, int part, int nparts, float fc, uint2* ovq, float* ovv,
                    unsigned long long* ovn, int ovcap, const int* skip, int* region_counter, hipStream_t st) {
    if (kcap == 6)
        return launch_knn_wide_t<6>(nwg, Xb, N, ldh, xnorm, thr, Dh, win_tiles, nwin, per_win, k1, maxn, partial, cnt, cap, wgq, wgv,
                                    qcap, wgq_count, part, nparts, fc, ovq, ovv, ovn, ovcap, skip, region_counter, st);
    AM_REQUIRE(kcap == 11, AM_ERR_UNSUPPORTED_K, "the wide k-NN sweep holds lists of 6 or 11 entries (got %d)", kcap);
    return launch_knn_wide_t<11>(nwg, Xb, N, ldh, xnorm, thr, Dh, win_tiles, nwin, per_win, k1, maxn, partial, cnt, cap, wgq, wgv,
                                 qcap, wgq_count, part, nparts, fc, ovq, ovv, ovn, ovcap, skip, region_counter, st);
}

}  // namespace am
