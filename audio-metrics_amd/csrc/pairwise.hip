// PRDC hot path: k-NN radii (reference prdc.py:4-14) and hypersphere membership
// counts (reference prdc.py:34-48) without ever materialising an N x M distance
// matrix.  Both kernels run the 128x128 f32-MFMA tile engine over
// (row block) x (column chunk) work items and fuse the reduction into the tile
// epilogue, working on SQUARED distances
//     d2(i,j) = max(fma(-2, <x_i,y_j>, |x_i|^2 + |y_j|^2), 0)
// (torch.cdist's matmul form before the sqrt).  The reference's comparisons on
// sqrt'ed values are reproduced exactly through per-row thresholds:
//     sqrt_rn(d2) < R   <=>   d2 < T(R),  T(R) = min { t : sqrt_rn(t) >= R }.
#include "pairwise_common.h"
#include <algorithm>
#include <vector>
#include <cmath>
#include <type_traits>

namespace am {

// ------------------------------------------------------------------ row norms
// |x|^2 in f32 with a fixed order (mirrored by oracle/exact_c): lane l of the
// row's wave fmaf-accumulates elements 4*(64t+l)+c (t = 0,1,..; c = 0..3), then
// a xor-butterfly (32,16,...,1) of plain adds.
__global__ void __launch_bounds__(256) row_sqnorm_kernel(const float* __restrict__ X, int64_t N, int64_t ld,
                                                         int D, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const float* x = X + row * ld;
    float acc = 0.f;
    for (int k = lane * 4; k < D; k += 256) {
        const f32x4 v = load_k4(x, k, D);
        acc = fmaf(v.x, v.x, acc);
        acc = fmaf(v.y, v.y, acc);
        acc = fmaf(v.z, v.z, acc);
        acc = fmaf(v.w, v.w, acc);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc = acc + __shfl_xor(acc, off);
    if (lane == 0) out[row] = acc;
}

// ------------------------------------------------- radius -> squared threshold
__device__ __forceinline__ float sqrt_rn(float x) { return (float)sqrt((double)x); }  // correctly rounded

__device__ __forceinline__ float threshold_of_radius(float R) {
    if (!(R > 0.f)) return 0.f;                     // R == 0 (or NaN): nothing is strictly inside
    if (isinf(R)) return R;
    float c = R * R;
    for (int it = 0; it < 8; ++it) {                // walk down while the predecessor still reaches R
        const float p = __uint_as_float(__float_as_uint(c) - 1u);
        if (c > 0.f && sqrt_rn(p) >= R) c = p; else break;
    }
    for (int it = 0; it < 8; ++it) {                // walk up until sqrt_rn(c) >= R
        if (sqrt_rn(c) < R) c = __uint_as_float(__float_as_uint(c) + 1u); else break;
    }
    return c;
}

__global__ void threshold_kernel(const float* __restrict__ R, int64_t n, float* __restrict__ T) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) T[i] = threshold_of_radius(R[i]);
}

// ------------------------------------------------------------- row sources
struct DenseRows {            // rows base + (tile0 + t) * 128 + local row, zero rows past n
    const float* base;
    int64_t ld, n, tile0;
    __device__ __forceinline__ const float* operator()(int t, int row) const {
        const int64_t g = (tile0 + t) * TB + row;
        return g < n ? base + g * ld : nullptr;
    }
};

struct WorkItem {
    int64_t prow0;     // first P row of this workgroup
    int64_t qtile0;    // first Q tile of this workgroup's column chunk
    int ntiles;
};

__device__ __forceinline__ WorkItem work_item(int64_t q_tiles, int nchunks, int order = 0) {
    int chunk = blockIdx.x % nchunks;              // consecutive blocks (= different XCDs) take different chunks
    int64_t rb = blockIdx.x / nchunks;
    if (order == 1) {                              // experiment: chunk-major (all row blocks of chunk 0 first)
        const int64_t nrb = gridDim.x / nchunks;
        chunk = (int)(blockIdx.x / nrb);
        rb = blockIdx.x % nrb;
    }
    WorkItem w;
    w.prow0 = rb * TB;
    w.qtile0 = q_tiles * chunk / nchunks;
    w.ntiles = (int)(q_tiles * (chunk + 1) / nchunks - w.qtile0);
    return w;
}

// XCD-grouped form of the same (row block, column chunk) grid for the f16 filter kernels, whose operand traffic is
// 16x the exact kernels' per unit of time and therefore has to come from L2: block b runs on XCD b % 8
// (MI355X_MICROARCH.md, workgroup dispatch), and the ~64 workgroups resident on one XCD should share few row blocks
// and few column chunks.  The b / 8-th work item of an XCD is item (b / 8) % 64 of that XCD's (b / 8) / 64-th GROUP of
// grp_rows row blocks x (64 / grp_rows) chunks: a group keeps grp_rows x 128 P rows (f16: 128 KB per block) resident
// in the 4 MB L2 while its chunks stream past, and each Q tile is fetched once per group instead of once per
// workgroup.  Grid: cross_grouped_blocks().  Items past the grid's edge come back with ntiles = 0.
static inline int64_t cross_grouped_blocks(int64_t row_blocks, int nchunks, int grp_rows) {
    const int grp_chunks = 64 / grp_rows;
    const int64_t groups = ceil_div(row_blocks, grp_rows) * ceil_div(nchunks, grp_chunks);
    return ceil_div(groups, 8) * 8 * 64;
}
__device__ __forceinline__ WorkItem work_item_grouped(int64_t q_tiles, int nchunks, int64_t row_blocks, int grp_rows) {
    const int grp_chunks = 64 / grp_rows;
    const int64_t cgroups = (nchunks + grp_chunks - 1) / grp_chunks;
    const int xcd = blockIdx.x & 7;
    const int64_t seq = blockIdx.x >> 3;
    const int64_t g = (seq >> 6) * 8 + xcd;
    const int within = (int)(seq & 63);
    const int64_t rb = (g / cgroups) * grp_rows + within / grp_chunks;
    const int chunk = (int)((g % cgroups) * grp_chunks + within % grp_chunks);
    WorkItem w;
    w.prow0 = rb * TB;
    if (rb >= row_blocks || chunk >= nchunks) {
        w.qtile0 = 0;
        w.ntiles = 0;
        return w;
    }
    w.qtile0 = q_tiles * chunk / nchunks;
    w.ntiles = (int)(q_tiles * (chunk + 1) / nchunks - w.qtile0);
    return w;
}

// the generic (gather) pipeline numbers tiles locally; shift them to absolute Q tile indices
template <class Epi>
struct OffsetEpilogue {
    Epi& e;
    int64_t q0;
    __device__ __forceinline__ void aux_issue(int t, int64_t qt) { e.aux_issue(t, q0 + qt); }
    __device__ __forceinline__ void aux_commit(int t) { e.aux_commit(t); }
    __device__ __forceinline__ void finish(int t, int64_t qt, f32x16 (&acc)[2][2]) { e.finish(t, q0 + qt, acc); }
};

// ------------------------------------------------------------ k-NN epilogue
template <int KCAP>
struct KnnEpilogue {
    const float* qnorm;
    int64_t nq, qtile0;
    float* aux;                 // LDS [2][128] : |y_j|^2 of the tile (+inf past nq)
    float xn[2];
    float best[2][KCAP];
    float aux_reg;
    float dscale = -2.f;        // -2 for f32 operands; -2 / (operand scales) for the scaled f16 filter pre-pass
    const LaneInfo& L;

    __device__ __forceinline__ KnnEpilogue(const LaneInfo& l) : L(l) {}
    __device__ __forceinline__ void aux_issue(int, int64_t qtile) {
        if (L.tid < TB) {
            const int64_t j = qtile * TB + L.tid;
            aux_reg = j < nq ? qnorm[j] : INFINITY;
        }
    }
    __device__ __forceinline__ void aux_commit(int t) {
        if (L.tid < TB) aux[(t & 1) * TB + L.tid] = aux_reg;
    }
    __device__ __forceinline__ void finish(int t, int64_t, f32x16 (&acc)[2][2]) {
        const float* a = aux + (t & 1) * TB + L.wm * 64 + L.h * 4;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f32x4 yn[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) yn[g4] = *reinterpret_cast<const f32x4*>(a + mt * 32 + g4 * 8);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                // max(.,0) commutes with min, so the tile minimum is clamped once
                float tmin = INFINITY;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg)
                    tmin = fminf(tmin, fmaf(dscale, acc[mt][nt][reg], xn[nt] + yn[reg >> 2][reg & 3]));
                tmin = fmaxf(tmin, 0.f);
                // common case after warm-up: no lane of the wave improves its list with this 32x32 tile
                if (__any(tmin < best[nt][KCAP - 1])) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const float d2 = clamp0(fmaf(dscale, acc[mt][nt][reg], xn[nt] + yn[reg >> 2][reg & 3]));
                        if (__any(d2 < best[nt][KCAP - 1])) list_insert<KCAP>(best[nt], d2);
                    }
                }
            }
        }
    }
};

// partial[(chunk * n_rows + row) * KCAP + s] = s-th smallest d2 of `row` inside column chunk `chunk`
template <int KCAP, int V, bool KTAIL>
__global__ void __launch_bounds__(ENGINE_THREADS, 2)
knn_partial_kernel(const float* __restrict__ X, int64_t N, int64_t ldx, const float* __restrict__ xnorm,
                   const float* __restrict__ Y, int64_t M, int64_t ldy, const float* __restrict__ ynorm,
                   int D, int nchunks, int qstride, float* __restrict__ partial, const unsigned* __restrict__ half_scale,
                   const int* __restrict__ run_flag, const int* __restrict__ active_rows, int min_active) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // (launched behind the f16 filter path as its data-dependent fallback: returns at once unless that path gave up)
    if (run_flag != nullptr && *run_flag == 0) return;
    const LaneInfo L;
    // qstride > 1: only every qstride-th column tile is visited (cheap upper bounds for the symmetric kernel)
    const int64_t q_tiles = ((M + TB - 1) / TB + qstride - 1) / qstride;
    const WorkItem w = work_item(q_tiles, nchunks);
    // (batched fix-up of the filter path: X holds *active_rows gathered rows - a device-side count -, the grid was sized for the
    // buffer's capacity; nothing to do for fewer than min_active rows, which take the row-at-a-time kernel)
    if (active_rows != nullptr && (*active_rows < min_active || w.prow0 >= *active_rows)) return;

    KnnEpilogue<KCAP> epi(L);
    epi.qnorm = ynorm;
    epi.nq = M;
    epi.qtile0 = w.qtile0;
    epi.aux = lds + ENGINE_LDS_FLOATS;
    if constexpr ((V & EV_F16) != 0) epi.dscale = half_unscale(half_scale[2], half_scale[3]);   // scaled f16 operands
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int64_t i = w.prow0 + L.wn * 64 + nt * 32 + L.r;
        epi.xn[nt] = i < N ? xnorm[i] : 0.f;
#pragma unroll
        for (int s = 0; s < KCAP; ++s) epi.best[nt][s] = INFINITY;
    }
    if constexpr (V & EV_EARLY) {
        dense_pipeline_early<V, KTAIL>(Y, M, ldy, LinearTiles{w.qtile0, qstride}, X, N, ldx, w.prow0, w.ntiles, D, lds, L, epi);
    } else if constexpr (V & EV_RSRC) {
        dense_pipeline<V>(Y, M, ldy, w.qtile0, X, N, ldx, w.prow0, w.ntiles, D, lds, L, epi);
    } else {
        const DenseRows qsrc{Y, ldy, M, w.qtile0};
        const DenseRows psrc_base{X, ldx, N, w.prow0 / TB};
        auto psrc = [&](int, int row) { return psrc_base(0, row); };
        OffsetEpilogue<decltype(epi)> oe{epi, w.qtile0};
        tile_pipeline(qsrc, psrc, w.ntiles, D, lds, L, oe);
    }

    // merge the 4 lists that cover each P row (2 half-waves x 2 Q-half waves) through LDS
    float* mg = lds;                                   // [128][4][KCAP], staging slabs are free now
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        float* dst = mg + ((L.wn * 64 + nt * 32 + L.r) * 4 + (L.wm * 2 + L.h)) * KCAP;
#pragma unroll
        for (int s = 0; s < KCAP; ++s) dst[s] = epi.best[nt][s];
    }
    __syncthreads();
    if (L.tid < TB) {
        const int64_t i = w.prow0 + L.tid;
        if (i < N) {
            const float* src = mg + L.tid * 4 * KCAP;
            float m[KCAP];
#pragma unroll
            for (int s = 0; s < KCAP; ++s) m[s] = src[s];
            for (int s = KCAP; s < 4 * KCAP; ++s) list_insert<KCAP>(m, src[s]);
            const int chunk = blockIdx.x % nchunks;
            float* out = partial + ((int64_t)chunk * N + i) * KCAP;
#pragma unroll
            for (int s = 0; s < KCAP; ++s) out[s] = m[s];
        }
    }
}

// radius[i] = sqrt_rn( (k+1)-th smallest d2 over all chunks )
template <int KCAP>
__global__ void knn_merge_kernel(const float* __restrict__ partial, int64_t N, int nchunks, int k1, int squared,
                                 float* __restrict__ radii, const int* __restrict__ run_flag,
                                 const int* __restrict__ active_rows = nullptr, int min_active = 0) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N || (run_flag != nullptr && *run_flag == 0)) return;
    if (active_rows != nullptr && (*active_rows < min_active || i >= *active_rows)) return;
    float m[KCAP];
#pragma unroll
    for (int s = 0; s < KCAP; ++s) m[s] = partial[i * KCAP + s];
    for (int c = 1; c < nchunks; ++c) {
        const float* src = partial + ((int64_t)c * N + i) * KCAP;
        for (int s = 0; s < KCAP; ++s) list_insert<KCAP>(m, src[s]);
    }
    float r2 = m[0];
#pragma unroll
    for (int s = 1; s < KCAP; ++s)
        if (s == k1 - 1) r2 = m[s];
    radii[i] = squared ? r2 : sqrt_rn(r2);
}

// ------------------------------------------------------------------------------------------------
// Symmetric k-NN (Y == X).  d2(i,j) == d2(j,i) bit for bit (products commute, same inner order), so only
// the tile pairs (pb, qt) with qt in the CYCLIC HALF-RANGE pb, pb+1, ..., pb+T/2 (mod T) are multiplied.
// Each tile serves two directions:
//   * rows of the P block: per-lane sorted lists, exactly as in the general kernel;
//   * rows of the Q block (the mirrored entries): a value can only matter if it is <= an UPPER BOUND
//     thr[j] of that row's final (k+1)-th smallest d2 (<=, not <: a bound can be exactly tight, and then the
//     entry that defines it may sit in the mirrored half).  Bounds come from a 1/32 column-sample pre-pass of
//     the general kernel and are tightened (atomicMin) by every workgroup that finishes a row block.
//     The few survivors are appended to a per-row candidate buffer; rows whose buffer overflows are
//     recomputed exactly by knn_fixup_kernel.  The final value is the (k+1)-th smallest of a multiset that
//     provably contains every entry below it, so it equals the general kernel's result bit for bit.
template <int KCAP>
struct KnnSymEpilogue {
    const float* qnorm;
    const float* thr;
    int64_t n, pblock;
    float* aux;                 // LDS [2][2][128] : |x_j|^2 and thr[j] of the tile
    float* cand;
    int* cnt;
    int cap;
    uint2* wgq;                 // this workgroup's append region in global memory: (row, d2 bits)
    int* qn;                    // LDS slot counter of that region
    int qcap;
    int ablate;                 // timing experiments only (AM_KNN_SYM_ABL): 1 = no mirrored test, 2 = no lane-local lists
    float xn[2];
    float flt[2];               // bound of this lane's own rows at workgroup start: larger values cannot matter
    bool rowok[2];
    float best[2][KCAP];
    float aux_n, aux_t;
    const LaneInfo& L;

    __device__ __forceinline__ KnnSymEpilogue(const LaneInfo& l) : L(l) {}
    // A surviving mirrored entry.  The slot comes from an LDS atomic (no global round trip) and the store
    // is fire-and-forget; knn_sym_scatter_kernel later files the entries under their rows.  Only when the
    // region is full does the entry go to the row's candidate buffer directly (returning global atomic).
    __device__ __forceinline__ void push(int64_t j, float d2) {
        const int slot = atomicAdd(qn, 1);
        if (slot < qcap) {
            wgq[slot] = make_uint2((unsigned)j, __float_as_uint(d2));
        } else {
            const int s2 = atomicAdd(cnt + j, 1);
            if (s2 < cap) cand[j * cap + s2] = d2;
        }
    }
    __device__ __forceinline__ void aux_issue(int, int64_t qtile) {
        if (L.tid < TB) {
            const int64_t j = qtile * TB + L.tid;
            aux_n = j < n ? qnorm[j] : INFINITY;
            aux_t = j < n ? __hip_atomic_load(thr + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1.f;   // -1: never hit
        }
    }
    __device__ __forceinline__ void aux_commit(int t) {
        if (L.tid < TB) {
            aux[(t & 1) * 2 * TB + L.tid] = aux_n;
            aux[(t & 1) * 2 * TB + TB + L.tid] = aux_t;
        }
    }
    __device__ __forceinline__ void finish(int t, int64_t qtile, f32x16 (&acc)[2][2]) {
        const float* a = aux + (t & 1) * 2 * TB + L.wm * 64 + L.h * 4;
        const bool mirror = (qtile != pblock) && !(ablate & 1);   // the diagonal tile already holds both directions
        const int64_t jbase = qtile * TB + L.wm * 64 + L.h * 4;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f32x4 yn[4], tq[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                yn[g4] = *reinterpret_cast<const f32x4*>(a + mt * 32 + g4 * 8);
                tq[g4] = *reinterpret_cast<const f32x4*>(a + TB + mt * 32 + g4 * 8);
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                // one pass: tile minimum for the lane's own row (clamp commutes with min) and, for the mirrored
                // direction, the smallest margin u - thr[j] (u <= thr  <=>  max(u,0) <= thr since valid thr >= 0;
                // rows past the end carry thr = -1 and |y|^2 = +inf)
                float tmin = INFINITY, marg = INFINITY;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const float u = fmaf(-2.f, acc[mt][nt][reg], xn[nt] + yn[reg >> 2][reg & 3]);
                    tmin = fminf(tmin, u);
                    marg = fminf(marg, u - tq[reg >> 2][reg & 3]);
                }
                tmin = fmaxf(tmin, 0.f);
                // <= : an entry EQUAL to the bound may be the (k+1)-th smallest itself
                if (!(ablate & 2) && __any(tmin < best[nt][KCAP - 1] && tmin <= flt[nt])) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const float d2 = clamp0(fmaf(-2.f, acc[mt][nt][reg], xn[nt] + yn[reg >> 2][reg & 3]));
                        const float v = d2 <= flt[nt] ? d2 : INFINITY;
                        if (__any(v < best[nt][KCAP - 1])) list_insert<KCAP>(best[nt], v);
                    }
                }
                // <= again: if a bound is exactly tight, the entry that defines it may live in the mirrored half
                if (mirror && __any(rowok[nt] && marg <= 0.f)) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const float d2 = clamp0(fmaf(-2.f, acc[mt][nt][reg], xn[nt] + yn[reg >> 2][reg & 3]));
                        if (rowok[nt] && d2 <= tq[reg >> 2][reg & 3])
                            push(jbase + mt * 32 + (reg >> 2) * 8 + (reg & 3), d2);
                    }
                }
            }
        }
    }
};

template <int KCAP, bool KTAIL>
__global__ void __launch_bounds__(ENGINE_THREADS, 2)
knn_sym_kernel(const float* __restrict__ X, int64_t N, int64_t ld, const float* __restrict__ xnorm, float* thr, int D,
               int win_tiles, int nwin, int per_win, int k1, float* __restrict__ partial, float* __restrict__ cand,
               int* __restrict__ cnt, int cap, uint2* __restrict__ wgq, int qcap, int* __restrict__ wgq_count, int ablate,
               int part, int nparts) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const LaneInfo L;
    const int64_t T = (N + TB - 1) / TB;
    const SymWork sw = sym_work(T, win_tiles, nwin, per_win, part, nparts);
    if (sw.ntiles == 0) {                              // nothing of this window belongs to this block
        if (L.tid == 0) wgq_count[blockIdx.x] = 0;
        return;
    }
    const int W = sw.W, ntiles = sw.ntiles;
    const int64_t pb = sw.pb, qa = sw.qa;
    const int chunk = W;                               // partial-list slot

    KnnSymEpilogue<KCAP> epi(L);
    epi.qnorm = xnorm;
    epi.thr = thr;
    epi.n = N;
    epi.pblock = pb;
    epi.aux = lds + ENGINE_LDS_FLOATS;
    epi.cand = cand;
    epi.cnt = cnt;
    epi.cap = cap;
    epi.wgq = wgq + (int64_t)blockIdx.x * qcap;
    epi.qn = reinterpret_cast<int*>(lds + ENGINE_LDS_FLOATS + 4 * TB);
    epi.qcap = qcap;
    epi.ablate = ablate;
    if (L.tid == 0) *epi.qn = 0;                    // visible after the pipeline's first barrier
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int64_t i = pb * TB + L.wn * 64 + nt * 32 + L.r;
        epi.rowok[nt] = i < N;
        epi.xn[nt] = i < N ? xnorm[i] : 0.f;
        epi.flt[nt] = i < N ? __hip_atomic_load(thr + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
#pragma unroll
        for (int s = 0; s < KCAP; ++s) epi.best[nt][s] = INFINITY;
    }
    dense_pipeline_early<EV_DEFAULT, KTAIL>(X, N, ld, LinearTiles{qa}, X, N, ld, pb * TB, ntiles, D, lds, L, epi);
    float* mg = lds;                                   // [128][4][KCAP]
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        float* dst = mg + ((L.wn * 64 + nt * 32 + L.r) * 4 + (L.wm * 2 + L.h)) * KCAP;
#pragma unroll
        for (int s = 0; s < KCAP; ++s) dst[s] = epi.best[nt][s];
    }
    __syncthreads();
    if (L.tid == 0) wgq_count[blockIdx.x] = min(*epi.qn, qcap);
    if (L.tid < TB) {
        const int64_t i = pb * TB + L.tid;
        if (i < N) {
            const float* src = mg + L.tid * 4 * KCAP;
            float m[KCAP];
#pragma unroll
            for (int s = 0; s < KCAP; ++s) m[s] = src[s];
            for (int s = KCAP; s < 4 * KCAP; ++s) list_insert<KCAP>(m, src[s]);
            float* out = partial + ((int64_t)chunk * N + i) * KCAP;
            // write-through (sc1) stores: workgroups on OTHER XCDs read these lists below while this kernel is
            // still running, and a plain store would sit dirty in this XCD's L2 (per-XCD L2s are not coherent)
#pragma unroll
            for (int s = 0; s < KCAP; ++s) __hip_atomic_store(out + s, m[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // Publish a new upper bound for the row: the (k+1)-th smallest over this window AND the lists the
            // higher windows have already written (distinct columns, so their union is a set of true entries;
            // a slot that is still +inf or stale only makes the bound looser, never wrong).
            // (a window's KCAP values are fetched together: written as load-insert-load-insert the agent-scope loads were
            // kept strictly serial - s_waitcnt vmcnt(0) behind each - and every one of them misses L2)
            for (int w2 = W + 1; w2 < nwin; ++w2) {
                const float* src2 = partial + ((int64_t)w2 * N + i) * KCAP;
                float v[KCAP];
#pragma unroll
                for (int s = 0; s < KCAP; ++s) v[s] = __hip_atomic_load(src2 + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int s = 0; s < KCAP; ++s) list_insert<KCAP>(m, v[s]);
            }
            float kth = m[0];
#pragma unroll
            for (int s = 1; s < KCAP; ++s)
                if (s == k1 - 1) kth = m[s];
            atomicMin(reinterpret_cast<unsigned*>(thr) + i, __float_as_uint(kth));
        }
    }
}

// files the per-workgroup append regions under their rows (massively parallel, latency hidden)
__global__ void __launch_bounds__(256) knn_sym_scatter_kernel(const uint2* __restrict__ wgq, int qcap,
                                                              const int* __restrict__ wgq_count, float* __restrict__ cand,
                                                              int* __restrict__ cnt, int cap) {
    const int n = wgq_count[blockIdx.x];
    const uint2* q = wgq + (int64_t)blockIdx.x * qcap;
    for (int e = threadIdx.x; e < n; e += 256) {
        const uint2 v = q[e];
        const int slot = atomicAdd(cnt + v.x, 1);
        if (slot < cap) cand[(int64_t)v.x * cap + slot] = __uint_as_float(v.y);
    }
}

template <int KCAP>
__global__ void knn_sym_merge_kernel(const float* __restrict__ partial, int64_t N, int nchunks, int k1,
                                     const float* __restrict__ cand, const int* __restrict__ cnt, int cap,
                                     float* __restrict__ radii, int* __restrict__ ov_list, int* __restrict__ ov_count,
                                     float* __restrict__ out_lists) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int c = cnt[i];
    if (c > cap) {                                     // candidate buffer overflowed: exact recomputation later
        if (out_lists != nullptr) out_lists[i * KCAP] = NAN;         // partitioned form: flag the row for every rank
        else ov_list[atomicAdd(ov_count, 1)] = (int)i;
        return;
    }
    float m[KCAP];
#pragma unroll
    for (int s = 0; s < KCAP; ++s) m[s] = partial[i * KCAP + s];
    for (int ch = 1; ch < nchunks; ++ch) {
        const float* src = partial + ((int64_t)ch * N + i) * KCAP;
        for (int s = 0; s < KCAP; ++s) list_insert<KCAP>(m, src[s]);
    }
    const float* cs = cand + i * (int64_t)cap;
    for (int s = 0; s < c; ++s) list_insert<KCAP>(m, cs[s]);
    if (out_lists != nullptr) {                        // partitioned form: this rank's KCAP smallest entries of the row
#pragma unroll
        for (int s = 0; s < KCAP; ++s) out_lists[i * KCAP + s] = m[s];
        return;
    }
    float r2 = m[0];
#pragma unroll
    for (int s = 1; s < KCAP; ++s)
        if (s == k1 - 1) r2 = m[s];
    radii[i] = sqrt_rn(r2);
}

// Partitioned form, last step: merge the per-rank lists of every row; a NaN marker from any rank sends the row
// to the exact fix-up kernel.
template <int KCAP>
__global__ void knn_lists_finish_kernel(const float* __restrict__ lists, int nparts, int64_t N, int k1,
                                        float* __restrict__ radii, int* __restrict__ ov_list, int* __restrict__ ov_count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float m[KCAP];
#pragma unroll
    for (int s = 0; s < KCAP; ++s) m[s] = INFINITY;
    bool flagged = false;
    for (int p = 0; p < nparts; ++p) {
        const float* src = lists + ((int64_t)p * N + i) * KCAP;
        if (src[0] != src[0]) { flagged = true; break; }
        for (int s = 0; s < KCAP; ++s) list_insert<KCAP>(m, src[s]);
    }
    if (flagged) {
        ov_list[atomicAdd(ov_count, 1)] = (int)i;
        return;
    }
    float r2 = m[0];
#pragma unroll
    for (int s = 1; s < KCAP; ++s)
        if (s == k1 - 1) r2 = m[s];
    radii[i] = sqrt_rn(r2);
}

// Exact recomputation of single rows (candidate-buffer overflow): one workgroup per row; every thread walks
// columns j = tid, tid+256, ... with the engine's fmaf order (two columns at a time for ILP, float4 loads),
// keeps a sorted list, and the 256 lists are merged pairwise through LDS.
// FEW rows (fewer than the grid has workgroups; round 4): one workgroup streaming all N rows for ONE overflowed row takes
// 10 ms at 100 000 x 512 and 100 ms at 1M - a single straggler row cost more than the whole call (2 flagged rows in 3 calls
// at 200 000 rows: 119 instead of 93 ms per PRDC pass).  The columns of a row are then split among gridDim / rows
// workgroups, each leaves the smallest values of its slice in split_lists, knn_fixup_merge_kernel selects.
__device__ __forceinline__ int fixup_split(int rows, int grid) { return (rows > 0 && rows < grid) ? grid / rows : 1; }

template <int KCAP>
__global__ void __launch_bounds__(256) knn_fixup_kernel(const float* __restrict__ X, int64_t N, int64_t ld,
                                                        const float* __restrict__ xnorm, int D, int k1,
                                                        const int* __restrict__ ov_list, const int* __restrict__ ov_count,
                                                        float* __restrict__ radii, int batched_from, int batched_cap,
                                                        float* __restrict__ split_lists) {
    extern __shared__ __attribute__((aligned(16))) float xrow[];      // D padded to a multiple of 32
    __shared__ float lists[256 * KCAP];
    const int n_ov = *ov_count;
    // batched_from > 0 (filter path): lists of batched_from rows or more are recomputed by the general MFMA kernel on a gathered
    // copy (knn_gather_rows_kernel), up to batched_cap rows - this kernel keeps the short lists and whatever exceeds the copy
    const int first_row = (batched_from > 0 && n_ov >= batched_from) ? (n_ov < batched_cap ? n_ov : batched_cap) : 0;
    const int dp = (D + 31) / 32 * 32;
    auto dot = [&](const float* __restrict__ ya, const float* __restrict__ yb, float& da, float& db) {
        float a = 0.f, b = 0.f;
        for (int c = 0; c < dp; c += 8) {
            const f32x4 a0 = load_k4(ya, c, D), a1 = load_k4(ya, c + 4, D);
            const f32x4 b0 = load_k4(yb, c, D), b1 = load_k4(yb, c + 4, D);
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(xrow + c), x1 = *reinterpret_cast<const f32x4*>(xrow + c + 4);
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {                          // index order 8c+s, 8c+4+s
                a = fmaf(a0[s2], x0[s2], a);
                b = fmaf(b0[s2], x0[s2], b);
                a = fmaf(a1[s2], x1[s2], a);
                b = fmaf(b1[s2], x1[s2], b);
            }
        }
        da = a;
        db = b;
    };
    const int nloc = n_ov - first_row;
    const int split = split_lists != nullptr ? fixup_split(nloc, (int)gridDim.x) : 1;
    for (int64_t item = blockIdx.x; item < (int64_t)nloc * split; item += gridDim.x) {
        const int ov = first_row + (int)(item / split), piece = (int)(item % split);
        const int64_t j_begin = N * piece / split, j_end = N * (piece + 1) / split;
        const int64_t i = ov_list[ov];
        for (int k = threadIdx.x; k < dp; k += 256) xrow[k] = k < D ? X[i * ld + k] : 0.f;
        __syncthreads();
        float m[KCAP];
#pragma unroll
        for (int s = 0; s < KCAP; ++s) m[s] = INFINITY;
        const float xi = xnorm[i];
        for (int64_t j = j_begin + threadIdx.x; j < j_end; j += 512) {
            const int64_t j2 = j + 256;
            float da, db;
            dot(X + j * ld, j2 < j_end ? X + j2 * ld : nullptr, da, db);
            const float d2a = clamp0(fmaf(-2.f, da, xi + xnorm[j]));
            if (d2a < m[KCAP - 1]) list_insert<KCAP>(m, d2a);
            if (j2 < j_end) {
                const float d2b = clamp0(fmaf(-2.f, db, xi + xnorm[j2]));
                if (d2b < m[KCAP - 1]) list_insert<KCAP>(m, d2b);
            }
        }
#pragma unroll
        for (int s = 0; s < KCAP; ++s) lists[threadIdx.x * KCAP + s] = m[s];
        __syncthreads();
        for (int stride = 128; stride >= 1; stride >>= 1) {           // pairwise merge of the sorted lists
            if ((int)threadIdx.x < stride) {
                const float* other = lists + (threadIdx.x + stride) * KCAP;
                for (int s = 0; s < KCAP; ++s) list_insert<KCAP>(m, other[s]);
#pragma unroll
                for (int s = 0; s < KCAP; ++s) lists[threadIdx.x * KCAP + s] = m[s];
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            if (split > 1) {
#pragma unroll
                for (int s = 0; s < KCAP; ++s) split_lists[item * KCAP + s] = m[s];
            } else {
                float r2 = m[0];
#pragma unroll
                for (int s = 1; s < KCAP; ++s)
                    if (s == k1 - 1) r2 = m[s];
                radii[i] = sqrt_rn(r2);
            }
        }
        __syncthreads();
    }
}

// second half of the few-rows form of knn_fixup_kernel (same grid size there): thread r merges the `split` lists of row r
template <int KCAP>
__global__ void __launch_bounds__(256) knn_fixup_merge_kernel(const float* __restrict__ split_lists, int k1, const int* __restrict__ ov_list,
                                                              const int* __restrict__ ov_count, float* __restrict__ radii,
                                                              int batched_from, int batched_cap, int fixup_grid) {
    const int n_ov = *ov_count;
    const int first_row = (batched_from > 0 && n_ov >= batched_from) ? (n_ov < batched_cap ? n_ov : batched_cap) : 0;
    const int nloc = n_ov - first_row, split = fixup_split(nloc, fixup_grid);
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (split <= 1 || r >= nloc) return;
    float m[KCAP];
#pragma unroll
    for (int s = 0; s < KCAP; ++s) m[s] = INFINITY;
    for (int piece = 0; piece < split; ++piece)
        for (int s = 0; s < KCAP; ++s) list_insert<KCAP>(m, split_lists[((int64_t)r * split + piece) * KCAP + s]);
    float r2 = m[0];
#pragma unroll
    for (int s = 1; s < KCAP; ++s)
        if (s == k1 - 1) r2 = m[s];
    radii[ov_list[first_row + r]] = sqrt_rn(r2);
}

constexpr int KNN_FIXUP_GRID = 256;
// both halves; split_lists: KNN_FIXUP_GRID x KCAP floats
template <int KCAP>
static void launch_knn_fixup(const float* X, int64_t N, int64_t ld, const float* xn, int D, int k1, const int* ov_list, const int* ov_count,
                             float* radii, int batched_from, int batched_cap, float* split_lists, hipStream_t st) {
    hipLaunchKernelGGL(knn_fixup_kernel<KCAP>, dim3(KNN_FIXUP_GRID), dim3(256), (size_t)((D + 31) / 32 * 32) * sizeof(float), st, X, N, ld,
                       xn, D, k1, ov_list, ov_count, radii, batched_from, batched_cap, split_lists);
    hipLaunchKernelGGL(knn_fixup_merge_kernel<KCAP>, dim3(1), dim3(256), 0, st, split_lists, k1, ov_list, ov_count, radii,
                       batched_from, batched_cap, KNN_FIXUP_GRID);
}

// --------------------------------------------------------- PRDC counts epilogue
struct CrossEpilogue {
    const float* qnorm;
    const float* qthr;
    int64_t nq, qtile0;
    float* aux;                 // LDS [2][2][128] : |c_j|^2 and T(r_cand[j]) of the tile
    int32_t* col_count;
    float xn[2], tr[2], rmin[2], margin[2];
    float aux_n, aux_t;
    const LaneInfo& L;

    __device__ __forceinline__ CrossEpilogue(const LaneInfo& l) : L(l) {}
    __device__ __forceinline__ void aux_issue(int, int64_t qtile) {
        if (L.tid < TB) {
            const int64_t j = qtile * TB + L.tid;
            aux_n = j < nq ? qnorm[j] : INFINITY;
            aux_t = j < nq ? qthr[j] : 0.f;
        }
    }
    __device__ __forceinline__ void aux_commit(int t) {
        if (L.tid < TB) {
            aux[(t & 1) * 2 * TB + L.tid] = aux_n;
            aux[(t & 1) * 2 * TB + TB + L.tid] = aux_t;
        }
    }
    __device__ __forceinline__ void finish(int t, int64_t qtile, f32x16 (&acc)[2][2]) {
        const float* a = aux + (t & 1) * 2 * TB + L.wm * 64 + L.h * 4;
        const int64_t jbase = qtile * TB + L.wm * 64;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f32x4 yn[4], tc[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                yn[g4] = *reinterpret_cast<const f32x4*>(a + mt * 32 + g4 * 8);
                tc[g4] = *reinterpret_cast<const f32x4*>(a + TB + mt * 32 + g4 * 8);
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                float tmin = INFINITY;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const float d2 = clamp0(fmaf(-2.f, acc[mt][nt][reg], xn[nt] + yn[reg >> 2][reg & 3]));
                    tmin = fminf(tmin, d2);
                    // d2 - T < 0  <=>  d2 < T (IEEE subtraction never rounds across zero); NaN (inf - inf) is dropped
                    margin[nt] = fminf(margin[nt], d2 - tc[reg >> 2][reg & 3]);
                }
                rmin[nt] = fminf(rmin[nt], tmin);
                // wave-uniform and rare: k*N inside-pairs among N*M
                if (__any(tmin < tr[nt])) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const float d2 = clamp0(fmaf(-2.f, acc[mt][nt][reg], xn[nt] + yn[reg >> 2][reg & 3]));
                        const unsigned long long mask = __ballot(d2 < tr[nt]);
                        if (mask != 0ull && L.lane == 0) {
                            const int64_t j = jbase + mt * 32 + (reg >> 2) * 8 + (reg & 3);
                            const int lo = __popcll(mask & 0xffffffffull);
                            const int hi = __popcll(mask >> 32);
                            if (lo) atomicAdd(col_count + j, lo);
                            if (hi) atomicAdd(col_count + j + 4, hi);
                        }
                    }
                }
            }
        }
    }
};

template <int V, bool KTAIL>
__global__ void __launch_bounds__(ENGINE_THREADS, 2)
prdc_cross_kernel(const float* __restrict__ R, int64_t Nr, int64_t ldr, const float* __restrict__ rnorm,
                  const float* __restrict__ rthr,
                  const float* __restrict__ C, int64_t Nc, int64_t ldc, const float* __restrict__ cnorm,
                  const float* __restrict__ cthr, int D, int nchunks,
                  int32_t* __restrict__ col_count, unsigned* __restrict__ row_min_bits,
                  unsigned* __restrict__ row_any, int order, const int* __restrict__ run_flag) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // fallback launch behind the filter-and-verify path (pairwise_fast.h): runs only if that path gave up
    if (run_flag != nullptr && *run_flag == 0) return;
    const LaneInfo L;
    const int64_t q_tiles = (Nc + TB - 1) / TB;
    const WorkItem w = work_item(q_tiles, nchunks, order);

    CrossEpilogue epi(L);
    epi.qnorm = cnorm;
    epi.qthr = cthr;
    epi.nq = Nc;
    epi.qtile0 = w.qtile0;
    epi.aux = lds + ENGINE_LDS_FLOATS;
    epi.col_count = col_count;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int64_t i = w.prow0 + L.wn * 64 + nt * 32 + L.r;
        epi.xn[nt] = i < Nr ? rnorm[i] : 0.f;
        epi.tr[nt] = i < Nr ? rthr[i] : 0.f;         // rows past Nr can never be "inside"
        epi.rmin[nt] = INFINITY;
        epi.margin[nt] = INFINITY;
    }
    if constexpr (V & EV_EARLY) {
        dense_pipeline_early<V, KTAIL>(C, Nc, ldc, LinearTiles{w.qtile0}, R, Nr, ldr, w.prow0, w.ntiles, D, lds, L, epi);
    } else if constexpr (V & EV_RSRC) {
        dense_pipeline<V>(C, Nc, ldc, w.qtile0, R, Nr, ldr, w.prow0, w.ntiles, D, lds, L, epi);
    } else {
        const DenseRows qsrc{C, ldc, Nc, w.qtile0};
        const DenseRows psrc_base{R, ldr, Nr, w.prow0 / TB};
        auto psrc = [&](int, int row) { return psrc_base(0, row); };
        OffsetEpilogue<decltype(epi)> oe{epi, w.qtile0};
        tile_pipeline(qsrc, psrc, w.ntiles, D, lds, L, oe);
    }

#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const float mn = fminf(epi.rmin[nt], __shfl_xor(epi.rmin[nt], 32));
        const float mg = fminf(epi.margin[nt], __shfl_xor(epi.margin[nt], 32));
        const bool any = mg < 0.f;
        const int64_t i = w.prow0 + L.wn * 64 + nt * 32 + L.r;
        if (L.h == 0 && i < Nr) {
            atomicMin(row_min_bits + i, __float_as_uint(mn));    // d2 >= 0: uint order == float order
            if (any) atomicOr(row_any + i, 1u);
        }
    }
}

// row_cover[i] = some candidate lies strictly inside reference row i's ball = (min_j d(i,j) < r_ref[i]), the
// reference's coverage predicate (prdc.py:45-47).  The exact kernel accumulates the row minimum and the flag is
// derived from it; the filter-and-verify path accumulates the flag itself (cov_w) and the minimum only on request.
// use_cov_flag: device flag, non-zero = the exact kernel ran after all (cov_w is void, row_min_bits is valid).
__global__ void prdc_finish_kernel(const unsigned* __restrict__ row_min_bits, const unsigned* __restrict__ row_any_w,
                                   const unsigned* __restrict__ cov_w, const int* __restrict__ exact_ran,
                                   const float* __restrict__ r_ref, int64_t Nr, float* __restrict__ row_min,
                                   uint8_t* __restrict__ row_any, uint8_t* __restrict__ row_cover) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Nr) return;
    const bool from_min = cov_w == nullptr || (exact_ran != nullptr && *exact_ran != 0);
    const float mn = sqrt_rn(__uint_as_float(row_min_bits[i]));
    if (row_min != nullptr) row_min[i] = mn;
    row_any[i] = row_any_w[i] ? 1 : 0;
    row_cover[i] = from_min ? (mn < r_ref[i] ? 1 : 0) : (cov_w[i] ? 1 : 0);
}

__global__ void fill_u32_kernel(unsigned* __restrict__ p, int64_t n, unsigned v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// four integer totals: { #cols count>0, #rows any, sum counts, #rows covered }
__global__ void __launch_bounds__(256) prdc_reduce_kernel(const int32_t* __restrict__ col_count, int64_t Nc,
                                                          const uint8_t* __restrict__ row_any,
                                                          const uint8_t* __restrict__ row_cover, int64_t Nr,
                                                          unsigned long long* __restrict__ out4) {
    unsigned long long v[4] = {0, 0, 0, 0};
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < Nc; j += stride) {
        const int c = col_count[j];
        v[0] += c > 0;
        v[2] += (unsigned long long)c;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < Nr; i += stride) {
        v[1] += row_any[i] != 0;
        v[3] += row_cover[i] != 0;
    }
    __shared__ unsigned long long red[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned long long x = v[q];
        for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off);
        if ((threadIdx.x & 63) == 0) red[q][threadIdx.x >> 6] = x;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const unsigned long long s = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        if (s) atomicAdd(out4 + threadIdx.x, s);
    }
}

// ------------------------------------------------------------------ host side
static int choose_chunks(int64_t p_rows, int64_t q_rows) {
    const int64_t row_blocks = ceil_div(p_rows, TB);
    const int64_t q_tiles = ceil_div(q_rows, TB);
    static const int target = env_int("AM_WG_TARGET", 8192);
    int64_t want = ceil_div(target, row_blocks);         // aim for >= 8192 workgroups: 16 rounds of 256 CUs x 2 keeps the tail ~2%
    if (want < 8) want = 8;                              // one chunk per XCD at least
    want = ceil_div(want, 8) * 8;
    if (want > q_tiles) want = q_tiles;
    if (want > 64) want = 64;
    return (int)(want < 1 ? 1 : want);
}

static int kcap_for(int k1) { return k1 <= 6 ? 6 : k1 <= 11 ? 11 : k1 <= 16 ? 16 : 32; }

constexpr size_t PAIRWISE_LDS_BYTES = (ENGINE_LDS_FLOATS + 4 * TB) * sizeof(float);

static int check_matrix(const float* p, int64_t n, int64_t ld, int D, const char* name) {
    AM_REQUIRE(p != nullptr, AM_ERR_BAD_ARG, "%s is null", name);
    AM_REQUIRE(n >= 1 && D >= 1, AM_ERR_BAD_SHAPE, "%s has shape %lld x %d", name, (long long)n, D);
    AM_REQUIRE(aligned16(p) && ld % 4 == 0 && ld >= D, AM_ERR_BAD_ARG,
               "%s must be 16-byte aligned with ld %% 4 == 0 and ld >= D (ld=%lld, D=%d)", name, (long long)ld, D);
    return AM_OK;
}

static int launch_norms(const float* X, int64_t N, int64_t ld, int D, float* out, hipStream_t st) {
    hipLaunchKernelGGL(row_sqnorm_kernel, dim3((unsigned)ceil_div(N, 4)), dim3(256), 0, st, X, N, ld, D, out);
    AM_LAUNCH_CHECK();
    return AM_OK;
}

template <int KCAP, int V, bool KTAIL>
static int launch_knn_vt(const float* X, int64_t N, int64_t ldx, const float* xn, const float* Y, int64_t M, int64_t ldy,
                         const float* yn, int D, int nchunks, int qstride, float* partial, hipStream_t st,
                         const unsigned* half_scale = nullptr, const int* run_flag = nullptr, const int* active_rows = nullptr,
                         int min_active = 0) {
    {
        AM_HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(&knn_partial_kernel<KCAP, V, KTAIL>), (int)PAIRWISE_LDS_BYTES));
    }
    const int64_t blocks = ceil_div(N, TB) * nchunks;
    const bool clocked = qstride == 1 && run_flag == nullptr && active_rows == nullptr;   // main pass only; qstride > 1 is the sampled pre-pass
    if (clocked) clock_begin(AM_KERNEL_KNN, st);
    hipLaunchKernelGGL((knn_partial_kernel<KCAP, V, KTAIL>), dim3((unsigned)blocks), dim3(ENGINE_THREADS),
                       PAIRWISE_LDS_BYTES, st, X, N, ldx, xn, Y, M, ldy, yn, D, nchunks, qstride, partial, half_scale, run_flag,
                       active_rows, min_active);
    if (clocked) clock_end(AM_KERNEL_KNN, st);
    AM_LAUNCH_CHECK();
    return AM_OK;
}

template <int KCAP, int V>
static int launch_knn_v(const float* X, int64_t N, int64_t ldx, const float* xn, const float* Y, int64_t M, int64_t ldy,
                        const float* yn, int D, int nchunks, int qstride, float* partial, hipStream_t st,
                        const int* run_flag = nullptr, const int* active_rows = nullptr, int min_active = 0) {
    // the inner-dimension tail (D % 32 != 0) is a separate instantiation so the common kernel carries no tail code
    if constexpr ((V & EV_EARLY) != 0) {
        if ((D % BK) != 0)
            return launch_knn_vt<KCAP, V, true>(X, N, ldx, xn, Y, M, ldy, yn, D, nchunks, qstride, partial, st, nullptr, run_flag,
                                                active_rows, min_active);
    }
    return launch_knn_vt<KCAP, V, false>(X, N, ldx, xn, Y, M, ldy, yn, D, nchunks, qstride, partial, st, nullptr, run_flag,
                                         active_rows, min_active);
}

template <int KCAP>
static int launch_knn(const float* X, int64_t N, int64_t ldx, const float* xn, const float* Y, int64_t M, int64_t ldy,
                      const float* yn, int D, int k1, int nchunks, int qstride, bool squared, float* partial,
                      float* out_r, hipStream_t st, const int* run_flag = nullptr, const int* active_rows = nullptr,
                      int min_active = 0) {
    int rc;
    if (run_flag != nullptr || active_rows != nullptr) {   // gated forms behind the filter path: production schedule only
        rc = launch_knn_v<KCAP, EV_DEFAULT>(X, N, ldx, xn, Y, M, ldy, yn, D, nchunks, qstride, partial, st, run_flag, active_rows,
                                            min_active);
        if (rc != AM_OK) return rc;
        hipLaunchKernelGGL(knn_merge_kernel<KCAP>, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, st,
                           partial, N, nchunks, k1, squared ? 1 : 0, out_r, run_flag, active_rows, min_active);
        AM_LAUNCH_CHECK();
        return AM_OK;
    }
#ifdef AM_DEV_KNOBS
    if constexpr (KCAP == 6) {                       // older schedules stay selectable for A/B runs (k <= 5 kernel only)
        switch (qstride == 1 ? engine_variant() : EV_DEFAULT) {
            case 0: rc = launch_knn_v<KCAP, 0>(X, N, ldx, xn, Y, M, ldy, yn, D, nchunks, 1, partial, st); break;
            case 3: rc = launch_knn_v<KCAP, 3>(X, N, ldx, xn, Y, M, ldy, yn, D, nchunks, 1, partial, st); break;
            default: rc = launch_knn_v<KCAP, EV_DEFAULT>(X, N, ldx, xn, Y, M, ldy, yn, D, nchunks, qstride, partial, st); break;
        }
    } else
#endif
    {
        rc = launch_knn_v<KCAP, EV_DEFAULT>(X, N, ldx, xn, Y, M, ldy, yn, D, nchunks, qstride, partial, st);
    }
    if (rc != AM_OK) return rc;
    hipLaunchKernelGGL(knn_merge_kernel<KCAP>, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, st,
                       partial, N, nchunks, k1, squared ? 1 : 0, out_r, static_cast<const int*>(nullptr));
    AM_LAUNCH_CHECK();
    return AM_OK;
}

// ---- symmetric path -------------------------------------------------------------------------------
template <int KCAP>
static int launch_knn_sym(const float* X, int64_t N, int64_t ld, const float* xn, float* thr, int D, int k1, int win_tiles,
                          int nwin, int per_win, float* partial, float* cand, int* cnt, int cap, uint2* wgq, int qcap,
                          int* wgq_count, int* ov_list, int* ov_count, float* out_r, hipStream_t st, int part = 0,
                          int nparts = 1, float* out_lists = nullptr) {
    const unsigned nwg = (unsigned)nwin * (unsigned)per_win;
    // slots of windows that do not touch a row stay +inf
    const int64_t nlist = (int64_t)nwin * N * KCAP;
    hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)ceil_div(nlist, 256)), dim3(256), 0, st,
                       reinterpret_cast<unsigned*>(partial), nlist, 0x7f800000u);
    AM_LAUNCH_CHECK();
    auto launch = [&](auto kernel) -> int {
        AM_HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), (int)PAIRWISE_LDS_BYTES + 16));
        clock_begin(AM_KERNEL_KNN, st);
        hipLaunchKernelGGL(kernel, dim3(nwg), dim3(ENGINE_THREADS), PAIRWISE_LDS_BYTES + 16, st,
                           X, N, ld, xn, thr, D, win_tiles, nwin, per_win, k1, partial, cand, cnt, cap, wgq, qcap, wgq_count,
                           env_int("AM_KNN_SYM_ABL", 0), part, nparts);
        clock_end(AM_KERNEL_KNN, st);
        AM_LAUNCH_CHECK();
        return AM_OK;
    };
    int rc = ((D % BK) != 0) ? launch(&knn_sym_kernel<KCAP, true>) : launch(&knn_sym_kernel<KCAP, false>);
    if (rc != AM_OK) return rc;
    hipLaunchKernelGGL(knn_sym_scatter_kernel, dim3(nwg), dim3(256), 0, st, wgq, qcap, wgq_count, cand, cnt, cap);
    AM_LAUNCH_CHECK();
    hipLaunchKernelGGL(knn_sym_merge_kernel<KCAP>, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, st, partial, N, nwin, k1,
                       cand, cnt, cap, out_r, ov_list, ov_count, out_lists);
    AM_LAUNCH_CHECK();
    if (out_lists != nullptr) return AM_OK;            // partitioned form: fix-up happens after the lists are merged
    // (the per-window lists have been merged: their memory - at least N x KCAP floats - holds the split lists of a few-row fix-up)
    launch_knn_fixup<KCAP>(X, N, ld, xn, D, k1, ov_list, ov_count, out_r, 0, 0, partial, st);
    AM_LAUNCH_CHECK();
    return AM_OK;
}

}  // namespace am

using namespace am;

// ---- k-NN planning (shared by the workspace query and the launcher) ---------------------------------
namespace am { static bool knn_fast_enabled(int64_t N, int D); }   // pairwise_fast.h
struct KnnPlan {
    int tile_rows;              // 128, or 256 when the f16 filter sweep runs on the wide engine
    bool sym;
    int kcap, nchunks;          // main pass
    int pre_chunks, pre_stride; // sampling pre-pass (symmetric path)
    int cap;                    // candidate slots per row (symmetric path)
    int qcap;                   // entries of each workgroup's append region (symmetric path)
    int win_tiles, nwin, per_win;   // symmetric path: column-tile windows and row blocks per window
    int64_t nwg;                    // workgroups of the symmetric sweep (nwin * per_win)
    int pre_windows;                // symmetric path: rows of this many top windows get a sampled bound
};

static KnnPlan plan_knn(int64_t N, int64_t M, int D, int k, bool self, bool partitioned = false) {
    static const int sym_min = env_int("AM_KNN_SYM_MIN_ROWS", 8192);
    static const int sym_min_dim = env_int("AM_KNN_SYM_MIN_DIM", 128);
    static const int cap_env = env_int("AM_KNN_SYM_CAP", 0);
    // Column sample of the bound pre-pass: every `stride`-th tile.  16 up to ~32 000 rows; larger sets keep about eight sampled
    // 256-row tiles (stride up to 48): the pre-pass costs N x N / stride pairs, and a bound from a third of the columns queues
    // 60 % more pairs at almost no cost (the regions grow with the stride) - 100 000 x 512: PRDC pass 25.94 -> 25.54 ms, CLAP-shaped
    // 27.09 -> 26.52; 1M x 512: one k-NN call 650 -> 590 ms (profiles/r4/knn_sample_stride.txt).  Same results for any stride.
    static const int stride_env = env_int("AM_KNN_SYM_STRIDE", 0);
    int stride = 16;
    if (stride_env > 0) stride = stride_env;
    // (A rank of a PARTITIONED run sees an n-th of each row's columns: the bounds its own sweep publishes tighten little, the
    // sampled bound carries the filter - with stride 48 a rank's eighth of the sweep queued 2.4x the pairs, 1.34 -> 1.56 ms.)
    // Narrow rows: the pre-pass's matrix work shrinks with D, what a looser bound costs the sweep's epilogue does not - a denser
    // sample pays (100 000 rows, stride 16 / 24 / 32 / 48: D = 64 2.34 / 2.38 / 2.42 / 2.45 ms, D = 128 2.93 / 2.91 / 2.93 / 2.97,
    // D = 256 4.08 / 4.02 / 4.03 / 4.07, D = 384 5.16 / 5.04 / 5.03 / 5.02; profiles/r5/ab_knn_stride2.txt).
    else if (self && !partitioned && knn_fast_enabled(N, D)) {
        const int64_t cap = D >= 384 ? 48 : D >= 192 ? 32 : D >= 96 ? 24 : 16;
        stride = (int)std::min<int64_t>(cap, std::max<int64_t>(16, ceil_div(N, 256) / 8));
    }
    const int pre_windows = 0;
    KnnPlan p;
    p.tile_rows = TB;
    p.kcap = kcap_for(k + 1);
    // the mirrored-candidate machinery costs per PAIR, the saved MFMA work scales with D: worth it for
    // wide embeddings and enough rows (measured crossover, tools/ab_knn.py)
    p.sym = self && N >= 2 * TB && ((D >= sym_min_dim && N >= sym_min) || knn_fast_enabled(N, D));
    // k + 1 > 16 needs 32 list registers per row and lane: the exact symmetric kernel then spills 1.4 KB per lane and runs
    // 4x SLOWER than the general kernel (measured: 19.7 vs 5.2 ms at 20 000 x 512, 432 vs 91 ms at 100 000 x 512, k = 16 / 20 /
    // 31) - such k take the symmetric plan only where it leads to the f16 filter sweep (whose lists only steer)
    if (p.kcap > 16 && !knn_fast_enabled(N, D)) p.sym = false;
    // survivors per row are ~ (k+1) * (1 + stride/2 ... ) with a heavy tail: 64 slots per list entry
    // (larger sets: the tail of the per-row counts grows with N - 1198 at 1M rows against 196 at 100k, k = 5)
    p.cap = cap_env > 0 ? cap_env : std::max(256, 64 * (k + 1)) * (int)(1 + N / 300000);
    p.pre_stride = stride;
    p.pre_chunks = 1;
    p.qcap = 0;
    p.nwg = 0;
    p.pre_windows = pre_windows;
    if (!p.sym) {
        p.nchunks = choose_chunks(N, M);
        return p;
    }
    // the f16 filter sweep of k <= 10 runs on the 256 x 256 engine: row blocks and column tiles of 256 rows
    static const int wide_on = env_int("AM_KNN_WIDE", 1);
    const bool wide = wide_on != 0 && knn_fast_enabled(N, D) && p.kcap <= KNN_WIDE_MAX_KCAP && N >= 4 * 256;
    if (wide) p.tile_rows = 256;
    const int64_t T = ceil_div(N, p.tile_rows);
    const int64_t sample_tiles = ceil_div(ceil_div(N, TB), stride);          // the sampled pre-pass stays on the 128-row engine
    p.pre_chunks = (int)std::min<int64_t>(sample_tiles, 16);
    // A row's half-range (T/2+1 tiles) is cut into ~`slices` windows; a workgroup publishes the (k+1)-th
    // smallest of ITS window as the row's new bound, so windows much shorter than the pre-pass sample
    // (N/stride columns) would publish nothing useful -> at most 16 slices.
    static const int target128 = env_int("AM_WG_TARGET", 8192);
    static const int target256 = env_int("AM_KNN_WIDE_WG_TARGET", 4096);      // one workgroup per CU there
    const int target = wide ? target256 : target128;
    static const int max_slices = env_int("AM_KNN_SYM_MAX_SLICES", 16);
    int64_t slices = ceil_div(target, T);
    // at least ~11 slices (22 windows) however large the set: the published bounds tighten window by window, and with
    // few, long windows the queues and per-row buffers of the first ones flood (measured at 1M rows)
    slices = std::max<int64_t>(slices, std::max<int64_t>(4, std::min<int64_t>(11, T / 8)));
    slices = std::min<int64_t>(slices, std::min<int64_t>(T / 2, max_slices));
    slices = std::max<int64_t>(slices, 1);
    const int64_t half = T / 2 + 1;
    p.win_tiles = (int)ceil_div(half, slices);
    p.nwin = (int)ceil_div(T, p.win_tiles);
    p.per_win = (int)std::min<int64_t>(T, half + p.win_tiles);   // row blocks that can own a tile of one window
    p.nwg = (int64_t)p.nwin * p.per_win;
    // Which rows need a sampled bound?  A row relies on the bounds its own block published from the windows
    // above it; those exist once the windows that are `in_flight` ahead have finished.  With ~512 resident
    // workgroups (256 CUs x 2) about 512/per_win windows run concurrently: rows in the top in_flight+2 windows
    // (and every row when a window cannot even fill a quarter of the GPU) get the sampled bound.  This is a
    // PERFORMANCE heuristic only: a row that meets a mirrored test with no bound yet floods its candidate
    // buffer, overflows, and is recomputed exactly by knn_fixup_kernel.
    static const int pre_env = env_int("AM_KNN_SYM_PRE_WINDOWS", 0);
    if (pre_env > 0) p.pre_windows = pre_env;
    else if (p.per_win < 256) p.pre_windows = p.nwin;
    else p.pre_windows = std::min<int>(p.nwin, (int)ceil_div(512, p.per_win) + 2);
    p.nchunks = p.nwin;                               // partial-list slots
    // expected survivors per workgroup when only the sample bound is known:
    //   pairs = 128 * win_tiles * 128, hit rate = (k+1) / (N / stride); keep 8x head-room
    const double expect = (double)p.tile_rows * p.tile_rows * p.win_tiles * (double)(k + 1) * stride / (double)N;
    static const int qcap_env = env_int("AM_KNN_SYM_QCAP", 0);
    p.qcap = qcap_env > 0 ? qcap_env : (int)std::min(32768.0, std::max(256.0, 8.0 * expect));   // (the f16 filter path queues both directions)
    p.qcap = (p.qcap + 15) / 16 * 16;                  // the f16 filter sweeps split half of a region evenly among their 4 / 8 waves
    return p;
}

struct KnnBuffers {
    float *xn, *yn, *thr, *partial, *cand;
    int *cnt, *ov_list, *ov_count, *wgq_count;
    uint2* wgq;
};

#include "pairwise_fast.h"      // f16 filter + exact verification forms of the two PRDC kernels

static size_t carve_knn(Carver& c, int64_t N, int64_t M, const KnnPlan& p, KnnBuffers& b) {
    b.xn = c.take<float>(N);
    b.yn = c.take<float>(M);
    const size_t lists = (size_t)std::max(p.nchunks, p.pre_chunks) * N * p.kcap;
    b.partial = c.take<float>(lists);
    if (p.sym) {
        b.thr = c.take<float>(N);
        b.cand = c.take<float>((size_t)N * p.cap);
        b.cnt = c.take<int>(N + 1);               // [N] = overflow counter
        b.ov_list = c.take<int>(N);
        b.ov_count = b.cnt ? b.cnt + N : nullptr;
        const size_t nwg = (size_t)p.nwg;
        b.wgq = c.take<uint2>(nwg * p.qcap);
        b.wgq_count = c.take<int>(nwg * 9);        // the f16 filter sweeps count per wave + the shared part (up to nine parts per workgroup)
    } else {
        b.thr = b.cand = nullptr;
        b.cnt = b.ov_list = b.ov_count = b.wgq_count = nullptr;
        b.wgq = nullptr;
    }
    return c.off;
}

static size_t knn_rows_ws(int64_t N, int64_t M);      // workspace of the row-at-a-time kernel (k > AM_MAX_K), below

extern "C" size_t am_knn_workspace_bytes(int64_t N, int64_t M, int D, int k) {
    if (N < 1 || M < 1 || D < 1 || k < 1 || (int64_t)k + 1 > M) return 0;
    if (k > AM_MAX_K) return knn_rows_ws(N, M);
    // sized for the symmetric path whenever the shapes allow it (the caller may pass Y == X)
    const KnnPlan p = plan_knn(N, M, D, k, N == M);
    Carver c(nullptr, 0);
    KnnBuffers b;
    carve_knn(c, N, M, p, b);
    if (p.sym && knn_fast_enabled(N, D)) carve_knn_fast(c, N, D, p);      // the filter path's own buffers
    return c.off;
}

template <int KCAP>
static int run_knn(const float* X, int64_t N, int64_t ldx, const float* Y, int64_t M, int64_t ldy, int D, int k1,
                   const KnnPlan& p, const KnnBuffers& b, bool self, float* out_r, hipStream_t st) {
    int rc;
    if (!p.sym)
        return launch_knn<KCAP>(X, N, ldx, b.xn, Y, M, ldy, self ? b.xn : b.yn, D, k1, p.nchunks, 1, false, b.partial, out_r,
                                st);
    // 1) upper bounds thr[i] >= final r2[i] (squared domain).  Rows below the top windows need none: by the
    //    time their own window is reached they have published bounds over whole windows above them.  The rows
    //    of the top pre_windows windows (whose half-ranges wrap around to the windows processed last) get the
    //    (k+1)-th smallest over every pre_stride-th column tile from the general kernel.
    const int64_t row_lo = std::max<int64_t>(0, (int64_t)(p.nwin - p.pre_windows) * p.win_tiles * TB);
    hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, st,
                       reinterpret_cast<unsigned*>(b.thr), N, 0x7f800000u);
    AM_LAUNCH_CHECK();
    if (row_lo < N &&
        (rc = launch_knn<KCAP>(X + row_lo * ldx, N - row_lo, ldx, b.xn + row_lo, X, N, ldx, b.xn, D, k1, p.pre_chunks,
                               p.pre_stride, true, b.partial, b.thr + row_lo, st)) != AM_OK)
        return rc;
    AM_HIP_TRY(hipMemsetAsync(b.cnt, 0, (size_t)(N + 1) * sizeof(int), st));
    // 2) half of the tile pairs + mirrored candidates, 3) merge, 4) exact fix-up of overflowed rows
    rc = launch_knn_sym<KCAP>(X, N, ldx, b.xn, b.thr, D, k1, p.win_tiles, p.nwin, p.per_win, b.partial, b.cand, b.cnt, p.cap,
                              b.wgq, p.qcap, b.wgq_count, b.ov_list, b.ov_count, out_r, st);
#ifdef AM_DEV_KNOBS
    static const int debug = env_int("AM_KNN_DEBUG", 0);
    if (rc == AM_OK && debug) {                      // development aid: candidate statistics (synchronises!)
        std::vector<int> cnt(N + 1);
        const size_t nwg = (size_t)p.nwin * p.per_win;
        std::vector<int> wq(nwg);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(cnt.data(), b.cnt, (N + 1) * sizeof(int), hipMemcpyDeviceToHost);
        (void)hipMemcpy(wq.data(), b.wgq_count, nwg * sizeof(int), hipMemcpyDeviceToHost);
        long long tot = 0, mx = 0, wtot = 0, wmx = 0, wfull = 0;
        for (int64_t i = 0; i < N; ++i) { tot += cnt[i]; mx = std::max<long long>(mx, cnt[i]); }
        for (size_t i = 0; i < nwg; ++i) { wtot += wq[i]; wmx = std::max<long long>(wmx, wq[i]); wfull += wq[i] >= p.qcap; }
        {
            const int64_t rows_per_win = (int64_t)p.win_tiles * TB;
            fprintf(stderr, "[am knn sym] mean candidates per row, by window of the row:");
            for (int w = 0; w < p.nwin; ++w) {
                long long t2 = 0, n2 = 0;
                for (int64_t i = w * rows_per_win; i < std::min<int64_t>(N, (w + 1) * rows_per_win); ++i) { t2 += cnt[i]; ++n2; }
                fprintf(stderr, " %.0f", n2 ? (double)t2 / n2 : 0.0);
            }
            fprintf(stderr, "\n");
            std::vector<float> th(N), rr(N);
            (void)hipMemcpy(th.data(), b.thr, N * sizeof(float), hipMemcpyDeviceToHost);
            (void)hipMemcpy(rr.data(), out_r, N * sizeof(float), hipMemcpyDeviceToHost);
            {
                std::vector<float> pl((size_t)p.nwin * N * p.kcap);
                (void)hipMemcpy(pl.data(), b.partial, pl.size() * sizeof(float), hipMemcpyDeviceToHost);
                double acc2 = 0; long long finite = 0;
                for (int64_t i = 0; i < N; ++i) {
                    std::vector<float> all;
                    for (int w = 0; w < p.nwin; ++w)
                        for (int s2 = 0; s2 < p.kcap; ++s2) {
                            const float v = pl[((size_t)w * N + i) * p.kcap + s2];
                            if (std::isfinite(v)) { all.push_back(v); ++finite; }
                        }
                    std::sort(all.begin(), all.end());
                    const float cum = (int)all.size() >= k1 ? all[k1 - 1] : INFINITY;
                    acc2 += (th[i] - cum) / cum;
                }
                fprintf(stderr, "[am knn sym] lane-local lists: %.1f finite entries per row; mean (final bound - cumulative)/cumulative x1e4 = %.2f\n",
                        (double)finite / N, 1e4 * acc2 / N);
            }
            fprintf(stderr, "[am knn sym] mean (final bound - r2)/r2 by window, x1e4:");
            for (int w = 0; w < p.nwin; ++w) {
                double t2 = 0; long long n2 = 0;
                for (int64_t i = w * rows_per_win; i < std::min<int64_t>(N, (w + 1) * rows_per_win); ++i) {
                    const double r2 = (double)rr[i] * rr[i];
                    t2 += (th[i] - r2) / r2; ++n2; }
                fprintf(stderr, " %.1f", n2 ? 1e4 * t2 / n2 : 0.0);
            }
            fprintf(stderr, "\n");
        }
        fprintf(stderr, "[am knn sym] N=%lld k1=%d windows=%d stride=%d cap=%d qcap=%d | candidates/row mean %.1f max %lld | "
                        "overflow rows %d | wg queue mean %.1f max %lld full %lld of %zu\n",
                (long long)N, k1, p.nchunks, p.pre_stride, p.cap, p.qcap, (double)tot / N, mx, cnt[N], (double)wtot / nwg, wmx,
                wfull, nwg);
    }
#endif
    return rc;
}


// ---- nearest_k beyond the per-lane lists (k > AM_MAX_K) -------------------------------------------------------------------
// The tile kernels keep the k+1 smallest values of a row in registers, which stops at 32 slots.  The reference takes any
// k (prdc.py:18: torch.kthvalue on the full distance row), so larger k run here: one workgroup per row (grid-stride) walks
// all M columns with the exact engine's fmaf order - the same values, bit for bit, as the tile kernels' - writes the row
// of squared distances to its own scratch line, and finds the (k+1)-th smallest by a most-significant-byte-first radix
// select over the f32 bit patterns (non-negative values order like their bit patterns; four histogram passes).
// O(N M D) on the vector ALUs instead of the matrix cores: a correctness path for a rarely used argument range
// (AudioMetrics.evaluate caps k at 10, audio_metrics.py:263), not a fast one.
__global__ void __launch_bounds__(256) knn_rows_kernel(const float* __restrict__ X, int64_t N, int64_t ldx,
                                                       const float* __restrict__ xnorm, const float* __restrict__ Y, int64_t M,
                                                       int64_t ldy, const float* __restrict__ ynorm, int D, int k1,
                                                       float* __restrict__ scratch, float* __restrict__ radii) {
    extern __shared__ __attribute__((aligned(16))) float xrow[];      // D padded to a multiple of 32
    __shared__ unsigned hist[256];
    __shared__ unsigned sel_prefix, sel_rank;
    const int dp = (D + 31) / 32 * 32;
    float* line = scratch + (int64_t)blockIdx.x * M;
    for (int64_t i = blockIdx.x; i < N; i += gridDim.x) {
        __syncthreads();
        for (int k = threadIdx.x; k < dp; k += 256) xrow[k] = k < D ? X[i * ldx + k] : 0.f;
        __syncthreads();
        const float xi = xnorm[i];
        for (int64_t j = threadIdx.x; j < M; j += 256) {
            const float* y = Y + j * ldy;
            float a = 0.f;
            for (int c = 0; c < dp; c += 8) {                         // index order 8c+s, 8c+4+s (tile_engine.h)
                const f32x4 a0 = load_k4(y, c, D), a1 = load_k4(y, c + 4, D);
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(xrow + c), x1 = *reinterpret_cast<const f32x4*>(xrow + c + 4);
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    a = fmaf(a0[s2], x0[s2], a);
                    a = fmaf(a1[s2], x1[s2], a);
                }
            }
            line[j] = clamp0(fmaf(-2.f, a, xi + ynorm[j]));
        }
        if (threadIdx.x == 0) { sel_prefix = 0u; sel_rank = (unsigned)(k1 - 1); }      // 0-based rank of the wanted value
        for (int shift = 24; shift >= 0; shift -= 8) {
            hist[threadIdx.x] = 0u;
            __syncthreads();                                          // (also makes the row of distances visible)
            const unsigned prefix = sel_prefix, mask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
            for (int64_t j = threadIdx.x; j < M; j += 256) {
                const unsigned bits = __float_as_uint(line[j]);
                if ((bits & mask) == prefix) atomicAdd(&hist[(bits >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                unsigned rank = sel_rank, bin = 0;
                while (bin < 255u && rank >= hist[bin]) rank -= hist[bin++];
                sel_rank = rank;
                sel_prefix = prefix | (bin << shift);
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) radii[i] = sqrt_rn(__uint_as_float(sel_prefix));
    }
}

static size_t knn_rows_ws(int64_t N, int64_t M) {
    Carver c(nullptr, 0);
    c.take<float>(N);
    c.take<float>(M);
    c.take<float>((size_t)std::min<int64_t>(N, 1024) * M);
    return c.off;
}

static int run_knn_rows(const float* X, int64_t N, int64_t ldx, const float* Y, int64_t M, int64_t ldy, int D, int k1, bool self,
                        float* out_r, void* ws, size_t ws_bytes, hipStream_t st, const PreparedSet* prep) {
    int rc;
    Carver c(ws, ws_bytes);
    float* xn = c.take<float>(N);
    float* yn = c.take<float>(M);
    const int64_t grid = std::min<int64_t>(N, 1024);
    float* scratch = c.take<float>((size_t)grid * M);
    AM_REQUIRE(c.ok(), AM_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", c.off, ws_bytes);
    if ((rc = norms_of(prep, X, N, ldx, D, xn, st)) != AM_OK) return rc;
    if (!self && (rc = launch_norms(Y, M, ldy, D, yn, st)) != AM_OK) return rc;
    hipLaunchKernelGGL(knn_rows_kernel, dim3((unsigned)grid), dim3(256), (size_t)((D + 31) / 32 * 32) * sizeof(float), st, X, N, ldx,
                       xn, Y, M, ldy, self ? xn : yn, D, k1, scratch, out_r);
    AM_LAUNCH_CHECK();
    return AM_OK;
}

static int knn_radii_impl(const float* X, int64_t N, int64_t ldx, const float* Y, int64_t M, int64_t ldy, int D, int k,
                          float* out_r, void* ws, size_t ws_bytes, am_stream_t stream, const PreparedSet* prep) {
    int rc;
    if ((rc = check_matrix(X, N, ldx, D, "X")) != AM_OK) return rc;
    if ((rc = check_matrix(Y, M, ldy, D, "Y")) != AM_OK) return rc;
    AM_REQUIRE(out_r != nullptr, AM_ERR_BAD_ARG, "out_r is null");
    AM_REQUIRE(k >= 1, AM_ERR_BAD_SHAPE, "nearest_k must be >= 1 (got %d)", k);
    AM_REQUIRE((int64_t)k + 1 <= M, AM_ERR_BAD_SHAPE, "k + 1 = %d exceeds the %lld available rows (kthvalue out of range)",
               k + 1, (long long)M);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool self = (Y == X && M == N && ldy == ldx);
    if (k > AM_MAX_K)                              // beyond the per-lane lists of the tile kernels: one row at a time
        return run_knn_rows(X, N, ldx, Y, M, ldy, D, k + 1, self, out_r, ws, ws_bytes, st, prep);
    KnnPlan p = plan_knn(N, M, D, k, self);
    Carver c(ws, ws_bytes);
    KnnBuffers b;
    carve_knn(c, N, M, p, b);
    bool fast = p.sym && knn_fast_enabled(N, D);
    KnnFastBuffers fb{};
    if (fast) fb = carve_knn_fast(c, N, D, p);
    if (!c.ok() && p.sym) {                        // a caller that sized the workspace for Y != X: general path
        p = plan_knn(N, M, D, k, false);
        c = Carver(ws, ws_bytes);
        carve_knn(c, N, M, p, b);
        fast = false;
    }
    AM_REQUIRE(c.ok(), AM_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", c.off, ws_bytes);
    if ((rc = norms_of(prep, X, N, ldx, D, b.xn, st)) != AM_OK) return rc;
    if (!self && (rc = launch_norms(Y, M, ldy, D, b.yn, st)) != AM_OK) return rc;
    const int k1 = k + 1;
    if (fast) {                                    // f16 filter sweep + exact verification (pairwise_fast.h), same bits
        switch (p.kcap) {
            case 6:  return run_knn_fast<6>(X, N, ldx, D, k1, p, b, fb, out_r, st, 0, 1, nullptr, nullptr, prep);
            case 11: return run_knn_fast<11>(X, N, ldx, D, k1, p, b, fb, out_r, st, 0, 1, nullptr, nullptr, prep);
            case 16: return run_knn_fast<16>(X, N, ldx, D, k1, p, b, fb, out_r, st, 0, 1, nullptr, nullptr, prep);
            default: return run_knn_fast<32>(X, N, ldx, D, k1, p, b, fb, out_r, st, 0, 1, nullptr, nullptr, prep);
        }
    }
    switch (p.kcap) {
        case 6:  return run_knn<6>(X, N, ldx, Y, M, ldy, D, k1, p, b, self, out_r, st);
        case 11: return run_knn<11>(X, N, ldx, Y, M, ldy, D, k1, p, b, self, out_r, st);
        case 16: return run_knn<16>(X, N, ldx, Y, M, ldy, D, k1, p, b, self, out_r, st);
        default: return run_knn<32>(X, N, ldx, Y, M, ldy, D, k1, p, b, self, out_r, st);
    }
}

// ---- float64 rows through the f16 filter sweep (declared in am_common.h; called by am_knn_radii_f64) ---------------------------
namespace am {

__global__ void __launch_bounds__(256) cast64_kernel(const double* __restrict__ X, int64_t N, int64_t ld, int D, float* __restrict__ out,
                                                     int64_t ldo) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = idx / ldo;
    const int col = (int)(idx % ldo);
    if (row < N) out[row * ldo + col] = col < D ? (float)X[row * ld + col] : 0.f;
}

// the shapes whose float32 twin takes the filter sweep on the 256-row engine (k <= 10) - and rows that survive the rounding:
// |x| up to ~1e38 and down to the f32 subnormals is checked on the device (unscalable operands raise the fallback flag)
bool knn64_filter_eligible(int64_t N, int D, int k) {
    if (k < 1 || k > 10 || N < 2 || D < 1 || (int64_t)k + 1 > N) return false;
    if (!knn_fast_enabled(N, D)) return false;
    const KnnPlan p = plan_knn(N, N, D, k, true);
    return p.sym && p.tile_rows == WIDE_TILE_ROWS;
}

struct Knn64Layout {
    float* x32 = nullptr;
    int *pair_start = nullptr, *pair_n = nullptr;
    void *ws32 = nullptr, *ws64 = nullptr;
    size_t ws32_bytes = 0, ws64_bytes = 0, total = 0;
    int64_t ld32 = 0;
};
static bool knn64_layout(Carver& c, int64_t N, int D, int k, Knn64Layout& L) {
    L.ld32 = (int64_t)round_up((size_t)D, (size_t)4);
    L.x32 = c.take<float>((size_t)N * L.ld32);
    L.pair_start = c.take<int>((size_t)N);
    L.pair_n = c.take<int>((size_t)N);
    L.ws32_bytes = round_up(am_knn_workspace_bytes(N, N, D, k), (size_t)256);
    L.ws32 = c.take<char>(L.ws32_bytes);
    L.ws64_bytes = round_up(knn64_self_workspace(N, k), (size_t)256);
    L.ws64 = c.take<char>(L.ws64_bytes);
    L.total = c.off;
    return c.ok();
}
size_t knn64_filter_workspace(int64_t N, int D, int k) {
    Carver c(nullptr, 0);
    Knn64Layout L;
    knn64_layout(c, N, D, k, L);
    return L.total;
}

int knn64_filter(const double* X, int64_t N, int64_t ld, int D, int k, double* out_r, void* ws, size_t ws_bytes, hipStream_t st) {
    Carver c(ws, ws_bytes);
    Knn64Layout L;
    AM_REQUIRE(knn64_layout(c, N, D, k, L), AM_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", L.total, ws_bytes);
    hipLaunchKernelGGL(cast64_kernel, dim3((unsigned)ceil_div(N * L.ld32, 256)), dim3(256), 0, st, X, N, ld, D, L.x32, L.ld32);
    AM_LAUNCH_CHECK();
    int rc;
    const KnnPlan p = plan_knn(N, N, D, k, true);
    Carver c32(L.ws32, L.ws32_bytes);
    KnnBuffers b;
    carve_knn(c32, N, N, p, b);
    KnnFastBuffers fb = carve_knn_fast(c32, N, D, p);
    AM_REQUIRE(c32.ok(), AM_ERR_WORKSPACE, "float32 part of the workspace too small: need %zu bytes, have %zu", c32.off, L.ws32_bytes);
    if ((rc = norms_of(nullptr, L.x32, N, L.ld32, D, b.xn, st)) != AM_OK) return rc;
    Knn64Hook h{X, ld, out_r, L.pair_start, L.pair_n, L.ws64, L.ws64_bytes, k};
    if (p.kcap == 6) return run_knn_fast<6>(L.x32, N, L.ld32, D, k + 1, p, b, fb, nullptr, st, 0, 1, nullptr, nullptr, nullptr, &h);
    return run_knn_fast<11>(L.x32, N, L.ld32, D, k + 1, p, b, fb, nullptr, st, 0, 1, nullptr, nullptr, nullptr, &h);
}

__global__ void __launch_bounds__(256) narrow64_kernel(const double* __restrict__ v, int64_t n, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (float)v[i];
}

bool prdc64_filter_eligible(int64_t Nr, int64_t Nc, int D) {
    return Nr >= 1 && Nc >= 1 && cross_fast_enabled(Nr, Nc, D) && plan_cross_fast(Nr, Nc, D).wide;
}

struct Prdc64Layout {
    float *r32 = nullptr, *c32 = nullptr, *rn = nullptr, *cn = nullptr, *rt = nullptr, *ct = nullptr;
    unsigned* rmin = nullptr;
    CrossFastPlan plan{};
    CrossFastBuffers buf{};
    int64_t ld32 = 0;
    size_t total = 0;
};
static bool prdc64_layout(Carver& c, int64_t Nr, int64_t Nc, int D, Prdc64Layout& L) {
    L.ld32 = (int64_t)round_up((size_t)D, (size_t)4);
    L.r32 = c.take<float>((size_t)Nr * L.ld32);
    L.c32 = c.take<float>((size_t)Nc * L.ld32);
    L.rn = c.take<float>(Nr);
    L.rt = c.take<float>(Nr);
    L.cn = c.take<float>(Nc);
    L.ct = c.take<float>(Nc);
    L.rmin = c.take<unsigned>(Nr);
    L.plan = plan_cross_fast(Nr, Nc, D);
    L.buf = carve_cross_fast(c, Nr, Nc, D, L.plan);
    L.total = c.off;
    return c.ok();
}
size_t prdc64_filter_workspace(int64_t Nr, int64_t Nc, int D) {
    Carver c(nullptr, 0);
    Prdc64Layout L;
    prdc64_layout(c, Nr, Nc, D, L);
    return L.total;
}

int prdc64_filter(const double* R, int64_t Nr, int64_t ldr, const double* C, int64_t Nc, int64_t ldc, int D, const double* rt,
                  const double* ct, int32_t* col_count, unsigned* row_any, unsigned* row_cover, const int** fail_flag, void* ws,
                  size_t ws_bytes, hipStream_t st) {
    Carver c(ws, ws_bytes);
    Prdc64Layout L;
    AM_REQUIRE(prdc64_layout(c, Nr, Nc, D, L), AM_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", L.total, ws_bytes);
    hipLaunchKernelGGL(cast64_kernel, dim3((unsigned)ceil_div(Nr * L.ld32, 256)), dim3(256), 0, st, R, Nr, ldr, D, L.r32, L.ld32);
    hipLaunchKernelGGL(cast64_kernel, dim3((unsigned)ceil_div(Nc * L.ld32, 256)), dim3(256), 0, st, C, Nc, ldc, D, L.c32, L.ld32);
    hipLaunchKernelGGL(narrow64_kernel, dim3((unsigned)ceil_div(Nr, 256)), dim3(256), 0, st, rt, Nr, L.rt);
    hipLaunchKernelGGL(narrow64_kernel, dim3((unsigned)ceil_div(Nc, 256)), dim3(256), 0, st, ct, Nc, L.ct);
    hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)ceil_div(Nr, 256)), dim3(256), 0, st, L.rmin, Nr, 0x7f800000u);
    AM_LAUNCH_CHECK();
    int rc;
    if ((rc = norms_of(nullptr, L.r32, Nr, L.ld32, D, L.rn, st)) != AM_OK) return rc;
    if ((rc = norms_of(nullptr, L.c32, Nc, L.ld32, D, L.cn, st)) != AM_OK) return rc;
    const Prdc64Hook h{R, C, ldr, ldc, rt, ct};
    if ((rc = run_cross_fast(L.r32, Nr, L.ld32, L.rn, L.rt, L.c32, Nc, L.ld32, L.cn, L.ct, D, L.plan, L.buf, col_count, L.rmin, row_any,
                             row_cover, false, st, nullptr, nullptr, &h)) != AM_OK)
        return rc;
    *fail_flag = L.buf.ov_count + 1;
    return AM_OK;
}

}  // namespace am

extern "C" int am_knn_radii_f32(const float* X, int64_t N, int64_t ldx, const float* Y, int64_t M, int64_t ldy,
                                int D, int k, float* out_r, void* ws, size_t ws_bytes, am_stream_t stream) {
    return knn_radii_impl(X, N, ldx, Y, M, ldy, D, k, out_r, ws, ws_bytes, stream, nullptr);
}

static bool prepared_ok(const am_prepared_set* p) { return p != nullptr && p->norms != nullptr && p->stats != nullptr && p->half != nullptr; }
static PreparedSet prepared_of(const am_prepared_set* p) { return PreparedSet{p->norms, p->stats, p->half}; }

extern "C" int64_t am_prepared_half_ld(int D) { return D < 1 ? 0 : half_ld(D); }

extern "C" int am_prepare_set_f32(const float* X, int64_t N, int64_t ld, int D, float* norms, uint32_t* stats4, uint16_t* half,
                                  am_stream_t stream) {
    int rc;
    if ((rc = check_matrix(X, N, ld, D, "X")) != AM_OK) return rc;
    AM_REQUIRE(norms && stats4 && half, AM_ERR_BAD_ARG, "null output pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if ((rc = launch_norms(X, N, ld, D, norms, st)) != AM_OK) return rc;
    AM_HIP_TRY(hipMemsetAsync(stats4, 0, 4 * sizeof(unsigned), st));
    return launch_to_half(X, N, ld, D, norms, stats4, 0, half, st);
}

extern "C" int am_knn_radii_prepared_f32(const float* X, int64_t N, int64_t ldx, int D, const am_prepared_set* prepared, int k,
                                         float* out_r, void* ws, size_t ws_bytes, am_stream_t stream) {
    AM_REQUIRE(prepared_ok(prepared), AM_ERR_BAD_ARG, "prepared set has a null member");
    const PreparedSet ps = prepared_of(prepared);
    return knn_radii_impl(X, N, ldx, X, N, ldx, D, k, out_r, ws, ws_bytes, stream, &ps);
}

extern "C" int am_filter_stats_enable(int64_t* device_slots) {
    if (device_slots == nullptr) {
        g_filter_stats = nullptr;
        g_filter_stats_device = -1;
        return AM_OK;
    }
    int dev = -1;
    AM_HIP_TRY(hipGetDevice(&dev));
    g_filter_stats = reinterpret_cast<long long*>(device_slots);
    g_filter_stats_device = dev;
    return AM_OK;
}

// which form am_knn_radii_f32 / am_prdc_counts_f32 take for a shape: 0 exact general, 1 exact symmetric,
// 2 f16 filter + verify on the 128 x 128 engine, 3 the same on the 256 x 256 engine
extern "C" int am_knn_path(int64_t N, int64_t M, int D, int k, int self) {
    if (N < 1 || M < 1 || D < 1 || k < 1) return -1;
    if (k > AM_MAX_K) return 4;                     // row-at-a-time kernel (knn_rows_kernel)
    const KnnPlan p = plan_knn(N, M, D, k, self != 0 && N == M);
    return !p.sym ? 0 : (knn_fast_enabled(N, D) ? (p.tile_rows == 256 ? 3 : 2) : 1);
}
extern "C" int am_prdc_path(int64_t Nr, int64_t Nc, int D) {
    if (Nr < 1 || Nc < 1 || D < 1) return -1;
    if (!cross_fast_enabled(Nr, Nc, D)) return 0;
    return plan_cross_fast(Nr, Nc, D).wide ? 3 : 2;
}

// which of the two 256-row engines multiplies the tiles of a path-3 filter pass for rows of D elements (benchmark support:
// the kernel name a profile shows): 0 wide_engine.h (knn_wide_kernel / cross_wide_kernel), 1 pstat_engine.h
// (knn_pstat_kernel / cross_pstat_kernel)
extern "C" int am_filter_engine(int D) {
    if (D < 1) return 0;
    const int Dh = (int)(half_ld(D) / 2);
    return wide_stationary(Dh) ? (pstat64_supported(Dh) ? 2 : 1) : 0;
}

// ---- partitioned symmetric k-NN (multi-GPU; every rank holds the full set) --------------------------
extern "C" int am_knn_sym_eligible(int64_t N, int D, int k) {
    if (N < 1 || D < 1 || k < 1 || k > AM_MAX_K) return 0;
    return plan_knn(N, N, D, k, true).sym ? 1 : 0;
}

extern "C" int am_knn_list_width(int k) { return (k < 1 || k > AM_MAX_K) ? 0 : kcap_for(k + 1); }

extern "C" size_t am_knn_part_workspace_bytes(int64_t N, int D, int k) {
    if (N < 1 || D < 1 || k < 1 || k > AM_MAX_K) return 0;
    const KnnPlan p = plan_knn(N, N, D, k, true, true);
    Carver c(nullptr, 0);
    KnnBuffers b;
    carve_knn(c, N, N, p, b);
    if (p.sym && knn_fast_enabled(N, D)) carve_knn_fast(c, N, D, p);
    return c.off;
}

static int knn_bounds_impl(const float* X, int64_t N, int64_t ld, int D, int k, int64_t row0, int64_t nrows, float* out_bound_sq,
                           void* ws, size_t ws_bytes, am_stream_t stream, const PreparedSet* prep) {
    int rc;
    if ((rc = check_matrix(X, N, ld, D, "X")) != AM_OK) return rc;
    AM_REQUIRE(out_bound_sq != nullptr, AM_ERR_BAD_ARG, "out_bound_sq is null");
    AM_REQUIRE(k >= 1 && k <= AM_MAX_K, AM_ERR_UNSUPPORTED_K, "nearest_k %d outside [1, %d]", k, AM_MAX_K);
    AM_REQUIRE(row0 >= 0 && nrows >= 1 && row0 + nrows <= N, AM_ERR_BAD_SHAPE, "row range [%lld, +%lld) outside %lld rows",
               (long long)row0, (long long)nrows, (long long)N);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const KnnPlan p = plan_knn(N, N, D, k, true, true);
    AM_REQUIRE(p.sym, AM_ERR_BAD_SHAPE, "the symmetric k-NN path does not apply to %lld x %d (see am_knn_sym_eligible)",
               (long long)N, D);
    Carver c(ws, ws_bytes);
    float* xn = c.take<float>(N);
    float* partial = c.take<float>((size_t)p.pre_chunks * nrows * p.kcap);
    const bool fast = knn_fast_enabled(N, D);
    uint16_t* xb = fast ? c.take<uint16_t>((size_t)N * half_ld(D)) : nullptr;
    unsigned* maxn = fast ? c.take<unsigned>(4) : nullptr;
    AM_REQUIRE(c.ok(), AM_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", c.off, ws_bytes);
    if ((rc = norms_of(prep, X, N, ld, D, xn, st)) != AM_OK) return rc;
    if (fast) {
        // f16 sample pass (pairwise_fast.h): kthA + E_i bounds the (k+1)-th smallest true value of the sample, hence of the row
        AM_HIP_TRY(hipMemsetAsync(maxn, 0, 4 * sizeof(unsigned), st));
        if (prep != nullptr) {
            hipLaunchKernelGGL(prepared_stats_kernel, dim3(1), dim3(64), 0, st, prep->stats, maxn, 0, 2, 3);
        } else {
            if ((rc = launch_to_half(X, N, ld, D, xn, maxn, 0, xb, st)) != AM_OK) return rc;
            AM_HIP_TRY(hipMemcpyAsync(maxn + 3, maxn + 2, sizeof(unsigned), hipMemcpyDeviceToDevice, st));
        }
        const int64_t ldh = half_ld(D) / 2;
        const float* Xb = reinterpret_cast<const float*>(prep != nullptr ? prep->half : xb);
        auto go = [&](auto kcap_tag) -> int {
            constexpr int KC = decltype(kcap_tag)::value;
            int chunks = p.pre_chunks, r2;
            if (p.tile_rows == WIDE_TILE_ROWS && (KC == 6 || KC == 11)) {
                // the shard's rows against every pre_stride-th 256-row tile on the 256-row engine, as on one GPU
                const int64_t samples = ceil_div(ceil_div(N, WIDE_TILE_ROWS), p.pre_stride);
                chunks = std::min(wide_sample_chunks(ceil_div(nrows, WIDE_TILE_ROWS), samples), p.pre_chunks);
                r2 = launch_knn_wide_sample(KC, Xb, N, ldh, xn, (int)ldh, p.pre_stride, chunks, maxn, partial, row0, nrows, st);
            } else {
                r2 = launch_knn_vt<KC, EV_FAST, false>(Xb + row0 * ldh, nrows, ldh, xn + row0, Xb, N, ldh, xn, (int)ldh, p.pre_chunks,
                                                       p.pre_stride, partial, st, maxn);
            }
            if (r2 != AM_OK) return r2;
            hipLaunchKernelGGL(knn_merge_kernel<KC>, dim3((unsigned)ceil_div(nrows, 256)), dim3(256), 0, st, partial, nrows,
                               chunks, k + 1, 1, out_bound_sq, static_cast<const int*>(nullptr));
            hipLaunchKernelGGL(knn_fast_bound_kernel, dim3((unsigned)ceil_div(nrows, 256)), dim3(256), 0, st, out_bound_sq,
                               out_bound_sq, xn + row0, nrows, maxn, fast_c(D));
            AM_LAUNCH_CHECK();
            return AM_OK;
        };
        switch (p.kcap) {
            case 6:  return go(std::integral_constant<int, 6>{});
            case 11: return go(std::integral_constant<int, 11>{});
            case 16: return go(std::integral_constant<int, 16>{});
            default: return go(std::integral_constant<int, 32>{});
        }
    }
    const float* Xr = X + row0 * ld;
    switch (p.kcap) {
        case 6:  return launch_knn<6>(Xr, nrows, ld, xn + row0, X, N, ld, xn, D, k + 1, p.pre_chunks, p.pre_stride, true, partial, out_bound_sq, st);
        case 11: return launch_knn<11>(Xr, nrows, ld, xn + row0, X, N, ld, xn, D, k + 1, p.pre_chunks, p.pre_stride, true, partial, out_bound_sq, st);
        case 16: return launch_knn<16>(Xr, nrows, ld, xn + row0, X, N, ld, xn, D, k + 1, p.pre_chunks, p.pre_stride, true, partial, out_bound_sq, st);
        default: return launch_knn<32>(Xr, nrows, ld, xn + row0, X, N, ld, xn, D, k + 1, p.pre_chunks, p.pre_stride, true, partial, out_bound_sq, st);
    }
}

template <int KCAP>
static int run_knn_part(const float* X, int64_t N, int64_t ld, int D, int k1, int part, int nparts, float* bounds,
                        float* out_lists, const KnnPlan& p, const KnnBuffers& b, hipStream_t st) {
    AM_HIP_TRY(hipMemsetAsync(b.cnt, 0, (size_t)(N + 1) * sizeof(int), st));
    return launch_knn_sym<KCAP>(X, N, ld, b.xn, bounds, D, k1, p.win_tiles, p.nwin, p.per_win, b.partial, b.cand, b.cnt, p.cap,
                                b.wgq, p.qcap, b.wgq_count, b.ov_list, b.ov_count, nullptr, st, part, nparts, out_lists);
}

extern "C" int am_knn_bounds_f32(const float* X, int64_t N, int64_t ld, int D, int k, int64_t row0, int64_t nrows,
                                 float* out_bound_sq, void* ws, size_t ws_bytes, am_stream_t stream) {
    return knn_bounds_impl(X, N, ld, D, k, row0, nrows, out_bound_sq, ws, ws_bytes, stream, nullptr);
}
extern "C" int am_knn_bounds_prepared_f32(const float* X, int64_t N, int64_t ld, int D, const am_prepared_set* prepared, int k,
                                          int64_t row0, int64_t nrows, float* out_bound_sq, void* ws, size_t ws_bytes,
                                          am_stream_t stream) {
    AM_REQUIRE(prepared_ok(prepared), AM_ERR_BAD_ARG, "prepared set has a null member");
    const PreparedSet ps = prepared_of(prepared);
    return knn_bounds_impl(X, N, ld, D, k, row0, nrows, out_bound_sq, ws, ws_bytes, stream, &ps);
}

static int knn_sym_part_impl(const float* X, int64_t N, int64_t ld, int D, int k, int part, int nparts, float* bounds_sq,
                             float* out_lists, void* ws, size_t ws_bytes, am_stream_t stream, const PreparedSet* prep) {
    int rc;
    if ((rc = check_matrix(X, N, ld, D, "X")) != AM_OK) return rc;
    AM_REQUIRE(bounds_sq && out_lists, AM_ERR_BAD_ARG, "null pointer");
    AM_REQUIRE(k >= 1 && k <= AM_MAX_K, AM_ERR_UNSUPPORTED_K, "nearest_k %d outside [1, %d]", k, AM_MAX_K);
    AM_REQUIRE((int64_t)k + 1 <= N, AM_ERR_BAD_SHAPE, "k + 1 exceeds the number of rows");
    AM_REQUIRE(nparts >= 1 && part >= 0 && part < nparts, AM_ERR_BAD_ARG, "part %d of %d", part, nparts);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const KnnPlan p = plan_knn(N, N, D, k, true, true);
    AM_REQUIRE(p.sym, AM_ERR_BAD_SHAPE, "the symmetric k-NN path does not apply to %lld x %d (see am_knn_sym_eligible)",
               (long long)N, D);
    Carver c(ws, ws_bytes);
    KnnBuffers b;
    carve_knn(c, N, N, p, b);
    const bool fast = knn_fast_enabled(N, D);
    KnnFastBuffers fb{};
    if (fast) fb = carve_knn_fast(c, N, D, p);
    AM_REQUIRE(c.ok(), AM_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", c.off, ws_bytes);
    if ((rc = norms_of(prep, X, N, ld, D, b.xn, st)) != AM_OK) return rc;
    if (fast) {                                    // f16 filter sweep of this rank's row blocks + exact verification
        switch (p.kcap) {
            case 6:  return run_knn_fast<6>(X, N, ld, D, k + 1, p, b, fb, nullptr, st, part, nparts, bounds_sq, out_lists, prep);
            case 11: return run_knn_fast<11>(X, N, ld, D, k + 1, p, b, fb, nullptr, st, part, nparts, bounds_sq, out_lists, prep);
            case 16: return run_knn_fast<16>(X, N, ld, D, k + 1, p, b, fb, nullptr, st, part, nparts, bounds_sq, out_lists, prep);
            default: return run_knn_fast<32>(X, N, ld, D, k + 1, p, b, fb, nullptr, st, part, nparts, bounds_sq, out_lists, prep);
        }
    }
    switch (p.kcap) {
        case 6:  return run_knn_part<6>(X, N, ld, D, k + 1, part, nparts, bounds_sq, out_lists, p, b, st);
        case 11: return run_knn_part<11>(X, N, ld, D, k + 1, part, nparts, bounds_sq, out_lists, p, b, st);
        case 16: return run_knn_part<16>(X, N, ld, D, k + 1, part, nparts, bounds_sq, out_lists, p, b, st);
        default: return run_knn_part<32>(X, N, ld, D, k + 1, part, nparts, bounds_sq, out_lists, p, b, st);
    }
}

extern "C" int am_knn_sym_part_f32(const float* X, int64_t N, int64_t ld, int D, int k, int part, int nparts,
                                   float* bounds_sq, float* out_lists, void* ws, size_t ws_bytes, am_stream_t stream) {
    return knn_sym_part_impl(X, N, ld, D, k, part, nparts, bounds_sq, out_lists, ws, ws_bytes, stream, nullptr);
}
extern "C" int am_knn_sym_part_prepared_f32(const float* X, int64_t N, int64_t ld, int D, const am_prepared_set* prepared, int k,
                                            int part, int nparts, float* bounds_sq, float* out_lists, void* ws, size_t ws_bytes,
                                            am_stream_t stream) {
    AM_REQUIRE(prepared_ok(prepared), AM_ERR_BAD_ARG, "prepared set has a null member");
    const PreparedSet ps = prepared_of(prepared);
    return knn_sym_part_impl(X, N, ld, D, k, part, nparts, bounds_sq, out_lists, ws, ws_bytes, stream, &ps);
}

// more flagged rows than the batched fix-up's copy holds: all of them go through the exact general kernel (every row's
// value there is the same bits as the lists'), the list is emptied
__global__ void knn_finish_route_kernel(int* __restrict__ ov_count, int capacity, int* __restrict__ run_exact) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const bool all = *ov_count > capacity;
    *run_exact = all ? 1 : 0;
    if (all) *ov_count = 0;
}

struct ListsFinishBuffers {
    float* xn;
    int *ov_list, *ov_count, *run_exact;
    KnnFixup fix;
    float* xpartial;
};

static ListsFinishBuffers carve_lists_finish(Carver& c, int64_t N, int D, int kcap) {
    ListsFinishBuffers b;
    b.xn = c.take<float>(N);
    b.ov_list = c.take<int>(N + 2);
    b.ov_count = b.ov_list ? b.ov_list + N : nullptr;
    b.run_exact = b.ov_list ? b.ov_list + N + 1 : nullptr;
    b.fix = carve_knn_fixup(c, N, D, kcap);
    b.xpartial = c.take<float>((size_t)choose_chunks(N, N) * N * kcap);
    return b;
}

// Flagged rows (a rank's candidate buffer or queue overflowed for them: blocks of identical rows, ties around a hub): a few -
// one row at a time; up to N / 8 - the batched fix-up on the matrix cores (run_knn_fixup); more - the exact general kernel
// over all rows, behind a device-side flag.  (Round 3 took every flagged row one at a time: 42 us per row at 100 000 x 512.)
template <int KCAP>
static int run_lists_finish(const float* lists, int nparts, const float* X, int64_t N, int64_t ld, int D, int k1,
                            const ListsFinishBuffers& b, float* out_r, hipStream_t st) {
    int rc;
    hipLaunchKernelGGL(knn_lists_finish_kernel<KCAP>, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, st, lists, nparts, N, k1,
                       out_r, b.ov_list, b.ov_count);
    hipLaunchKernelGGL(knn_finish_route_kernel, dim3(1), dim3(64), 0, st, b.ov_count, (int)b.fix.capacity, b.run_exact);
    AM_LAUNCH_CHECK();
    if ((rc = run_knn_fixup<KCAP>(X, N, ld, b.xn, D, k1, b.ov_list, b.ov_count, b.fix, out_r, st)) != AM_OK) return rc;
    return launch_knn<KCAP>(X, N, ld, b.xn, X, N, ld, b.xn, D, k1, choose_chunks(N, N), 1, false, b.xpartial, out_r, st, b.run_exact);
}

extern "C" size_t am_knn_lists_finish_workspace_bytes(int64_t N, int D, int k) {
    if (N < 1 || D < 1 || k < 1 || k > AM_MAX_K) return 0;
    Carver c(nullptr, 0);
    carve_lists_finish(c, N, D, kcap_for(k + 1));
    return c.off;
}

extern "C" int am_knn_lists_finish_f32(const float* lists, int nparts, const float* X, int64_t N, int64_t ld, int D, int k,
                                       float* out_r, void* ws, size_t ws_bytes, am_stream_t stream) {
    int rc;
    if ((rc = check_matrix(X, N, ld, D, "X")) != AM_OK) return rc;
    AM_REQUIRE(lists && out_r, AM_ERR_BAD_ARG, "null pointer");
    AM_REQUIRE(k >= 1 && k <= AM_MAX_K && nparts >= 1, AM_ERR_BAD_ARG, "k=%d nparts=%d", k, nparts);
    hipStream_t st = static_cast<hipStream_t>(stream);
    Carver c(ws, ws_bytes);
    const ListsFinishBuffers b = carve_lists_finish(c, N, D, kcap_for(k + 1));
    AM_REQUIRE(c.ok(), AM_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", c.off, ws_bytes);
    if ((rc = launch_norms(X, N, ld, D, b.xn, st)) != AM_OK) return rc;
    AM_HIP_TRY(hipMemsetAsync(b.ov_count, 0, 2 * sizeof(int), st));
    switch (kcap_for(k + 1)) {
        case 6:  return run_lists_finish<6>(lists, nparts, X, N, ld, D, k + 1, b, out_r, st);
        case 11: return run_lists_finish<11>(lists, nparts, X, N, ld, D, k + 1, b, out_r, st);
        case 16: return run_lists_finish<16>(lists, nparts, X, N, ld, D, k + 1, b, out_r, st);
        default: return run_lists_finish<32>(lists, nparts, X, N, ld, D, k + 1, b, out_r, st);
    }
}

extern "C" size_t am_prdc_workspace_bytes(int64_t Nr, int64_t Nc, int D) {
    if (Nr < 1 || Nc < 1 || D < 1) return 0;
    Carver c(nullptr, 0);
    c.take<float>(Nr); c.take<float>(Nr);          // |r|^2, T(r_ref)
    c.take<float>(Nc); c.take<float>(Nc);          // |c|^2, T(r_cand)
    c.take<unsigned>(Nr); c.take<unsigned>(Nr);    // row_min bits, row_any words
    c.take<unsigned>(Nr);                          // row_cover words
    if (cross_fast_enabled(Nr, Nc, D)) carve_cross_fast(c, Nr, Nc, D, plan_cross_fast(Nr, Nc, D));
    return c.off;
}

static int prdc_counts_impl(const float* R, int64_t Nr, int64_t ldr, const float* C, int64_t Nc, int64_t ldc, int D,
                            const float* r_ref, const float* r_cand, int32_t* out_col_count, uint8_t* out_row_any,
                            uint8_t* out_row_cover, float* out_row_min, void* ws, size_t ws_bytes, am_stream_t stream,
                            const PreparedSet* prep_r, const PreparedSet* prep_c) {
    int rc;
    if ((rc = check_matrix(R, Nr, ldr, D, "R")) != AM_OK) return rc;
    if ((rc = check_matrix(C, Nc, ldc, D, "C")) != AM_OK) return rc;
    AM_REQUIRE(r_ref && r_cand && out_col_count && out_row_any && out_row_cover, AM_ERR_BAD_ARG, "null radius/output pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    Carver c(ws, ws_bytes);
    float* rn = c.take<float>(Nr);
    float* rt = c.take<float>(Nr);
    float* cn = c.take<float>(Nc);
    float* ct = c.take<float>(Nc);
    unsigned* rmin = c.take<unsigned>(Nr);
    unsigned* rany = c.take<unsigned>(Nr);
    unsigned* rcov = c.take<unsigned>(Nr);
    const bool fast = cross_fast_enabled(Nr, Nc, D);
    CrossFastPlan fplan{};
    CrossFastBuffers fbuf{};
    if (fast) {
        fplan = plan_cross_fast(Nr, Nc, D);
        fbuf = carve_cross_fast(c, Nr, Nc, D, fplan);
    }
    AM_REQUIRE(c.ok(), AM_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", c.off, ws_bytes);
    if ((rc = norms_of(prep_r, R, Nr, ldr, D, rn, st)) != AM_OK) return rc;
    if ((rc = norms_of(prep_c, C, Nc, ldc, D, cn, st)) != AM_OK) return rc;
    hipLaunchKernelGGL(threshold_kernel, dim3((unsigned)ceil_div(Nr, 256)), dim3(256), 0, st, r_ref, Nr, rt);
    hipLaunchKernelGGL(threshold_kernel, dim3((unsigned)ceil_div(Nc, 256)), dim3(256), 0, st, r_cand, Nc, ct);
    hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)ceil_div(Nr, 256)), dim3(256), 0, st, rmin, Nr, 0x7f800000u);
    AM_LAUNCH_CHECK();
    AM_HIP_TRY(hipMemsetAsync(rany, 0, (size_t)Nr * sizeof(unsigned), st));
    AM_HIP_TRY(hipMemsetAsync(rcov, 0, (size_t)Nr * sizeof(unsigned), st));
    AM_HIP_TRY(hipMemsetAsync(out_col_count, 0, (size_t)Nc * sizeof(int32_t), st));
    const int* run_flag = nullptr;
    if (fast) {
        // f16 filter pass + exact verification of the queued pairs (pairwise_fast.h): same outputs, bit for bit.
        // The exact kernel is still launched behind it, but its workgroups return at once unless the filter path
        // raised its device-side fail flag (both queues overflowed).
        if ((rc = run_cross_fast(R, Nr, ldr, rn, rt, C, Nc, ldc, cn, ct, D, fplan, fbuf, out_col_count, rmin, rany, rcov,
                                 out_row_min != nullptr, st, prep_r, prep_c)) != AM_OK)
            return rc;
        run_flag = fbuf.ov_count + 1;
    }
    const int nchunks = choose_chunks(Nr, Nc);
    const int64_t blocks = ceil_div(Nr, TB) * nchunks;
    auto launch = [&](auto kernel) -> int {
        AM_HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), (int)PAIRWISE_LDS_BYTES));
        if (!fast) clock_begin(AM_KERNEL_PRDC_CROSS, st);
        hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(ENGINE_THREADS), PAIRWISE_LDS_BYTES, st, R, Nr, ldr, rn, rt,
                           C, Nc, ldc, cn, ct, D, nchunks, out_col_count, rmin, rany, env_int("AM_CROSS_ORDER", 0), run_flag);
        if (!fast) clock_end(AM_KERNEL_PRDC_CROSS, st);
        AM_LAUNCH_CHECK();
        return AM_OK;
    };
#ifdef AM_DEV_KNOBS
    if (engine_variant() == 0) rc = launch(&prdc_cross_kernel<0, false>);
    else if ((D % BK) != 0) rc = launch(&prdc_cross_kernel<EV_DEFAULT, true>);
    else if (engine_variant() == (EV_DEFAULT | EV_LDS)) rc = launch(&prdc_cross_kernel<EV_DEFAULT | EV_LDS, false>);
    else rc = launch(&prdc_cross_kernel<EV_DEFAULT, false>);
#else
    if ((D % BK) != 0) rc = launch(&prdc_cross_kernel<EV_DEFAULT, true>);
    else rc = launch(&prdc_cross_kernel<EV_DEFAULT, false>);
#endif
    if (rc != AM_OK) return rc;
    hipLaunchKernelGGL(prdc_finish_kernel, dim3((unsigned)ceil_div(Nr, 256)), dim3(256), 0, st, rmin, rany,
                       fast ? rcov : static_cast<unsigned*>(nullptr), run_flag, r_ref, Nr, out_row_min, out_row_any, out_row_cover);
    AM_LAUNCH_CHECK();
    return AM_OK;
}

extern "C" int am_prdc_counts_f32(const float* R, int64_t Nr, int64_t ldr, const float* C, int64_t Nc, int64_t ldc,
                                  int D, const float* r_ref, const float* r_cand, int32_t* out_col_count,
                                  uint8_t* out_row_any, uint8_t* out_row_cover, float* out_row_min, void* ws,
                                  size_t ws_bytes, am_stream_t stream) {
    return prdc_counts_impl(R, Nr, ldr, C, Nc, ldc, D, r_ref, r_cand, out_col_count, out_row_any, out_row_cover, out_row_min, ws,
                            ws_bytes, stream, nullptr, nullptr);
}
extern "C" int am_prdc_counts_prepared_f32(const float* R, int64_t Nr, int64_t ldr, const am_prepared_set* prepared_r, const float* C,
                                           int64_t Nc, int64_t ldc, const am_prepared_set* prepared_c, int D, const float* r_ref,
                                           const float* r_cand, int32_t* out_col_count, uint8_t* out_row_any,
                                           uint8_t* out_row_cover, float* out_row_min, void* ws, size_t ws_bytes,
                                           am_stream_t stream) {
    AM_REQUIRE(prepared_ok(prepared_r) && prepared_ok(prepared_c), AM_ERR_BAD_ARG, "prepared set has a null member");
    const PreparedSet pr = prepared_of(prepared_r), pc = prepared_of(prepared_c);
    return prdc_counts_impl(R, Nr, ldr, C, Nc, ldc, D, r_ref, r_cand, out_col_count, out_row_any, out_row_cover, out_row_min, ws,
                            ws_bytes, stream, &pr, &pc);
}

extern "C" int am_prdc_reduce(const int32_t* col_count, int64_t Nc, const uint8_t* row_any, const uint8_t* row_cover,
                              int64_t Nr, int64_t* out4, am_stream_t stream) {
    AM_REQUIRE(col_count && out4 && (Nr == 0 || (row_any && row_cover)), AM_ERR_BAD_ARG, "null pointer");
    AM_REQUIRE(Nc >= 1 && Nr >= 0, AM_ERR_BAD_SHAPE, "empty input");     // Nr == 0: column totals only (multi-GPU second pass)
    hipStream_t st = static_cast<hipStream_t>(stream);
    AM_HIP_TRY(hipMemsetAsync(out4, 0, 4 * sizeof(int64_t), st));
    const int64_t n = Nc > Nr ? Nc : Nr;
    int blocks = (int)ceil_div(n, 256 * 8);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(prdc_reduce_kernel, dim3(blocks), dim3(256), 0, st, col_count, Nc, row_any, row_cover, Nr,
                       reinterpret_cast<unsigned long long*>(out4));
    AM_LAUNCH_CHECK();
    return AM_OK;
}
