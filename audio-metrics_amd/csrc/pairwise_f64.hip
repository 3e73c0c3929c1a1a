// float64 forms of the pairwise kernels: k-NN radii, hypersphere membership counts, kernel distance.
//
// The reference computes every stage in the dtype of the rows it is given (data.py:39-44, prdc.py:12-13,34-48,
// kd.py:112-116), and float64 rows are what its PCA projection hands on (projection.py:20-21: scikit-learn's float64
// product; audio_metrics.py:163-182) - the path of every reference test and of examples/2_musdb.py - as well as what its
// own test embedder yields.  Until round 5 such rows were narrowed to f32 for KD / PRDC; these entry points keep them f64.
//
// One tile engine on the f64 matrix cores (v_mfma_f64_16x16x4_f64: 2048 flop per instruction, 64 cycles per SIMD - it is
// the instruction, not its operands, that bounds these kernels, so the operand path is kept simple):
//   * workgroup = 256 threads = 4 wave64, tile = 64 "P" rows x 64 "Q" rows; wave w owns P rows 16 w .. 16 w + 15 as the
//     MFMA B operand (column = lane & 15: everything a lane accumulates belongs to ONE P row - per-row lists, flags and
//     minima need no cross-lane traffic in the tile loop) against all 64 Q rows (4 A tiles, register r of tile m = Q row
//     16 m + (lane >> 4) + 4 r): 4 accumulators of 4 doubles;
//   * operands go through LDS in 16-element slabs, row stride 17 doubles (conflict-free ds_read_b64 for the fragment
//     pattern row = lane & 15, element = 4 s + (lane >> 4)), two stages, the global loads of slab s + 1 in flight under the
//     MFMAs of slab s, one barrier per slab; rows need no alignment (8-byte loads);
//   * dot(i, j) = the MFMA chain over the inner index, d2(i, j) = max(fma(-2, dot, |x_i|^2 + |y_j|^2), 0) - torch.cdist's
//     matmul form in f64 (prdc.py:12,34); a NaN distance (non-finite row) is carried as +inf (am_common.h: clamp0);
//     d2(i, j) and d2(j, i) are bit-identical (products commute, same inner order).
// Radii = sqrt_rn of the (k+1)-th smallest d2 (prdc.py:13); memberships compare d2 with T(R) = min{t : sqrt_rn(t) >= R},
// which is exactly the reference's strict `sqrt(d2) < R` (prdc.py:36-47).
#include "am_common.h"
#include <algorithm>

namespace am {

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int FT = 64;                 // tile rows of either operand
constexpr int FK = 16;                 // inner-dimension slab
constexpr int FLD = 17;                // padded LDS row stride (doubles)
constexpr int FTILE = FT * FLD;        // one operand slab
constexpr int FENGINE_DOUBLES = 4 * FTILE;   // two stages x (Q slab, P slab) = 34 816 bytes
constexpr int FTHREADS = 256;

__device__ __forceinline__ double clamp0d(double v) { return v != v ? __builtin_inf() : fmax(v, 0.0); }

struct FLane {
    int tid, lane, wave, l15, l4;
    __device__ __forceinline__ FLane() {
        tid = threadIdx.x;
        lane = tid & 63;
        wave = tid >> 6;
        l15 = lane & 15;
        l4 = lane >> 4;
    }
    __device__ __forceinline__ int prow() const { return wave * 16 + l15; }                 // P row of the tile this lane owns
    __device__ __forceinline__ int qrow(int m, int r) const { return m * 16 + l4 + 4 * r; }   // Q row behind acc[m][r]
};

// rows base + (row0 + local) of a dense matrix, nullptr (= a zero row) past n
struct DenseRows64 {
    const double* base;
    int64_t ld, n, row0;
    __device__ __forceinline__ const double* operator()(int row) const {
        const int64_t g = row0 + row;
        return g < n ? base + g * ld : nullptr;
    }
};
// rows gathered through an index list: local row -> X[idx[pos0 + row]], zero rows past m
struct GatherRows64 {
    const double* base;
    int64_t ld;
    const int64_t* idx;
    int m, pos0;
    __device__ __forceinline__ const double* operator()(int row) const {
        const int p = pos0 + row;
        return p < m ? base + idx[p] * ld : nullptr;
    }
};

// acc[m][r] += <Q row 16 m + l4 + 4 r, P row 16 wave + l15> over the D inner elements.  All 256 threads take part.
// The caller's LDS writes before this call (side data of the tile) become visible at the first barrier inside.
template <class QRows, class PRows>
__device__ __forceinline__ void f64_tile(const QRows& qrows, const PRows& prows, int D, double* __restrict__ lds, const FLane& L,
                                         f64x4 (&acc)[4]) {
    const int srow = L.tid >> 2, scol = (L.tid & 3) * 4;             // this thread stages 4 elements of one row per operand
    const double* qsrc = qrows(srow);
    const double* psrc = prows(srow);
    const int nslabs = (D + FK - 1) / FK;
    double qv[4], pv[4];
    auto fetch = [&](int s) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = s * FK + scol + j;
            qv[j] = (qsrc != nullptr && k < D) ? qsrc[k] : 0.0;
            pv[j] = (psrc != nullptr && k < D) ? psrc[k] : 0.0;
        }
    };
    fetch(0);
    for (int s = 0; s < nslabs; ++s) {
        double* st = lds + (s & 1) * 2 * FTILE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            st[srow * FLD + scol + j] = qv[j];
            st[FTILE + srow * FLD + scol + j] = pv[j];
        }
        __syncthreads();
        if (s + 1 < nslabs) fetch(s + 1);
        const double* q = st + L.l15 * FLD + L.l4;
        const double* p = st + FTILE + (L.wave * 16 + L.l15) * FLD + L.l4;
#pragma unroll
        for (int ks = 0; ks < FK / 4; ++ks) {
            const double b = p[ks * 4];
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(q[m * 16 * FLD + ks * 4], b, acc[m], 0, 0, 0);
        }
        // (no second barrier: the next slab goes to the other stage, which every wave finished reading before it arrived at
        // the barrier above)
    }
    __syncthreads();                                                 // the stages are free for the next tile / the caller
}

__device__ __forceinline__ void zero4(f64x4 (&acc)[4]) {
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = f64x4{0.0, 0.0, 0.0, 0.0};
}

// ascending list of the CAP smallest values seen (a value >= best[CAP-1] falls through)
template <int CAP>
__device__ __forceinline__ void list_insert64(double (&best)[CAP], double x) {
#pragma unroll
    for (int i = 0; i < CAP; ++i) {
        const double lo = fmin(best[i], x);
        x = fmax(best[i], x);
        best[i] = lo;
    }
}

// ---- squared row norms (one wave per row, fixed order: lane-strided partial sums, butterfly)
__global__ void __launch_bounds__(256) row_sqnorm64_kernel(const double* __restrict__ X, int64_t N, int64_t ld, int D, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const double* x = X + row * ld;
    double s = 0.0;
    for (int k = lane; k < D; k += 64) s = fma(x[k], x[k], s);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) out[row] = s;
}

// ---- radius -> squared threshold: T(R) = min{t : sqrt_rn(t) >= R}  (so that  sqrt_rn(d2) < R  <=>  d2 < T(R))
__device__ __forceinline__ double threshold_of_radius64(double R) {
    if (!(R > 0.0)) return R == R ? 0.0 : R;                       // R <= 0: nothing is closer (d >= 0); NaN stays NaN (compares false)
    if (isinf(R)) return R;
    double c = R * R;
    if (isinf(c)) c = 1.7976931348623157e308;
    if (c < 1.0e-290) {
        // R * R in or near the subnormal range (R below ~1e-145: radii of near-duplicate rows): its rounding error is many ulps
        // of the result and the 8-step walk below could not reach T(R).  sqrt_rn is monotone and, for 0 < R < 1, sqrt_rn(R) >= R:
        // bisect the bit pattern of t over [0, R] (positive doubles order like their bit patterns).
        long long lo = 0, hi = __double_as_longlong(R);            // sqrt(lo) = 0 < R <= sqrt(hi)
        while (hi - lo > 1) {
            const long long mid = lo + (hi - lo) / 2;
            if (__dsqrt_rn(__longlong_as_double(mid)) >= R) hi = mid; else lo = mid;
        }
        return __longlong_as_double(hi);
    }
    for (int it = 0; it < 8; ++it) {                               // walk down while the predecessor still reaches R
        const double p = __longlong_as_double(__double_as_longlong(c) - 1);
        if (c > 0.0 && __dsqrt_rn(p) >= R) c = p; else break;
    }
    for (int it = 0; it < 8; ++it) {                               // walk up until sqrt_rn(c) >= R
        if (__dsqrt_rn(c) < R) c = __longlong_as_double(__double_as_longlong(c) + 1); else break;
    }
    return c;
}
__global__ void threshold64_kernel(const double* __restrict__ R, int64_t n, double* __restrict__ T) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) T[i] = threshold_of_radius64(R[i]);
}

// =====================================================================================================================
// A9  k-NN radii.  Grid = (64-row blocks of X) x nchunks column chunks; partial[(chunk * N + row) * KCAP + s] = the chunk's
// smallest values of the row, ascending, the first KCAP - (k+1) slots -inf pads (so slot KCAP-1 is the (k+1)-th smallest).
template <int KCAP>
__global__ void __launch_bounds__(FTHREADS) knn64_kernel(const double* __restrict__ X, int64_t N, int64_t ldx, const double* __restrict__ xn,
                                                         const double* __restrict__ Y, int64_t M, int64_t ldy, const double* __restrict__ yn,
                                                         int D, int k1, int nchunks, double* __restrict__ partial,
                                                         const int* __restrict__ run_flag) {
    extern __shared__ __attribute__((aligned(16))) double lds64[];
    if (run_flag != nullptr && *run_flag == 0) return;               // (behind the f16 filter route: only when that gave up)
    double* qnorm = lds64 + FENGINE_DOUBLES;                          // [64] norms of the Q tile
    const FLane L;
    const int64_t pb = blockIdx.x / nchunks;
    const int chunk = blockIdx.x % nchunks;
    const int64_t qtiles = (M + FT - 1) / FT;
    const int64_t t0 = qtiles * chunk / nchunks, t1 = qtiles * (chunk + 1) / nchunks;
    const int64_t prow = pb * FT + L.prow();
    const double pn = prow < N ? xn[prow] : __builtin_inf();
    double best[KCAP];
#pragma unroll
    for (int s = 0; s < KCAP; ++s) best[s] = s < KCAP - k1 ? -__builtin_inf() : __builtin_inf();
    const DenseRows64 prows{X, ldx, N, pb * FT};
    for (int64_t t = t0; t < t1; ++t) {
        if (L.tid < FT) {
            const int64_t j = t * FT + L.tid;
            qnorm[L.tid] = j < M ? yn[j] : __builtin_inf();          // past the end: +inf, never among the smallest
        }
        f64x4 acc[4];
        zero4(acc);
        f64_tile(DenseRows64{Y, ldy, M, t * FT}, prows, D, lds64, L, acc);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double d2 = clamp0d(fma(-2.0, acc[m][r], pn + qnorm[L.qrow(m, r)]));
                if (__any(d2 < best[KCAP - 1])) list_insert64<KCAP>(best, d2);
            }
        __syncthreads();                                             // qnorm is rewritten by the next tile
    }
    // the four lanes of a P row (l4 = 0..3) hold lists over disjoint columns: merge through LDS
    double* mg = lds64;                                              // [64][4][KCAP]
    double* dst = mg + ((size_t)L.prow() * 4 + L.l4) * KCAP;
#pragma unroll
    for (int s = 0; s < KCAP; ++s) dst[s] = best[s];
    __syncthreads();
    if (L.tid < FT) {
        const int64_t i = pb * FT + L.tid;
        if (i < N) {
            const double* src = mg + (size_t)L.tid * 4 * KCAP;
            double m[KCAP];
#pragma unroll
            for (int s = 0; s < KCAP; ++s) m[s] = src[s];
            for (int s = KCAP; s < 4 * KCAP; ++s)
                if (src[s] > -__builtin_inf()) list_insert64<KCAP>(m, src[s]);
            double* out = partial + ((int64_t)chunk * N + i) * KCAP;
#pragma unroll
            for (int s = 0; s < KCAP; ++s) out[s] = m[s];
        }
    }
}

template <int KCAP>
__global__ void __launch_bounds__(256) knn64_merge_kernel(const double* __restrict__ partial, int64_t N, int nchunks, double* __restrict__ out_r,
                                                          const int* __restrict__ run_flag) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N || (run_flag != nullptr && *run_flag == 0)) return;
    double m[KCAP];
#pragma unroll
    for (int s = 0; s < KCAP; ++s) m[s] = partial[i * KCAP + s];
    for (int c = 1; c < nchunks; ++c) {
        const double* src = partial + ((int64_t)c * N + i) * KCAP;
        for (int s = 0; s < KCAP; ++s)
            if (src[s] > -__builtin_inf()) list_insert64<KCAP>(m, src[s]);
    }
    out_r[i] = __dsqrt_rn(m[KCAP - 1]);
}

// k + 1 > 32 (the reference takes any k, prdc.py:18; its evaluate() caps k at 10): the squared distances of a block of rows
// are written out and the (k+1)-th smallest of each row is found by a radix select over the 64-bit patterns (d2 >= 0, so
// the unsigned order of the bits is the order of the values).  A correctness path.
__global__ void __launch_bounds__(FTHREADS) dist64_block_kernel(const double* __restrict__ X, int64_t N, int64_t ldx, const double* __restrict__ xn,
                                                                int64_t row0, int64_t nrows, const double* __restrict__ Y, int64_t M,
                                                                int64_t ldy, const double* __restrict__ yn, int D, double* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double lds64[];
    const FLane L;
    const int64_t qtiles = (M + FT - 1) / FT;
    const int64_t pb = blockIdx.x / qtiles, t = blockIdx.x % qtiles;
    const int64_t prel = pb * FT + L.prow();
    const double pn = prel < nrows ? xn[row0 + prel] : 0.0;
    f64x4 acc[4];
    zero4(acc);
    f64_tile(DenseRows64{Y, ldy, M, t * FT}, DenseRows64{X, ldx, row0 + nrows, row0 + pb * FT}, D, lds64, L, acc);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t j = t * FT + L.qrow(m, r);
            if (prel < nrows && j < M) out[prel * M + j] = clamp0d(fma(-2.0, acc[m][r], pn + yn[j]));
        }
}

__global__ void __launch_bounds__(256) select64_kernel(const double* __restrict__ d2, int64_t M, int k1, double* __restrict__ out_r, int64_t row0) {
    __shared__ unsigned hist[256];
    __shared__ unsigned long long prefix_s;
    __shared__ unsigned rank_s;
    const unsigned long long* v = reinterpret_cast<const unsigned long long*>(d2 + (int64_t)blockIdx.x * M);
    unsigned long long prefix = 0ull;
    unsigned rank = (unsigned)k1;                                   // the rank-th smallest (1-based) among the values matching `prefix`
    for (int byte = 7; byte >= 0; --byte) {
        hist[threadIdx.x] = 0u;
        __syncthreads();
        const unsigned long long mask = byte == 7 ? 0ull : (~0ull << ((byte + 1) * 8));
        for (int64_t j = threadIdx.x; j < M; j += 256) {
            const unsigned long long b = v[j];
            if ((b & mask) == prefix) atomicAdd(&hist[(b >> (byte * 8)) & 255u], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned acc = 0u, d = 0u;
            for (; d < 256u; ++d) {
                if (acc + hist[d] >= rank) break;
                acc += hist[d];
            }
            prefix_s = prefix | ((unsigned long long)d << (byte * 8));
            rank_s = rank - acc;
        }
        __syncthreads();
        prefix = prefix_s;
        rank = rank_s;
        __syncthreads();
    }
    if (threadIdx.x == 0) out_r[row0 + blockIdx.x] = __dsqrt_rn(__longlong_as_double((long long)prefix));
}

// =====================================================================================================================
// A10  membership counts.  P rows = reference rows i (lane-local), Q rows = candidate rows j.
__global__ void __launch_bounds__(FTHREADS) prdc64_kernel(const double* __restrict__ R, int64_t Nr, int64_t ldr, const double* __restrict__ rn,
                                                          const double* __restrict__ rt, const double* __restrict__ C, int64_t Nc,
                                                          int64_t ldc, const double* __restrict__ cn, const double* __restrict__ ct, int D,
                                                          int nchunks, int32_t* __restrict__ col_count, unsigned* __restrict__ row_any,
                                                          unsigned* __restrict__ row_cover, unsigned long long* __restrict__ row_min_bits,
                                                          const int* __restrict__ run_flag) {
    extern __shared__ __attribute__((aligned(16))) double lds64[];
    if (run_flag != nullptr && *run_flag == 0) return;               // (behind the f16 filter route: only when that gave up)
    double* qnorm = lds64 + FENGINE_DOUBLES;                          // [64] |c_j|^2
    double* qthr = qnorm + FT;                                       // [64] T(r_cand[j]); -inf past the end: no witness there
    const FLane L;
    const int64_t pb = blockIdx.x / nchunks;
    const int chunk = blockIdx.x % nchunks;
    const int64_t qtiles = (Nc + FT - 1) / FT;
    const int64_t t0 = qtiles * chunk / nchunks, t1 = qtiles * (chunk + 1) / nchunks;
    const int64_t prow = pb * FT + L.prow();
    const bool ok = prow < Nr;
    const double pn = ok ? rn[prow] : 0.0, pt = ok ? rt[prow] : -__builtin_inf();
    bool anyf = false, covf = false;
    double mn = __builtin_inf();
    const DenseRows64 prows{R, ldr, Nr, pb * FT};
    for (int64_t t = t0; t < t1; ++t) {
        if (L.tid < FT) {
            const int64_t j = t * FT + L.tid;
            qnorm[L.tid] = j < Nc ? cn[j] : __builtin_inf();
            qthr[L.tid] = j < Nc ? ct[j] : -__builtin_inf();
        }
        f64x4 acc[4];
        zero4(acc);
        f64_tile(DenseRows64{C, ldc, Nc, t * FT}, prows, D, lds64, L, acc);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            bool inside[4];
            bool hit = false;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = L.qrow(m, r);
                const double d2 = clamp0d(fma(-2.0, acc[m][r], pn + qnorm[q]));   // +inf past the end of the candidates
                mn = fmin(mn, d2);
                anyf = anyf || (ok && d2 < qthr[q]);
                inside[r] = ok && d2 < pt;
                hit = hit || inside[r];
            }
            covf = covf || hit;
            if (!__any(hit)) continue;                               // (wave-uniform: memberships are rare)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // lanes with l4 = g hold candidate row 16 m + g + 4 r: the 16 bits of group g count its members among this
                // wave's 16 reference rows
                const unsigned long long mask = __ballot(inside[r]);
                const int cnt = __popc((unsigned)(mask >> (16 * L.l4)) & 0xffffu);
                if (L.l15 == 0 && cnt > 0) atomicAdd(col_count + t * FT + L.qrow(m, r), cnt);
            }
        }
        __syncthreads();                                             // qnorm / qthr are rewritten by the next tile
    }
    // the four lanes of a reference row saw disjoint candidates
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
        mn = fmin(mn, __shfl_xor(mn, off));
        const int other_any = __shfl_xor((int)anyf, off), other_cov = __shfl_xor((int)covf, off);   // unconditionally: every lane takes part
        anyf = anyf || other_any != 0;
        covf = covf || other_cov != 0;
    }
    if (L.l4 == 0 && ok) {
        if (anyf) atomicOr(row_any + prow, 1u);
        if (covf) atomicOr(row_cover + prow, 1u);
        if (row_min_bits != nullptr) atomicMin(row_min_bits + prow, (unsigned long long)__double_as_longlong(mn));   // mn >= 0: bit order = value order
    }
}

__global__ void __launch_bounds__(256) prdc64_finish_kernel(const unsigned* __restrict__ row_any, const unsigned* __restrict__ row_cover,
                                                            const unsigned long long* __restrict__ row_min_bits, int64_t Nr,
                                                            uint8_t* __restrict__ out_any, uint8_t* __restrict__ out_cover,
                                                            double* __restrict__ out_min) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= Nr) return;
    out_any[i] = row_any[i] != 0u;
    out_cover[i] = row_cover[i] != 0u;
    if (out_min != nullptr) out_min[i] = __longlong_as_double((long long)row_min_bits[i]);
}

__global__ void __launch_bounds__(256) fill_u64_kernel(unsigned long long* __restrict__ p, int64_t n, unsigned long long v) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

// =====================================================================================================================
// A6-A8  kernel distance.  One workgroup per 64 x 64 tile of one Gram block of one subset; Kxx / Kyy: upper-triangular
// tiles only, off-diagonal tiles weighted 2, diagonal ENTRIES dropped (kd.py:62-69); rows gathered through the subset's
// index list.  partial[s * per_subset + b] = weighted tile sum; kd64_reduce_kernel adds them in a fixed order.
// MODE 0: polynomial (dot * gamma + coef0)^degree (kd.py:112-116); MODE 1: RBF exp(-(|x|^2 + |y|^2 - 2 dot) * gamma),
// gamma = 1 / (2 sigma^2) (kd.py:86-109).
template <int MODE>
__global__ void __launch_bounds__(FTHREADS) kd64_kernel(const double* __restrict__ X, int64_t ldx, const double* __restrict__ Y, int64_t ldy, int D,
                                                        const int64_t* __restrict__ idx1, const int64_t* __restrict__ idx2, int m, int T,
                                                        int ntri, double gamma, double coef0, int degree, double* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) double lds64[];
    double* qnorm = lds64 + FENGINE_DOUBLES;                          // [64] (RBF: squared norms of the Q rows)
    double* pnorm = qnorm + FT;
    double* red = pnorm + FT;                                        // [4] wave sums
    const FLane L;
    const int per_subset = 2 * ntri + T * T;
    const int s = blockIdx.x / per_subset;
    int b = blockIdx.x % per_subset;
    int which, tq, tp;                                               // which: 0 = XX, 1 = YY, 2 = XY
    if (b < 2 * ntri) {
        which = b / ntri;
        int t = b % ntri;
        tq = 0;
        while (t >= T - tq) { t -= T - tq; ++tq; }
        tp = tq + t;
    } else {
        which = 2;
        b -= 2 * ntri;
        tq = b / T;
        tp = b % T;
    }
    const int64_t* i1 = idx1 + (int64_t)s * m;
    const int64_t* i2 = idx2 + (int64_t)s * m;
    // Kxy[a][b] = k(x_a, y_b): Q rows (register axis) from set 1, P rows (lane axis) from set 2
    const GatherRows64 qsrc{which == 1 ? Y : X, which == 1 ? ldy : ldx, which == 1 ? i2 : i1, m, tq * FT};
    const GatherRows64 psrc{which == 0 ? X : Y, which == 0 ? ldx : ldy, which == 0 ? i1 : i2, m, tp * FT};
    if (MODE == 1) {                                                 // norms of the tile's rows, one wave per row at a time
        for (int row = L.wave; row < 2 * FT; row += 4) {
            const double* x = row < FT ? qsrc(row) : psrc(row - FT);
            double n2 = 0.0;
            if (x != nullptr)
                for (int k = L.lane; k < D; k += 64) n2 = fma(x[k], x[k], n2);
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) n2 += __shfl_xor(n2, off);
            if (L.lane == 0) (row < FT ? qnorm : pnorm)[row & (FT - 1)] = n2;
        }
    }
    f64x4 acc[4];
    zero4(acc);
    f64_tile(qsrc, psrc, D, lds64, L, acc);
    const bool drop_diag = which != 2 && tq == tp;
    const int p = tp * FT + L.prow();
    double sum = 0.0;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = tq * FT + L.qrow(mt, r);
            double kv;
            if (MODE == 0) {
                const double base = acc[mt][r] * gamma + coef0;
                kv = 1.0;
                for (int d = 0; d < degree; ++d) kv *= base;
            } else {
                double d2 = (qnorm[L.qrow(mt, r)] + pnorm[L.prow()]) - 2.0 * acc[mt][r];
                d2 = d2 < 0.0 ? 0.0 : d2;
                kv = exp(-d2 * gamma);
            }
            if (p < m && q < m && !(drop_diag && p == q)) sum += kv;
        }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    if (L.lane == 0) red[L.wave] = sum;
    __syncthreads();
    if (L.tid == 0) {
        const double weight = (which != 2 && tq != tp) ? 2.0 : 1.0;
        partial[blockIdx.x] = weight * (((red[0] + red[1]) + red[2]) + red[3]);
    }
}

__global__ void __launch_bounds__(64) kd64_reduce_kernel(const double* __restrict__ partial, int ntri, int T, int m, double* __restrict__ out_mmd) {
    const int s = blockIdx.x;
    const int per_subset = 2 * ntri + T * T;
    const double* p = partial + (int64_t)s * per_subset;
    if (threadIdx.x != 0) return;
    double sxx = 0.0, syy = 0.0, sxy = 0.0;
    for (int i = 0; i < ntri; ++i) sxx += p[i];
    for (int i = 0; i < ntri; ++i) syy += p[ntri + i];
    for (int i = 0; i < T * T; ++i) sxy += p[2 * ntri + i];
    const double dm = (double)m;
    out_mmd[s] = (sxx + syy) / (dm * (dm - 1.0)) - 2.0 * sxy / (dm * dm);      // kd.py:77-79
}

// ------------------------------------------------------------------------------------------------------------ host side
static int kcap_for(int k1) { return k1 <= 8 ? 8 : (k1 <= 16 ? 16 : 32); }
static size_t lds_bytes_knn(int kcap) { return std::max<size_t>((size_t)(FENGINE_DOUBLES + FT) * 8, (size_t)FT * 4 * kcap * 8); }
constexpr size_t LDS_BYTES_PRDC = (size_t)(FENGINE_DOUBLES + 2 * FT) * 8;
constexpr size_t LDS_BYTES_KD = (size_t)(FENGINE_DOUBLES + 2 * FT + 4) * 8;

static int chunks_for(int64_t row_blocks, int64_t col_tiles) {
    // enough workgroups for ~8 per CU, at least 8 column tiles per workgroup
    int64_t c = std::max<int64_t>(1, ceil_div(2048, row_blocks));
    c = std::min<int64_t>(c, std::max<int64_t>(1, col_tiles / 8));
    return (int)std::min<int64_t>(c, 64);
}

struct Knn64Plan {
    int kcap, nchunks;
    bool select;                     // k + 1 > 32: distance blocks + radix select
    int64_t block_rows;
};
static Knn64Plan plan_knn64(int64_t N, int64_t M, int k) {
    Knn64Plan p;
    p.select = k + 1 > 32;
    p.kcap = kcap_for(k + 1);
    p.nchunks = chunks_for(ceil_div(N, FT), ceil_div(M, FT));
    // the select path materialises block_rows x M doubles at a time: ~256 MB
    p.block_rows = std::max<int64_t>(FT, std::min<int64_t>(ceil_div(N, FT) * FT, ((int64_t)1 << 25) / std::max<int64_t>(M, 1) / FT * FT));
    return p;
}

template <int KCAP>
static int launch_knn64(const double* X, int64_t N, int64_t ldx, const double* xn, const double* Y, int64_t M, int64_t ldy, const double* yn,
                        int D, int k1, int nchunks, double* partial, double* out_r, hipStream_t st, const int* run_flag = nullptr) {
    const size_t lds = lds_bytes_knn(KCAP);
    AM_HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(&knn64_kernel<KCAP>), (int)lds));
    hipLaunchKernelGGL(knn64_kernel<KCAP>, dim3((unsigned)(ceil_div(N, FT) * nchunks)), dim3(FTHREADS), lds, st, X, N, ldx, xn, Y, M, ldy, yn,
                       D, k1, nchunks, partial, run_flag);
    AM_LAUNCH_CHECK();
    hipLaunchKernelGGL(knn64_merge_kernel<KCAP>, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, st, partial, N, nchunks, out_r, run_flag);
    AM_LAUNCH_CHECK();
    return AM_OK;
}

// the general f64 kernels of a set against itself, launched behind a device flag (knn64_filter in pairwise.hip: the f16
// filter route's fallback - they return at once unless *run_flag != 0); k + 1 <= 32
size_t knn64_self_workspace(int64_t N, int k) {
    const Knn64Plan p = plan_knn64(N, N, k);
    Carver c(nullptr, 0);
    c.take<double>((size_t)N);
    c.take<double>((size_t)p.nchunks * N * p.kcap);
    return c.off;
}
int knn64_self_gated(const double* X, int64_t N, int64_t ld, int D, int k, double* out_r, void* ws, size_t ws_bytes, const int* run_flag,
                     hipStream_t st) {
    const Knn64Plan p = plan_knn64(N, N, k);
    Carver c(ws, ws_bytes);
    double* xn = c.take<double>((size_t)N);
    double* scratch = c.take<double>((size_t)p.nchunks * N * p.kcap);
    AM_REQUIRE(c.ok() && !p.select, AM_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", c.off, ws_bytes);
    hipLaunchKernelGGL(row_sqnorm64_kernel, dim3((unsigned)ceil_div(N, 4)), dim3(256), 0, st, X, N, ld, D, xn);
    AM_LAUNCH_CHECK();
    switch (p.kcap) {
        case 8: return launch_knn64<8>(X, N, ld, xn, X, N, ld, xn, D, k + 1, p.nchunks, scratch, out_r, st, run_flag);
        case 16: return launch_knn64<16>(X, N, ld, xn, X, N, ld, xn, D, k + 1, p.nchunks, scratch, out_r, st, run_flag);
        default: return launch_knn64<32>(X, N, ld, xn, X, N, ld, xn, D, k + 1, p.nchunks, scratch, out_r, st, run_flag);
    }
}

}  // namespace am

using namespace am;

extern "C" size_t am_knn_f64_workspace_bytes(int64_t N, int64_t M, int D, int k) {
    if (N < 1 || M < 1 || k < 1) return 0;
    const Knn64Plan p = plan_knn64(N, M, k);
    Carver c(nullptr, 0);
    c.take<double>((size_t)N);
    c.take<double>((size_t)M);
    if (p.select) c.take<double>((size_t)p.block_rows * M);
    else c.take<double>((size_t)p.nchunks * N * p.kcap);
    // (a set against itself may take the f16 filter route: sized for whichever is larger)
    return N == M && knn64_filter_eligible(N, D, k) ? std::max(c.off, knn64_filter_workspace(N, D, k)) : c.off;
}

extern "C" int am_knn_radii_f64(const double* X, int64_t N, int64_t ldx, const double* Y, int64_t M, int64_t ldy, int D, int k,
                                double* out_r, void* ws, size_t ws_bytes, am_stream_t stream) {
    AM_REQUIRE(X && Y && out_r, AM_ERR_BAD_ARG, "null pointer");
    AM_REQUIRE(N >= 1 && M >= 1 && D >= 1, AM_ERR_BAD_SHAPE, "X is %lld x %d, Y has %lld rows", (long long)N, D, (long long)M);
    AM_REQUIRE(ldx >= D && ldy >= D, AM_ERR_BAD_ARG, "ld < D (ldx=%lld ldy=%lld D=%d)", (long long)ldx, (long long)ldy, D);
    AM_REQUIRE(k >= 1 && (int64_t)k + 1 <= M, AM_ERR_BAD_SHAPE, "nearest_k=%d needs 1 <= k and k + 1 <= %lld rows (torch.kthvalue would raise)", k,
               (long long)M);
    hipStream_t st = static_cast<hipStream_t>(stream);
    // a large set against itself: candidates from the f16 filter sweep of the float32 path, their distances and the selection
    // in f64 (pairwise.hip: knn64_filter) - where the caller's workspace holds it
    if (X == Y && N == M && ldx == ldy && knn64_filter_eligible(N, D, k) && ws_bytes >= knn64_filter_workspace(N, D, k))
        return knn64_filter(X, N, ldx, D, k, out_r, ws, ws_bytes, st);
    const Knn64Plan p = plan_knn64(N, M, k);
    Carver c(ws, ws_bytes);
    double* xn = c.take<double>((size_t)N);
    double* yn = c.take<double>((size_t)M);
    double* scratch = p.select ? c.take<double>((size_t)p.block_rows * M) : c.take<double>((size_t)p.nchunks * N * p.kcap);
    AM_REQUIRE(c.ok(), AM_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", c.off, ws_bytes);
    hipLaunchKernelGGL(row_sqnorm64_kernel, dim3((unsigned)ceil_div(N, 4)), dim3(256), 0, st, X, N, ldx, D, xn);
    const bool self = X == Y && N == M && ldx == ldy;
    if (self) yn = xn;
    else hipLaunchKernelGGL(row_sqnorm64_kernel, dim3((unsigned)ceil_div(M, 4)), dim3(256), 0, st, Y, M, ldy, D, yn);
    AM_LAUNCH_CHECK();
    if (p.select) {
        const size_t lds = (size_t)FENGINE_DOUBLES * 8;
        const int64_t qtiles = ceil_div(M, FT);
        for (int64_t row0 = 0; row0 < N; row0 += p.block_rows) {
            const int64_t nrows = std::min<int64_t>(p.block_rows, N - row0);
            hipLaunchKernelGGL(dist64_block_kernel, dim3((unsigned)(ceil_div(nrows, FT) * qtiles)), dim3(FTHREADS), lds, st, X, N, ldx, xn, row0,
                               nrows, Y, M, ldy, yn, D, scratch);
            AM_LAUNCH_CHECK();
            hipLaunchKernelGGL(select64_kernel, dim3((unsigned)nrows), dim3(256), 0, st, scratch, M, k + 1, out_r, row0);
            AM_LAUNCH_CHECK();
        }
        return AM_OK;
    }
    switch (p.kcap) {
        case 8: return launch_knn64<8>(X, N, ldx, xn, Y, M, ldy, yn, D, k + 1, p.nchunks, scratch, out_r, st);
        case 16: return launch_knn64<16>(X, N, ldx, xn, Y, M, ldy, yn, D, k + 1, p.nchunks, scratch, out_r, st);
        default: return launch_knn64<32>(X, N, ldx, xn, Y, M, ldy, yn, D, k + 1, p.nchunks, scratch, out_r, st);
    }
}

extern "C" size_t am_prdc_f64_workspace_bytes(int64_t Nr, int64_t Nc, int D) {
    if (Nr < 1 || Nc < 1) return 0;
    Carver c(nullptr, 0);
    c.take<double>((size_t)Nr);
    c.take<double>((size_t)Nr);
    c.take<double>((size_t)Nc);
    c.take<double>((size_t)Nc);
    c.take<unsigned>((size_t)Nr);
    c.take<unsigned>((size_t)Nr);
    c.take<unsigned long long>((size_t)Nr);
    if (prdc64_filter_eligible(Nr, Nc, D)) c.take<char>(prdc64_filter_workspace(Nr, Nc, D));     // (the f16 filter route's part)
    return c.off;
}

extern "C" int am_prdc_counts_f64(const double* R, int64_t Nr, int64_t ldr, const double* C, int64_t Nc, int64_t ldc, int D,
                                  const double* r_ref, const double* r_cand, int32_t* out_col_count, uint8_t* out_row_any,
                                  uint8_t* out_row_cover, double* out_row_min, void* ws, size_t ws_bytes, am_stream_t stream) {
    AM_REQUIRE(R && C && r_ref && r_cand && out_col_count && out_row_any && out_row_cover, AM_ERR_BAD_ARG, "null pointer");
    AM_REQUIRE(Nr >= 1 && Nc >= 1 && D >= 1, AM_ERR_BAD_SHAPE, "reference %lld x %d, candidate %lld rows", (long long)Nr, D, (long long)Nc);
    AM_REQUIRE(ldr >= D && ldc >= D, AM_ERR_BAD_ARG, "ld < D (ldr=%lld ldc=%lld D=%d)", (long long)ldr, (long long)ldc, D);
    hipStream_t st = static_cast<hipStream_t>(stream);
    Carver c(ws, ws_bytes);
    double* rn = c.take<double>((size_t)Nr);
    double* rt = c.take<double>((size_t)Nr);
    double* cn = c.take<double>((size_t)Nc);
    double* ct = c.take<double>((size_t)Nc);
    unsigned* rany = c.take<unsigned>((size_t)Nr);
    unsigned* rcov = c.take<unsigned>((size_t)Nr);
    unsigned long long* rmin = c.take<unsigned long long>((size_t)Nr);
    // large problems without a row minimum: the float32 path's f16 filter pass decides what it can, the pairs inside its band are
    // evaluated in f64 (pairwise.hip: prdc64_filter); the general kernel below then runs behind the route's fail flag
    const size_t filter_bytes = prdc64_filter_eligible(Nr, Nc, D) ? prdc64_filter_workspace(Nr, Nc, D) : 0;
    void* filter_ws = nullptr;
    if (filter_bytes != 0 && out_row_min == nullptr && c.off + filter_bytes <= ws_bytes) filter_ws = c.take<char>(filter_bytes);
    AM_REQUIRE(c.ok(), AM_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", c.off, ws_bytes);
    hipLaunchKernelGGL(row_sqnorm64_kernel, dim3((unsigned)ceil_div(Nr, 4)), dim3(256), 0, st, R, Nr, ldr, D, rn);
    hipLaunchKernelGGL(row_sqnorm64_kernel, dim3((unsigned)ceil_div(Nc, 4)), dim3(256), 0, st, C, Nc, ldc, D, cn);
    hipLaunchKernelGGL(threshold64_kernel, dim3((unsigned)ceil_div(Nr, 256)), dim3(256), 0, st, r_ref, Nr, rt);
    hipLaunchKernelGGL(threshold64_kernel, dim3((unsigned)ceil_div(Nc, 256)), dim3(256), 0, st, r_cand, Nc, ct);
    AM_LAUNCH_CHECK();
    AM_HIP_TRY(hipMemsetAsync(out_col_count, 0, (size_t)Nc * sizeof(int32_t), st));
    AM_HIP_TRY(hipMemsetAsync(rany, 0, (size_t)Nr * sizeof(unsigned), st));
    AM_HIP_TRY(hipMemsetAsync(rcov, 0, (size_t)Nr * sizeof(unsigned), st));
    if (out_row_min != nullptr) {
        hipLaunchKernelGGL(fill_u64_kernel, dim3((unsigned)ceil_div(Nr, 256)), dim3(256), 0, st, rmin, Nr, 0x7ff0000000000000ull);
        AM_LAUNCH_CHECK();
    }
    const int* run_flag = nullptr;
    if (filter_ws != nullptr) {
        int rc = prdc64_filter(R, Nr, ldr, C, Nc, ldc, D, rt, ct, out_col_count, rany, rcov, &run_flag, filter_ws, filter_bytes, st);
        if (rc != AM_OK) return rc;
    }
    const int nchunks = chunks_for(ceil_div(Nr, FT), ceil_div(Nc, FT));
    AM_HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(&prdc64_kernel), (int)LDS_BYTES_PRDC));
    hipLaunchKernelGGL(prdc64_kernel, dim3((unsigned)(ceil_div(Nr, FT) * nchunks)), dim3(FTHREADS), LDS_BYTES_PRDC, st, R, Nr, ldr, rn, rt, C, Nc,
                       ldc, cn, ct, D, nchunks, out_col_count, rany, rcov, out_row_min != nullptr ? rmin : nullptr, run_flag);
    AM_LAUNCH_CHECK();
    hipLaunchKernelGGL(prdc64_finish_kernel, dim3((unsigned)ceil_div(Nr, 256)), dim3(256), 0, st, rany, rcov, rmin, Nr, out_row_any, out_row_cover,
                       out_row_min);
    AM_LAUNCH_CHECK();
    return AM_OK;
}

extern "C" size_t am_kd_f64_workspace_bytes(int S, int m) {
    if (S < 1 || m < 1) return 0;
    const int T = (int)ceil_div(m, FT), ntri = T * (T + 1) / 2;
    Carver c(nullptr, 0);
    c.take<double>((size_t)S * (2 * ntri + T * T));
    return c.off;
}

static int run_kd64(int mode, const double* X, int64_t N1, int64_t ldx, const double* Y, int64_t N2, int64_t ldy, int D, const int64_t* idx1,
                    const int64_t* idx2, int S, int m, double gamma, double coef0, int degree, double* out_mmd, void* ws, size_t ws_bytes,
                    hipStream_t st) {
    AM_REQUIRE(X && Y && idx1 && idx2 && out_mmd, AM_ERR_BAD_ARG, "null pointer");
    AM_REQUIRE(N1 >= 1 && N2 >= 1 && D >= 1 && S >= 1 && m >= 1, AM_ERR_BAD_SHAPE, "N1=%lld N2=%lld D=%d S=%d m=%d",
               (long long)N1, (long long)N2, D, S, m);
    AM_REQUIRE(ldx >= D && ldy >= D, AM_ERR_BAD_ARG, "ld < D");
    AM_REQUIRE(degree >= 0, AM_ERR_BAD_ARG, "degree=%d", degree);
    const int T = (int)ceil_div(m, FT), ntri = T * (T + 1) / 2;
    Carver c(ws, ws_bytes);
    double* partial = c.take<double>((size_t)S * (2 * ntri + T * T));
    AM_REQUIRE(c.ok(), AM_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", c.off, ws_bytes);
    const unsigned blocks = (unsigned)((size_t)S * (2 * ntri + T * T));
    if (mode == 0) {
        AM_HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(&kd64_kernel<0>), (int)LDS_BYTES_KD));
        hipLaunchKernelGGL(kd64_kernel<0>, dim3(blocks), dim3(FTHREADS), LDS_BYTES_KD, st, X, ldx, Y, ldy, D, idx1, idx2, m, T, ntri, gamma, coef0,
                           degree, partial);
    } else {
        AM_HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(&kd64_kernel<1>), (int)LDS_BYTES_KD));
        hipLaunchKernelGGL(kd64_kernel<1>, dim3(blocks), dim3(FTHREADS), LDS_BYTES_KD, st, X, ldx, Y, ldy, D, idx1, idx2, m, T, ntri, gamma, coef0,
                           degree, partial);
    }
    AM_LAUNCH_CHECK();
    hipLaunchKernelGGL(kd64_reduce_kernel, dim3((unsigned)S), dim3(64), 0, st, partial, ntri, T, m, out_mmd);
    AM_LAUNCH_CHECK();
    return AM_OK;
}

extern "C" int am_kd_poly_f64(const double* X, int64_t N1, int64_t ldx, const double* Y, int64_t N2, int64_t ldy, int D, const int64_t* idx1,
                              const int64_t* idx2, int S, int m, double gamma, double coef0, int degree, double* out_mmd, void* ws,
                              size_t ws_bytes, am_stream_t stream) {
    return run_kd64(0, X, N1, ldx, Y, N2, ldy, D, idx1, idx2, S, m, gamma, coef0, degree, out_mmd, ws, ws_bytes, static_cast<hipStream_t>(stream));
}

extern "C" int am_kd_rbf_f64(const double* X, int64_t N1, int64_t ldx, const double* Y, int64_t N2, int64_t ldy, int D, const int64_t* idx1,
                             const int64_t* idx2, int S, int m, double sigma, double* out_mmd, void* ws, size_t ws_bytes, am_stream_t stream) {
    AM_REQUIRE(sigma > 0.0, AM_ERR_BAD_ARG, "sigma=%g", sigma);
    return run_kd64(1, X, N1, ldx, Y, N2, ldy, D, idx1, idx2, S, m, 1.0 / (2.0 * sigma * sigma), 0.0, 0, out_mmd, ws, ws_bytes,
                    static_cast<hipStream_t>(stream));
}
