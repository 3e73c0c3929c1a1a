// Per-set sufficient statistics: column means and unbiased covariance of an
// N x D f32 embedding matrix (reference data.py:37-58), and the Chan merge of two
// (n, mean, cov) triples (reference data.py:77-94).
//
//   pass 1  column sums, f64 accumulation                     (HBM-bound, N*D*4 bytes)
//   pass 2  centred scatter  S = sum_n (x_n-mu)(x_n-mu)^T      (f32 MFMA, 2*N*D^2 flop)
//           - only the upper-triangular 128x128 output tiles are computed,
//           - each workgroup owns (tile, row slab); products run on
//             v_mfma_f32_32x32x2_f32 in chains of FLUSH*32 rows that are then
//             added into f64 accumulators, so rounding does not grow with N,
//           - per-slab f64 partial tiles go to the workspace and are summed in a
//             fixed order (deterministic, no float atomics).
#include "am_common.h"
#include "tile_engine.h"
#include <algorithm>

namespace am {

constexpr int SC_ROWS = 32;          // rows of X per stage (MFMA k extent 2 -> 16 steps)
constexpr int SC_LD = 128;           // LDS slab row stride (floats); reads are lane-consecutive
constexpr int SC_SLAB = SC_ROWS * SC_LD;
constexpr int SC_FLUSH = 8;          // stages per f32 chain (256 rows)

// ---------------------------------------------------------------- column sums
__global__ void __launch_bounds__(256) colsum_partial_kernel(const float* __restrict__ X, int64_t N, int64_t ld,
                                                             int D, int64_t rows_per_block,
                                                             double* __restrict__ partial) {
    __shared__ double red[256 * 4];
    const int tid = threadIdx.x;
    const int cgs = (D + 3) / 4;
    const int cpp = cgs < 256 ? cgs : 256;
    const int rpar = 256 / cpp;
    const int my_cg = tid % cpp, my_r = tid / cpp;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = (r0 + rows_per_block < N) ? r0 + rows_per_block : N;
    for (int cg0 = 0; cg0 < cgs; cg0 += cpp) {
        const int cg = cg0 + my_cg;
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        if (cg < cgs && my_r < rpar) {
            // four loads in flight per thread (one at a time left the pass latency-bound at 3.5 TB/s); the sums keep their order
            int64_t row = r0 + my_r;
            for (; row + 3 * rpar < r1; row += 4 * rpar) {
                f32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = load_k4(X + (row + u * rpar) * ld, cg * 4, D);
#pragma unroll
                for (int u = 0; u < 4; ++u) { s0 += (double)v[u].x; s1 += (double)v[u].y; s2 += (double)v[u].z; s3 += (double)v[u].w; }
            }
            for (; row < r1; row += rpar) {
                const f32x4 v = load_k4(X + row * ld, cg * 4, D);
                s0 += (double)v.x; s1 += (double)v.y; s2 += (double)v.z; s3 += (double)v.w;
            }
        }
        red[tid * 4 + 0] = s0; red[tid * 4 + 1] = s1; red[tid * 4 + 2] = s2; red[tid * 4 + 3] = s3;
        __syncthreads();
        if (my_r == 0 && cg < cgs) {
            for (int rr = 1; rr < rpar; ++rr) {
                const int o = (rr * cpp + my_cg) * 4;
                s0 += red[o]; s1 += red[o + 1]; s2 += red[o + 2]; s3 += red[o + 3];
            }
            double* out = partial + (int64_t)blockIdx.x * D + cg * 4;
            out[0] = s0;
            if (cg * 4 + 1 < D) out[1] = s1;
            if (cg * 4 + 2 < D) out[2] = s2;
            if (cg * 4 + 3 < D) out[3] = s3;
        }
        __syncthreads();
    }
}

// out[d] = (sum_b partial[b][d]) * scale.  One workgroup per 16 columns; sixteen threads per column split the partial rows
// (eight loads in flight each) and combine through LDS in a fixed order (deterministic).
constexpr int CSR_COLS = 16, CSR_PARTS = 16;
__global__ void __launch_bounds__(256) colsum_reduce_kernel(const double* __restrict__ partial, int nblocks, int D,
                                                            double scale, double* __restrict__ out) {
    __shared__ double red[CSR_PARTS][CSR_COLS];
    const int c = threadIdx.x % CSR_COLS, part = threadIdx.x / CSR_COLS;
    const int d = blockIdx.x * CSR_COLS + c;
    double s = 0;
    if (d < D) {
        int b = part;
        for (; b + 7 * CSR_PARTS < nblocks; b += 8 * CSR_PARTS) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(int64_t)(b + u * CSR_PARTS) * D + d];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; b < nblocks; b += CSR_PARTS) s += partial[(int64_t)b * D + d];
    }
    red[part][c] = s;
    __syncthreads();
    if (part == 0 && d < D) {
        double t = red[0][c];
        for (int q = 1; q < CSR_PARTS; ++q) t += red[q][c];
        out[d] = t * scale;
    }
}

// ------------------------------------------------------------ centred scatter
__device__ __forceinline__ void tri_decode(int t, int T, int& tp, int& tq) {   // t-th pair with tp <= tq
    tp = 0;
    int rem = t;
    while (rem >= T - tp) { rem -= T - tp; ++tp; }
    tq = tp + rem;
}

// (2 waves per SIMD: two workgroups per CU - 2 x 64 KB of LDS - cover each other's barriers, f64 flushes and load latency;
// with one wave per SIMD every such stall idled the matrix pipe: 74 TF)
template <bool FULLD>                               // D is a multiple of the tile width: no column masks
__global__ void __launch_bounds__(ENGINE_THREADS, 2)
scatter_partial_kernel(const float* __restrict__ X, int64_t N, int64_t ld, int D, const double* __restrict__ mean,
                       int64_t rows_per_slab, int ntri, double* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float lds[4 * SC_SLAB];    // [2 stages][A slab, B slab]
    const LaneInfo L;
    const int T = (D + TB - 1) / TB;
    const int tri = blockIdx.x % ntri;
    const int slab = blockIdx.x / ntri;
    int tp, tq;
    tri_decode(tri, T, tp, tq);
    const int64_t r0 = (int64_t)slab * rows_per_slab;
    const int64_t r1 = (r0 + rows_per_slab < N) ? r0 + rows_per_slab : N;
    const int nstages = (int)((r1 - r0 + SC_ROWS - 1) / SC_ROWS);

    // staging role: 4 rows (tid/32 + 8q) x one float4 column group (tid%32) per operand
    const int srow = L.tid >> 5, sc4 = (L.tid & 31) * 4;
    const int colA = tp * TB + sc4, colB = tq * TB + sc4;
    f32x4 muA, muB;
#pragma unroll
    for (int e = 0; e < 4; ++e) {                   // (clamped index: eight independent loads instead of eight branches)
        muA[e] = (float)mean[colA + e < D ? colA + e : D - 1];
        muB[e] = (float)mean[colB + e < D ? colB + e : D - 1];
    }
    // issue(): the eight loads of the next stage, nothing else - the values are first touched in commit(), after the
    // stage's MFMAs.  (Centring them right behind each load, as the first version did, made the compiler wait for every
    // load in turn: eight exposed memory round trips per stage, 47 % of the wave time parked, 43 % matrix-pipe busy.)
    // Rows past the slab and column groups past the row are read from a valid address and masked in commit().
    f32x4 ra[4], rb[4];
    // one buffer descriptor per stage (its 32 rows; rows past the slab's end read as zero in hardware): each thread keeps
    // two constant 32-bit offsets and the row group travels in the scalar offset - no 64-bit address arithmetic per load,
    // and no limit on the slab's size in bytes
    const unsigned voA = (unsigned)((srow * ld + (colA + 3 < ld ? colA : 0)) * 4);
    const unsigned voB = (unsigned)((srow * ld + (colB + 3 < ld ? colB : 0)) * 4);
    const unsigned row_bytes = (unsigned)(ld * 4);
    auto issue = [&](int st) {
        const int64_t row0 = r0 + (int64_t)st * SC_ROWS;
        const int64_t left = r1 - row0;
        const float* p0 = X + row0 * ld;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<uintptr_t>(p0) & 0xffffffffu));
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<uintptr_t>(p0) >> 32));
        const unsigned bytes = __builtin_amdgcn_readfirstlane((unsigned)((left < SC_ROWS ? left : SC_ROWS) * ld * 4));
        TileRsrc xr;
        xr.rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>((static_cast<uintptr_t>(hi) << 32) | lo), 0, (int)bytes, 0x00020000);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ra[q] = rsrc_load(xr, voA, (unsigned)(q * 8) * row_bytes);
            rb[q] = rsrc_load(xr, voB, (unsigned)(q * 8) * row_bytes);
        }
    };
    const bool fullA = colA + 3 < D, fullB = colB + 3 < D;
    auto centre = [&](const f32x4& v, const f32x4& mu, int col, bool full, bool row_ok) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (row_ok && (full || col + e < D)) ? v[e] - mu[e] : 0.f;
        return o;
    };
    auto commit = [&](int st) {
        float* s = lds + (st & 1) * 2 * SC_SLAB + srow * SC_LD + sc4;
        const bool whole = r0 + (int64_t)(st + 1) * SC_ROWS <= r1;       // wave-uniform: every row of the stage exists
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (FULLD && whole) {                                          // the common case: plain subtractions
                *reinterpret_cast<f32x4*>(s + q * 8 * SC_LD) = ra[q] - muA;
                *reinterpret_cast<f32x4*>(s + SC_SLAB + q * 8 * SC_LD) = rb[q] - muB;
                continue;
            }
            const bool row_ok = whole || r0 + (int64_t)st * SC_ROWS + q * 8 + srow < r1;
            *reinterpret_cast<f32x4*>(s + q * 8 * SC_LD) = centre(ra[q], muA, colA, fullA, row_ok);
            *reinterpret_cast<f32x4*>(s + SC_SLAB + q * 8 * SC_LD) = centre(rb[q], muB, colB, fullB, row_ok);
        }
    };

    f32x16 acc[2][2];
    zero_acc(acc);
    double acc64[2][2][16];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc64[a][b][i] = 0.0;

    if (nstages > 0) {
        issue(0);
        commit(0);
    }
    __syncthreads();
    for (int st = 0; st < nstages; ++st) {
        const bool more = st + 1 < nstages;
        if (more) issue(st + 1);
        const float* sA = lds + (st & 1) * 2 * SC_SLAB + L.wm * 64 + L.r;
        const float* sB = sA - L.wm * 64 + SC_SLAB + L.wn * 64;
#pragma unroll
        for (int ks = 0; ks < SC_ROWS / 2; ++ks) {
            const int o = (ks * 2 + L.h) * SC_LD;
            const float a0 = sA[o], a1 = sA[o + 32];
            const float b0 = sB[o], b1 = sB[o + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if ((st % SC_FLUSH) == SC_FLUSH - 1 || !more) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc64[a][b][i] += (double)acc[a][b][i];
            zero_acc(acc);
        }
        if (more) commit(st + 1);
        __syncthreads();
    }
    // partial tile, row-major 128x128 f64: row = output row p (MFMA m), col = q (MFMA n = lane&31)
    double* out = partial + ((int64_t)slab * ntri + tri) * (TB * TB);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int p = L.wm * 64 + mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * L.h;
                const int q = L.wn * 64 + nt * 32 + L.r;
                out[p * TB + q] = acc64[mt][nt][i];
            }
}

// out = scale * sum_slab partial[slab]: one thread per element of an upper-triangular tile; it writes out[p][q] and, for
// an off-diagonal tile, the mirrored out[q][p] (each partial element is read once: the first version computed every
// output element on its own and read the off-diagonal tiles twice)
__global__ void __launch_bounds__(256) scatter_reduce_kernel(const double* __restrict__ partial, int nslabs, int ntri, int D, double scale,
                                                             double* __restrict__ out) {
    const int T = (D + TB - 1) / TB;
    const int tri = blockIdx.y;
    int tp, tq;
    tri_decode(tri, T, tp, tq);
    const int e = blockIdx.x * 256 + threadIdx.x;              // element of the 128 x 128 tile, row-major
    const int a = tp * TB + e / TB, b = tq * TB + e % TB;
    if (a >= D || b >= D) return;
    const double* src = partial + (int64_t)tri * (TB * TB) + e;
    const int64_t step = (int64_t)ntri * (TB * TB);
    double s = 0;
    int sl = 0;
    for (; sl + 7 < nslabs; sl += 8) {             // eight loads in flight, summed in slab order
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(sl + u) * step];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; sl < nslabs; ++sl) s += src[sl * step];
    s *= scale;
    out[(int64_t)a * D + b] = s;
    if (tp != tq) out[(int64_t)b * D + a] = s;
}

__global__ void zero_f64_kernel(double* __restrict__ p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.0;
}

// ------------------------------------------------------------------ Chan merge
__global__ void merge_cov_kernel(int64_t n1, const double* __restrict__ mean1, const double* cov1, int64_t n2,
                                 const double* __restrict__ mean2, const double* __restrict__ cov2, int D,
                                 double* out_cov) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= D) return;
    const double n = (double)(n1 + n2);
    const double w1 = (double)(n1 - 1) / (n - 1);
    const double w2 = (double)(n2 - 1) / (n - 1);
    const double wd = ((double)n1 * (double)n2 / n) / (n - 1);
    const double di = mean1[i] - mean2[i], dj = mean1[j] - mean2[j];
    const int64_t o = (int64_t)i * D + j;
    // same association as the reference: (w1*cov1 + w2*cov2) + wd*outer
    out_cov[o] = (w1 * cov1[o] + w2 * cov2[o]) + wd * (di * dj);
}

__global__ void merge_mean_kernel(int64_t n1, const double* mean1, int64_t n2, const double* __restrict__ mean2, int D,
                                  double* out_mean) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D) return;
    out_mean[i] = ((double)n1 * mean1[i] + (double)n2 * mean2[i]) / (double)(n1 + n2);
}


// ------------------------------------------------------------------ streaming add (one launch per batch)
// The embedding pipeline feeds AudioMetricsData.add() with batches of <= 32 rows (reference embed.py:231-236), each of
// which is: batch mean / covariance (data.py:37-47), Chan merge into the running (n, mean, cov) (data.py:77-94), row
// append (data.py:68-72).  As separate entry points that is >= 6 launches and three temporaries per batch; here it is ONE
// launch: workgroup (I, J) owns the 32 x 32 tile of the running covariance, stages the batch's columns I and J in LDS,
// computes their means in f64, centres in f32 (as am_stats_f32 does: exact f32 products of f32-centred values), sums the
// products in f64, applies the reference's merge formula with its association order to its tile in place, and copies its
// share of the rows into the stored matrix.  The means are read from mean_in and written to mean_out (two buffers the
// caller alternates): other workgroups still need the old mean of columns I for their own delta terms.
constexpr int PUSH_TILE = 32;
constexpr int PUSH_MAX_ROWS = 128;

__global__ void __launch_bounds__(256) stats_push_kernel(const float* __restrict__ E, int b, int D, int64_t ld, int64_t n_old,
                                                         const double* __restrict__ mean_in, double* __restrict__ mean_out,
                                                         double* __restrict__ cov, float* __restrict__ rows_out, int64_t ld_out) {
    __shared__ float ea[PUSH_MAX_ROWS][PUSH_TILE + 1], eb[PUSH_MAX_ROWS][PUSH_TILE + 1];
    __shared__ double mu[2][PUSH_TILE];
    const int tid = threadIdx.x;
    const int tj = blockIdx.x, ti = blockIdx.y;
    const int i0 = ti * PUSH_TILE, j0 = tj * PUSH_TILE;
    // stage the two column strips of the batch (columns past D as zeros)
    for (int e = tid; e < b * PUSH_TILE; e += 256) {
        const int r = e / PUSH_TILE, c = e % PUSH_TILE;
        ea[r][c] = i0 + c < D ? E[(int64_t)r * ld + i0 + c] : 0.f;
        eb[r][c] = j0 + c < D ? E[(int64_t)r * ld + j0 + c] : 0.f;
    }
    __syncthreads();
    if (tid < 2 * PUSH_TILE) {                                  // batch means of the 64 columns, f64 sums in row order
        const int which = tid / PUSH_TILE, c = tid % PUSH_TILE;
        double s = 0;
        for (int r = 0; r < b; ++r) s += (double)(which ? eb[r][c] : ea[r][c]);
        mu[which][c] = s / (double)b;
    }
    __syncthreads();
    for (int e = tid; e < b * PUSH_TILE; e += 256) {            // centre in f32 against the f32-rounded mean (am_stats_f32's arithmetic)
        const int r = e / PUSH_TILE, c = e % PUSH_TILE;
        ea[r][c] -= (float)mu[0][c];
        eb[r][c] -= (float)mu[1][c];
    }
    __syncthreads();
    const double n1 = (double)n_old, n2 = (double)b, n = n1 + n2;
    const int c = tid % PUSH_TILE;
#pragma unroll
    for (int q = 0; q < PUSH_TILE / 8; ++q) {
        const int r = tid / PUSH_TILE + 8 * q;                  // tile row (column of strip A)
        const int gi = i0 + r, gj = j0 + c;
        if (gi >= D || gj >= D) continue;
        double sum = 0;
        for (int k = 0; k < b; ++k) sum = fma((double)ea[k][r], (double)eb[k][c], sum);
        const double cov2 = b > 1 ? sum / (double)(b - 1) : 0.0;       // data.py:40-42: one row -> zero covariance
        const int64_t o = (int64_t)gi * D + gj;
        if (n_old == 0) {
            cov[o] = cov2;
        } else {                                                 // data.py:83-92, same association order
            const double w1 = (n1 - 1) / (n - 1), w2 = (n2 - 1) / (n - 1), wd = (n1 * n2 / n) / (n - 1);
            const double di = mean_in[gi] - mu[0][r], dj = mean_in[gj] - mu[1][c];
            cov[o] = (w1 * cov[o] + w2 * cov2) + wd * (di * dj);
        }
    }
    if (ti == tj && tid < PUSH_TILE && i0 + tid < D) {          // the diagonal workgroups write the new means
        const int g = i0 + tid;
        mean_out[g] = n_old == 0 ? mu[0][tid] : (n1 * mean_in[g] + n2 * mu[0][tid]) / n;
    }
    if (rows_out != nullptr) {                                   // append: this workgroup's share of the b x D elements
        const int64_t total = (int64_t)b * D;
        const int64_t nwg = (int64_t)gridDim.x * gridDim.y, wg = (int64_t)ti * gridDim.x + tj;
        const int64_t per = (total + nwg - 1) / nwg;
        const int64_t lo = wg * per, hi = lo + per < total ? lo + per : total;
        for (int64_t e = lo + tid; e < hi; e += 256) {
            const int64_t r = e / D, col = e % D;
            rows_out[r * ld_out + col] = E[r * ld + col];
        }
    }
}

// ------------------------------------------------------------------ host side
struct StatsPlan {
    int cs_blocks;
    int64_t cs_rows;
    int ntri, nslabs;
    int64_t slab_rows;
};

static StatsPlan plan_stats(int64_t N, int D) {
    StatsPlan p;
    int64_t b = ceil_div(N, 64);
    if (b > 512) b = 512;
    if (b < 1) b = 1;
    p.cs_rows = ceil_div(N, b);
    p.cs_blocks = (int)ceil_div(N, p.cs_rows);
    const int T = (int)ceil_div(D, TB);
    p.ntri = T * (T + 1) / 2;
    int64_t s = std::max<int64_t>(512 / p.ntri, 1);          // one round of 256 CUs x 2 resident workgroups (the kernel is bound by the
                                                             // matrix pipe; more slabs only add partial tiles to write and reduce)
    const int64_t max_s = ceil_div(N, SC_ROWS * SC_FLUSH);   // at least one full f32 chain per slab
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    p.slab_rows = ceil_div(ceil_div(N, s), SC_ROWS) * SC_ROWS;
    p.nslabs = (int)ceil_div(N, p.slab_rows);
    return p;
}

static size_t stats_ws(int64_t N, int D, const StatsPlan& p) {
    Carver c(nullptr, 0);
    c.take<double>((size_t)p.cs_blocks * D);
    c.take<double>((size_t)p.nslabs * p.ntri * TB * TB);
    c.take<double>(D);
    return c.off;
}

static int check_x(const float* X, int64_t N, int D, int64_t ld) {
    AM_REQUIRE(X != nullptr, AM_ERR_BAD_ARG, "X is null");
    AM_REQUIRE(N >= 1 && D >= 1, AM_ERR_BAD_SHAPE, "X has shape %lld x %d", (long long)N, D);
    AM_REQUIRE(aligned16(X) && ld % 4 == 0 && ld >= D, AM_ERR_BAD_ARG,
               "X must be 16-byte aligned with ld %% 4 == 0 and ld >= D (ld=%lld, D=%d)", (long long)ld, D);
    return AM_OK;
}

// ---- float64 rows (data.py:37-58 computes in the dtype it is given: the reference's own test embedder and the output of its
//      PCA projection are f64).  Column sums per row block + the same fixed-order reduction as above; the centred scatter
//      on v_mfma_f64_16x16x4_f64: a workgroup owns a 32 x 32 block of the upper triangle of the D x D matrix (wave w the
//      16 x 16 tile (w >> 1, w & 1)) over one slab of rows, lane (l15, l4) feeding row n0 + l4, column c0 + l15 - no LDS, the
//      operands come straight from the cache (a rare path: ~2 ms for 100 000 x 512).  Per-slab partial blocks, reduced in a
//      fixed order and mirrored (deterministic).
constexpr int S64_TILE = 32;
__global__ void __launch_bounds__(256) colsum64_partial_kernel(const double* __restrict__ X, int64_t N, int64_t ld, int D,
                                                               int64_t rows_per_block, double* __restrict__ partial) {
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = (r0 + rows_per_block < N) ? r0 + rows_per_block : N;
    for (int d = threadIdx.x; d < D; d += 256) {
        double s = 0.0;
        for (int64_t row = r0; row < r1; ++row) s += X[row * ld + d];
        partial[(int64_t)blockIdx.x * D + d] = s;
    }
}

__global__ void __launch_bounds__(256) scatter64_partial_kernel(const double* __restrict__ X, int64_t N, int64_t ld, int D,
                                                                const double* __restrict__ mean, int64_t slab_rows, int ntri,
                                                                double* __restrict__ partial) {
    const int T = (D + S64_TILE - 1) / S64_TILE;
    int tp, tq;
    tri_decode((int)(blockIdx.x % ntri), T, tp, tq);
    const int64_t slab = blockIdx.x / ntri;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int ca = tp * S64_TILE + (wave >> 1) * 16 + l15, cb = tq * S64_TILE + (wave & 1) * 16 + l15;
    const double ma = ca < D ? mean[ca] : 0.0, mb = cb < D ? mean[cb] : 0.0;
    const int64_t r0 = slab * slab_rows;
    const int64_t r1 = (r0 + slab_rows < N) ? r0 + slab_rows : N;
    typedef double f64x4s __attribute__((ext_vector_type(4)));
    f64x4s acc = {0.0, 0.0, 0.0, 0.0};
    for (int64_t n0 = r0; n0 < r1; n0 += 16) {
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {                             // four independent row groups in flight
            const int64_t n = n0 + u * 4 + l4;
            const bool in = n < r1;
            a[u] = (in && ca < D) ? X[n * ld + ca] - ma : 0.0;
            b[u] = (in && cb < D) ? X[n * ld + cb] - mb : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
    }
    // C layout: row (first factor's column) = (lane >> 4) + 4 reg, column (second factor's column) = lane & 15
    double* out = partial + ((int64_t)slab * ntri + (blockIdx.x % ntri)) * (S64_TILE * S64_TILE);
#pragma unroll
    for (int r = 0; r < 4; ++r) out[((wave >> 1) * 16 + l4 + 4 * r) * S64_TILE + (wave & 1) * 16 + l15] = acc[r];
}

__global__ void __launch_bounds__(256) scatter64_reduce_kernel(const double* __restrict__ partial, int nslabs, int ntri, int D,
                                                               double scale, double* __restrict__ out) {
    const int T = (D + S64_TILE - 1) / S64_TILE;
    int tp, tq;
    tri_decode((int)blockIdx.y, T, tp, tq);
    const int e = blockIdx.x * 256 + threadIdx.x;                 // element of the 32 x 32 block
    const int i = tp * S64_TILE + e / S64_TILE, j = tq * S64_TILE + e % S64_TILE;
    if (i >= D || j >= D) return;
    double s = 0.0;
    for (int sl = 0; sl < nslabs; ++sl) s += partial[((int64_t)sl * ntri + blockIdx.y) * (S64_TILE * S64_TILE) + e];
    s *= scale;
    if (tp != tq || j >= i) out[(int64_t)i * D + j] = s;           // diagonal blocks: the upper triangle decides (exactly symmetric)
    if (tp != tq || j > i) out[(int64_t)j * D + i] = s;
}

struct Stats64Plan {
    int cs_blocks, ntri, nslabs;
    int64_t cs_rows, slab_rows;
};
static Stats64Plan plan_stats64(int64_t N, int D) {
    Stats64Plan p;
    int64_t b = std::min<int64_t>(std::max<int64_t>(ceil_div(N, 256), 1), 512);
    p.cs_rows = ceil_div(N, b);
    p.cs_blocks = (int)ceil_div(N, p.cs_rows);
    const int T = (int)ceil_div(D, S64_TILE);
    p.ntri = T * (T + 1) / 2;
    int64_t s = std::max<int64_t>(2048 / p.ntri, 1);                 // ~8 workgroups per CU in all
    s = std::min<int64_t>(s, std::max<int64_t>(ceil_div(N, 256), 1));   // at least 256 rows per slab
    s = std::min<int64_t>(s, 64);
    p.slab_rows = ceil_div(ceil_div(N, s), 16) * 16;
    p.nslabs = (int)ceil_div(N, p.slab_rows);
    return p;
}
static size_t stats64_ws(int D, const Stats64Plan& p) {
    Carver c(nullptr, 0);
    c.take<double>((size_t)p.cs_blocks * D);
    c.take<double>((size_t)p.nslabs * p.ntri * S64_TILE * S64_TILE);
    return c.off;
}

static int run_colsum(const float* X, int64_t N, int D, int64_t ld, double scale, double* out, double* partial,
                      const StatsPlan& p, hipStream_t st) {
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(p.cs_blocks), dim3(256), 0, st, X, N, ld, D, p.cs_rows, partial);
    AM_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)ceil_div(D, CSR_COLS)), dim3(256), 0, st, partial, p.cs_blocks, D,
                       scale, out);
    AM_LAUNCH_CHECK();
    return AM_OK;
}

static int run_scatter(const float* X, int64_t N, int D, int64_t ld, const double* mean, double scale, double* out,
                       double* partial, const StatsPlan& p, hipStream_t st) {
    if (D % TB == 0)
        hipLaunchKernelGGL(scatter_partial_kernel<true>, dim3((unsigned)(p.ntri * p.nslabs)), dim3(ENGINE_THREADS), 0, st, X, N, ld,
                           D, mean, p.slab_rows, p.ntri, partial);
    else
        hipLaunchKernelGGL(scatter_partial_kernel<false>, dim3((unsigned)(p.ntri * p.nslabs)), dim3(ENGINE_THREADS), 0, st, X, N, ld,
                           D, mean, p.slab_rows, p.ntri, partial);
    AM_LAUNCH_CHECK();
    hipLaunchKernelGGL(scatter_reduce_kernel, dim3((unsigned)(TB * TB / 256), (unsigned)p.ntri), dim3(256), 0, st, partial,
                       p.nslabs, p.ntri, D, scale, out);
    AM_LAUNCH_CHECK();
    return AM_OK;
}

}  // namespace am

using namespace am;

extern "C" size_t am_stats_workspace_bytes(int64_t N, int D) {
    if (N < 1 || D < 1) return 0;
    return stats_ws(N, D, plan_stats(N, D));
}

extern "C" int am_colsum_f32(const float* X, int64_t N, int D, int64_t ld, double* colsum, void* ws, size_t ws_bytes,
                             am_stream_t stream) {
    int rc;
    if ((rc = check_x(X, N, D, ld)) != AM_OK) return rc;
    AM_REQUIRE(colsum != nullptr, AM_ERR_BAD_ARG, "colsum is null");
    const StatsPlan p = plan_stats(N, D);
    Carver c(ws, ws_bytes);
    double* cs_part = c.take<double>((size_t)p.cs_blocks * D);
    AM_REQUIRE(c.ok(), AM_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", c.off, ws_bytes);
    return run_colsum(X, N, D, ld, 1.0, colsum, cs_part, p, static_cast<hipStream_t>(stream));
}

extern "C" int am_scatter_f32(const float* X, int64_t N, int D, int64_t ld, const double* mean, double* scatter,
                              void* ws, size_t ws_bytes, am_stream_t stream) {
    int rc;
    if ((rc = check_x(X, N, D, ld)) != AM_OK) return rc;
    AM_REQUIRE(mean && scatter, AM_ERR_BAD_ARG, "mean/scatter is null");
    const StatsPlan p = plan_stats(N, D);
    Carver c(ws, ws_bytes);
    c.take<double>((size_t)p.cs_blocks * D);
    double* sc_part = c.take<double>((size_t)p.nslabs * p.ntri * TB * TB);
    AM_REQUIRE(c.ok(), AM_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", c.off, ws_bytes);
    return run_scatter(X, N, D, ld, mean, 1.0, scatter, sc_part, p, static_cast<hipStream_t>(stream));
}

extern "C" int am_stats_f32(const float* X, int64_t N, int D, int64_t ld, double* mean, double* cov, void* ws,
                            size_t ws_bytes, am_stream_t stream) {
    int rc;
    if ((rc = check_x(X, N, D, ld)) != AM_OK) return rc;
    AM_REQUIRE(mean && cov, AM_ERR_BAD_ARG, "mean/cov is null");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const StatsPlan p = plan_stats(N, D);
    Carver c(ws, ws_bytes);
    double* cs_part = c.take<double>((size_t)p.cs_blocks * D);
    double* sc_part = c.take<double>((size_t)p.nslabs * p.ntri * TB * TB);
    AM_REQUIRE(c.ok(), AM_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", c.off, ws_bytes);
    if ((rc = run_colsum(X, N, D, ld, 1.0 / (double)N, mean, cs_part, p, st)) != AM_OK) return rc;
    if (N == 1) {                                          // data.py:40-42
        hipLaunchKernelGGL(zero_f64_kernel, dim3((unsigned)ceil_div((int64_t)D * D, 256)), dim3(256), 0, st, cov,
                           (int64_t)D * D);
        AM_LAUNCH_CHECK();
        return AM_OK;
    }
    return run_scatter(X, N, D, ld, mean, 1.0 / (double)(N - 1), cov, sc_part, p, st);
}

extern "C" size_t am_stats_f64_workspace_bytes(int64_t N, int D) {
    if (N < 1 || D < 1) return 0;
    return stats64_ws(D, plan_stats64(N, D));
}

// D above ~11 500 would need more than 65 535 triangle blocks in grid.y of the reduce (and the operands are read straight
// from memory with a row stride): far beyond any embedding width, refused instead of failing at launch
static int check_x64(const double* X, int64_t N, int D, int64_t ld) {
    AM_REQUIRE(X != nullptr, AM_ERR_BAD_ARG, "X is null");
    AM_REQUIRE(N >= 1 && D >= 1, AM_ERR_BAD_SHAPE, "X has shape %lld x %d", (long long)N, D);
    AM_REQUIRE(D <= 8192, AM_ERR_BAD_SHAPE, "float64 statistics take rows of up to 8192 elements (D=%d)", D);
    AM_REQUIRE(ld >= D, AM_ERR_BAD_ARG, "ld=%lld < D=%d", (long long)ld, D);
    return AM_OK;
}

static int run_colsum64(const double* X, int64_t N, int D, int64_t ld, double scale, double* out, double* cs_part, const Stats64Plan& p,
                        hipStream_t st) {
    hipLaunchKernelGGL(colsum64_partial_kernel, dim3(p.cs_blocks), dim3(256), 0, st, X, N, ld, D, p.cs_rows, cs_part);
    AM_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)ceil_div(D, CSR_COLS)), dim3(256), 0, st, cs_part, p.cs_blocks, D, scale, out);
    AM_LAUNCH_CHECK();
    return AM_OK;
}

static int run_scatter64(const double* X, int64_t N, int D, int64_t ld, const double* mean, double scale, double* out, double* sc_part,
                         const Stats64Plan& p, hipStream_t st) {
    hipLaunchKernelGGL(scatter64_partial_kernel, dim3((unsigned)(p.ntri * p.nslabs)), dim3(256), 0, st, X, N, ld, D, mean, p.slab_rows,
                       p.ntri, sc_part);
    AM_LAUNCH_CHECK();
    hipLaunchKernelGGL(scatter64_reduce_kernel, dim3(S64_TILE * S64_TILE / 256, (unsigned)p.ntri), dim3(256), 0, st, sc_part,
                       p.nslabs, p.ntri, D, scale, out);
    AM_LAUNCH_CHECK();
    return AM_OK;
}

extern "C" int am_colsum_f64(const double* X, int64_t N, int D, int64_t ld, double* colsum, void* ws, size_t ws_bytes, am_stream_t stream) {
    int rc;
    if ((rc = check_x64(X, N, D, ld)) != AM_OK) return rc;
    AM_REQUIRE(colsum != nullptr, AM_ERR_BAD_ARG, "colsum is null");
    const Stats64Plan p = plan_stats64(N, D);
    Carver c(ws, ws_bytes);
    double* cs_part = c.take<double>((size_t)p.cs_blocks * D);
    AM_REQUIRE(c.ok(), AM_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", c.off, ws_bytes);
    return run_colsum64(X, N, D, ld, 1.0, colsum, cs_part, p, static_cast<hipStream_t>(stream));
}

extern "C" int am_scatter_f64(const double* X, int64_t N, int D, int64_t ld, const double* mean, double* scatter, void* ws, size_t ws_bytes,
                              am_stream_t stream) {
    int rc;
    if ((rc = check_x64(X, N, D, ld)) != AM_OK) return rc;
    AM_REQUIRE(mean && scatter, AM_ERR_BAD_ARG, "mean/scatter is null");
    const Stats64Plan p = plan_stats64(N, D);
    Carver c(ws, ws_bytes);
    c.take<double>((size_t)p.cs_blocks * D);
    double* sc_part = c.take<double>((size_t)p.nslabs * p.ntri * S64_TILE * S64_TILE);
    AM_REQUIRE(c.ok(), AM_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", c.off, ws_bytes);
    return run_scatter64(X, N, D, ld, mean, 1.0, scatter, sc_part, p, static_cast<hipStream_t>(stream));
}

extern "C" int am_stats_f64(const double* X, int64_t N, int D, int64_t ld, double* mean, double* cov, void* ws, size_t ws_bytes,
                            am_stream_t stream) {
    int rc;
    if ((rc = check_x64(X, N, D, ld)) != AM_OK) return rc;
    AM_REQUIRE(mean && cov, AM_ERR_BAD_ARG, "mean/cov is null");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const Stats64Plan p = plan_stats64(N, D);
    Carver c(ws, ws_bytes);
    double* cs_part = c.take<double>((size_t)p.cs_blocks * D);
    double* sc_part = c.take<double>((size_t)p.nslabs * p.ntri * S64_TILE * S64_TILE);
    AM_REQUIRE(c.ok(), AM_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", c.off, ws_bytes);
    if ((rc = run_colsum64(X, N, D, ld, 1.0 / (double)N, mean, cs_part, p, st)) != AM_OK) return rc;
    if (N == 1) {                                          // data.py:40-42
        hipLaunchKernelGGL(zero_f64_kernel, dim3((unsigned)ceil_div((int64_t)D * D, 256)), dim3(256), 0, st, cov, (int64_t)D * D);
        AM_LAUNCH_CHECK();
        return AM_OK;
    }
    return run_scatter64(X, N, D, ld, mean, 1.0 / (double)(N - 1), cov, sc_part, p, st);
}

extern "C" int am_stats_merge_f64(int64_t n1, const double* mean1, const double* cov1, int64_t n2, const double* mean2,
                                  const double* cov2, int D, double* out_mean, double* out_cov, am_stream_t stream) {
    AM_REQUIRE(mean1 && cov1 && mean2 && cov2 && out_mean && out_cov, AM_ERR_BAD_ARG, "null pointer");
    AM_REQUIRE(D >= 1 && n1 >= 1 && n2 >= 1, AM_ERR_BAD_SHAPE, "n1=%lld n2=%lld D=%d", (long long)n1, (long long)n2, D);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(merge_cov_kernel, dim3((unsigned)ceil_div(D, 128), (unsigned)D), dim3(128), 0, st, n1, mean1, cov1,
                       n2, mean2, cov2, D, out_cov);
    AM_LAUNCH_CHECK();
    hipLaunchKernelGGL(merge_mean_kernel, dim3((unsigned)ceil_div(D, 128)), dim3(128), 0, st, n1, mean1, n2, mean2, D,
                       out_mean);
    AM_LAUNCH_CHECK();
    return AM_OK;
}

extern "C" int am_stats_push_max_rows(void) { return PUSH_MAX_ROWS; }

extern "C" int am_stats_push_f32(const float* E, int64_t b, int D, int64_t ld, int64_t n_old, const double* mean_in,
                                 double* mean_out, double* cov, float* rows_out, int64_t ld_out, am_stream_t stream) {
    AM_REQUIRE(E != nullptr && mean_out != nullptr && cov != nullptr, AM_ERR_BAD_ARG, "null pointer");
    AM_REQUIRE(b >= 1 && b <= PUSH_MAX_ROWS && D >= 1, AM_ERR_BAD_SHAPE, "batch of %lld rows x %d (1 .. %d rows per push)",
               (long long)b, D, PUSH_MAX_ROWS);
    AM_REQUIRE(ld >= D && n_old >= 0, AM_ERR_BAD_ARG, "ld=%lld < D=%d or n_old=%lld < 0", (long long)ld, D, (long long)n_old);
    AM_REQUIRE(n_old == 0 || (mean_in != nullptr && mean_in != mean_out), AM_ERR_BAD_ARG,
               "the running mean is read from mean_in and written to a different mean_out");
    AM_REQUIRE(rows_out == nullptr || ld_out >= D, AM_ERR_BAD_ARG, "ld_out=%lld < D=%d", (long long)ld_out, D);
    const unsigned t = (unsigned)ceil_div(D, PUSH_TILE);
    hipLaunchKernelGGL(stats_push_kernel, dim3(t, t), dim3(256), 0, static_cast<hipStream_t>(stream), E, (int)b, D, ld, n_old,
                       mean_in, mean_out, cov, rows_out, ld_out);
    AM_LAUNCH_CHECK();
    return AM_OK;
}
