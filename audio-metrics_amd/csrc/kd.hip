// Kernel distance (KID-style unbiased MMD^2 with a polynomial kernel), reference
// kd.py:38-83 (mmd2), 112-116 (polynomial_kernel), 119-124, 178-187 (subset loop).
//
// One launch covers all S subsets.  Per subset the three m x m Gram blocks
// Kxx, Kyy, Kxy are cut into 128x128 tiles (Kxx / Kyy: upper-triangular tiles
// only, off-diagonal tiles weighted 2) and every tile is one workgroup of the
// f32-MFMA tile engine fed by rows GATHERED through the subset's index list.
// Epilogue: K = (dot*gamma + coef0)^degree in f64, summed in f64 (the reference
// forms K in f32 and sums with numpy's pairwise f32 sums; f64 keeps the device
// on the exact side of the reference's own rounding noise).
#include "am_common.h"
#include "tile_engine.h"

namespace am {

struct GatherRows {           // local row -> X[idx[tile*128 + row]], zero rows past m
    const float* base;
    int64_t ld;
    const int64_t* idx;
    int m, tile;
    __device__ __forceinline__ const float* operator()(int, int row) const {
        const int p = tile * TB + row;
        return p < m ? base + idx[p] * ld : nullptr;
    }
};

// kernel value from the f32 dot product: polynomial (dot*gamma + coef0)^degree (kd.py:112-116) or, with
// rbf != 0, exp(-(|x|^2 + |y|^2 - 2 dot) / (2 sigma^2)) (kd.py:86-109; gamma then holds 1/(2 sigma^2) and
// qn / pn the f64 squared norms of the subset rows, +inf for padded rows so that they contribute exactly 0).
struct KdEpilogue {
    double gamma, coef0;
    int degree, m;
    int rbf;
    const double* qn;          // [m] squared norms of the Q rows of this subset (RBF only)
    const double* pn;
    int q0, p0;                // first subset position of the Q (register) / P (lane) rows of this tile
    bool drop_diag;
    double sum;
    const LaneInfo& L;
    __device__ __forceinline__ KdEpilogue(const LaneInfo& l) : L(l) {}
    __device__ __forceinline__ void aux_issue(int, int64_t) {}
    __device__ __forceinline__ void aux_commit(int) {}
    __device__ __forceinline__ double kval(float dot) const {
        const double base = (double)dot * gamma + coef0;
        double k = 1.0;
        for (int d = 0; d < degree; ++d) k *= base;
        return k;
    }
    // Sums ALL 128x128 entries of the tile; padded (zero) rows give exactly kval(0) each, which the
    // caller subtracts analytically.  Diagonal entries of Kxx / Kyy are removed here (valid ones only).
    __device__ __forceinline__ void finish_rbf(f32x16 (&acc)[2][2]) {
        double s = 0.0;
        double pnorm[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int p = p0 + L.wn * 64 + nt * 32 + L.r;
            pnorm[nt] = p < m ? pn[p] : INFINITY;
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int q = q0 + L.wm * 64 + mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * L.h;
                const double qnorm = q < m ? qn[q] : INFINITY;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int p = p0 + L.wn * 64 + nt * 32 + L.r;
                    double d2 = (qnorm + pnorm[nt]) - 2.0 * (double)acc[mt][nt][i];
                    d2 = d2 < 0.0 ? 0.0 : d2;
                    const double k = exp(-d2 * gamma);                  // padded rows: exp(-inf) = 0
                    s += (drop_diag && p == q) ? 0.0 : k;
                }
            }
        sum = s;
    }
    __device__ __forceinline__ void finish(int, int64_t, f32x16 (&acc)[2][2]) {
        if (rbf) {                                   // compile-time constant per kernel instantiation
            finish_rbf(acc);
            return;
        }
        double s = 0.0;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) s += kval(acc[mt][nt][i]);
        if (drop_diag && L.wm == L.wn) {                 // wave-uniform: only waves that straddle the diagonal
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int q = q0 + L.wm * 64 + mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * L.h;
                    const int p = p0 + L.wn * 64 + mt * 32 + L.r;
                    if (q == p && p < m) s -= kval(acc[mt][mt][i]);
                }
        }
        sum = s;
    }
};

// partial[(s * blocks_per_subset) + b] = weighted tile sum
// MODE bits: 1 = inner-dimension tail (D % 32 != 0), 2 = RBF kernel, 4 = generic pointer pipeline (matrices >= 4 GiB)
template <int MODE>
__global__ void __launch_bounds__(ENGINE_THREADS, 2)
kd_tile_kernel(const float* __restrict__ X, int64_t ldx, const float* __restrict__ Y, int64_t ldy, int D,
               const int64_t* __restrict__ idx1, const int64_t* __restrict__ idx2, int m, int T, int ntri,
               double gamma, double coef0, int degree, double* __restrict__ partial, int rbf,
               const double* __restrict__ sub_norm1, const double* __restrict__ sub_norm2, int64_t N1, int64_t N2) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const LaneInfo L;
    const int per_subset = 2 * ntri + T * T;
    const int s = blockIdx.x / per_subset;
    int b = blockIdx.x % per_subset;
    int which, tq, tp;                       // which: 0 = XX, 1 = YY, 2 = XY
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
    const GatherRows qsrc{which == 1 ? Y : X, which == 1 ? ldy : ldx, which == 1 ? i2 : i1, m, tq};
    const GatherRows psrc{which == 0 ? X : Y, which == 0 ? ldx : ldy, which == 0 ? i1 : i2, m, tp};
    const int64_t n_q_rows = which == 1 ? N2 : N1, n_p_rows = which == 0 ? N1 : N2;
    KdEpilogue epi(L);
    epi.gamma = gamma;
    epi.coef0 = coef0;
    epi.degree = degree;
    epi.m = m;
    epi.q0 = tq * TB;
    epi.p0 = tp * TB;
    epi.drop_diag = (which != 2) && (tq == tp);
    epi.sum = 0.0;
    epi.rbf = (MODE & 2) ? 1 : 0;
    epi.qn = rbf ? (which == 1 ? sub_norm2 : sub_norm1) + (int64_t)s * m : nullptr;
    epi.pn = rbf ? (which == 0 ? sub_norm1 : sub_norm2) + (int64_t)s * m : nullptr;
    // single tile per workgroup; rows gathered through the subset's index list.  One buffer descriptor spans the
    // whole matrix (the 32-bit offsets need N*ld*4 < 4 GiB; larger sets take the generic pointer pipeline)
    if constexpr ((MODE & 4) == 0) {
        const int srow = L.tid >> 3, scol = (L.tid & 7) * 4;
        auto make = [&](const GatherRows& g, int64_t n_rows) {
            TileAddr a;
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<uintptr_t>(g.base) & 0xffffffffu));
            const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<uintptr_t>(g.base) >> 32));
            const unsigned bytes = __builtin_amdgcn_readfirstlane((unsigned)((uint64_t)n_rows * (uint64_t)g.ld * 4u));
            a.rs.rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>((static_cast<uintptr_t>(hi) << 32) | lo), 0,
                                                          (int)bytes, 0x00020000);
            // four independent index loads (a conditional load used at once made the compiler wait for each in turn: eight
            // exposed memory round trips at the top of every workgroup, whose whole life is one 128 x 128 tile)
            int64_t ix[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int p = g.tile * TB + q * 32 + srow;
                ix[q] = g.idx[p < g.m ? p : g.m - 1];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int p = g.tile * TB + q * 32 + srow;
                a.vo[q] = p < g.m ? (unsigned)((ix[q] * g.ld + scol) * 4) : 0xffffffffu;      // padded rows read as 0
            }
            return a;
        };
        const TileAddr qa = make(qsrc, n_q_rows), pa = make(psrc, n_p_rows);
        auto qaddr = [&](int t) {
            TileAddr a = qa;
            if (t > 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) a.vo[q] = 0xffffffffu;   // the pipeline prefetches one tile past the end
            }
            return a;
        };
        addr_pipeline_early<0, (MODE & 1) != 0>(qaddr, pa, 1, D, 0, lds, L, epi);
    } else {
        tile_pipeline(qsrc, psrc, 1, D, lds, L, epi);
    }

    double v = epi.sum;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    double* red = reinterpret_cast<double*>(lds);          // staging slabs are idle after the pipeline's last barrier
    if (L.lane == 0) red[L.tid >> 6] = v;
    __syncthreads();
    if (L.tid == 0) {
        const double w = (which != 2 && tq != tp) ? 2.0 : 1.0;   // symmetric blocks: count the mirrored tile too
        const int vq = min(TB, m - tq * TB), vp = min(TB, m - tp * TB);
        const double pad = rbf ? 0.0 : (double)(TB * TB - vq * vp) * epi.kval(0.f);
        partial[blockIdx.x] = w * ((((red[0] + red[1]) + red[2]) + red[3]) - pad);
    }
}

__global__ void kd_finish_kernel(const double* __restrict__ partial, int S, int m, int T, int ntri,
                                 double* __restrict__ out) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const int per_subset = 2 * ntri + T * T;
    const double* p = partial + (int64_t)s * per_subset;
    double sxx = 0, syy = 0, sxy = 0;
    for (int i = 0; i < ntri; ++i) sxx += p[i];
    for (int i = 0; i < ntri; ++i) syy += p[ntri + i];
    for (int i = 0; i < T * T; ++i) sxy += p[2 * ntri + i];
    const double dm = (double)m;
    // kd.py:77-79 (unbiased): within-set sums over m(m-1) off-diagonal pairs, cross term over m^2
    out[s] = (sxx + syy) / (dm * (dm - 1.0)) - 2.0 * sxy / (dm * dm);
}

// sub_norm[s][p] = |X[idx[s][p]]|^2 in f64 (one wave per subset row)
__global__ void __launch_bounds__(256) kd_gather_norms_kernel(const float* __restrict__ X, int64_t ld, int D,
                                                              const int64_t* __restrict__ idx, int64_t total,
                                                              double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= total) return;
    const float* x = X + idx[e] * ld;
    double acc = 0.0;
    for (int k = lane * 4; k < D; k += 256) {
        const f32x4 v = load_k4(x, k, D);
        acc += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) out[e] = acc;
}

constexpr size_t KD_LDS_BYTES = ENGINE_LDS_FLOATS * sizeof(float);

static int run_kd(const float* X, int64_t N1, int64_t ldx, const float* Y, int64_t N2, int64_t ldy, int D, const int64_t* idx1,
                  const int64_t* idx2, int S, int m, double gamma, double coef0, int degree, int rbf, double* out_mmd, void* ws,
                  size_t ws_bytes, hipStream_t st) {
    AM_REQUIRE(X && Y && idx1 && idx2 && out_mmd, AM_ERR_BAD_ARG, "null pointer");
    AM_REQUIRE(N1 >= 1 && N2 >= 1 && D >= 1 && S >= 1 && m >= 1, AM_ERR_BAD_SHAPE,
               "N1=%lld N2=%lld D=%d S=%d m=%d", (long long)N1, (long long)N2, D, S, m);
    AM_REQUIRE(m <= N1 && m <= N2, AM_ERR_BAD_SHAPE, "subset size %d exceeds a set size", m);
    AM_REQUIRE(aligned16(X) && aligned16(Y) && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= D && ldy >= D, AM_ERR_BAD_ARG,
               "X/Y must be 16-byte aligned with ld %% 4 == 0 and ld >= D");
    const int T = (int)ceil_div(m, TB);
    const int ntri = T * (T + 1) / 2;
    const int per_subset = 2 * ntri + T * T;
    Carver c(ws, ws_bytes);
    double* partial = c.take<double>((size_t)S * per_subset);
    double *n1 = nullptr, *n2 = nullptr;
    if (rbf) {
        n1 = c.take<double>((size_t)S * m);
        n2 = c.take<double>((size_t)S * m);
    }
    AM_REQUIRE(c.ok(), AM_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", c.off, ws_bytes);
    if (rbf) {
        const int64_t total = (int64_t)S * m;
        hipLaunchKernelGGL(kd_gather_norms_kernel, dim3((unsigned)ceil_div(total, 4)), dim3(256), 0, st, X, ldx, D, idx1, total, n1);
        hipLaunchKernelGGL(kd_gather_norms_kernel, dim3((unsigned)ceil_div(total, 4)), dim3(256), 0, st, Y, ldy, D, idx2, total, n2);
        AM_LAUNCH_CHECK();
    }
    const bool generic = (uint64_t)N1 * (uint64_t)ldx * 4u >= 0xffffffffull || (uint64_t)N2 * (uint64_t)ldy * 4u >= 0xffffffffull;
    const int mode = (((D % BK) != 0 && !generic) ? 1 : 0) | (rbf ? 2 : 0) | (generic ? 4 : 0);
    auto launch = [&](auto kernel) -> int {
        AM_HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), (int)KD_LDS_BYTES));
        hipLaunchKernelGGL(kernel, dim3((unsigned)((int64_t)S * per_subset)), dim3(ENGINE_THREADS), KD_LDS_BYTES, st,
                           X, ldx, Y, ldy, D, idx1, idx2, m, T, ntri, gamma, coef0, degree, partial, rbf, n1, n2, N1, N2);
        AM_LAUNCH_CHECK();
        return AM_OK;
    };
    int rc;
    switch (mode) {
        case 0: rc = launch(&kd_tile_kernel<0>); break;
        case 1: rc = launch(&kd_tile_kernel<1>); break;
        case 2: rc = launch(&kd_tile_kernel<2>); break;
        case 3: rc = launch(&kd_tile_kernel<3>); break;
        case 4: rc = launch(&kd_tile_kernel<4>); break;
        default: rc = launch(&kd_tile_kernel<6>); break;
    }
    if (rc != AM_OK) return rc;
    hipLaunchKernelGGL(kd_finish_kernel, dim3((unsigned)ceil_div(S, 64)), dim3(64), 0, st, partial, S, m, T, ntri,
                       out_mmd);
    AM_LAUNCH_CHECK();
    return AM_OK;
}

}  // namespace am

using namespace am;

extern "C" size_t am_kd_workspace_bytes(int S, int m) {
    if (S < 1 || m < 1) return 0;
    const int T = (int)ceil_div(m, TB);
    Carver c(nullptr, 0);
    c.take<double>((size_t)S * (T * (T + 1) + T * T));
    return c.off;
}

extern "C" int am_kd_poly_f32(const float* X, int64_t N1, int64_t ldx, const float* Y, int64_t N2, int64_t ldy, int D,
                              const int64_t* idx1, const int64_t* idx2, int S, int m, double gamma, double coef0,
                              int degree, double* out_mmd, void* ws, size_t ws_bytes, am_stream_t stream) {
    AM_REQUIRE(degree >= 0 && degree <= 16, AM_ERR_BAD_ARG, "degree %d outside [0, 16]", degree);
    return run_kd(X, N1, ldx, Y, N2, ldy, D, idx1, idx2, S, m, gamma, coef0, degree, 0, out_mmd, ws, ws_bytes,
                  static_cast<hipStream_t>(stream));
}

extern "C" size_t am_kd_rbf_workspace_bytes(int S, int m) {
    if (S < 1 || m < 1) return 0;
    const int T = (int)ceil_div(m, TB);
    Carver c(nullptr, 0);
    c.take<double>((size_t)S * (T * (T + 1) + T * T));
    c.take<double>((size_t)S * m);
    c.take<double>((size_t)S * m);
    return c.off;
}

extern "C" int am_kd_rbf_f32(const float* X, int64_t N1, int64_t ldx, const float* Y, int64_t N2, int64_t ldy, int D,
                             const int64_t* idx1, const int64_t* idx2, int S, int m, double sigma, double* out_mmd,
                             void* ws, size_t ws_bytes, am_stream_t stream) {
    AM_REQUIRE(sigma > 0, AM_ERR_BAD_ARG, "sigma must be positive");
    return run_kd(X, N1, ldx, Y, N2, ldy, D, idx1, idx2, S, m, 1.0 / (2.0 * sigma * sigma), 0.0, 0, 1, out_mmd, ws, ws_bytes,
                  static_cast<hipStream_t>(stream));
}
