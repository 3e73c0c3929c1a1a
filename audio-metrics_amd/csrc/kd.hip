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
#include "pairwise_common.h"
#include "wide_engine.h"
#include <algorithm>

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


// ------------------------------------------------------------------------------------------------
// Split-f16 form of the polynomial kernel distance on the 256 x 256 f16 engine (wide_engine.h).
//
// The f32 MFMA runs at 1/16 of the f16 rate, and kd_tile_kernel already sits at 96 % of it (2.1 ms per evaluate at
// S = 100, m = 1000, D = 512).  Here every gathered row x is scaled by its own exact power of two (largest |element| in
// [2^13, 2^14)) and split into two f16 planes, hi = rn16(x) and lo = rn16(x - hi) - x = hi + lo up to 2^-22 |x| - and
// the dot product is accumulated as  <hi, hi'> + <lo, hi'> + <hi, lo'>  by three stages of the unchanged pipeline per
// 64-element slab (SplitSlabs below: the Q operand walks the planes hi, lo, hi, the P operand hi, hi, lo).  f16 x f16
// products are exact in the f32 accumulator, so what is left out is the lo x lo' term and the two plane roundings:
// |error of a dot product| <= ~3 * 2^-22 |x| |y| - the size of one f32 rounding of the f32 chain it replaces - and the
// kernel values and every sum are formed in f64 exactly as before.  Measured against the reference's own outputs: see
// tests/test_gpu_parity.py::test_kd_vs_golden (same tolerances as the f32 form).
// Layout: planes[(s * 2 + set) * MP + p] = [hi: DP f16 | lo: DP f16], MP = m rounded up to 256, DP = D rounded up to 64,
// rows p >= m zero; unscale[...] = 2^-e of the row (the factor that undoes its scaling).
constexpr int KW = 256;

__global__ void __launch_bounds__(256) kd_split_gather_kernel(const float* __restrict__ X, int64_t ldx, const float* __restrict__ Y,
                                                              int64_t ldy, int D, int DP, const int64_t* __restrict__ idx1,
                                                              const int64_t* __restrict__ idx2, int S, int m, int MP,
                                                              uint16_t* __restrict__ planes, float* __restrict__ unscale) {
    const int lane = threadIdx.x & 63;
    const int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);       // one wave per gathered row
    if (e >= (int64_t)S * 2 * MP) return;
    const int p = (int)(e % MP), set = (int)((e / MP) & 1);
    const int64_t s = e / (2 * (int64_t)MP);
    uint16_t* dst = planes + e * 2 * DP;
    if (p >= m) {                                                          // padded row: zeros, contributes kval(0)
        for (int k = lane * 8; k < 2 * DP; k += 512) *reinterpret_cast<uint4*>(dst + k) = make_uint4(0u, 0u, 0u, 0u);
        if (lane == 0) unscale[e] = 0.f;
        return;
    }
    const float* x = set ? Y + idx2[s * m + p] * ldy : X + idx1[s * m + p] * ldx;
    float mx = 0.f;
    for (int k = lane * 4; k < D; k += 256) {
        const f32x4 v = load_k4(x, k, D);
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    const int ex = half_scale_exp(__float_as_uint(mx));                    // 13 - floor(log2(max)), clamped (pairwise_common.h)
    const float up = __uint_as_float((unsigned)(127 + ex) << 23);
    if (lane == 0) unscale[e] = __uint_as_float((unsigned)(127 - ex) << 23);
    for (int k = lane * 4; k < DP; k += 256) {
        const f32x4 v = load_k4(x, k, D);                                  // zero past D
        _Float16 hi[4], lo[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float t = v[c] * up;
            hi[c] = (_Float16)t;
            lo[c] = (_Float16)(t - (float)hi[c]);
        }
        *reinterpret_cast<uint2*>(dst + k) = *reinterpret_cast<const uint2*>(hi);
        *reinterpret_cast<uint2*>(dst + DP + k) = *reinterpret_cast<const uint2*>(lo);
    }
}

struct SplitSlabs {                   // three stages per 64-element slab: Q planes hi, lo, hi against P planes hi, hi, lo
    unsigned plane_bytes;
    __device__ __forceinline__ int count(int Dh) const { return 3 * (Dh / WROW); }
    __device__ __forceinline__ unsigned q(int kt) const { return (unsigned)((kt / 3) * WROW * 4) + ((kt % 3) == 1 ? plane_bytes : 0u); }
    __device__ __forceinline__ unsigned p(int kt) const { return (unsigned)((kt / 3) * WROW * 4) + ((kt % 3) == 2 ? plane_bytes : 0u); }
};

struct KdWideEpilogue {
    double gamma, coef0;
    int degree, m;
    const float* qun;           // unscale factors of the Q rows of this (subset, set)
    float* aux;                 // LDS [256]
    float pun[2];               // unscale factors of this lane's two P rows
    int q0, p0;
    bool drop_diag;
    double sum;
    float aux_v;
    const WLane& L;
    __device__ __forceinline__ KdWideEpilogue(const WLane& l) : L(l) {}
    __device__ __forceinline__ void aux_issue(int, int64_t) {
        if (L.tid < KW) aux_v = qun[q0 + L.tid];
    }
    __device__ __forceinline__ void aux_commit(int) {
        if (L.tid < KW) aux[L.tid] = aux_v;
    }
    // degree 3 (kd.py:22, the only degree this form is built for: a run-time power loop per accumulator element makes the
    // epilogue 128 small loops, and the register allocator then spills accumulators inside the MAIN loop)
    __device__ __forceinline__ double kval(double dot) const {
        const double base = dot * gamma + coef0;
        return base * base * base;
    }
    // sums ALL 256 x 256 entries of the tile (padded rows give exactly kval(0) each: subtracted by the caller) and takes
    // the valid diagonal entries of a diagonal Kxx / Kyy tile out again
    __device__ __forceinline__ void finish(int, int64_t, f32x16 (&acc)[4][2]) {
        const float* a = aux + L.wm * 128 + L.h * 4;
        double s = 0.0;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            f32x4 qs[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) qs[g4] = *reinterpret_cast<const f32x4*>(a + mt * 32 + g4 * 8);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                double part = 0.0;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const float dot = acc[mt][nt][reg] * (qs[reg >> 2][reg & 3] * pun[nt]);      // exact powers of two
                    part += kval((double)dot);
                }
                s += part;
                // (one accumulator tile at a time: left to itself the scheduler converts all 128 accumulators to f64 first
                // and spills 170 registers)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (drop_diag && L.wm == (L.wn >> 1)) {          // wave-uniform: only waves whose 128 x 64 block meets the diagonal
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    if ((L.wn & 1) * 2 + nt != mt) continue;     // Q rows wm*128 + mt*32 + .. against P rows wn*64 + nt*32 + ..
                    const float qsel[4] = {a[mt * 32 + 0], a[mt * 32 + 8], a[mt * 32 + 16], a[mt * 32 + 24]};
                    (void)qsel;
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int q = q0 + L.wm * 128 + mt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * L.h;
                        const int p = p0 + L.wn * 64 + nt * 32 + L.r;
                        if (q == p && p < m) {
                            const float dot = acc[mt][nt][reg] * (a[mt * 32 + (reg >> 2) * 8 + (reg & 3)] * pun[nt]);
                            s -= kval((double)dot);
                        }
                    }
                }
        }
        sum = s;
    }
};

constexpr size_t KD_WIDE_LDS_BYTES = (WENGINE_LDS_WORDS + KW) * sizeof(float);

__global__ void __launch_bounds__(WTHREADS, 1)
kd_wide_kernel(const float* __restrict__ planes, const float* __restrict__ unscale, int DP, int m, int MP, int T, int ntri,
               double gamma, double coef0, int degree, double* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const WLane L;
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
    // Kxy[a][b] = k(x_a, y_b): Q rows (register axis) from set 1 (X), P rows (lane axis) from set 2 (Y)
    const int qset = which == 1 ? 1 : 0, pset = which == 0 ? 0 : 1;
    const int64_t ldw = DP;                                              // words per row: two planes of DP f16
    const float* Q = planes + ((int64_t)s * 2 + qset) * MP * ldw;
    const float* P = planes + ((int64_t)s * 2 + pset) * MP * ldw;
    KdWideEpilogue epi(L);
    epi.gamma = gamma;
    epi.coef0 = coef0;
    epi.degree = degree;
    epi.m = m;
    epi.qun = unscale + ((int64_t)s * 2 + qset) * MP;
    epi.aux = lds + WENGINE_LDS_WORDS;
    epi.q0 = tq * KW;
    epi.p0 = tp * KW;
    epi.drop_diag = (which != 2) && (tq == tp);
    epi.sum = 0.0;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) epi.pun[nt] = unscale[((int64_t)s * 2 + pset) * MP + tp * KW + L.wn * 64 + nt * 32 + L.r];
    const SplitSlabs slabs{(unsigned)(DP * 2)};
    wide_pipeline(Q, MP, ldw, WideSingleTile{tq}, P, MP, ldw, (int64_t)tp * KW, 1, DP / 2, lds, L, epi, slabs);
    double v = epi.sum;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    double* red = reinterpret_cast<double*>(lds);          // the engine's buffers are idle after the pipeline's last barrier
    if (L.lane == 0) red[L.wave] = v;
    __syncthreads();
    if (L.tid == 0) {
        double t = 0.0;
        for (int w = 0; w < 8; ++w) t += red[w];
        const double w2 = (which != 2 && tq != tp) ? 2.0 : 1.0;   // symmetric blocks: count the mirrored tile too
        const int vq = max(0, min(KW, m - tq * KW)), vp = max(0, min(KW, m - tp * KW));
        const double pad = (double)(KW * KW - vq * vp) * epi.kval(0.0);
        partial[blockIdx.x] = w2 * (t - pad);
    }
}

static bool kd_wide_eligible(int S, int m, int D, int rbf, int degree = 3) {
    (void)S;
    return !rbf && degree == 3 && m >= 512 && D >= 128 && D <= 8192;   // (smaller: a subset is one or two tiles, the f32 form is as fast)
}
static size_t kd_wide_ws(int S, int m, int D) {
    const int MP = (int)ceil_div(m, KW) * KW, DP = (int)ceil_div(D, 64) * 64, T = MP / KW;
    Carver c(nullptr, 0);
    c.take<double>((size_t)S * (T * (T + 1) + T * T));
    c.take<uint16_t>((size_t)S * 2 * MP * 2 * DP);
    c.take<float>((size_t)S * 2 * MP);
    return c.off;
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
    if (kd_wide_eligible(S, m, D, rbf, degree)) {        // split-f16 form on the 256 x 256 engine
        const int MP = (int)ceil_div(m, KW) * KW, DP = (int)ceil_div(D, 64) * 64, T = MP / KW;
        const int ntri = T * (T + 1) / 2, per_subset = 2 * ntri + T * T;
        Carver c(ws, ws_bytes);
        double* partial = c.take<double>((size_t)S * per_subset);
        uint16_t* planes = c.take<uint16_t>((size_t)S * 2 * MP * 2 * DP);
        float* unscale = c.take<float>((size_t)S * 2 * MP);
        // the form is a function of the SHAPES alone (identical inputs give identical bits whatever buffer they come with): a
        // workspace sized by the D-less am_kd_workspace_bytes is an error here, not a switch to the f32 kernel
        AM_REQUIRE(c.ok(), AM_ERR_WORKSPACE, "workspace too small for the split-f16 form these shapes take: need %zu bytes "
                   "(am_kd_poly_workspace_bytes), have %zu", c.off, ws_bytes);
        const int64_t rows = (int64_t)S * 2 * MP;
        hipLaunchKernelGGL(kd_split_gather_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, st, X, ldx, Y, ldy, D, DP, idx1,
                           idx2, S, m, MP, planes, unscale);
        AM_LAUNCH_CHECK();
        AM_HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(&kd_wide_kernel), (int)KD_WIDE_LDS_BYTES));
        hipLaunchKernelGGL(kd_wide_kernel, dim3((unsigned)((int64_t)S * per_subset)), dim3(WTHREADS), KD_WIDE_LDS_BYTES, st,
                           reinterpret_cast<const float*>(planes), unscale, DP, m, MP, T, ntri, gamma, coef0, degree, partial);
        AM_LAUNCH_CHECK();
        hipLaunchKernelGGL(kd_finish_kernel, dim3((unsigned)ceil_div(S, 64)), dim3(64), 0, st, partial, S, m, T, ntri, out_mmd);
        AM_LAUNCH_CHECK();
        return AM_OK;
    }
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
    // the f32 form's workspace: enough for every shape that does NOT take the split-f16 form (m < 512, D < 128 or D > 8192,
    // degree != 3).  This query does not know the feature width; am_kd_poly_workspace_bytes(S, m, D) is the one that is
    // right for every shape, and am_kd_poly_f32 returns AM_ERR_WORKSPACE (with the size it needs) for a short buffer.
    if (S < 1 || m < 1) return 0;
    const int T = (int)ceil_div(m, TB);
    Carver c(nullptr, 0);
    c.take<double>((size_t)S * (T * (T + 1) + T * T));
    return c.off;
}

extern "C" size_t am_kd_poly_workspace_bytes(int S, int m, int D) {
    if (S < 1 || m < 1 || D < 1) return 0;
    const int T = (int)ceil_div(m, TB);
    Carver c(nullptr, 0);
    c.take<double>((size_t)S * (T * (T + 1) + T * T));
    return std::max(c.off, kd_wide_eligible(S, m, D, 0) ? kd_wide_ws(S, m, D) : (size_t)0);
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
