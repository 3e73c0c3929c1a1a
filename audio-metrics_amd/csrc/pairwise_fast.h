// Filter-and-verify form of the PRDC kernels (included by pairwise.hip).
//
// The exact kernels of pairwise.hip spend all their time in f32 MFMAs (157 TF peak, 96 % of the sustained rate
// reached).  Almost none of the N x M distances they compute matter: a pair is relevant only if its squared
// distance lies below a row or column threshold (a k-NN bound, a hypersphere radius, the running row minimum).
// The kernels here find the relevant pairs with a 16x cheaper f16 MFMA pass and then evaluate exactly those
// pairs with the f32 arithmetic of the exact engine, so the results are BIT-IDENTICAL to pairwise.hip's:
//
//   1. X, Y are copied once to f16 (round-to-nearest-even) after an exact power-of-two scaling that puts each
//      matrix's largest |element| M into [2^13, 2^14) (half_scale_exp).  A scaled element v becomes v^ = v (1 + d) + e
//      with |d| <= 2^-11 and |e| <= eta = 2^-14 <= 2^-27 M (the smallest normal f16: covers subnormal operands being
//      flushed by the matrix core).  f16 x f16 products are exact in f32, so the f16 dot product, scaled back, differs
//      from <x, y> by at most
//          (2^-10 + 2^-22) S + (1 + 2^-11) 2^-27 (M_y |x|_1 + M_x |y|_1) + D 2^-54 M_x M_y  +  D 2^-22 S ,
//      S = sum_k |x_k y_k|; the last term is the accumulation inside the matrix core, priced at one rounding of
//      relative size 2^-23 (truncation) per added term with a factor 2 to spare, whatever the order.  With
//      S <= (|x|^2 + |y|^2) / 2, |x|_1 <= sqrt(D) |x|, 2ab <= a^2 + b^2 and M_x^2, M_y^2 <= G := the largest squared
//      row norm of EITHER matrix, the approximate squared distance a = fma(-2 / (scales), dot', |x|^2 + |y|^2) (same
//      f32 norms and the same rounding of their sum as the exact value t, whose own fmaf chain is off by at most
//      D 2^-24 S) satisfies
//          |a - t| <= fast_c(D) (|x|^2 + G),     fast_c(D) = 2^-10 + 2^-19 + 2^-25 sqrt(D) + D 2^-21 ,
//      for D <= 4096 (wider inputs take the exact kernels); 2^-19 covers the roundings of a, t, the thresholds
//      below and of the f32 norms the bound itself is computed from.  At D = 512 fast_c = 1.2229e-3.
//      The operand-stationary engine (pstat_engine.h) starts every accumulator at c0 = -|y|^2 / 2 (in the units of the scaled
//      dot product, an exact scaling) instead of adding |y|^2 afterwards: c0 is then one more term of the matrix core's
//      accumulation, whose partial sums are bounded by |c0| + S, so its part of the bound on a = -2 acc + |x|^2 becomes
//      (D + 1) 2^-21 (|y|^2 / 2 + S) <= (D + 1) 2^-21 (|x|^2 / 2 + |y|^2) <= (D + 1) 2^-21 (|x|^2 + G): the same form with D + 1
//      for D (fast_c below counts D + 1 terms for both engines); that a no longer adds |x|^2 + |y|^2 with the exact value's
//      rounding of the sum is a difference of 2^-24 (|x|^2 + |y|^2), inside the 2^-19 term.  am_filter_stats slot 9 MEASURES
//      |a - t| / (fast_c (|x|^2 + G)) on every verified pair (tests/test_gpu_routes.py).
//   2. A pair is QUEUED when a <= threshold + eps, which every pair with t <(=) threshold satisfies.
//   3. Queued pairs are evaluated with the engine's fmaf order (the chain oracle/exact_c reproduces), and the
//      reductions of the exact kernels are applied to those values.  Queue overflow falls back to the exact
//      computation (per row for the k-NN radii, for the whole call for the membership counts), as does a matrix
//      whose M is not a finite number within 2^+-60.
#pragma once

namespace am {

constexpr int EV_FAST = EV_DEFAULT | EV_F16 | EV_LDS;       // 128-row engine, f16 operands, LDS-direct fills (no ds_write: the store path bounds the staged form)
constexpr int FAST_MAX_DIM = 4096;                                            // fast_c's derivation holds up to here
constexpr int FAST_MIN_DIM = 1;                                               // (32 until round 4: see knn_fast_enabled)
// Mantissa bits dropped from the f16 copies (round to nearest even on the bit pattern, after the f32 -> f16 rounding):
// the matrix cores draw less power on operands with fewer significant bits and the chip, which runs these kernels at its
// power limit (1.8-1.9 GHz), clocks higher.  An element then carries |d| <= u = 2^-(11-n) + 2^-11 instead of 2^-11, and the
// first term of fast_c becomes 2u + u^2.
constexpr int FAST_DROP_BITS = 0;
static inline int half_drop_bits() {
    static const int n = std::min(std::max(env_int("AM_HALF_DROP_BITS", FAST_DROP_BITS), 0), 6);
    return n;
}
static inline float fast_c(int D) {
    const int n = half_drop_bits();
    const float rest = 1.9073486328125e-06f + 2.98023223876953125e-08f * sqrtf((float)D) + (float)(D + 1) * 4.76837158203125e-07f;
    if (n == 0) return 0.0009765625f + rest;
    const float u = ldexpf(1.f, -(11 - n)) + ldexpf(1.f, -11);
    return 2.f * u + u * u + rest;
}
constexpr int FAST_LDH_ALIGN = 64;                                            // f16 row stride: whole 128-B slabs

static inline int64_t half_ld(int D) { return (int64_t)(D + FAST_LDH_ALIGN - 1) / FAST_LDH_ALIGN * FAST_LDH_ALIGN; }

// largest |element| of a matrix through its bit pattern (sign cleared; NaN/inf sort above every finite value)
__global__ void __launch_bounds__(256) maxabs_bits_kernel(const float* __restrict__ X, int64_t N, int64_t ld, int D,
                                                          unsigned* __restrict__ out) {
    // one wave per row at a time (the access pattern of row_sqnorm_kernel: no index divisions, whole rows coalesced)
    const int lane = threadIdx.x & 63;
    unsigned m = 0u;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < N; row += (int64_t)gridDim.x * 4) {
        const float* x = X + row * ld;
        for (int k = lane * 4; k < D; k += 256) {
            const f32x4 v = load_k4(x, k, D);
            m = max(max(m, __float_as_uint(v.x) & 0x7fffffffu), __float_as_uint(v.y) & 0x7fffffffu);
            m = max(max(m, __float_as_uint(v.z) & 0x7fffffffu), __float_as_uint(v.w) & 0x7fffffffu);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off));
    __shared__ unsigned wave_max[4];                       // one atomic per workgroup: the target is a single address
    if (lane == 0) wave_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3]));
        if (m != 0u) atomicMax(out, m);
    }
}

// ---- scaled f32 -> f16 copy (RNE), zero-padded to ldh columns; one thread per 8 elements
__global__ void __launch_bounds__(256) to_half_kernel(const float* __restrict__ X, int64_t N, int64_t ld, int D, int64_t ldh,
                                                      const unsigned* __restrict__ maxabs_bits, uint16_t* __restrict__ Xh, int drop) {
    const int64_t per_row = ldh / 8;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = idx / per_row;
    if (row >= N) return;
    const float sc = __uint_as_float((unsigned)(127 + half_scale_exp(*maxabs_bits)) << 23);
    const int c = (int)(idx % per_row) * 8;
    const f32x4 a = load_k4(X + row * ld, c, D), b = load_k4(X + row * ld, c + 4, D);
    auto h = [&](float v) {
        unsigned b = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)(v * sc));
        if (drop > 0) {                               // RNE on the magnitude bits (a carry into the exponent is the right value)
            const unsigned mag = b & 0x7fffu;
            b = (b & 0x8000u) | ((mag + (1u << (drop - 1)) - 1u + ((mag >> drop) & 1u)) & ~((1u << drop) - 1u));
        }
        return b;
    };
    uint4 o;
    o.x = h(a.x) | (h(a.y) << 16);
    o.y = h(a.z) | (h(a.w) << 16);
    o.z = h(b.x) | (h(b.y) << 16);
    o.w = h(b.z) | (h(b.w) << 16);
    *reinterpret_cast<uint4*>(Xh + row * ldh + c) = o;
}

// max of non-negative floats (squared norms) through their bit patterns
__global__ void __launch_bounds__(256) max_bits_kernel(const float* __restrict__ v, int64_t n, unsigned* __restrict__ out) {
    unsigned m = 0u;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        m = max(m, __float_as_uint(v[i]));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off));
    if ((threadIdx.x & 63) == 0 && m != 0u) atomicMax(out, m);
}

// ---- optional filter statistics (am_filter_stats_enable): how much work the filter passes left for the exact kernels.
// int64 slots: 0 k-NN calls, 1 entries queued by the sweep, 2 of those through the spill queue, 3 pairs evaluated exactly,
// 4 rows recomputed by the exact fix-up kernel; 5 membership calls, 6 pairs queued, 7 of those through the overflow queue,
// 8 calls handed to the exact kernel (both queues overflowed or the operands could not be scaled);
// 9 the largest |f16 value - exact value| / (fast_c (|x|^2 + G)) over every pair the k-NN verification evaluated, as f32 bits
// (the MEASURED form of the bound of item 1 above: must stay <= 1), 10 the number of pairs it was measured on.
constexpr int FILTER_STATS_SLOTS = 16;
static long long* g_filter_stats = nullptr;                 // caller-owned device buffer, nullptr = off
static int g_filter_stats_device = -1;

static long long* filter_stats_for_current_device() {
    if (g_filter_stats == nullptr) return nullptr;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != g_filter_stats_device) return nullptr;
    return g_filter_stats;
}

__global__ void __launch_bounds__(256) filter_stats_kernel(const int* __restrict__ wgq_count, int64_t nwg, long long* __restrict__ stats,
                                                           int slot_calls, int slot_queued, const unsigned long long* spill64,
                                                           const int* spill32, int spill_cap, int slot_spill, const int* extra_a,
                                                           int slot_a, const int* extra_b, int slot_b) {
    long long q = 0;
    for (int64_t i = threadIdx.x; i < nwg; i += 256) q += wgq_count[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off);
    __shared__ long long red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long spill = 0;
        if (spill64 != nullptr) spill = (long long)min(*spill64, (unsigned long long)spill_cap);
        if (spill32 != nullptr) spill = min(*spill32, spill_cap);
        atomicAdd(reinterpret_cast<unsigned long long*>(stats + slot_calls), 1ull);
        atomicAdd(reinterpret_cast<unsigned long long*>(stats + slot_queued), (unsigned long long)(red[0] + red[1] + red[2] + red[3] + spill));
        atomicAdd(reinterpret_cast<unsigned long long*>(stats + slot_spill), (unsigned long long)spill);
        if (extra_a != nullptr) atomicAdd(reinterpret_cast<unsigned long long*>(stats + slot_a), (unsigned long long)*extra_a);
        if (extra_b != nullptr) atomicAdd(reinterpret_cast<unsigned long long*>(stats + slot_b), (unsigned long long)*extra_b);
    }
}

// stats[which] (0/1: largest squared norm, 2/3: largest |element|) of one matrix, then its scaled f16 copy
static int launch_to_half(const float* X, int64_t N, int64_t ld, int D, const float* norms, unsigned* stats, int which,
                          uint16_t* Xh, hipStream_t st) {
    const int64_t ldh = half_ld(D);
    hipLaunchKernelGGL(max_bits_kernel, dim3(256), dim3(256), 0, st, norms, N, stats + which);
    hipLaunchKernelGGL(maxabs_bits_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(N, 4), 1024)), dim3(256), 0, st, X, N, ld, D,
                       stats + 2 + which);
    const int64_t threads = N * (ldh / 8);
    hipLaunchKernelGGL(to_half_kernel, dim3((unsigned)ceil_div(threads, 256)), dim3(256), 0, st, X, N, ld, D, ldh,
                       stats + 2 + which, Xh, half_drop_bits());
    AM_LAUNCH_CHECK();
    return AM_OK;
}

// ---- prepared sets (am_prepare_set_f32): what the entry points above derive from a set before their tile kernels run -
// squared row norms, {largest squared norm, -, largest |element|, -} as bit patterns, and the scaled f16 copy - computed
// once by the caller and handed to the *_prepared_* entry points (a row shard of a prepared set is the same three
// pointers offset by the shard's first row; the statistics of the whole set remain valid bounds for the shard).
struct PreparedSet {
    const float* norms;
    const unsigned* stats;        // [4] as written by launch_to_half(..., which = 0, ...)
    const uint16_t* half;
};

// maxn[slot_norm] = stats[0], maxn[slot_abs] = stats[2]
__global__ void prepared_stats_kernel(const unsigned* __restrict__ stats, unsigned* __restrict__ maxn, int slot_norm, int slot_abs,
                                      int slot_abs2) {
    if (threadIdx.x == 0) {
        maxn[slot_norm] = stats[0];
        maxn[slot_abs] = stats[2];
        if (slot_abs2 >= 0) maxn[slot_abs2] = stats[2];
    }
}

// the norms of a set: from the prepared set (a device-to-device copy of N floats) or computed
static int norms_of(const PreparedSet* prep, const float* X, int64_t N, int64_t ld, int D, float* out, hipStream_t st) {
    if (prep != nullptr) {
        AM_HIP_TRY(hipMemcpyAsync(out, prep->norms, (size_t)N * sizeof(float), hipMemcpyDeviceToDevice, st));
        return AM_OK;
    }
    return launch_norms(X, N, ld, D, out, st);
}

// The exact engine's value for one pair: f32 fmaf chain over the inner index in the order 8c+0, 8c+4, 8c+1, ...
// (tile_engine.h, "K order"); xs = the row held in LDS, zero-padded to a multiple of 8.
__device__ __forceinline__ float exact_pair_dot(const float* __restrict__ xs, const float* __restrict__ y, int D) {
    float acc = 0.f;
    const int dp = (D + 7) / 8 * 8;
    for (int c = 0; c < dp; c += 8) {
        const f32x4 y0 = load_k4(y, c, D), y1 = load_k4(y, c + 4, D);
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(xs + c), x1 = *reinterpret_cast<const f32x4*>(xs + c + 4);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc = fmaf(y0[s], x0[s], acc);
            acc = fmaf(y1[s], x1[s], acc);
        }
    }
    return acc;
}

// ------------------------------------------------------------------------------------------------
// Membership counts, filter pass.  P rows = reference rows i (lane-local), Q rows = candidate rows j.
// With G = the largest squared row norm of either set, E_i = fast_c (|r_i|^2 + G) and E'_j = fast_c (G + |c_j|^2)  (>= eps of every pair),
// T_i = T(r_ref[i]), T'_j = T(r_cand[j]) the strict "<" thresholds of the exact kernel:
//   column counts   a <  T_i - E_i   the pair is inside for certain: counted at once (as the exact kernel does)
//                   a <= T_i + E_i   otherwise ambiguous: QUEUED
//   row "any"       a <  T'_j - E'_j  certain witness: the row is flagged, and flagged rows skip this test
//                   a <= T'_j + E'_j  otherwise ambiguous: QUEUED
//   row minimum     a <= m_i + 2 E_i  QUEUED, m_i = an upper bound of min_j max(a_ij, 0) over the columns seen so
//                   far (global array, atomicMin): the true minimiser j* has
//                   a_ij* <= t_ij* + eps <= t_ij + eps <= max(a_ij, 0) + 2 eps for every j.
// A queue entry is (i, j | COUNTED) - COUNTED = the pair was already counted as certain.  PRE = sampled
// pre-pass (every qstride-th column tile): only m_i and the certain "any" flags are produced.
constexpr int FAST_AUX_FLOATS = 6 * TB;                                    // LDS [2][3][128]
constexpr size_t FAST_LDS_BYTES = (ENGINE_LDS_FLOATS + FAST_AUX_FLOATS) * sizeof(float) + 16;

struct CrossFastEpilogue {
    const float* qnorm;
    const float* qthr;
    int64_t nq;
    float fc;                   // fast_c(D)
    float rnmax_c;              // fast_c * G
    float* aux;                 // LDS [2][3][128] : |c_j|^2, T'_j + E'_j, T'_j - E'_j of the tile
    int32_t* col_count;
    uint2* wgq;                 // this workgroup's append region
    int* qn;                    // LDS slot counter
    int qcap;
    uint2* ovq;                 // global overflow queue
    int* ov_count;
    int ovcap;
    int* fail;
    int dbg;
    float dsc;                  // -2 / (operand scales)
    int64_t prow[2];
    float xn[2], thi[2], tlo[2], e2[2], m[2];
    bool rowok[2], anyf[2], covf[2];
    float aux_n, aux_hi, aux_lo;
    const LaneInfo& L;

    __device__ __forceinline__ CrossFastEpilogue(const LaneInfo& l) : L(l) {}
    __device__ __forceinline__ void push(int64_t i, unsigned jflag) {
        const int slot = atomicAdd(qn, 1);
        if (slot < qcap) {
            wgq[slot] = make_uint2((unsigned)i, jflag);
        } else if (*reinterpret_cast<volatile int*>(fail) == 0) {   // region full: spill to the global queue
            // (once the fail flag is up nothing is counted any more: on inputs the bound cannot decide - e.g. sets three
            // orders of magnitude apart - billions of pairs arrive here and would wrap the counter)
            const unsigned s2 = atomicAdd(reinterpret_cast<unsigned*>(ov_count), 1u);
            if (s2 < (unsigned)ovcap) ovq[s2] = make_uint2((unsigned)i, jflag);
            else *fail = 1;                                 // -> the exact kernel redoes the whole call
        }
    }
    __device__ __forceinline__ void aux_issue(int, int64_t qtile) {
        if (L.tid < TB) {
            const int64_t j = qtile * TB + L.tid;
            if (j < nq) {
                const float e = fmaf(fc, qnorm[j], rnmax_c);
                aux_n = qnorm[j];
                aux_hi = qthr[j] + e;
                aux_lo = qthr[j] - e;
            } else {
                aux_n = INFINITY;                           // a = +inf: never below anything
                aux_hi = -INFINITY;
                aux_lo = -INFINITY;
            }
        }
    }
    __device__ __forceinline__ void aux_commit(int t) {
        if (L.tid < TB) {
            float* d = aux + (t & 1) * 3 * TB + L.tid;
            d[0] = aux_n;
            d[TB] = aux_hi;
            d[2 * TB] = aux_lo;
        }
    }
    template <bool PRE, bool WANT_MIN>
    __device__ __forceinline__ void finish_impl(int t, int64_t qtile, f32x16 (&acc)[2][2]) {
        const float* a = aux + (t & 1) * 3 * TB + L.wm * 64 + L.h * 4;
        const int64_t jbase = qtile * TB + L.wm * 64 + L.h * 4;
        if (dbg & 8) return;                                   // timing experiment: MFMA pipeline only
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f32x4 yn[4], th[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                yn[g4] = *reinterpret_cast<const f32x4*>(a + mt * 32 + g4 * 8);
                th[g4] = *reinterpret_cast<const f32x4*>(a + (PRE ? 2 : 1) * TB + mt * 32 + g4 * 8);   // PRE: T'-E', main: T'+E'
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                float tmin = INFINITY, marg = INFINITY;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const float u = fmaf(dsc, acc[mt][nt][reg], xn[nt] + yn[reg >> 2][reg & 3]);
                    tmin = fminf(tmin, u);
                    marg = fminf(marg, u - th[reg >> 2][reg & 3]);          // +inf - (-inf) = +inf past nq
                }
                if constexpr (WANT_MIN) m[nt] = fminf(m[nt], fmaxf(tmin, 0.f));
                if constexpr (PRE) {
                    anyf[nt] = anyf[nt] || (marg < 0.f);                     // some a < T'_j - E'_j
                } else {
                    const float prow_thr = WANT_MIN ? fmaxf(thi[nt], m[nt] + e2[nt]) : thi[nt];
                    if (!(dbg & 4) && __any(rowok[nt] && (tmin <= prow_thr || (!anyf[nt] && marg <= 0.f)))) {
                        const float* alo = a + 2 * TB + mt * 32;
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) {
                            const float u = fmaf(dsc, acc[mt][nt][reg], xn[nt] + yn[reg >> 2][reg & 3]);
                            const int64_t j = jbase + mt * 32 + (reg >> 2) * 8 + (reg & 3);
                            const bool sure = rowok[nt] && u < tlo[nt];
                            const unsigned long long mask = __ballot(sure);
                            if (mask != 0ull && L.lane == 0) {               // lanes 0-31: column j, lanes 32-63: column j + 4
                                const int lo = __popcll(mask & 0xffffffffull);
                                const int hi = __popcll(mask >> 32);
                                if (lo) atomicAdd(col_count + j - L.h * 4, lo);
                                if (hi) atomicAdd(col_count + j - L.h * 4 + 4, hi);
                            }
                            covf[nt] = covf[nt] || sure;                              // inside for certain: the row is covered
                            bool want = rowok[nt] && !sure && u <= thi[nt];           // ambiguous count
                            if constexpr (WANT_MIN) want = want || (rowok[nt] && u <= m[nt] + e2[nt]);   // row-minimum candidate
                            if (rowok[nt] && !anyf[nt] && u <= th[reg >> 2][reg & 3]) {
                                if (u < alo[(reg >> 2) * 8 + (reg & 3)]) { anyf[nt] = true; if (dbg & 1) want = true; }   // certain witness
                                else want = true;                                     // ambiguous "any"
                            }
                            if (want) push(prow[nt], (unsigned)j | (sure ? FAST_COUNTED : 0u));
                        }
                    }
                }
            }
        }
    }
};

template <bool PRE, bool WANT_MIN>
struct CrossFastShim {            // picks the epilogue body at compile time
    CrossFastEpilogue& e;
    __device__ __forceinline__ void aux_issue(int t, int64_t q) { e.aux_issue(t, q); }
    __device__ __forceinline__ void aux_commit(int t) { e.aux_commit(t); }
    __device__ __forceinline__ void finish(int t, int64_t q, f32x16 (&acc)[2][2]) {
        e.template finish_impl<PRE, WANT_MIN>(t, q, acc);
    }
};

// Rb / Cb: f16 copies viewed as f32 words (ld and Dh in words, Dh % 32 == 0).
template <bool PRE, bool WANT_MIN>
__global__ void __launch_bounds__(ENGINE_THREADS, 2)
cross_fast_kernel(const float* __restrict__ Rb, int64_t Nr, int64_t ldr, const float* __restrict__ rnorm,
                  const float* __restrict__ rthr, const float* __restrict__ Cb, int64_t Nc, int64_t ldc,
                  const float* __restrict__ cnorm, const float* __restrict__ cthr, int Dh, int nchunks, int qstride,
                  const unsigned* __restrict__ maxn, unsigned* __restrict__ rmin_approx, unsigned* __restrict__ row_any,
                  unsigned* __restrict__ row_cover, int32_t* __restrict__ col_count, uint2* __restrict__ wgq, int qcap, int* __restrict__ wgq_count,
                  uint2* __restrict__ ovq, int* __restrict__ ov_count, int ovcap, int* __restrict__ fail, int dbg, float fc,
                  int grp_rows) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const LaneInfo L;
    const int64_t q_tiles = ((Nc + TB - 1) / TB + qstride - 1) / qstride;
    const WorkItem w = grp_rows > 0 ? work_item_grouped(q_tiles, nchunks, (Nr + TB - 1) / TB, grp_rows) : work_item(q_tiles, nchunks);
    if (w.ntiles == 0) {                                   // padding item of the grouped grid
        if (!PRE && L.tid == 0) wgq_count[blockIdx.x] = 0;
        return;
    }
    int* qn = reinterpret_cast<int*>(lds + ENGINE_LDS_FLOATS + FAST_AUX_FLOATS);
    if (L.tid == 0) *qn = 0;

    const float gmax = fmaxf(__uint_as_float(maxn[0]), __uint_as_float(maxn[1]));
    CrossFastEpilogue epi(L);
    epi.fc = fc;
    epi.qnorm = cnorm;
    epi.qthr = cthr;
    epi.nq = Nc;
    epi.rnmax_c = fc * gmax;
    epi.aux = lds + ENGINE_LDS_FLOATS;
    epi.col_count = col_count;
    epi.wgq = wgq + (int64_t)blockIdx.x * qcap;
    epi.qn = qn;
    epi.qcap = qcap;
    epi.ovq = ovq;
    epi.ov_count = ov_count;
    epi.ovcap = ovcap;
    epi.fail = fail;
    epi.dbg = dbg;
    epi.dsc = half_unscale(maxn[2], maxn[3]);
    if (blockIdx.x == 0 && L.tid == 0 && !(half_scale_ok(maxn[2]) && half_scale_ok(maxn[3]))) *fail = 1;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int64_t i = w.prow0 + L.wn * 64 + nt * 32 + L.r;
        const bool ok = i < Nr;
        epi.prow[nt] = i;
        epi.rowok[nt] = ok;
        epi.xn[nt] = ok ? rnorm[i] : 0.f;
        const float e = fc * ((ok ? rnorm[i] : 0.f) + gmax);
        epi.thi[nt] = ok ? rthr[i] + e : -INFINITY;
        epi.tlo[nt] = ok ? rthr[i] - e : -INFINITY;
        epi.e2[nt] = 2.f * e;
        epi.m[nt] = (WANT_MIN && ok) ? __uint_as_float(rmin_approx[i]) : INFINITY;
        epi.anyf[nt] = ok ? (row_any[i] != 0u && !(dbg & 2)) : true;
        epi.covf[nt] = false;
    }
    CrossFastShim<PRE, WANT_MIN> shim{epi};
    dense_pipeline_early<EV_FAST, false>(Cb, Nc, ldc, LinearTiles{w.qtile0, qstride}, Rb, Nr, ldr, w.prow0, w.ntiles, Dh,
                                         lds, L, shim);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const float mn = fminf(epi.m[nt], __shfl_xor(epi.m[nt], 32));
        const int other = __shfl_xor((int)epi.anyf[nt], 32);                  // unconditionally: every lane must take part
        const int other_c = __shfl_xor((int)epi.covf[nt], 32);
        const bool any = epi.anyf[nt] || other != 0;
        const bool cov = epi.covf[nt] || other_c != 0;
        if (L.h == 0 && epi.rowok[nt]) {
            if constexpr (WANT_MIN) atomicMin(rmin_approx + epi.prow[nt], __float_as_uint(mn));   // mn >= 0: uint order == float order
            if (any) atomicOr(row_any + epi.prow[nt], 1u);
            if (cov) atomicOr(row_cover + epi.prow[nt], 1u);
        }
    }
    if constexpr (!PRE) {
        __syncthreads();
        if (L.tid == 0) wgq_count[blockIdx.x] = *qn < qcap ? *qn : qcap;
    }
}

// Exact value of one queued pair applied to the exact kernel's accumulators.
__device__ __forceinline__ void cross_apply(float t, int64_t j, unsigned jflag, float ti, const float* __restrict__ cthr,
                                            int32_t* __restrict__ col_count, float& mn, bool& any, bool& cov) {
    mn = fminf(mn, t);
    any = any || (t < cthr[j]);
    if (t < ti) {
        cov = true;
        if (!(jflag & FAST_COUNTED)) atomicAdd(col_count + j, 1);
    }
}

// Exact dot products of 64 pairs, one per lane, with COALESCED row fetches (shared by the two verification kernels).
// A wave takes 64 pairs at a time.  Its 128 rows are fetched in 128-byte pieces (8 lanes per row piece, 8 rows per load
// instruction) into the wave's own LDS tile, then lane p reads the two pieces of pair p back and runs the 32 products of the
// chain in the exact engine's order 8c+s, 8c+4+s (row stride 36 floats: the reads are bank-conflict free).  One 16-byte load
// per lane and row - the first version of both kernels - makes every load instruction touch 64 different cache lines: the
// texture-address unit, not the memory, then bounds the kernel (4.4x slower per pair, measured on the membership queue of
// the CLAP-shaped sets: 1.2 ms for 0.96 M pairs against 0.31 ms for 0.74 M in the k-NN verification).
constexpr int VERIFY_LD = 36;                                       // floats per 32-float piece in LDS
constexpr size_t VERIFY_LDS_BYTES = (size_t)4 * 128 * VERIFY_LD * sizeof(float);   // 4 waves x 128 rows

// ia / ib: this lane's row of A / of B (any valid row for lanes without a pair); tile: the wave's 128 x VERIFY_LD floats
__device__ __forceinline__ float wave_pair_dot(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
                                               unsigned ia, unsigned ib, int D, float* tile, int lane) {
    const int dp = (D + 7) / 8 * 8;
    const int grp = lane >> 3, slot = lane & 7;
    // the row this lane fetches in load instruction rr: tile row rr * 8 + grp (rows 0..63: A rows of the pairs, 64..127: B rows)
    const float* src[16];
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
        const int trow = rr * 8 + grp;
        const unsigned idx = (unsigned)__shfl((int)(trow < 64 ? ia : ib), trow & 63);
        src[rr] = trow < 64 ? A + (int64_t)idx * lda : B + (int64_t)idx * ldb;
    }
    float acc = 0.f;
    f32x4 piece[16];
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) piece[rr] = load_k4(src[rr], slot * 4, D);
    for (int c = 0; c < dp; c += 32) {
        __builtin_amdgcn_wave_barrier();                         // the previous piece has been consumed (same wave: LDS ops stay in order)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr)
            *reinterpret_cast<f32x4*>(tile + (rr * 8 + grp) * VERIFY_LD + slot * 4) = piece[rr];
        if (c + 32 < dp) {                                       // the next piece is on its way while this one is multiplied
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) piece[rr] = load_k4(src[rr], c + 32 + slot * 4, D);
        }
        __builtin_amdgcn_wave_barrier();
        const float* ua = tile + lane * VERIFY_LD;
        const float* va = tile + (64 + lane) * VERIFY_LD;
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
            const f32x4 u0 = *reinterpret_cast<const f32x4*>(ua + 4 * q), u1 = *reinterpret_cast<const f32x4*>(ua + 4 * q + 4);
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(va + 4 * q), v1 = *reinterpret_cast<const f32x4*>(va + 4 * q + 4);
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                acc = fmaf(v0[s2], u0[s2], acc);
                acc = fmaf(v1[s2], u1[s2], acc);
            }
        }
    }
    return acc;
}

// exact evaluation of the regions of cross_wide_kernel: a wave takes 64 queued pairs at a time (wave_pair_dot).  The work
// items (region, first entry) were listed by the filter kernel itself, so the grid is the same for ten entries per region
// (independent sets: one workgroup per region would leave three of four waves and most of the LDS idle) and for thousands.
constexpr unsigned CROSS_VERIFY_GRID = 1024;
__global__ void __launch_bounds__(256) cross_verify_regions_kernel(const float* __restrict__ R, int64_t ldr,
                                                                   const float* __restrict__ rnorm, const float* __restrict__ rthr,
                                                                   const float* __restrict__ C, int64_t ldc,
                                                                   const float* __restrict__ cnorm, const float* __restrict__ cthr,
                                                                   int D, const uint2* __restrict__ wgq, int qcap,
                                                                   const int* __restrict__ wgq_count, const uint2* __restrict__ items,
                                                                   const int* __restrict__ item_count, int32_t* __restrict__ col_count,
                                                                   unsigned* __restrict__ row_min_bits,
                                                                   unsigned* __restrict__ row_any, unsigned* __restrict__ row_cover,
                                                                   const int* __restrict__ fail) {
    extern __shared__ __attribute__((aligned(16))) float vlds[];
    if (*fail) return;                                    // (the exact kernel takes the whole call: cross_fast_decide_kernel)
    const int nitems = *item_count;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* tile = vlds + wave * 128 * VERIFY_LD;
    for (int it = blockIdx.x * 4 + wave; it < nitems; it += gridDim.x * 4) {      // (wave-uniform: every lane of a wave takes part)
        const uint2 item = items[it];
        const int e = (int)item.y + lane;
        const bool valid = e < wgq_count[item.x];
        const uint2 v = valid ? wgq[(int64_t)item.x * qcap + e] : make_uint2(0u, 0u);
        const unsigned i = v.x, j = v.y & ~FAST_COUNTED;
        const float acc = wave_pair_dot(R, ldr, C, ldc, i, j, D, tile, lane);
        if (valid) {
            const float t = clamp0(fmaf(-2.f, acc, rnorm[i] + cnorm[j]));
            float mn = INFINITY;
            bool any = false, cov = false;
            cross_apply(t, j, v.y, rthr[i], cthr, col_count, mn, any, cov);
            if (row_min_bits != nullptr) atomicMin(row_min_bits + i, __float_as_uint(mn));
            if (any) atomicOr(row_any + i, 1u);
            if (cov) atomicOr(row_cover + i, 1u);
        }
    }
}

// Verification of one workgroup region (same grid as the filter pass): the entries are bucketed by reference
// row in LDS, then each wave takes rows - the row goes to LDS once, every lane evaluates one candidate with the
// exact engine's fmaf chain - and the exact reductions are applied.
__global__ void __launch_bounds__(256) cross_verify_kernel(const float* __restrict__ R, int64_t Nr, int64_t ldr,
                                                           const float* __restrict__ rnorm, const float* __restrict__ rthr,
                                                           const float* __restrict__ C, int64_t ldc,
                                                           const float* __restrict__ cnorm, const float* __restrict__ cthr, int D,
                                                           int nchunks, int grp_rows, const uint2* __restrict__ wgq, int qcap,
                                                           const int* __restrict__ wgq_count, int32_t* __restrict__ col_count,
                                                           unsigned* __restrict__ row_min_bits, unsigned* __restrict__ row_any,
                                                           unsigned* __restrict__ row_cover) {
    extern __shared__ __attribute__((aligned(16))) float vlds[];       // [4][dp] rows, then qcap sorted entries
    __shared__ int bucket[TB], start[TB];
    const int n = wgq_count[blockIdx.x];
    if (n == 0) return;
    const int dp = (D + 7) / 8 * 8;
    unsigned* sorted = reinterpret_cast<unsigned*>(vlds + 4 * dp);
    const uint2* q = wgq + (int64_t)blockIdx.x * qcap;
    // row block of this region: the filter kernel's work mapping (q_tiles is irrelevant for prow0)
    const int64_t prow0 = grp_rows > 0 ? work_item_grouped(1, nchunks, (Nr + TB - 1) / TB, grp_rows).prow0
                                       : (int64_t)(blockIdx.x / nchunks) * TB;
    if (threadIdx.x < TB) bucket[threadIdx.x] = 0;
    __syncthreads();
    constexpr int PER = 8;                                             // qcap <= 256 * PER
    int pos[PER];
#pragma unroll
    for (int s = 0; s < PER; ++s) {
        const int e = threadIdx.x + s * 256;
        pos[s] = e < n ? atomicAdd(&bucket[(int)(q[e].x - prow0)], 1) : 0;
    }
    __syncthreads();
    if (threadIdx.x < 64) {                                            // exclusive scan of the 128 bucket sizes
        const int a0 = bucket[2 * threadIdx.x], a1 = bucket[2 * threadIdx.x + 1];
        int v = a0 + a1;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(v, off);
            if ((int)threadIdx.x >= off) v += o;
        }
        start[2 * threadIdx.x] = v - a0 - a1;
        start[2 * threadIdx.x + 1] = v - a1;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < PER; ++s) {
        const int e = threadIdx.x + s * 256;
        if (e < n) sorted[start[(int)(q[e].x - prow0)] + pos[s]] = q[e].y;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* xs = vlds + wave * dp;
    for (int lr = wave; lr < TB; lr += 4) {
        const int cnt = bucket[lr];
        if (cnt == 0) continue;
        const int64_t i = prow0 + lr;
        for (int k = lane; k < dp; k += 64) xs[k] = k < D ? R[i * ldr + k] : 0.f;
        __builtin_amdgcn_wave_barrier();                               // same wave wrote the row: LDS ops stay in order
        const float xi = rnorm[i], ti = rthr[i];
        float mn = INFINITY;
        bool any = false, cov = false;
        for (int e0 = 0; e0 < cnt; e0 += 64) {
            const int e = e0 + lane;
            if (e < cnt) {
                const unsigned jf = sorted[start[lr] + e];
                const int64_t j = jf & ~FAST_COUNTED;
                const float t = clamp0(fmaf(-2.f, exact_pair_dot(xs, C + j * ldc, D), xi + cnorm[j]));
                cross_apply(t, j, jf, ti, cthr, col_count, mn, any, cov);
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mn = fminf(mn, __shfl_xor(mn, off));
        const bool wave_any = __any(any), wave_cov = __any(cov);
        if (lane == 0) {
            if (row_min_bits != nullptr) atomicMin(row_min_bits + i, __float_as_uint(mn));   // t >= 0: uint order == float order
            if (wave_any) atomicOr(row_any + i, 1u);
            if (wave_cov) atomicOr(row_cover + i, 1u);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Entries that did not fit their region: one thread per pair, both rows from global memory.
__global__ void __launch_bounds__(256) cross_verify_overflow_kernel(const float* __restrict__ R, int64_t ldr,
                                                                    const float* __restrict__ rnorm, const float* __restrict__ rthr,
                                                                    const float* __restrict__ C, int64_t ldc,
                                                                    const float* __restrict__ cnorm, const float* __restrict__ cthr,
                                                                    int D, const uint2* __restrict__ ovq,
                                                                    const int* __restrict__ ov_count, int ovcap,
                                                                    const int* __restrict__ fail, int32_t* __restrict__ col_count,
                                                                    unsigned* __restrict__ row_min_bits,
                                                                    unsigned* __restrict__ row_any, unsigned* __restrict__ row_cover) {
    if (*fail) return;
    const int n = *ov_count < ovcap ? *ov_count : ovcap;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
        const uint2 v = ovq[e];
        const int64_t i = v.x, j = v.y & ~FAST_COUNTED;
        const float* x = R + i * ldr;
        const float* y = C + j * ldc;
        float acc = 0.f;
        const int dp = (D + 7) / 8 * 8;
        for (int c = 0; c < dp; c += 8) {
            const f32x4 y0 = load_k4(y, c, D), y1 = load_k4(y, c + 4, D);
            const f32x4 x0 = load_k4(x, c, D), x1 = load_k4(x, c + 4, D);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc = fmaf(y0[s], x0[s], acc);
                acc = fmaf(y1[s], x1[s], acc);
            }
        }
        const float t = clamp0(fmaf(-2.f, acc, rnorm[i] + cnorm[j]));
        float mn = INFINITY;
        bool any = false, cov = false;
        cross_apply(t, j, v.y, rthr[i], cthr, col_count, mn, any, cov);
        if (row_min_bits != nullptr) atomicMin(row_min_bits + i, __float_as_uint(mn));
        if (any) atomicOr(row_any + i, 1u);
        if (cov) atomicOr(row_cover + i, 1u);
    }
}

// ---- float64 rows (round 5): the membership filter's queued pairs evaluated in f64 ------------------------------------------
// The filter pass runs on float32-rounded copies of the two sets with float32-rounded thresholds; what it decides WITHOUT
// looking again - "certainly inside" (a < T - E), "certainly a witness", "certainly outside" (not queued) - holds for the
// float64 rows because its band is widened by the two roundings (run_cross_fast: fast_c + 2^-18: a distance moves by at most
// 2^-21 (|x|^2 + |y|^2) when the rows are rounded, a threshold by 2^-24 of itself); every pair inside the band is queued and
// evaluated here as a sum of squared differences in f64 against the f64 thresholds T(R) (pairwise_f64.hip).
struct Prdc64Hook {
    const double *R, *C;
    int64_t ldr, ldc;
    const double *rt, *ct;          // thresholds T(r_ref[i]), T(r_cand[j]) in f64
};

__device__ __forceinline__ void cross_apply64(const double* __restrict__ x, const double* __restrict__ y, int D, int64_t i, int64_t j,
                                              unsigned jflag, const double* __restrict__ rt, const double* __restrict__ ct,
                                              int32_t* __restrict__ col_count, unsigned* __restrict__ row_any,
                                              unsigned* __restrict__ row_cover) {
    double t = 0.0;
    for (int d = 0; d < D; ++d) {
        const double u = x[d] - y[d];
        t = fma(u, u, t);
    }
    if (t < ct[j]) atomicOr(row_any + i, 1u);               // (a NaN distance is inside nothing)
    if (t < rt[i]) {
        atomicOr(row_cover + i, 1u);
        if (!(jflag & FAST_COUNTED)) atomicAdd(col_count + j, 1);
    }
}

__global__ void __launch_bounds__(256) cross_verify_regions64_kernel(Prdc64Hook h, int D, const uint2* __restrict__ wgq, int qcap,
                                                                     const int* __restrict__ wgq_count, const uint2* __restrict__ items,
                                                                     const int* __restrict__ item_count, int32_t* __restrict__ col_count,
                                                                     unsigned* __restrict__ row_any, unsigned* __restrict__ row_cover,
                                                                     const int* __restrict__ fail) {
    if (*fail) return;
    const int nitems = *item_count;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int it = blockIdx.x * 4 + wave; it < nitems; it += gridDim.x * 4) {
        const uint2 item = items[it];
        const int e = (int)item.y + lane;
        if (e >= wgq_count[item.x]) continue;
        const uint2 v = wgq[(int64_t)item.x * qcap + e];
        const int64_t i = v.x, j = v.y & ~FAST_COUNTED;
        cross_apply64(h.R + i * h.ldr, h.C + j * h.ldc, D, i, j, v.y, h.rt, h.ct, col_count, row_any, row_cover);
    }
}

__global__ void __launch_bounds__(256) cross_verify_overflow64_kernel(Prdc64Hook h, int D, const uint2* __restrict__ ovq,
                                                                      const int* __restrict__ ov_count, int ovcap,
                                                                      const int* __restrict__ fail, int32_t* __restrict__ col_count,
                                                                      unsigned* __restrict__ row_any, unsigned* __restrict__ row_cover) {
    if (*fail) return;
    const int n = *ov_count < ovcap ? *ov_count : ovcap;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
        const uint2 v = ovq[e];
        const int64_t i = v.x, j = v.y & ~FAST_COUNTED;
        cross_apply64(h.R + i * h.ldr, h.C + j * h.ldc, D, i, j, v.y, h.rt, h.ct, col_count, row_any, row_cover);
    }
}

// Data-dependent fallback of the membership filter (round 4), the counterpart of check B of the k-NN filter: when the filter
// pass has queued more pairs than an exact verification is worth - both sets drawn around the SAME tight clusters: every
// candidate of a reference row's cluster lies inside the error band of that row's radius, 400 undecidable pairs per row at
// 20 000 rows, most of them through the one-pair-per-thread overflow path: 96 ms against 7.6 ms for the exact kernel at
// 20 000 x 512, 90 against 2 ms at 20 000 x 64 (tools/threshold_sweep.py, data family `shared`) - the fail flag is raised, the
// verification kernels return at once and the exact kernel, always launched behind, really runs.  An exactly verified pair
// costs about as much as 80 pairs of the exact kernel and the filter pass itself an eighth of it: the limit is 1 / 128 of all
// pairs, or more entries in the overflow queue than its slow path should see.
__global__ void cross_fast_decide_kernel(const int* __restrict__ ov_count, int ovcap, int* __restrict__ fail, long long limit_total,
                                         int limit_overflow) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int spilled = ov_count[0] < ovcap ? ov_count[0] : ovcap;
    const long long total = (long long)ov_count[2] * 64 + spilled;           // (regions are listed in batches of 64 entries)
    if (total > limit_total || ov_count[0] > limit_overflow) *fail = 1;
}

// Both queues overflowed (pathological inputs: e.g. one huge cluster of duplicates): wipe the accumulators so the
// exact kernel, which then really runs (its workgroups return at once otherwise), starts from a clean state.
__global__ void __launch_bounds__(256) cross_fail_reset_kernel(const int* __restrict__ fail, int32_t* __restrict__ col_count,
                                                               int64_t Nc, unsigned* __restrict__ row_min_bits,
                                                               unsigned* __restrict__ row_any, unsigned* __restrict__ row_cover,
                                                               int64_t Nr) {
    if (!*fail) return;
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < Nc; j += (int64_t)gridDim.x * 256) col_count[j] = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < Nr; i += (int64_t)gridDim.x * 256) {
        row_min_bits[i] = 0x7f800000u;          // the exact kernel always accumulates the minimum
        row_any[i] = 0u;
        row_cover[i] = 0u;
    }
}

struct CrossFastPlan {
    int nchunks, pre_chunks, qstride, qcap, ovcap, grp_rows;
    int64_t blocks;
    bool wide;                  // main pass on the 256 x 256 engine (cross_wide_kernel)
};

static CrossFastPlan plan_cross_fast(int64_t Nr, int64_t Nc, int D) {
    CrossFastPlan p;
    p.nchunks = choose_chunks(Nr, Nc);
    static const int grp = env_int("AM_FAST_GROUP_ROWS", 8);                     // 0: plain (row block, chunk) order
    p.grp_rows = (grp == 1 || grp == 2 || grp == 4 || grp == 8 || grp == 16 || grp == 32 || grp == 64) ? grp : 0;
    p.blocks = p.grp_rows > 0 ? cross_grouped_blocks(ceil_div(Nr, TB), p.nchunks, p.grp_rows) : ceil_div(Nr, TB) * p.nchunks;
    static const int wide = env_int("AM_FAST_WIDE", 1);
    p.wide = wide != 0;
    if (p.wide) {
        // Row blocks of a group (the 32 workgroups of an XCD = grp_rows row blocks x 32 / grp_rows column chunks).  wide_engine.h
        // streams the P blocks too: eight of them (2 MB of f16) stay in an XCD's L2 beside the Q window.  The stationary
        // engines hold P in registers - the L2 only serves Q - so ALL 32 workgroups walk ONE column chunk: every Q tile is
        // fetched once per 32 row blocks instead of once per 8 (100 000 x 512: 9.22 / 8.87 / 8.79 / 8.54 ms per launch at
        // 4 / 8 / 16 / 32, CLAP-shaped k = 10 10.56 -> 10.14; D = 128 flat; profiles/r6/wide_bench_cross_group.txt).
        static const int wgrp_env = env_int("AM_WIDE_GROUP_ROWS", 0);
        const int wgrp = wgrp_env > 0 ? wgrp_env : (wide_stationary((int)(half_ld(D) / 2)) ? 32 : 8);
        static const int wtarget = env_int("AM_WIDE_WG_TARGET", 6144);               // ~24 rounds of 256 CUs x 1 workgroup
        p.grp_rows = (wgrp == 1 || wgrp == 2 || wgrp == 4 || wgrp == 8 || wgrp == 16 || wgrp == 32) ? wgrp : 8;
        const int grp_chunks = 32 / p.grp_rows;
        const int64_t rbw = ceil_div(Nr, WIDE_TILE_ROWS), qtw = ceil_div(Nc, WIDE_TILE_ROWS);
        int64_t want = std::max<int64_t>(ceil_div(wtarget, rbw), 1);
        want = ceil_div(want, grp_chunks) * grp_chunks;
        // at least ~8 column tiles per workgroup (row shards of a multi-GPU run would otherwise get 3-tile work items
        // whose pipeline fill and epilogue set-up dominate)
        want = std::min<int64_t>(want, std::max<int64_t>(qtw / 8 / grp_chunks * grp_chunks, grp_chunks));
        want = std::min<int64_t>(want, std::max<int64_t>(qtw / grp_chunks * grp_chunks, 1));
        want = std::min<int64_t>(want, qtw);
        p.nchunks = (int)want;
        p.blocks = wide_grouped_blocks(rbw, p.nchunks, p.grp_rows);
    }
    // Sampled "any" pre-pass: every stride-th candidate tile.  16 until round 5; since the main pass's hit path got cheap (scalar
    // gates, accumulator start values) a witness found there costs little, and what the pre-pass is still good for - taking
    // the rows with MANY witnesses out of the "any" direction early - needs about six sampled tiles: at 100 000 x 512 stride
    // 16 / 32 / 64 / 128 / no pre-pass give 9.88 / 9.75 / 9.68 / 9.75 / 9.96 ms per call on the bench sets and 10.72 / 10.48 /
    // 10.47 / 10.32 / 10.29 ms on CLAP-shaped ones with k = 10 (profiles/r5/ab_pre_stride.txt); narrow rows are flat.
    static const int stride_env = env_int("AM_FAST_PRE_STRIDE", 0);
    p.qstride = stride_env > 0 ? stride_env
                               : (p.wide ? (int)std::min<int64_t>(std::max<int64_t>(ceil_div(Nc, WIDE_TILE_ROWS) / 6, 16), 64) : 16);
    const int64_t sample_tiles = ceil_div(ceil_div(Nc, TB), p.qstride);
    p.pre_chunks = (int)std::min<int64_t>(sample_tiles, 8);
    static const int qcap = std::min(env_int("AM_FAST_QCAP", 2048), 2048);       // cross_verify_kernel: <= 256 * 8
    static const int ovcap = env_int("AM_FAST_OVCAP", 1 << 22);
    p.qcap = qcap;
    // The 256-row engine's regions are verified in batches of 64 (no limit from a sort in LDS): twice the size, so that a
    // few candidate columns that EVERY reference row has to look at exactly - a block of identical low-norm rows, silence in
    // a stem dataset, is everybody's nearest neighbour at one and the same distance - stay in the regions (coalesced
    // verification) instead of spilling into the one-pair-per-thread overflow path (tools/dups_probe.py)
    if (p.wide && env_int("AM_FAST_QCAP", 0) == 0) p.qcap = 4096;
    p.ovcap = ovcap;
    return p;
}

struct CrossFastBuffers {
    uint16_t *rb, *cb;
    unsigned *maxn, *rmin_approx;
    uint2 *wgq, *ovq, *items;       // items: (region, first entry) per batch of 64 entries (wide path)
    int *wgq_count, *ov_count;      // ov_count[0] = overflow entries, ov_count[1] = fail flag, ov_count[2] = number of items
};

static CrossFastBuffers carve_cross_fast(Carver& c, int64_t Nr, int64_t Nc, int D, const CrossFastPlan& p) {
    CrossFastBuffers b;
    b.rb = c.take<uint16_t>((size_t)Nr * half_ld(D));
    b.cb = c.take<uint16_t>((size_t)Nc * half_ld(D));
    b.maxn = c.take<unsigned>(4);
    b.rmin_approx = c.take<unsigned>(Nr);
    b.wgq = c.take<uint2>((size_t)p.blocks * p.qcap);
    b.ovq = c.take<uint2>((size_t)p.ovcap);
    b.items = c.take<uint2>((size_t)p.blocks * ((p.qcap + 63) / 64));
    b.wgq_count = c.take<int>(p.blocks);
    b.ov_count = c.take<int>(4);
    return b;
}

static bool cross_fast_enabled(int64_t Nr, int64_t Nc, int D) {
    static const int on = env_int("AM_PRDC_FAST", 1);
    // Where the membership filter starts to pay (round 4: tools/threshold_sweep.py on randn, unit-norm and clustered sets,
    // profiles/r4/threshold_sweep.txt): 2^24 pairs for D >= 256; for narrower rows the exact kernel's MFMA work shrinks with D
    // while the filter's fixed passes do not - 2^26 pairs (8192^2) for 128 <= D < 256, 1e8 (10 000^2) for 32 <= D < 128, 2^28 (16 384^2) below.  (Round 3 used
    // 2^24 for every width: 0.05 - 0.08 ms too slow at 6 000 - 8 000 rows x 64 / 128.)
    static const int min_pairs_env = env_int("AM_FAST_MIN_PAIRS_LOG2", 0);
    const int64_t min_pairs_n = min_pairs_env > 0 ? ((int64_t)1 << min_pairs_env)
                                                  : (D >= 256 ? (int64_t)1 << 24 : D >= 128 ? (int64_t)1 << 26 : D >= 32 ? (int64_t)100000000 : (int64_t)1 << 28);
    const size_t verify_lds = (size_t)(4 * ((D + 7) / 8 * 8) + 2048) * sizeof(float);
    return on != 0 && D >= FAST_MIN_DIM && D <= FAST_MAX_DIM && verify_lds <= 60 * 1024 && Nr * Nc >= min_pairs_n &&
           Nr < ((int64_t)1 << 31) && Nc < ((int64_t)1 << 31);
}

// rn, rt, cn, ct: norms and thresholds already computed; col_count / rmin / rany: the exact kernel's accumulators,
// initialised by the caller (0, +inf bits, 0).  On return `*fail_flag` (device) tells the exact kernel whether it has
// to run after all.
static int run_cross_fast(const float* R, int64_t Nr, int64_t ldr, const float* rn, const float* rt, const float* C, int64_t Nc,
                          int64_t ldc, const float* cn, const float* ct, int D, const CrossFastPlan& p, const CrossFastBuffers& b,
                          int32_t* col_count, unsigned* rmin, unsigned* rany, unsigned* rcov, bool want_min, hipStream_t st,
                          const PreparedSet* prep_r = nullptr, const PreparedSet* prep_c = nullptr, const Prdc64Hook* h64 = nullptr) {
    int rc;
    // (float64 route: R, C are float32-rounded copies and rt, ct rounded thresholds; both roundings are priced into the band)
    const float fc = h64 != nullptr ? fast_c(D) + 3.814697265625e-06f : fast_c(D);
    AM_REQUIRE(h64 == nullptr || (p.wide && !want_min), AM_ERR_BAD_ARG, "the float64 route is the 256-row form without row minima");
    AM_HIP_TRY(hipMemsetAsync(b.maxn, 0, 4 * sizeof(unsigned), st));
    AM_HIP_TRY(hipMemsetAsync(b.ov_count, 0, 4 * sizeof(int), st));
    if (prep_r != nullptr) hipLaunchKernelGGL(prepared_stats_kernel, dim3(1), dim3(64), 0, st, prep_r->stats, b.maxn, 0, 2, -1);
    else if ((rc = launch_to_half(R, Nr, ldr, D, rn, b.maxn, 0, b.rb, st)) != AM_OK) return rc;
    if (prep_c != nullptr) hipLaunchKernelGGL(prepared_stats_kernel, dim3(1), dim3(64), 0, st, prep_c->stats, b.maxn, 1, 3, -1);
    else if ((rc = launch_to_half(C, Nc, ldc, D, cn, b.maxn, 1, b.cb, st)) != AM_OK) return rc;
    hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)ceil_div(Nr, 256)), dim3(256), 0, st, b.rmin_approx, Nr, 0x7f800000u);
    AM_LAUNCH_CHECK();
    const int64_t ldb = half_ld(D);
    const int Dh = (int)(ldb / 2);
    const float* Rb = reinterpret_cast<const float*>(prep_r != nullptr ? prep_r->half : b.rb);
    const float* Cb = reinterpret_cast<const float*>(prep_c != nullptr ? prep_c->half : b.cb);
    int* fail = b.ov_count + 1;
    {
        const void* kernels[] = {reinterpret_cast<const void*>(&cross_fast_kernel<true, true>),
                                 reinterpret_cast<const void*>(&cross_fast_kernel<true, false>),
                                 reinterpret_cast<const void*>(&cross_fast_kernel<false, true>),
                                 reinterpret_cast<const void*>(&cross_fast_kernel<false, false>)};
        for (const void* k : kernels)
            AM_HIP_TRY(ensure_dynamic_lds(k, (int)FAST_LDS_BYTES));
    }
    const int dbg = env_int("AM_FAST_DBG", 0);
    // sampled pre-pass over every 16th column tile: certain "any" witnesses (and, when the row minimum is wanted,
    // an approximate minimum that bounds its candidate queue)
    auto launch_filter = [&](auto kernel, unsigned grid, int nchunks, int qstride, int grp_rows) {
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(ENGINE_THREADS), FAST_LDS_BYTES, st, Rb, Nr, ldb / 2, rn, rt, Cb, Nc, ldb / 2,
                           cn, ct, Dh, nchunks, qstride, b.maxn, b.rmin_approx, rany, rcov, col_count, b.wgq, p.qcap, b.wgq_count,
                           b.ovq, b.ov_count, p.ovcap, fail, dbg, fc, grp_rows);
    };
    const unsigned pre_grid = (unsigned)(ceil_div(Nr, TB) * p.pre_chunks);
    static const int pre_any = env_int("AM_FAST_PRE_ANY", 1);          // 1: on the engine of the main pass, 2: 128-row engine, 0: none
    if (want_min) launch_filter(&cross_fast_kernel<true, true>, pre_grid, p.pre_chunks, p.qstride, 0);
    else if (pre_any == 1 && p.wide) {
        // chunks of the sampled tiles per row block: the count that needs the fewest tile-times on 256 CUs (rounds of
        // workgroups x tiles per workgroup, one tile-time of fill and drain per round)
        const int64_t samples = ceil_div(ceil_div(Nc, WIDE_TILE_ROWS), p.qstride), rbs = ceil_div(Nr, WIDE_TILE_ROWS);
        int chunks = 1;
        int64_t best_cost = INT64_MAX;
        for (int c = 1; c <= 12 && c <= samples; ++c) {
            const int64_t cost = ceil_div(rbs * c, 256) * (ceil_div(samples, c) + 1);
            if (cost < best_cost) {
                best_cost = cost;
                chunks = c;
            }
        }
        if ((rc = launch_cross_wide_sample(Rb, Nr, ldb / 2, rn, Cb, Nc, ldb / 2, cn, ct, Dh, p.qstride, chunks, b.maxn, rany, fc,
                                           st)) != AM_OK)
            return rc;
    } else if (pre_any) launch_filter(&cross_fast_kernel<true, false>, pre_grid, p.pre_chunks, p.qstride, 0);
    AM_LAUNCH_CHECK();
    unsigned* rmin_or_null = want_min ? rmin : nullptr;
    if (p.wide) {
        // budgets of the data-dependent fallback (cross_fast_decide_kernel).  The overflow queue's budget is enforced INSIDE
        // the filter pass: entries past it raise the fail flag at once and the workgroups not yet started return at their
        // first instruction - on inputs the bound cannot decide most of the pass's time went into those entries
        // (an overflow-queue entry costs ~20 ns in its one-pair-per-thread kernel, a pair of the exact kernel 2 D / 140 TF:
        // the budget is what costs a quarter of the exact kernel's time)
        const long long pairs = (long long)Nr * (long long)Nc;
        const long long limit_total = std::max<long long>(pairs / 128, 65536);
        const int limit_overflow = (int)std::min<long long>(std::max<long long>((pairs >> 22) * D, 65536), p.ovcap);
        clock_begin(AM_KERNEL_PRDC_CROSS, st);
        // rows of up to 512 f16: the operand-stationary engine (pstat_engine.h), wider ones: both operands through LDS
        const auto launch_main = wide_stationary(Dh) ? &launch_cross_pstat : &launch_cross_wide;
        if ((rc = launch_main(want_min, (unsigned)p.blocks, Rb, Nr, ldb / 2, rn, rt, Cb, Nc, ldb / 2, cn, ct, Dh, p.nchunks,
                                    p.grp_rows, b.maxn, b.rmin_approx, rany, rcov, col_count, b.wgq, p.qcap, b.wgq_count, b.items,
                                    b.ovq, b.ov_count, limit_overflow, fail, fc, st)) != AM_OK)
            return rc;
        clock_end(AM_KERNEL_PRDC_CROSS, st);
        hipLaunchKernelGGL(cross_fast_decide_kernel, dim3(1), dim3(64), 0, st, b.ov_count, p.ovcap, fail, limit_total, limit_overflow);
        clock_begin(AM_KERNEL_PRDC_VERIFY, st);
        if (h64 != nullptr) {
            hipLaunchKernelGGL(cross_verify_regions64_kernel, dim3(CROSS_VERIFY_GRID), dim3(256), 0, st, *h64, D, b.wgq, p.qcap, b.wgq_count,
                               b.items, b.ov_count + 2, col_count, rany, rcov, fail);
        } else {
            AM_HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(&cross_verify_regions_kernel), (int)VERIFY_LDS_BYTES));
            hipLaunchKernelGGL(cross_verify_regions_kernel, dim3(CROSS_VERIFY_GRID), dim3(256), VERIFY_LDS_BYTES, st, R, ldr, rn, rt, C, ldc,
                               cn, ct, D, b.wgq, p.qcap, b.wgq_count, b.items, b.ov_count + 2, col_count, rmin_or_null, rany, rcov, fail);
        }
        clock_end(AM_KERNEL_PRDC_VERIFY, st);
    } else {
        clock_begin(AM_KERNEL_PRDC_CROSS, st);
        if (want_min) launch_filter(&cross_fast_kernel<false, true>, (unsigned)p.blocks, p.nchunks, 1, p.grp_rows);
        else launch_filter(&cross_fast_kernel<false, false>, (unsigned)p.blocks, p.nchunks, 1, p.grp_rows);
        clock_end(AM_KERNEL_PRDC_CROSS, st);
        AM_LAUNCH_CHECK();
        const size_t verify_lds = (size_t)(4 * ((D + 7) / 8 * 8) + p.qcap) * sizeof(float);
        clock_begin(AM_KERNEL_PRDC_VERIFY, st);
        hipLaunchKernelGGL(cross_verify_kernel, dim3((unsigned)p.blocks), dim3(256), verify_lds, st, R, Nr, ldr, rn, rt, C, ldc, cn,
                           ct, D, p.nchunks, p.grp_rows, b.wgq, p.qcap, b.wgq_count, col_count, rmin_or_null, rany, rcov);
        clock_end(AM_KERNEL_PRDC_VERIFY, st);
    }
    AM_LAUNCH_CHECK();
    if (h64 != nullptr)
        hipLaunchKernelGGL(cross_verify_overflow64_kernel, dim3(1024), dim3(256), 0, st, *h64, D, b.ovq, b.ov_count, p.ovcap, fail, col_count,
                           rany, rcov);
    else
    hipLaunchKernelGGL(cross_verify_overflow_kernel, dim3(1024), dim3(256), 0, st, R, ldr, rn, rt, C, ldc, cn, ct, D, b.ovq,
                       b.ov_count, p.ovcap, fail, col_count, rmin_or_null, rany, rcov);
    AM_LAUNCH_CHECK();
    hipLaunchKernelGGL(cross_fail_reset_kernel, dim3(256), dim3(256), 0, st, fail, col_count, Nc, rmin, rany, rcov, Nr);
    AM_LAUNCH_CHECK();
    if (long long* stats = filter_stats_for_current_device()) {
        hipLaunchKernelGGL(filter_stats_kernel, dim3(1), dim3(256), 0, st, b.wgq_count, (int64_t)p.blocks, stats, 5, 6,
                           (const unsigned long long*)nullptr, (const int*)b.ov_count, p.ovcap, 7, (const int*)fail, 8,
                           (const int*)nullptr, 0);
        AM_LAUNCH_CHECK();
    }
#ifdef AM_DEV_KNOBS
    static const int debug = env_int("AM_FAST_DEBUG", 0);
    if (debug) {                                       // development aid: synchronises
        AM_HIP_TRY(hipStreamSynchronize(st));
        std::vector<int> wc(p.blocks);
        int ovc[2] = {0, 0};
        AM_HIP_TRY(hipMemcpy(wc.data(), b.wgq_count, p.blocks * sizeof(int), hipMemcpyDeviceToHost));
        AM_HIP_TRY(hipMemcpy(ovc, b.ov_count, 2 * sizeof(int), hipMemcpyDeviceToHost));
        long long tot = 0, full = 0;
        int wmax = 0;
        for (int v : wc) { tot += v; full += (v >= p.qcap); wmax = std::max(wmax, v); }
        fprintf(stderr, "[cross_fast] blocks=%lld nchunks=%d queued=%lld (max/wg %d, full regions %lld) overflow queue=%d fail=%d\n",
                (long long)p.blocks, p.nchunks, tot, wmax, full, ovc[0], ovc[1]);
    }
#endif
    return AM_OK;
}

// ------------------------------------------------------------------------------------------------
// k-NN radii of a set against itself, filter pass: the symmetric sweep of knn_sym_kernel on the f16 copy.
// thr[i] is an upper bound of (true (k+1)-th smallest t of row i) + E_i, E_i = fast_c (|x_i|^2 + max_j |x_j|^2):
// every pair with t_ij <= the row's final value has a_ij <= thr[i], in whichever direction it is met.
//   own rows (lane-local):  queue (i, j) when a <= min(thr[i] at workgroup start, kthA + 2 E_i), kthA = the
//                           (k+1)-th smallest max(a, 0) this lane has seen (true values of those columns are
//                           <= max(a, 0) + E_i, so kthA + E_i bounds the row's final value from above);
//   mirrored (Q rows):      queue (j, i) when a <= thr[j].
// The per-lane lists are merged per row and window, published (kthA + 2 E_i, cumulative over the windows done)
// exactly like the exact kernel's, but they only steer the filter: the radii come from the exact values of the
// queued pairs (knn_fast_scatter_kernel -> knn_fast_prune_kernel -> knn_fast_verify_kernel -> knn_fast_select_kernel).
// Queue entry: (row a | FAST_BOTH, row b): the exact value t(a, b) is filed under row a, and under row b as well when
// FAST_BOTH is set (a pair that passes the own-row test of a and the mirrored test of b is evaluated once).
// Xb: f16 copy viewed as f32 words (ldh, Dh in words).  Same grid and work mapping as knn_sym_kernel.
template <int KCAP>
__global__ void __launch_bounds__(ENGINE_THREADS, 2) __attribute__((amdgpu_waves_per_eu(2, 2)))
knn_fast_kernel(const float* __restrict__ Xb, int64_t N, int64_t ldh, const float* __restrict__ xnorm, float* thr, int Dh,
                int win_tiles, int nwin, int per_win, int k1, const unsigned* __restrict__ maxn, float* __restrict__ partial,
                int* __restrict__ cnt, int cap, uint2* __restrict__ wgq, float* __restrict__ wgv, int qcap,
                int* __restrict__ wgq_count, int part, int nparts, float fc, uint2* __restrict__ ovq, float* __restrict__ ovv,
                unsigned long long* __restrict__ ovn, int ovcap, const int* __restrict__ skip, int* __restrict__ region_counter) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const LaneInfo L;
    const int64_t T = (N + TB - 1) / TB;
    const SymWork sw = sym_work(T, win_tiles, nwin, per_win, part, nparts);
    constexpr int NW = KnnFastEpilogue<KCAP>::NWAVES;
    if (sw.ntiles == 0 || (skip != nullptr && *skip != 0)) {      // (skip: the data-dependent fallback took over, see knn_fast_predict_kernel)
        if (region_counter == nullptr && L.tid <= NW) wgq_count[(int64_t)blockIdx.x * (NW + 1) + L.tid] = 0;
        return;
    }
    // Queue region of this workgroup: its own index - or, in a PARTITIONED run (region_counter given), the next free one:
    // only the workgroups of this rank's row blocks queue anything, so the memory of all regions is cut into as many (larger)
    // regions as there are active workgroups and handed out in arrival order (the counts were zeroed by the host).  With
    // regions indexed by workgroup a rank used 1 / nparts of them, and at 1M rows the busiest ones overflowed: 14 000 rows
    // went through the row-at-a-time fix-up (6 - 11 s per set on 8 ranks, tools/scale_model.py).
    int64_t region = blockIdx.x;
    if (region_counter != nullptr) {
        int* slot = reinterpret_cast<int*>(lds + ENGINE_LDS_FLOATS + 4 * TB) + 1;
        if (L.tid == 0) *slot = atomicAdd(region_counter, 1);
        __syncthreads();
        region = __builtin_amdgcn_readfirstlane(*slot);
    }
    const float nmax = __uint_as_float(maxn[0]);
    KnnFastEpilogue<KCAP> epi(L);
    epi.qnorm = xnorm;
    epi.thr = thr;
    epi.n = N;
    epi.pblock = sw.pb;
    epi.aux = lds + ENGINE_LDS_FLOATS;
    const int wave = __builtin_amdgcn_readfirstlane(L.tid >> 6);
    epi.wcap = qcap / (2 * NW);                       // private sub-regions: half of the workgroup's region in all
    epi.wgq = wgq + region * qcap + wave * epi.wcap;
    epi.wgv = wgv + region * qcap + wave * epi.wcap;
    epi.wq = 0;
    epi.shcap = qcap - NW * epi.wcap;                 // the shared part behind them
    epi.shq = wgq + region * qcap + NW * epi.wcap;
    epi.shv = wgv + region * qcap + NW * epi.wcap;
    epi.qn = reinterpret_cast<int*>(lds + ENGINE_LDS_FLOATS + 4 * TB);
    if (L.tid == 0) *epi.qn = 0;                    // visible after the pipeline's first barrier
    epi.ovq = ovq;
    epi.ovv = ovv;
    epi.ovn = ovn;
    epi.ovcap = ovcap;
    epi.cnt = cnt;
    epi.cap = cap;
    epi.dsc = half_unscale(maxn[2], maxn[2]);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int64_t i = sw.pb * TB + L.wn * 64 + nt * 32 + L.r;
        epi.prow[nt] = (unsigned)i;
        epi.xn[nt] = i < N ? xnorm[i] : INFINITY;
        epi.e2c = 2.f * fc;
        epi.e2n = 2.f * fc * nmax;
        epi.flt[nt] = i < N ? __hip_atomic_load(thr + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -INFINITY;
#pragma unroll
        for (int s = 0; s < KCAP; ++s) epi.best[nt][s] = s < KCAP - k1 ? -INFINITY : INFINITY;
    }
    dense_pipeline_early<EV_FAST, false>(Xb, N, ldh, LinearTiles{sw.qa}, Xb, N, ldh, sw.pb * TB, sw.ntiles, Dh, lds, L, epi);
    float* mg = lds;                                   // [128][4][KCAP]
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        float* dst = mg + ((L.wn * 64 + nt * 32 + L.r) * 4 + (L.wm * 2 + L.h)) * KCAP;
#pragma unroll
        for (int s = 0; s < KCAP; ++s) dst[s] = epi.best[nt][s];
    }
    __syncthreads();
    if (L.lane == 0) wgq_count[region * (NW + 1) + wave] = min(epi.wq, epi.wcap);
    if (L.tid == 0) wgq_count[region * (NW + 1) + NW] = min(*epi.qn, epi.shcap);
    if (L.tid < TB) {
        const int64_t i = sw.pb * TB + L.tid;
        if (i < N) {
            const float* src = mg + L.tid * 4 * KCAP;
            float m[KCAP];                             // same padding: m[KCAP-1] = (k+1)-th smallest of the finite entries
#pragma unroll
            for (int s = 0; s < KCAP; ++s) m[s] = src[s];
            for (int s = KCAP; s < 4 * KCAP; ++s)
                if (src[s] > -INFINITY) list_insert<KCAP>(m, src[s]);
            // CUMULATIVE list: this window's values merged with the cumulative list of the row block's previous window in
            // processing order (the nearest higher window in which the block owned tiles; windows are dispatched in
            // descending order) - distinct columns, so the (k+1)-th smallest still bounds the row's final value from above.
            // One list is read (KCAP independent loads) instead of the own lists of ALL higher windows one value at a time:
            // the compiler kept those agent-scope loads strictly serial (s_waitcnt vmcnt(0) behind each), up to 21 x KCAP
            // L2-missing round trips at the end of every workgroup - ~10 % of the kernel at 100k rows.
            // (A block still running, or not started, leaves +inf pads or a partly written list there: any subset is valid.)
            int prev = -1;
            for (int w2 = sw.W + 1; w2 < nwin && prev < 0; ++w2)
                if (sym_item(T, win_tiles, w2, sw.pb, part, nparts).ntiles > 0) prev = w2;
            if (prev >= 0) {
                const float* src2 = partial + ((int64_t)prev * N + i) * KCAP;
                float v[KCAP];
#pragma unroll
                for (int s = 0; s < KCAP; ++s) v[s] = __hip_atomic_load(src2 + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int s = 0; s < KCAP; ++s)
                    if (v[s] > -INFINITY) list_insert<KCAP>(m, v[s]);
            }
            // write-through stores / agent-scope loads: other XCDs read these lists while the kernel runs
            float* out = partial + ((int64_t)sw.W * N + i) * KCAP;
#pragma unroll
            for (int s = 0; s < KCAP; ++s) __hip_atomic_store(out + s, m[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float kthv = m[KCAP - 1];
            const float bound = kthv + 2.f * fc * (xnorm[i] + nmax);     // >= 0, so its bit pattern orders like the value
            // (a row taken out of the sweep keeps its -inf: knn_fast_mask_flat_kernel; nobody else writes thr[i])
            if (epi_row_in_sweep(thr, i)) atomicMin(reinterpret_cast<unsigned*>(thr) + i, __float_as_uint(bound));
        }
    }
}

// pre-pass result (approximate (k+1)-th smallest over the column sample) -> filter bound
__global__ void __launch_bounds__(256) knn_fast_bound_kernel(const float* __restrict__ in, float* __restrict__ thr,
                                                             const float* __restrict__ xnorm, int64_t n,
                                                             const unsigned* __restrict__ maxn, float factor /* 1 or 2 times fast_c(D) */) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) thr[i] = in[i] + factor * (xnorm[i] + __uint_as_float(maxn[0]));     // +inf stays +inf
}

// After the sweep: (1) the APPROXIMATE values of the queued pairs are filed under their rows
// (knn_fast_scatter_kernel); (2) per row, kq = the (k+1)-th smallest max(a, 0) among ITS filed entries bounds the
// row's final value: the k+1 entries behind it have true values <= kq + E_i, so every pair the row needs has
// a <= kq + 2 E_i - entries above that are dropped, the survivors go to one global pair list
// (knn_fast_prune_kernel; the bounds during the sweep come from a column sample and from half-finished lists, so
// ~4/5 of the queue is dropped here); (3) the exact value of each surviving pair is computed with the exact
// engine's fmaf chain and filed under its row (knn_fast_verify_kernel); (4) selection, (5) exact fix-up of rows
// whose buffers overflowed anywhere on the way (count marker > cap).
__device__ __forceinline__ void knn_file_approx(float* __restrict__ fval, unsigned* __restrict__ fidx, int* __restrict__ cnt,
                                                int cap, int64_t row, float v, unsigned partner) {
    const int slot = atomicAdd(cnt + row, 1);
    if (slot < cap) {
        fval[row * (int64_t)cap + slot] = v;
        fidx[row * (int64_t)cap + slot] = partner;
    }
}

// one workgroup per queue part: blockIdx = workgroup of the sweep * (nsub + 1) + part; parts 0 .. nsub-1 are the waves'
// private sub-regions (wcap entries each), part nsub is the shared part behind them
__global__ void __launch_bounds__(256) knn_fast_scatter_kernel(const uint2* __restrict__ wgq, const float* __restrict__ wgv,
                                                               int qcap, int wcap, int nsub, const int* __restrict__ wgq_count,
                                                               float* __restrict__ fval, unsigned* __restrict__ fidx,
                                                               int* __restrict__ cnt, int cap, const float* __restrict__ thr,
                                                               const int* __restrict__ skip) {
    if (skip != nullptr && *skip != 0) return;
    // thr[i] (the bound the sweep left behind: (k+1)-th smallest approximate value seen + 2E) admits every pair that can
    // be among row i's k+1 smallest; most entries were queued under the much looser bounds of the first windows and are
    // dropped here instead of being filed and pruned later
    const int n = wgq_count[blockIdx.x];
    const int64_t base = (int64_t)(blockIdx.x / (nsub + 1)) * qcap + (int64_t)(blockIdx.x % (nsub + 1)) * wcap;
    const uint2* q = wgq + base;
    const float* v = wgv + base;
    for (int e = threadIdx.x; e < n; e += 256) {
        const uint2 p = q[e];
        const unsigned a = p.x & ~FAST_BOTH;
        const float val = v[e];
        if (val <= thr[a]) knn_file_approx(fval, fidx, cnt, cap, a, val, p.y);
        if ((p.x & FAST_BOTH) && val <= thr[p.y]) knn_file_approx(fval, fidx, cnt, cap, p.y, val, a);
    }
}

__global__ void __launch_bounds__(256) knn_fast_scatter_spill_kernel(const uint2* __restrict__ ovq, const float* __restrict__ ovv,
                                                                     const unsigned long long* __restrict__ ovn, int ovcap,
                                                                     float* __restrict__ fval, unsigned* __restrict__ fidx,
                                                                     int* __restrict__ cnt, int cap, const float* __restrict__ thr,
                                                                     const int* __restrict__ skip) {
    if (skip != nullptr && *skip != 0) return;
    const int n = (int)min(*ovn, (unsigned long long)ovcap);
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
        const uint2 p = ovq[e];
        const unsigned a = p.x & ~FAST_BOTH;
        const float val = ovv[e];
        if (val <= thr[a]) knn_file_approx(fval, fidx, cnt, cap, a, val, p.y);
        if ((p.x & FAST_BOTH) && val <= thr[p.y]) knn_file_approx(fval, fidx, cnt, cap, p.y, val, a);
    }
}

// one thread per row; survivors are appended to pairs[] (one atomicAdd per wave)
template <int KCAP>
__global__ void __launch_bounds__(256) knn_fast_prune_kernel(const float* __restrict__ fval, const unsigned* __restrict__ fidx,
                                                             const int* __restrict__ cnt, int cap, int64_t N, int k1,
                                                             const float* __restrict__ xnorm, const unsigned* __restrict__ maxn,
                                                             uint2* __restrict__ pairs, float* __restrict__ pair_val, int pair_cap,
                                                             int* __restrict__ pair_count, int* __restrict__ cnt2, int partitioned,
                                                             float fc, int* __restrict__ gate, int* __restrict__ pair_start = nullptr,
                                                             int* __restrict__ pair_n = nullptr) {
    if (gate != nullptr && gate[0] != 0) return;          // (gate: data-dependent fallback, see knn_fast_predict_kernel)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int c = i < N ? cnt[i] : 0;
    // overflow on the way here; c < k1 cannot happen on one GPU (the true top k+1 are always queued) - in the
    // partitioned form a rank may hold fewer than k+1 entries of a row, which then all survive (kq = +inf)
    const bool bad = i < N && (c > cap || (c < k1 && !partitioned));
    if (gate != nullptr) {
        const unsigned long long nb = __ballot(bad);
        if (lane == 0 && nb != 0ull) atomicAdd(gate + 3, __popcll(nb));
    }
    const float* fv = fval + i * (int64_t)cap;
    int ns = 0;
    float thr = -INFINITY;
    if (i < N && !bad) {
        float m[KCAP];
#pragma unroll
        for (int s = 0; s < KCAP; ++s) m[s] = INFINITY;
        for (int s = 0; s < c; ++s) {
            const float v = fmaxf(fv[s], 0.f);
            if (v < m[KCAP - 1]) list_insert<KCAP>(m, v);
        }
        float kq = m[0];
#pragma unroll
        for (int s = 1; s < KCAP; ++s)
            if (s == k1 - 1) kq = m[s];
        thr = kq + 2.f * fc * (xnorm[i] + __uint_as_float(maxn[0]));
        for (int s = 0; s < c; ++s) ns += fv[s] <= thr;
    }
    int incl = ns;                                        // wave-inclusive prefix sum of the survivor counts
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    const int total = __shfl(incl, 63);
    int base = 0;
    if (lane == 63 && total > 0) base = atomicAdd(pair_count, total);
    base = __shfl(base, 63);
    if (i >= N) return;
    int at = base + incl - ns;
    if (pair_start != nullptr) {                          // (float64 route: where this row's pairs lie, -1 = no list)
        pair_start[i] = (bad || at + ns > pair_cap) ? -1 : at;
        pair_n[i] = ns;
    }
    if (bad || at + ns > pair_cap) {                      // overflow on the way here, or the pair list is full: -> exact fix-up
        cnt2[i] = cap + 1;
        // a range that straddles the end of the list is not written: mark its slots below the end as holes
        if (!bad) for (int e = at; e < pair_cap; ++e) pairs[e] = make_uint2(FAST_HOLE, 0u);
        return;
    }
    const unsigned* fi = fidx + i * (int64_t)cap;
    for (int s = 0; s < c; ++s)
        if (fv[s] <= thr) {
            pair_val[at] = fv[s];                              // the f16 value travels with the pair: the verification measures the bound on it
            pairs[at++] = make_uint2((unsigned)i, fi[s]);
        }
}

// One pair per lane, 64 pairs per wave at a time (wave_pair_dot): the exact engine's fmaf chain (t(a,b) == t(b,a) bit for
// bit: products commute and the inner order is the same).  Consecutive pairs share their first row (the list is written
// row by row); the partner rows come from L2 / Infinity Cache.
__device__ __forceinline__ void knn_file(float* __restrict__ cand, int* __restrict__ cnt, int cap, int64_t row, float t) {
    const int slot = atomicAdd(cnt + row, 1);
    if (slot < cap) cand[row * (int64_t)cap + slot] = t;
}

__global__ void __launch_bounds__(256) knn_fast_verify_kernel(const float* __restrict__ X, int64_t ld,
                                                              const float* __restrict__ xnorm, int D,
                                                              const uint2* __restrict__ pairs, const int* __restrict__ pair_count,
                                                              int pair_cap, float* __restrict__ cand, int* __restrict__ cnt2,
                                                              int cap, const int* __restrict__ skip, const float* __restrict__ pair_val,
                                                              float fc, const unsigned* __restrict__ maxn,
                                                              unsigned long long* __restrict__ bound_slots) {
    extern __shared__ __attribute__((aligned(16))) float vlds[];
    if (skip != nullptr && *skip != 0) return;
    // bound_slots (am_filter_stats_enable): the verification holds BOTH values of every surviving pair - the f16 matrix-core
    // value `a` the filter decided on and the exact f32 value `t` - so the error bound the filter rests on,
    // |a - t| <= fast_c(D) (|x|^2 + G) with x the smaller-normed row of the pair, is MEASURED here instead of assumed:
    // slot 0 = max of |a - t| / bound as f32 bits (<= 1 or the filter is unsound), slot 1 = pairs measured
    float worst = 0.f;
    unsigned long long measured = 0ull;
    const int n = min(*pair_count, pair_cap);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* tile = vlds + wave * 128 * VERIFY_LD;                     // rows 0..63: first rows of the pairs, 64..127: second rows
    for (int64_t base = ((int64_t)blockIdx.x * 4 + wave) * 64; base < n; base += (int64_t)gridDim.x * 256) {
        const int64_t e = base + lane;
        uint2 p = e < n ? pairs[e] : make_uint2(FAST_HOLE, 0u);
        const bool hole = p.x == FAST_HOLE;
        if (hole) p = make_uint2(0u, 0u);
        const float acc = wave_pair_dot(X, ld, X, ld, p.x, p.y, D, tile, lane);
        if (!hole) {
            const float nx = xnorm[p.x], ny = xnorm[p.y];
            const float t = fmaf(-2.f, acc, nx + ny);
            knn_file(cand, cnt2, cap, p.x, clamp0(t));
            if (bound_slots != nullptr) {
                const float bound = fc * (fminf(nx, ny) + __uint_as_float(maxn[0]));
                const float ratio = fabsf(pair_val[e] - t) / bound;
                if (ratio == ratio) worst = fmaxf(worst, ratio);   // (0 / 0 of an all-zero set: no statement)
                ++measured;
            }
        }
    }
    if (bound_slots != nullptr) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            worst = fmaxf(worst, __shfl_xor(worst, off));
            measured += __shfl_xor(measured, off);
        }
        if (lane == 0 && measured != 0ull) {
            atomicMax(bound_slots, (unsigned long long)__float_as_uint(worst));      // worst >= 0: bit order = value order
            atomicAdd(bound_slots + 1, measured);
        }
    }
}

// radius[i] = sqrt_rn( (k+1)-th smallest exact value filed under row i ); overflowed rows -> exact fix-up.
// Partitioned form (out_lists != nullptr): this rank's KCAP smallest exact values of the row (+inf padded), or a NaN
// in slot 0 when the row overflowed here (am_knn_lists_finish_f32 then recomputes it exactly).
template <int KCAP>
__global__ void knn_fast_select_kernel(const float* __restrict__ cand, const int* __restrict__ cnt, int cap, int64_t N, int k1,
                                       const unsigned* __restrict__ maxn, float* __restrict__ radii, int* __restrict__ ov_list,
                                       int* __restrict__ ov_count, float* __restrict__ out_lists, const int* __restrict__ skip) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N || (skip != nullptr && *skip != 0)) return;
    const int c = cnt[i];
    if (c > cap || (out_lists == nullptr && c < k1) || !half_scale_ok(maxn[2])) {
        if (out_lists != nullptr) out_lists[i * KCAP] = NAN;
        else ov_list[atomicAdd(ov_count, 1)] = (int)i;
        return;
    }
    float m[KCAP];
#pragma unroll
    for (int s = 0; s < KCAP; ++s) m[s] = INFINITY;
    const float* src = cand + i * cap;
    for (int s = 0; s < c; ++s) {
        const float v = src[s];
        if (v < m[KCAP - 1]) list_insert<KCAP>(m, v);
    }
    if (out_lists != nullptr) {
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

// ---- batched fix-up (round 4) -----------------------------------------------------------------------------------------
// Rows whose candidate buffers overflowed - a block of identical rows larger than the buffer: every one of them has all the
// others at distance zero; silent windows of a stem dataset embed to the same vector - were recomputed one row at a time on
// the vector ALUs (knn_fixup_kernel: 42 us per row at 100 000 x 512; 500 duplicates = 21 ms, 1000 = 42 ms, beside a 7 ms
// call).  From KNN_FIX_BATCH_FROM rows on they are gathered into a contiguous copy and go through the exact GENERAL kernel
// on the matrix cores (rows of the copy against all columns: 0.75 us per row), the radii are scattered back; same values bit
// for bit (the general kernel is the exact arithmetic).  The list is a device-side count: the grids are sized for the
// copy's capacity (N / 8 rows; more overflowed rows than that send the whole call to the exact kernel, check B) and the
// workgroups past the count return at once.
__global__ void __launch_bounds__(256) knn_gather_rows_kernel(const float* __restrict__ X, int64_t ld, int D, const float* __restrict__ xnorm,
                                                              const int* __restrict__ ov_list, const int* __restrict__ ov_count,
                                                              int batched_from, int capacity, float* __restrict__ rows, int64_t ldr,
                                                              float* __restrict__ norms) {
    const int n = *ov_count < capacity ? *ov_count : capacity;
    if (*ov_count < batched_from) return;
    for (int r = blockIdx.x; r < n; r += gridDim.x) {
        const int64_t src = ov_list[r];
        for (int c = threadIdx.x; c < (int)ldr; c += 256) rows[(int64_t)r * ldr + c] = c < D ? X[src * ld + c] : 0.f;
        if (threadIdx.x == 0) norms[r] = xnorm[src];
    }
}

__global__ void __launch_bounds__(256) knn_scatter_radii_kernel(const float* __restrict__ radii_of_copy, const int* __restrict__ ov_list,
                                                                const int* __restrict__ ov_count, int batched_from, int capacity,
                                                                float* __restrict__ radii) {
    const int n = *ov_count < capacity ? *ov_count : capacity;
    if (*ov_count < batched_from) return;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < n; r += gridDim.x * 256) radii[ov_list[r]] = radii_of_copy[r];
}

constexpr int KNN_FIX_BATCH_FROM = 32;                        // shorter lists of overflowed rows: one row at a time (knn_fixup_kernel)
static inline int64_t knn_fix_capacity(int64_t N) { return ceil_div(std::max<int64_t>(N / 8, 1024), TB) * TB; }

struct KnnFixup {                 // batched fix-up: gathered copy of the overflowed rows, their norms, the general kernel's
    float *rows, *norms, *partial, *radii;                                                    // partial lists and radii
    int64_t capacity;
};

static KnnFixup carve_knn_fixup(Carver& c, int64_t N, int D, int kcap) {
    KnnFixup x;
    x.capacity = knn_fix_capacity(N);
    x.rows = c.take<float>((size_t)x.capacity * ((D + 3) / 4 * 4));
    x.norms = c.take<float>((size_t)x.capacity);
    x.partial = c.take<float>((size_t)choose_chunks(x.capacity, N) * x.capacity * kcap);
    x.radii = c.take<float>((size_t)x.capacity);
    return x;
}

// radii of the rows in ov_list (a device-side count): one row at a time below KNN_FIX_BATCH_FROM rows, through the gathered
// copy from there on (rows past the copy's capacity: one at a time again - the callers keep the list below it)
// ---- float64 rows (round 5): the candidates of the f16 filter sweep, evaluated and selected in f64 --------------------------
// The sweep, the scatter and the prune step run on a float32-rounded copy of the rows exactly as for float32 sets; their
// product is, per row, the list of partners whose squared distance can be among the row's k + 1 smallest.  That statement is
// about the TRUE distances of the float64 rows as well: the f16 value a of a pair differs from the real-number distance of the
// rounded rows by at most fast_c (|x|^2 + G) (the derivation above bounds exactly that, the f32 chain's own error is a term of
// it), and rounding the rows to float32 first moves a distance by at most 2^-21 (|x|^2 + |y|^2) - run_knn_fast adds 2^-19 to
// fast_c for this route.  One wave per row then evaluates the row's pairs in f64, sum of squared differences (no norms, exact
// zero for the row itself), and takes the (k+1)-th smallest by repeated extraction.  A row without a usable list (overflow on
// the way, a block of identical rows taken out of the sweep, operands that cannot be scaled) raises gate[1]: the general f64
// kernels, launched behind it, then compute the whole call (pairwise_f64.hip: knn64_self_gated).
struct Knn64Hook {
    const double* X;
    int64_t ld;
    double* out_r;
    int *pair_start, *pair_n;       // [N] each
    void* gen_ws;                   // workspace of the gated general kernels
    size_t gen_ws_bytes;
    int k;
};
constexpr int KNN64_PER_LANE = 12;                       // 64 x 12 >= the candidate slots of a row (64 (k + 1), k <= 10)
constexpr int KNN64_ROWS_PER_WAVE = 8;

__global__ void __launch_bounds__(256) knn_fast_select64_kernel(const double* __restrict__ X, int64_t ld, int D,
                                                                const uint2* __restrict__ pairs, const int* __restrict__ pair_start,
                                                                const int* __restrict__ pair_n, int64_t N, int k1,
                                                                const unsigned* __restrict__ maxn, double* __restrict__ out_r,
                                                                int* __restrict__ gate, const float* __restrict__ pair_val,
                                                                const float* __restrict__ xnorm, float fc,
                                                                unsigned long long* __restrict__ bound_slots) {
    if (*reinterpret_cast<volatile int*>(gate + 1) != 0) return;     // the general kernels take the call anyway
    // bound_slots (am_filter_stats_enable; round 6): this route holds both values of every surviving pair too - the
    // f16 matrix-core value the sweep decided on (computed from the float32-ROUNDED rows) and the float64 sum of squared
    // differences of the float64 rows - so the widened bound it rests on, |a - t| <= (fast_c(D) + 2^-19) (|x|^2 + G), is
    // measured like the float32 route's (slot 0 = max ratio as f32 bits, slot 1 = pairs measured).  The row itself (distance 0,
    // a = rounding noise around 0) is a pair like any other.
    float worst = 0.f;
    unsigned long long measured = 0ull;
    const int lane = threadIdx.x & 63;
    // a wave takes KNN64_ROWS_PER_WAVE consecutive rows: the two statistics atomics are paid once per wave, not once per row
    // (100 000 waves on one address cost 2 ms per set)
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * KNN64_ROWS_PER_WAVE;
    for (int64_t row = row0; row < row0 + KNN64_ROWS_PER_WAVE && row < N; ++row) {
        const int start = pair_start[row], n = pair_n[row];
        if (start < 0 || n < k1 || n > 64 * KNN64_PER_LANE || !half_scale_ok(maxn[2])) {
            if (lane == 0) atomicOr(gate + 1, 1);
            continue;
        }
        const double* xi = X + row * ld;
        // EIGHT lanes share one candidate (round 6): a row has ~7 candidates, so one candidate per lane left nine lanes in ten
        // idle while the others walked a 512-byte row 8 bytes at a time (0.47 / 0.95 ms per set at 100 000 x 64 / 128 against
        // 0.16 ms for the float32 twin).  Lane group g = lane >> 3 takes candidate 8 it + g, its lane s = lane & 7 the s-th
        // eighth of the row (a contiguous piece: the group reads the row coalesced); partial sums in element order, then three
        // butterfly steps.  Candidate p's sum lands in v[p / 64] of lane p % 64, where the selection below expects it.
        const int grp = lane >> 3, sub = lane & 7;
        const int piece = (D + 7) / 8, d0 = sub * piece, d1 = min(D, d0 + piece);
        double v[KNN64_PER_LANE];
#pragma unroll
        for (int q = 0; q < KNN64_PER_LANE; ++q) {
            v[q] = __builtin_inf();
            if (64 * q >= n) continue;                                  // (wave-uniform)
            for (int it8 = 0; it8 < 8; ++it8) {
                const int p0 = 64 * q + 8 * it8;                        // candidates p0 .. p0 + 7 of this step
                if (p0 >= n) break;                                     // (wave-uniform)
                const int p = p0 + grp;
                double sum = 0.0;
                unsigned partner = 0u;
                if (p < n) {
                    partner = pairs[start + p].y;
                    const double* xj = X + (int64_t)partner * ld;
                    for (int d = d0; d < d1; ++d) {
                        const double t = xi[d] - xj[d];
                        sum = fma(t, t, sum);
                    }
                }
                sum += __shfl_xor(sum, 1);
                sum += __shfl_xor(sum, 2);
                sum += __shfl_xor(sum, 4);
                sum = (p < n && sum == sum) ? sum : __builtin_inf();    // (a NaN distance is nobody's neighbour: clamp0)
                if (bound_slots != nullptr && p < n && sub == 0) {
                    const float bound = fc * (fminf(xnorm[row], xnorm[partner]) + __uint_as_float(maxn[0]));
                    const float ratio = fabsf(pair_val[start + p] - (float)sum) / bound;
                    if (ratio == ratio && ratio < __builtin_inff()) worst = fmaxf(worst, ratio);
                    ++measured;
                }
                // candidate p0 + j sits in lane group j: lane 8 it8 + j of this q takes it
                const double got = __shfl(sum, (lane & 7) * 8);
                if ((lane >> 3) == it8) v[q] = got;
            }
        }
        double kth = __builtin_inf();
        for (int r = 0; r < k1; ++r) {
            double m = v[0];
            int mq = 0;
#pragma unroll
            for (int q = 1; q < KNN64_PER_LANE; ++q)
                if (v[q] < m) {
                    m = v[q];
                    mq = q;
                }
            double wm = m;
            int wl = lane;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const double om = __shfl_xor(wm, off);
                const int ol = __shfl_xor(wl, off);
                if (om < wm || (om == wm && ol < wl)) {
                    wm = om;
                    wl = ol;
                }
            }
            kth = wm;
            if (lane == wl) {
#pragma unroll
                for (int q = 0; q < KNN64_PER_LANE; ++q)
                    if (q == mq) v[q] = __builtin_inf();
            }
        }
        if (lane == 0) out_r[row] = __dsqrt_rn(kth);
    }
    if (bound_slots != nullptr) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            worst = fmaxf(worst, __shfl_xor(worst, off));
            measured += __shfl_xor(measured, off);
        }
        if (lane == 0 && measured != 0ull) {
            atomicMax(bound_slots, (unsigned long long)__float_as_uint(worst));
            atomicAdd(bound_slots + 1, measured);
        }
    }
}

template <int KCAP>
static int run_knn_fixup(const float* X, int64_t N, int64_t ld, const float* xn, int D, int k1, const int* ov_list, const int* ov_count,
                         const KnnFixup& x, float* out_r, hipStream_t st) {
    int rc;
    const int fcap = (int)x.capacity;
    const int64_t ldr = (D + 3) / 4 * 4;
    // (x.partial, at least capacity x KCAP floats, is free until the batched form's general kernel below: split lists of a few rows)
    launch_knn_fixup<KCAP>(X, N, ld, xn, D, k1, ov_list, ov_count, out_r, KNN_FIX_BATCH_FROM, fcap, x.partial, st);
    hipLaunchKernelGGL(knn_gather_rows_kernel, dim3(1024), dim3(256), 0, st, X, ld, D, xn, ov_list, ov_count, KNN_FIX_BATCH_FROM, fcap,
                       x.rows, ldr, x.norms);
    AM_LAUNCH_CHECK();
    if ((rc = launch_knn<KCAP>(x.rows, fcap, ldr, x.norms, X, N, ld, xn, D, k1, choose_chunks(fcap, N), 1, false, x.partial, x.radii,
                               st, nullptr, ov_count, KNN_FIX_BATCH_FROM)) != AM_OK)
        return rc;
    hipLaunchKernelGGL(knn_scatter_radii_kernel, dim3(64), dim3(256), 0, st, x.radii, ov_list, ov_count, KNN_FIX_BATCH_FROM, fcap, out_r);
    AM_LAUNCH_CHECK();
    return AM_OK;
}

// ---- data-dependent fallback of the k-NN filter path (round 4) -----------------------------------------------------
// The filter pays when the f16 values SEPARATE a row's nearest neighbours from the rest.  On tightly clustered data they
// do not: every member of a row's cluster lies inside the error band of its (k+1)-th neighbour, the queues keep them all,
// the prune step cannot drop them, and the exact verification touches hundreds of pairs per row (tools/threshold_sweep.py:
// 50 clusters of width 1e-3, 20 000 x 512, k = 5: 283 ms against 7.5 ms for the exact kernels - 34 000 rows overflowed their
// candidate buffers and went through the row-at-a-time fix-up).  Two device-side checks, no host round trip:
//   A  after the sample pass: a row whose smallest SAMPLED values (the whole list: 6 or 11) span less than the error band
//      2 E_i cannot be separated; if more than one row in eight is like that the sweep, scatter and prune kernels return at once;
//   B  after the prune step: more surviving pairs than max(8 (k+1) + 16, N / 160) per row, or more than N / 8 rows sent to
//      the fix-up.
// In either case verification, selection and fix-up return at once and the exact general kernel - always launched behind
// them, like the exact membership kernel behind its filter - really runs.  Same outputs bit for bit either way (both are
// the exact kernels' values); the checks only choose the cheaper route.  Single-GPU form only: the partitioned entry points
// exchange bounds and lists between ranks and have no common place for the decision.
template <int KCAP>
__global__ void __launch_bounds__(256) knn_fast_predict_kernel(const float* __restrict__ partial, int64_t N, int nchunks, int k1,
                                                               const float* __restrict__ xnorm, const unsigned* __restrict__ maxn,
                                                               float fc, int* __restrict__ gate, unsigned char* __restrict__ flat_rows) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool flat = false;
    if (i < N) {
        float m[KCAP];
#pragma unroll
        for (int s = 0; s < KCAP; ++s) m[s] = partial[i * KCAP + s];
        for (int c = 1; c < nchunks; ++c) {
            const float* src = partial + ((int64_t)c * N + i) * KCAP;
            for (int s = 0; s < KCAP; ++s) list_insert<KCAP>(m, src[s]);
        }
        // The WHOLE list, whatever k: m[0] is the row's own column when the sample holds it - a zero; from m[1] on the values
        // are neighbours either way.  KCAP - 1 sampled neighbours inside one band means 16 times as many in the whole row.
        // (The span up to the (k+1)-th value only - a single spacing of the order statistics for k = 2 - misfired on
        // well-separated low-dimensional sets.)
        (void)k1;
        const float band = 2.f * fc * (xnorm[i] + __uint_as_float(maxn[0]));
        flat = m[KCAP - 1] < INFINITY && m[KCAP - 1] - m[1] <= band;
        // The per-row verdict (knn_fast_mask_flat_kernel) asks for much more: a span of a sixteenth of the band - identical rows
        // give identical approximate values, span zero - because a chance coincidence of five or ten order statistics within
        // the whole band happens to one row in a few hundred of ordinary data (0.3 - 0.7 % of randn / unit-norm rows: measured),
        // and every masked row costs a row of the exact kernel.
        if (i < N) flat_rows[i] = (flat && m[KCAP - 1] - m[1] <= 0.0625f * band) ? 1 : 0;
    }
    const unsigned long long b = __ballot(flat);
    if ((threadIdx.x & 63) == 0 && b != 0ull) atomicAdd(gate + 2, __popcll(b));
}

// Rows the sample cannot separate (fewer than one in eight, or check A would have taken the whole call): taken out of the
// sweep - bound -inf: nothing is queued FOR them, while they stay everybody else's candidates - and marked as overflowed, so
// that selection hands them to the batched fix-up.  A block of identical rows (silence in a stem dataset) larger than a
// row's candidate buffer used to flood the queues: 10 000 duplicates among 100 000 rows took 90 ms (whole-call fallback),
// now the sweep plus 10 000 rows of the exact general kernel.
__global__ void __launch_bounds__(256) knn_fast_mask_flat_kernel(const unsigned char* __restrict__ flat_rows, const int* __restrict__ gate,
                                                                 float* __restrict__ thr, int* __restrict__ cnt, int64_t N) {
    if (gate[0] != 0) return;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < N && flat_rows[i]) {
        thr[i] = -INFINITY;
        cnt[i] = FAST_ROW_OVERFLOW;
    }
}

__global__ void knn_fast_decide_kernel(int* __restrict__ gate, int64_t N, int k1, int stage, const int* __restrict__ pair_count,
                                       const unsigned* __restrict__ maxn, int* __restrict__ ov_count) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (stage == 0) {
        gate[0] = (int64_t)gate[2] * 8 > N ? 1 : 0;
    } else if (stage == 1) {
        // an exactly verified pair costs about as much as 80 pairs of the exact kernel, which looks at N pairs per row: the
        // verification is worth it up to N / 160 survivors per row (half the exact kernel's time) - 625 at 100 000 rows, where a
        // block of 100 identical low-norm rows makes every row verify 100 tied neighbours (tools/dups_probe.py) - and never
        // fewer than 8 (k+1) + 16
        const int64_t per_row = N / 160 > (int64_t)(8 * k1 + 16) ? N / 160 : (int64_t)(8 * k1 + 16);
        const bool volume = (int64_t)*pair_count > N * per_row;
        const bool fixups = (int64_t)gate[3] * 8 > N;          // (more than the batched fix-up's copy holds)
        // (a scale outside the f16 range sends every row to the fix-up as well: the exact kernel is the better fix-up)
        gate[1] = (gate[0] != 0 || volume || fixups || !half_scale_ok(maxn[2])) ? 1 : 0;
    } else if (gate[1] != 0) {
        *ov_count = (int)(N < 0x7fffffff ? N : 0x7fffffff);     // statistics: every row took the exact kernel
    }
}

static bool knn_fast_enabled(int64_t N, int D) {
    static const int on = env_int("AM_KNN_FAST", 1);
    // Below: the exact kernels are faster.  Measured in round 3 (tools/size_sweep.py, k = 5, cold PRDC of two sets): the f16
    // filter sweep wins from ~6000 rows at D = 512 (1.01 -> 0.74 ms at 6000, 3.6 -> 1.5 ms at 16 000, 11.2 -> 3.7 ms at 32 000)
    // and from ~8000 rows at D = 128 (0.80 -> 0.71 ms at 8192, 4.0 -> 2.1 ms at 32 000).  Rounds 1-2 switched at 32 768 rows, a
    // threshold measured on the 128-row engine before the 256-row engine, the queue pruning and the prepared sets existed:
    // the sizes most evaluations have (8k - 32k clips) ran up to 3x slower than necessary.
    // Narrow rows (32 <= D < 128: what n_pca leaves) were excluded altogether ("the saved MFMA work scales with D"): from
    // 16 384 rows the sweep wins there too - 8.3 -> 2.9 ms at 50 000 x 64, 29.2 -> 7.8 ms at 100 000 x 64, break-even at 16 000.
    static const int min_rows_env = env_int("AM_KNN_FAST_MIN_ROWS", 0);
    // D < 32 (round 4; what n_pca = 8 ... 24 leaves: the reference's own tests project to 8 - 10 components): excluded until
    // then for no better reason than the 32-wide k-slab of the 128-row engine - the f16 copy pads a row to 64 elements anyway
    // and the bound holds for any D.  25.7 -> 8.2 ms at 100 000 x 8, 7.2 -> 3.1 ms at 50 000 x 16, break-even at ~14 000 rows
    // (profiles/r4/threshold_sweep_narrow.txt; bit-identical to the exact kernels on randn and unit-norm sets).
    const int64_t min_rows = min_rows_env > 0 ? min_rows_env : (D >= 256 ? 6144 : D >= 128 ? 8192 : D >= 32 ? 12000 : 16384);
    return on != 0 && N >= min_rows && D >= FAST_MIN_DIM && D <= FAST_MAX_DIM && N < ((int64_t)1 << 31);
}

struct KnnFastBuffers {           // on top of the symmetric path's KnnBuffers
    uint16_t* xb;                 // scaled f16 copy
    unsigned* maxn;               // [4] largest squared norm / largest |element|
    float* wgv;                   // approximate values of the queue entries
    unsigned* fidx;               // partner of each filed entry
    int *cnt2, *pair_count;
    uint2* ovq;                   // spill queue (entries past a region's capacity)
    float* ovv;
    unsigned long long* ovn;      // entries offered to it (64-bit: adversarial inputs offer billions)
    int* gate;                    // [8] data-dependent fallback: [0] skip the sweep, [1] skip verification / run the exact
                                  //     kernel, [2] rows the sample cannot separate, [3] rows the prune step sent to fix-up;
                                  //     [4] next free queue region of a partitioned run
    float* xpartial;              // partial lists of the gated exact kernel (choose_chunks(N, N) x N x kcap)
    unsigned char* flat_rows;     // [N] rows the sample cannot separate (check A's per-row verdict)
    KnnFixup fix;                 // batched fix-up of the overflowed rows
};
constexpr int KNN_FAST_OVCAP = 1 << 22;

static KnnFastBuffers carve_knn_fast(Carver& c, int64_t N, int D, const KnnPlan& p) {
    KnnFastBuffers f;
    f.xb = c.take<uint16_t>((size_t)N * half_ld(D));
    f.maxn = c.take<unsigned>(4);
    f.wgv = c.take<float>((size_t)p.nwg * p.qcap);
    f.fidx = c.take<unsigned>((size_t)N * p.cap);
    f.cnt2 = c.take<int>(N + 2);                  // [N] = pair counter
    f.pair_count = f.cnt2 ? f.cnt2 + N : nullptr;
    f.ovn = c.take<unsigned long long>(1);
    f.ovq = c.take<uint2>(KNN_FAST_OVCAP);
    f.ovv = c.take<float>(KNN_FAST_OVCAP);
    f.gate = c.take<int>(8);
    f.xpartial = c.take<float>((size_t)choose_chunks(N, N) * N * p.kcap);
    f.flat_rows = c.take<unsigned char>((size_t)N);
    f.fix = carve_knn_fixup(c, N, D, p.kcap);
    return f;
}

// X: N x D f32 (exact verification); plan / buffers of the symmetric path + the filter path's own.
// bounds_in != nullptr: upper bounds B_i of the rows' final values already computed (partitioned form:
// am_knn_bounds_f32 of every rank, all-gathered) -> filter bound B_i + E_i;
// out_lists != nullptr: partitioned form, emit per-row lists instead of radii.
// chunks per row block of the wide sample pass: the split of a row block's `samples` sampled column tiles that leaves the
// shortest schedule on 256 CUs (rounds of workgroups x tiles per workgroup)
static int wide_sample_chunks(int64_t row_blocks, int64_t samples) {
    int chunks = 1;
    int64_t best_cost = INT64_MAX;
    for (int c = 1; c <= 8 && c <= samples; ++c) {
        const int64_t cost = ceil_div(row_blocks * c, 256) * ceil_div(samples, c);
        if (cost < best_cost) {
            best_cost = cost;
            chunks = c;
        }
    }
    return chunks;
}

template <int KCAP>
static int run_knn_fast(const float* X, int64_t N, int64_t ld, int D, int k1, const KnnPlan& p, const KnnBuffers& b,
                        const KnnFastBuffers& f, float* out_r, hipStream_t st, int part = 0, int nparts = 1,
                        const float* bounds_in = nullptr, float* out_lists = nullptr, const PreparedSet* prep = nullptr,
                        const Knn64Hook* h64 = nullptr) {
    int rc;
    // (float64 route: X is the float32-rounded copy of h64->X; the rounding is priced into the bound, see Knn64Hook)
    const float fc = h64 != nullptr ? fast_c(D) + 1.9073486328125e-06f : fast_c(D);
    const int64_t ldh = half_ld(D) / 2;                              // row stride of the f16 copy in f32 words
    const int Dh = (int)ldh;
    const float* Xb = reinterpret_cast<const float*>(prep != nullptr ? prep->half : f.xb);
    unsigned* maxn = f.maxn;
    float* thr = b.thr;
    // data-dependent fallback to the exact general kernel (knn_fast_predict_kernel): the single-GPU form only
    static const int gate_on = env_int("AM_KNN_FAST_GATE", 1);
    int* gate = (gate_on != 0 && bounds_in == nullptr && out_lists == nullptr) ? f.gate : nullptr;
    const int* skip_sweep = gate;
    const int* skip_verify = gate != nullptr ? gate + 1 : nullptr;
    if (gate != nullptr) AM_HIP_TRY(hipMemsetAsync(gate, 0, 4 * sizeof(int), st));
    AM_HIP_TRY(hipMemsetAsync(maxn, 0, 4 * sizeof(unsigned), st));
    if (prep != nullptr) {
        hipLaunchKernelGGL(prepared_stats_kernel, dim3(1), dim3(64), 0, st, prep->stats, maxn, 0, 2, 3);
    } else {
        if ((rc = launch_to_half(X, N, ld, D, b.xn, maxn, 0, f.xb, st)) != AM_OK) return rc;
        AM_HIP_TRY(hipMemcpyAsync(maxn + 3, maxn + 2, sizeof(unsigned), hipMemcpyDeviceToDevice, st));   // both operands are X
    }
    // 1) filter bounds for every row from a sampled f16 pass of the general kernel (here a row's OWN entries are
    //    queued too, so - unlike in the exact symmetric kernel - every row needs a bound from the start)
    if (bounds_in != nullptr) {
        hipLaunchKernelGGL(knn_fast_bound_kernel, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, st, bounds_in, thr, b.xn, N, maxn, fc);
        AM_LAUNCH_CHECK();
    } else {
        int sample_chunks = p.pre_chunks;
        if (p.tile_rows == WIDE_TILE_ROWS) {              // the sample pass on the 256-row engine: every pre_stride-th 256-row tile
            const int64_t rb = ceil_div(N, WIDE_TILE_ROWS), samples = ceil_div(rb, p.pre_stride);
            sample_chunks = wide_sample_chunks(rb, samples);
            if ((rc = launch_knn_wide_sample(KCAP, Xb, N, ldh, b.xn, Dh, p.pre_stride, sample_chunks, maxn, b.partial, 0, N, st)) != AM_OK)
                return rc;
        } else if ((rc = launch_knn_vt<KCAP, EV_FAST, false>(Xb, N, ldh, b.xn, Xb, N, ldh, b.xn, Dh, p.pre_chunks, p.pre_stride,
                                                             b.partial, st, maxn)) != AM_OK) {
            return rc;
        }
        hipLaunchKernelGGL(knn_merge_kernel<KCAP>, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, st, b.partial, N, sample_chunks,
                           k1, 1, thr, static_cast<const int*>(nullptr));
        if (gate != nullptr) {                           // check A: can the f16 values separate the rows' neighbours at all?
            hipLaunchKernelGGL(knn_fast_predict_kernel<KCAP>, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, st, b.partial, N,
                               sample_chunks, k1, b.xn, maxn, fc, gate, f.flat_rows);
            hipLaunchKernelGGL(knn_fast_decide_kernel, dim3(1), dim3(64), 0, st, gate, N, k1, 0, f.pair_count, maxn, b.ov_count);
        }
        hipLaunchKernelGGL(knn_fast_bound_kernel, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, st, thr, thr, b.xn, N, maxn, 2.f * fc);
        AM_LAUNCH_CHECK();
    }
    AM_HIP_TRY(hipMemsetAsync(b.cnt, 0, (size_t)(N + 1) * sizeof(int), st));
    AM_HIP_TRY(hipMemsetAsync(f.cnt2, 0, (size_t)(N + 2) * sizeof(int), st));
    if (gate != nullptr) {
        hipLaunchKernelGGL(knn_fast_mask_flat_kernel, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, st, f.flat_rows, gate, thr, b.cnt, N);
        AM_LAUNCH_CHECK();
    }
    AM_HIP_TRY(hipMemsetAsync(f.ovn, 0, sizeof(unsigned long long), st));
    // 2) symmetric filter sweep
    const unsigned nwg = (unsigned)p.nwg;
    const int64_t nlist = (int64_t)p.nwin * N * KCAP;
    hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)ceil_div(nlist, 256)), dim3(256), 0, st,
                       reinterpret_cast<unsigned*>(b.partial), nlist, 0x7f800000u);
    AM_LAUNCH_CHECK();
    {
        AM_HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(&knn_fast_kernel<KCAP>), (int)PAIRWISE_LDS_BYTES + 16));
    }
    int qcap = p.qcap;                                                     // per workgroup: nsub private sub-regions + a shared part
    const int nsub = p.tile_rows == WIDE_TILE_ROWS ? 8 : 4;                  // KnnFastEpilogue::NWAVES of the engine that runs
    unsigned nregions = nwg;
    int* region_counter = nullptr;
    if (nparts > 1) {
        // a rank's workgroups share ALL the region memory: as many regions as it has active workgroups, handed out in
        // arrival order (see the kernels); every region nparts-ish times the one-GPU size
        const int64_t active = std::max<int64_t>(1, sym_active_items(ceil_div(N, p.tile_rows), p.win_tiles, p.nwin, p.per_win, part, nparts));
        const int64_t grown = std::min<int64_t>((int64_t)nwg * qcap / active, 1 << 18) / 16 * 16;
        qcap = (int)std::max<int64_t>(grown, qcap);
        nregions = (unsigned)std::min<int64_t>(active, (int64_t)nwg * p.qcap / qcap);
        region_counter = f.gate + 4;
        AM_HIP_TRY(hipMemsetAsync(region_counter, 0, sizeof(int), st));
        AM_HIP_TRY(hipMemsetAsync(b.wgq_count, 0, (size_t)nregions * (nsub + 1) * sizeof(int), st));
    }
    const int wcap = qcap / (2 * nsub);
    const unsigned nreg = nregions * (unsigned)(nsub + 1);                  // queue parts in all
    static const int ovcap = std::max(0, std::min(env_int("AM_KNN_FAST_OVCAP", KNN_FAST_OVCAP), KNN_FAST_OVCAP));   // (tests shrink it)
    clock_begin(AM_KERNEL_KNN, st);
    if (p.tile_rows == WIDE_TILE_ROWS) {
        const auto launch_sweep = wide_stationary(Dh) ? &launch_knn_pstat : &launch_knn_wide;
        if ((rc = launch_sweep(KCAP, nwg, Xb, N, ldh, b.xn, thr, Dh, p.win_tiles, p.nwin, p.per_win, k1, maxn, b.partial, b.cnt,
                                  p.cap, b.wgq, f.wgv, qcap, b.wgq_count, part, nparts, fc, f.ovq, f.ovv, f.ovn, ovcap,
                                  skip_sweep, region_counter, st)) != AM_OK)
            return rc;
    } else {
        hipLaunchKernelGGL(knn_fast_kernel<KCAP>, dim3(nwg), dim3(ENGINE_THREADS), PAIRWISE_LDS_BYTES + 16, st, Xb, N, ldh, b.xn,
                           thr, Dh, p.win_tiles, p.nwin, p.per_win, k1, maxn, b.partial, b.cnt, p.cap, b.wgq, f.wgv, qcap,
                           b.wgq_count, part, nparts, fc, f.ovq, f.ovv, f.ovn, ovcap, skip_sweep, region_counter);
    }
    clock_end(AM_KERNEL_KNN, st);
    AM_LAUNCH_CHECK();
    // 3) approximate values filed by row, 4) pruned against the row's own (k+1)-th smallest, 5) exact values of the
    //    survivors, 6) selection, 7) exact fix-up of overflowed rows
    hipLaunchKernelGGL(knn_fast_scatter_kernel, dim3(nreg), dim3(256), 0, st, b.wgq, f.wgv, qcap, wcap, nsub, b.wgq_count, b.cand,
                       f.fidx, b.cnt, p.cap, thr, skip_sweep);
    hipLaunchKernelGGL(knn_fast_scatter_spill_kernel, dim3(256), dim3(256), 0, st, f.ovq, f.ovv, f.ovn, ovcap, b.cand, f.fidx,
                       b.cnt, p.cap, thr, skip_sweep);
    AM_LAUNCH_CHECK();
    const int64_t pair_cap64 = std::min<int64_t>((int64_t)nwg * p.qcap, (int64_t)1 << 30);   // (the pair list reuses the region memory)
    const int pair_cap = (int)pair_cap64;
    hipLaunchKernelGGL(knn_fast_prune_kernel<KCAP>, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, st, b.cand, f.fidx, b.cnt, p.cap,
                       N, k1, b.xn, maxn, b.wgq, f.wgv, pair_cap, f.pair_count, f.cnt2, out_lists != nullptr ? 1 : 0, fc, gate,
                       h64 != nullptr ? h64->pair_start : nullptr, h64 != nullptr ? h64->pair_n : nullptr);
    if (gate != nullptr)                                 // check B: did the prune step leave a verifiable amount of work?
        hipLaunchKernelGGL(knn_fast_decide_kernel, dim3(1), dim3(64), 0, st, gate, N, k1, 1, f.pair_count, maxn, b.ov_count);
    AM_LAUNCH_CHECK();
    if (h64 != nullptr) {
        AM_REQUIRE(gate != nullptr && out_lists == nullptr, AM_ERR_BAD_ARG, "the float64 route is the single-GPU form");
        clock_begin(AM_KERNEL_KNN_VERIFY, st);
        long long* stats64 = filter_stats_for_current_device();
        hipLaunchKernelGGL(knn_fast_select64_kernel, dim3((unsigned)ceil_div(N, 4 * KNN64_ROWS_PER_WAVE)), dim3(256), 0, st, h64->X, h64->ld, D, b.wgq,
                           h64->pair_start, h64->pair_n, N, k1, maxn, h64->out_r, gate, (const float*)f.wgv, (const float*)b.xn, fc,
                           stats64 != nullptr ? reinterpret_cast<unsigned long long*>(stats64) + 9 : nullptr);
        clock_end(AM_KERNEL_KNN_VERIFY, st);
        AM_LAUNCH_CHECK();
        // the general f64 kernels behind the route: they return at once unless check A / B or a row without a list gave up
        if ((rc = knn64_self_gated(h64->X, N, h64->ld, D, h64->k, h64->out_r, h64->gen_ws, h64->gen_ws_bytes, gate + 1, st)) != AM_OK) return rc;
        hipLaunchKernelGGL(knn_fast_decide_kernel, dim3(1), dim3(64), 0, st, gate, N, k1, 2, f.pair_count, maxn, b.ov_count);
        AM_LAUNCH_CHECK();
        if (long long* stats = filter_stats_for_current_device()) {
            hipLaunchKernelGGL(filter_stats_kernel, dim3(1), dim3(256), 0, st, b.wgq_count, (int64_t)nreg, stats, 0, 1,
                               (const unsigned long long*)f.ovn, (const int*)nullptr, ovcap, 2, (const int*)f.pair_count, 3,
                               (const int*)b.ov_count, 4);
            AM_LAUNCH_CHECK();
        }
        return AM_OK;
    }
    clock_begin(AM_KERNEL_KNN_VERIFY, st);
    {
        AM_HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(&knn_fast_verify_kernel), (int)VERIFY_LDS_BYTES));
    }
    {
        long long* stats = filter_stats_for_current_device();      // (the approximate values of the pairs: f.wgv, free since the scatter)
        hipLaunchKernelGGL(knn_fast_verify_kernel, dim3(2048), dim3(256), VERIFY_LDS_BYTES, st, X, ld, b.xn, D, b.wgq, f.pair_count,
                           pair_cap, b.cand, f.cnt2, p.cap, skip_verify, (const float*)f.wgv, fc, (const unsigned*)maxn,
                           stats != nullptr ? reinterpret_cast<unsigned long long*>(stats) + 9 : nullptr);
    }
    clock_end(AM_KERNEL_KNN_VERIFY, st);
    AM_LAUNCH_CHECK();
    hipLaunchKernelGGL(knn_fast_select_kernel<KCAP>, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, st, b.cand, f.cnt2, p.cap, N, k1,
                       maxn, out_r, b.ov_list, b.ov_count, out_lists, skip_verify);
    AM_LAUNCH_CHECK();
    if (out_lists == nullptr) {
        // short lists of overflowed rows: one row at a time; from KNN_FIX_BATCH_FROM rows on: gathered copy + general kernel
        if ((rc = run_knn_fixup<KCAP>(X, N, ld, b.xn, D, k1, b.ov_list, b.ov_count, f.fix, out_r, st)) != AM_OK) return rc;
    }
    if (gate != nullptr) {
        // the exact general kernel behind the filter path: its workgroups return at once unless check A or B gave up
        if ((rc = launch_knn<KCAP>(X, N, ld, b.xn, X, N, ld, b.xn, D, k1, choose_chunks(N, N), 1, false, f.xpartial, out_r, st,
                                   gate + 1)) != AM_OK)
            return rc;
        hipLaunchKernelGGL(knn_fast_decide_kernel, dim3(1), dim3(64), 0, st, gate, N, k1, 2, f.pair_count, maxn, b.ov_count);
        AM_LAUNCH_CHECK();
    }
    if (long long* stats = filter_stats_for_current_device()) {
        hipLaunchKernelGGL(filter_stats_kernel, dim3(1), dim3(256), 0, st, b.wgq_count, (int64_t)nreg, stats, 0, 1,
                           (const unsigned long long*)f.ovn, (const int*)nullptr, ovcap, 2, (const int*)f.pair_count, 3,
                           out_lists == nullptr ? (const int*)b.ov_count : (const int*)nullptr, 4);
        AM_LAUNCH_CHECK();
    }
#ifdef AM_DEV_KNOBS
    static const int debug = env_int("AM_FAST_DEBUG", 0);
    if (debug) {                                       // development aid: synchronises
        AM_HIP_TRY(hipStreamSynchronize(st));
        std::vector<int> wc(nreg), cn(N + 1), c2(N + 2);
        AM_HIP_TRY(hipMemcpy(wc.data(), b.wgq_count, nreg * sizeof(int), hipMemcpyDeviceToHost));
        AM_HIP_TRY(hipMemcpy(cn.data(), b.cnt, (N + 1) * sizeof(int), hipMemcpyDeviceToHost));
        AM_HIP_TRY(hipMemcpy(c2.data(), f.cnt2, (N + 2) * sizeof(int), hipMemcpyDeviceToHost));
        long long tot = 0, full = 0, ctot = 0, bad = 0;
        int wmax = 0, cmax = 0;
        unsigned long long spilled = 0;
        AM_HIP_TRY(hipMemcpy(&spilled, f.ovn, sizeof(spilled), hipMemcpyDeviceToHost));
        for (size_t r = 0; r < wc.size(); ++r) {
            const int v = wc[r], capr = (int)(r % (nsub + 1)) == nsub ? qcap - nsub * wcap : wcap;
            tot += v; full += (v >= capr); wmax = std::max(wmax, v);
        }
        for (int64_t i = 0; i < N; ++i) { ctot += std::min(cn[i], p.cap); cmax = std::max(cmax, cn[i]); bad += c2[i] > p.cap; }
        fprintf(stderr, "[knn_fast] wgs=%u nwin=%d qcap=%d queued=%lld (max/wg %d, full regions %lld) filed=%lld max/row=%d "
                        "pairs verified=%d spilled=%llu rows to fix-up=%lld\n", nwg, p.nwin, qcap, tot, wmax, full, ctot, cmax, c2[N], spilled, bad);
    }
#endif
    return AM_OK;
}

}  // namespace am
