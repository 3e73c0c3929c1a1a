// Filter-and-verify form of the PRDC kernels (included by pairwise.hip).
//
// The exact kernels of pairwise.hip spend all their time in f32 MFMAs (157 TF peak, 96 % of the sustained rate
// reached).  Almost none of the N x M distances they compute matter: a pair is relevant only if its squared
// distance lies below a row or column threshold (a k-NN bound, a hypersphere radius, the running row minimum).
// The kernels here find the relevant pairs with a 16x cheaper bf16 MFMA pass and then evaluate exactly those
// pairs with the f32 arithmetic of the exact engine, so the results are BIT-IDENTICAL to pairwise.hip's:
//
//   1. X, Y are rounded to bf16 once (round-to-nearest-even).  For finite normal inputs
//      |bf16(v) - v| <= 2^-8 |v|, hence for the bf16 dot product accumulated in f32
//          |dot'(x,y) - <x,y>| <= (2^-7 + 2^-16) sum_k |x_k y_k| + (f32 accumulation, < 2^-14 |x||y|)
//                              <= (2^-7 + 2^-13) |x| |y|                                   (Cauchy-Schwarz)
//      and the approximate squared distance a = fma(-2, dot', |x|^2 + |y|^2) (same f32 norms and the same
//      rounding of their sum as the exact value t) satisfies
//          |a - t| <= 2 |dot' - dot_f32chain| + ulps <= FAST_C (|x|^2 + |y|^2) =: eps(x, y),
//      FAST_C = 2^-7 + 2^-10 + 2^-12 (the 2^-10 + 2^-12 slack covers the f32 chain's own error, the error of the
//      f32 norms and the rounding of the thresholds below, each < 2^-13 relative).
//   2. A pair is QUEUED when a <= threshold + eps, which every pair with t <(=) threshold satisfies.
//   3. Queued pairs are filed under their row, their exact t is computed with the engine's fmaf order
//      (exact_pair_d2, the chain oracle/exact_c reproduces), and the reductions of the exact kernels are applied
//      to those values.  Rows whose queue or candidate buffer overflowed are recomputed exactly against every
//      column (fix-up kernels), and their queued entries are ignored.
//
// Inputs are assumed finite and below bf16's overflow threshold (3.4e38); f32 denormals are outside the bound
// above only by absolute amounts below 2^-133 |y| and are ignored.
#pragma once

namespace am {

constexpr int EV_FAST = EV_DEFAULT | EV_BF16;
constexpr float FAST_C = 0.0078125f + 0.0009765625f + 0.000244140625f;       // 2^-7 + 2^-10 + 2^-12
constexpr int FAST_LDB_ALIGN = 64;                                            // bf16 row stride: whole 128-B slabs

static inline int64_t bf16_ld(int D) { return (int64_t)(D + FAST_LDB_ALIGN - 1) / FAST_LDB_ALIGN * FAST_LDB_ALIGN; }

// ---- f32 -> bf16 copy (RNE), zero-padded to ldb columns; one thread per 8 elements
__device__ __forceinline__ unsigned bf16_rne(float v) {
    unsigned u = __float_as_uint(v);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}

__global__ void __launch_bounds__(256) to_bf16_kernel(const float* __restrict__ X, int64_t N, int64_t ld, int D, int64_t ldb,
                                                      uint16_t* __restrict__ Xb) {
    const int64_t per_row = ldb / 8;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = idx / per_row;
    if (row >= N) return;
    const int c = (int)(idx % per_row) * 8;
    const f32x4 a = load_k4(X + row * ld, c, D), b = load_k4(X + row * ld, c + 4, D);
    uint4 o;
    o.x = bf16_rne(a.x) | (bf16_rne(a.y) << 16);
    o.y = bf16_rne(a.z) | (bf16_rne(a.w) << 16);
    o.z = bf16_rne(b.x) | (bf16_rne(b.y) << 16);
    o.w = bf16_rne(b.z) | (bf16_rne(b.w) << 16);
    *reinterpret_cast<uint4*>(Xb + row * ldb + c) = o;
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

static int launch_to_bf16(const float* X, int64_t N, int64_t ld, int D, uint16_t* Xb, hipStream_t st) {
    const int64_t ldb = bf16_ld(D);
    const int64_t threads = N * (ldb / 8);
    hipLaunchKernelGGL(to_bf16_kernel, dim3((unsigned)ceil_div(threads, 256)), dim3(256), 0, st, X, N, ld, D, ldb, Xb);
    AM_LAUNCH_CHECK();
    return AM_OK;
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
// With E_i = FAST_C (|r_i|^2 + max_j |c_j|^2) and E'_j = FAST_C (max_i |r_i|^2 + |c_j|^2)  (>= eps of every pair),
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
constexpr unsigned FAST_COUNTED = 0x80000000u;
constexpr int FAST_AUX_FLOATS = 6 * TB;                                    // LDS [2][3][128]
constexpr size_t FAST_LDS_BYTES = (ENGINE_LDS_FLOATS + FAST_AUX_FLOATS) * sizeof(float) + 16;

struct CrossFastEpilogue {
    const float* qnorm;
    const float* qthr;
    int64_t nq;
    float rnmax_c;              // FAST_C * max_i |r_i|^2
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
        } else {                                            // region full: spill to the global queue
            const int s2 = atomicAdd(ov_count, 1);
            if (s2 < ovcap) ovq[s2] = make_uint2((unsigned)i, jflag);
            else *fail = 1;                                 // -> the exact kernel redoes the whole call
        }
    }
    __device__ __forceinline__ void aux_issue(int, int64_t qtile) {
        if (L.tid < TB) {
            const int64_t j = qtile * TB + L.tid;
            if (j < nq) {
                const float e = fmaf(FAST_C, qnorm[j], rnmax_c);
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
                    const float u = fmaf(-2.f, acc[mt][nt][reg], xn[nt] + yn[reg >> 2][reg & 3]);
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
                            const float u = fmaf(-2.f, acc[mt][nt][reg], xn[nt] + yn[reg >> 2][reg & 3]);
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

// Rb / Cb: bf16 copies viewed as f32 words (ld and Dh in words, Dh % 32 == 0).
template <bool PRE, bool WANT_MIN>
__global__ void __launch_bounds__(ENGINE_THREADS, 2)
cross_fast_kernel(const float* __restrict__ Rb, int64_t Nr, int64_t ldr, const float* __restrict__ rnorm,
                  const float* __restrict__ rthr, const float* __restrict__ Cb, int64_t Nc, int64_t ldc,
                  const float* __restrict__ cnorm, const float* __restrict__ cthr, int Dh, int nchunks, int qstride,
                  const unsigned* __restrict__ maxn, unsigned* __restrict__ rmin_approx, unsigned* __restrict__ row_any,
                  unsigned* __restrict__ row_cover, int32_t* __restrict__ col_count, uint2* __restrict__ wgq, int qcap, int* __restrict__ wgq_count,
                  uint2* __restrict__ ovq, int* __restrict__ ov_count, int ovcap, int* __restrict__ fail, int dbg) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const LaneInfo L;
    const int64_t q_tiles = ((Nc + TB - 1) / TB + qstride - 1) / qstride;
    const WorkItem w = work_item(q_tiles, nchunks);
    int* qn = reinterpret_cast<int*>(lds + ENGINE_LDS_FLOATS + FAST_AUX_FLOATS);
    if (L.tid == 0) *qn = 0;

    const float rnmax = __uint_as_float(maxn[0]), cnmax = __uint_as_float(maxn[1]);
    CrossFastEpilogue epi(L);
    epi.qnorm = cnorm;
    epi.qthr = cthr;
    epi.nq = Nc;
    epi.rnmax_c = FAST_C * rnmax;
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
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int64_t i = w.prow0 + L.wn * 64 + nt * 32 + L.r;
        const bool ok = i < Nr;
        epi.prow[nt] = i;
        epi.rowok[nt] = ok;
        epi.xn[nt] = ok ? rnorm[i] : 0.f;
        const float e = FAST_C * ((ok ? rnorm[i] : 0.f) + cnmax);
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

// Verification of one workgroup region (same grid as the filter pass): the entries are bucketed by reference
// row in LDS, then each wave takes rows - the row goes to LDS once, every lane evaluates one candidate with the
// exact engine's fmaf chain - and the exact reductions are applied.
__global__ void __launch_bounds__(256) cross_verify_kernel(const float* __restrict__ R, int64_t Nr, int64_t ldr,
                                                           const float* __restrict__ rnorm, const float* __restrict__ rthr,
                                                           const float* __restrict__ C, int64_t ldc,
                                                           const float* __restrict__ cnorm, const float* __restrict__ cthr, int D,
                                                           int nchunks, const uint2* __restrict__ wgq, int qcap,
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
    const int64_t prow0 = (int64_t)(blockIdx.x / nchunks) * TB;        // work_item(): row block of this region
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
                const float t = fmaxf(fmaf(-2.f, exact_pair_dot(xs, C + j * ldc, D), xi + cnorm[j]), 0.f);
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
        const float t = fmaxf(fmaf(-2.f, acc, rnorm[i] + cnorm[j]), 0.f);
        float mn = INFINITY;
        bool any = false, cov = false;
        cross_apply(t, j, v.y, rthr[i], cthr, col_count, mn, any, cov);
        if (row_min_bits != nullptr) atomicMin(row_min_bits + i, __float_as_uint(mn));
        if (any) atomicOr(row_any + i, 1u);
        if (cov) atomicOr(row_cover + i, 1u);
    }
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
    int nchunks, pre_chunks, qstride, qcap, ovcap;
    int64_t blocks;
};

static CrossFastPlan plan_cross_fast(int64_t Nr, int64_t Nc) {
    CrossFastPlan p;
    p.nchunks = choose_chunks(Nr, Nc);
    p.blocks = ceil_div(Nr, TB) * p.nchunks;
    static const int stride = env_int("AM_FAST_PRE_STRIDE", 16);
    p.qstride = stride;
    const int64_t sample_tiles = ceil_div(ceil_div(Nc, TB), p.qstride);
    p.pre_chunks = (int)std::min<int64_t>(sample_tiles, 8);
    static const int qcap = std::min(env_int("AM_FAST_QCAP", 2048), 2048);       // cross_verify_kernel: <= 256 * 8
    static const int ovcap = env_int("AM_FAST_OVCAP", 1 << 22);
    p.qcap = qcap;
    p.ovcap = ovcap;
    return p;
}

struct CrossFastBuffers {
    uint16_t *rb, *cb;
    unsigned *maxn, *rmin_approx;
    uint2 *wgq, *ovq;
    int *wgq_count, *ov_count;      // ov_count[0] = overflow entries, ov_count[1] = fail flag
};

static CrossFastBuffers carve_cross_fast(Carver& c, int64_t Nr, int64_t Nc, int D, const CrossFastPlan& p) {
    CrossFastBuffers b;
    b.rb = c.take<uint16_t>((size_t)Nr * bf16_ld(D));
    b.cb = c.take<uint16_t>((size_t)Nc * bf16_ld(D));
    b.maxn = c.take<unsigned>(4);
    b.rmin_approx = c.take<unsigned>(Nr);
    b.wgq = c.take<uint2>((size_t)p.blocks * p.qcap);
    b.ovq = c.take<uint2>((size_t)p.ovcap);
    b.wgq_count = c.take<int>(p.blocks);
    b.ov_count = c.take<int>(4);
    return b;
}

static bool cross_fast_enabled(int64_t Nr, int64_t Nc, int D) {
    static const int on = env_int("AM_PRDC_FAST", 1);
    static const int64_t min_pairs = (int64_t)env_int("AM_FAST_MIN_PAIRS_LOG2", 24);
    const size_t verify_lds = (size_t)(4 * ((D + 7) / 8 * 8) + 2048) * sizeof(float);
    return on != 0 && D >= 32 && verify_lds <= 60 * 1024 && Nr * Nc >= ((int64_t)1 << min_pairs) &&
           Nr < ((int64_t)1 << 31) && Nc < ((int64_t)1 << 31);
}

// rn, rt, cn, ct: norms and thresholds already computed; col_count / rmin / rany: the exact kernel's accumulators,
// initialised by the caller (0, +inf bits, 0).  On return `*fail_flag` (device) tells the exact kernel whether it has
// to run after all.
static int run_cross_fast(const float* R, int64_t Nr, int64_t ldr, const float* rn, const float* rt, const float* C, int64_t Nc,
                          int64_t ldc, const float* cn, const float* ct, int D, const CrossFastPlan& p, const CrossFastBuffers& b,
                          int32_t* col_count, unsigned* rmin, unsigned* rany, unsigned* rcov, bool want_min, hipStream_t st) {
    int rc;
    if ((rc = launch_to_bf16(R, Nr, ldr, D, b.rb, st)) != AM_OK) return rc;
    if ((rc = launch_to_bf16(C, Nc, ldc, D, b.cb, st)) != AM_OK) return rc;
    AM_HIP_TRY(hipMemsetAsync(b.maxn, 0, 4 * sizeof(unsigned), st));
    AM_HIP_TRY(hipMemsetAsync(b.ov_count, 0, 4 * sizeof(int), st));
    hipLaunchKernelGGL(max_bits_kernel, dim3(256), dim3(256), 0, st, rn, Nr, b.maxn);
    hipLaunchKernelGGL(max_bits_kernel, dim3(256), dim3(256), 0, st, cn, Nc, b.maxn + 1);
    hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)ceil_div(Nr, 256)), dim3(256), 0, st, b.rmin_approx, Nr, 0x7f800000u);
    AM_LAUNCH_CHECK();
    const int64_t ldb = bf16_ld(D);
    const int Dh = (int)(ldb / 2);
    const float* Rb = reinterpret_cast<const float*>(b.rb);
    const float* Cb = reinterpret_cast<const float*>(b.cb);
    int* fail = b.ov_count + 1;
    static bool attr_done = false;
    if (!attr_done) {
        const void* kernels[] = {reinterpret_cast<const void*>(&cross_fast_kernel<true, true>),
                                 reinterpret_cast<const void*>(&cross_fast_kernel<true, false>),
                                 reinterpret_cast<const void*>(&cross_fast_kernel<false, true>),
                                 reinterpret_cast<const void*>(&cross_fast_kernel<false, false>)};
        for (const void* k : kernels)
            AM_HIP_TRY(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FAST_LDS_BYTES));
        attr_done = true;
    }
    const int dbg = env_int("AM_FAST_DBG", 0);
    // sampled pre-pass over every 16th column tile: certain "any" witnesses (and, when the row minimum is wanted,
    // an approximate minimum that bounds its candidate queue)
    auto launch_filter = [&](auto kernel, unsigned grid, int nchunks, int qstride) {
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(ENGINE_THREADS), FAST_LDS_BYTES, st, Rb, Nr, ldb / 2, rn, rt, Cb, Nc, ldb / 2,
                           cn, ct, Dh, nchunks, qstride, b.maxn, b.rmin_approx, rany, rcov, col_count, b.wgq, p.qcap, b.wgq_count,
                           b.ovq, b.ov_count, p.ovcap, fail, dbg);
    };
    const unsigned pre_grid = (unsigned)(ceil_div(Nr, TB) * p.pre_chunks);
    if (want_min) launch_filter(&cross_fast_kernel<true, true>, pre_grid, p.pre_chunks, p.qstride);
    else launch_filter(&cross_fast_kernel<true, false>, pre_grid, p.pre_chunks, p.qstride);
    AM_LAUNCH_CHECK();
    clock_begin(AM_KERNEL_PRDC_CROSS, st);
    if (want_min) launch_filter(&cross_fast_kernel<false, true>, (unsigned)p.blocks, p.nchunks, 1);
    else launch_filter(&cross_fast_kernel<false, false>, (unsigned)p.blocks, p.nchunks, 1);
    clock_end(AM_KERNEL_PRDC_CROSS, st);
    AM_LAUNCH_CHECK();
    unsigned* rmin_or_null = want_min ? rmin : nullptr;
    const size_t verify_lds = (size_t)(4 * ((D + 7) / 8 * 8) + p.qcap) * sizeof(float);
    hipLaunchKernelGGL(cross_verify_kernel, dim3((unsigned)p.blocks), dim3(256), verify_lds, st, R, Nr, ldr, rn, rt, C, ldc, cn, ct,
                       D, p.nchunks, b.wgq, p.qcap, b.wgq_count, col_count, rmin_or_null, rany, rcov);
    AM_LAUNCH_CHECK();
    hipLaunchKernelGGL(cross_verify_overflow_kernel, dim3(1024), dim3(256), 0, st, R, ldr, rn, rt, C, ldc, cn, ct, D, b.ovq,
                       b.ov_count, p.ovcap, fail, col_count, rmin_or_null, rany, rcov);
    AM_LAUNCH_CHECK();
    hipLaunchKernelGGL(cross_fail_reset_kernel, dim3(256), dim3(256), 0, st, fail, col_count, Nc, rmin, rany, rcov, Nr);
    AM_LAUNCH_CHECK();
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
    return AM_OK;
}

}  // namespace am
