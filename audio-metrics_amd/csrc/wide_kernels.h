// Bodies of the two 256-row f16 FILTER kernels of the PRDC path - the membership filter and the symmetric k-NN sweep -
// written once over an ENGINE policy (which pipeline multiplies the tiles, what its lanes own):
//   WideEng          wide_engine.h   both operands streamed through LDS, a wave owns 128 Q x 64 P rows   (pairwise_wide.hip)
//   PstatEng<KSLABS> pstat_engine.h  P block stationary in registers, a wave owns 128 Q x 32 P rows      (pairwise_pstat.hip)
// Algorithms, error bound and queue protocol are documented in pairwise_fast.h, which holds the 128-row forms of the same
// filters and everything that runs around these kernels.
#pragma once
#include "pairwise_common.h"
#include "wide_engine.h"
#include <algorithm>

namespace am {

// ---- the same filter on the 256 x 256 f16 engine (wide_engine.h): main pass only ---------------------------------
constexpr int WIDE_AUX_WORDS = 6 * WTB;                                    // LDS [2][3][256]: what finish() reads
constexpr int WIDE_RAW_WORDS = 4 * WTB;                                    // LDS [2][2][256]: DMA landing zone of the side data
template <class Eng> constexpr size_t wide_cross_lds_bytes() { return (Eng::LDS_WORDS + WIDE_AUX_WORDS + WIDE_RAW_WORDS) * sizeof(float) + 16; }

template <class Lane>
struct CrossWideEpilogue {
    static constexpr int NT = Lane::NT;
    static constexpr int MT = Lane::MT;               // 32-row Q tiles of a finish() call: L.wm counts units of MT * 32 columns
    // ACC_INIT (pstat_engine.h; the scheme of KnnFastEpilogue): accumulators start at |c_j|^2 / dsc, an element holds
    // a' = <r_i, c_j>' + |c_j|^2 / dsc and the approximate squared distance is dsc a' + |r_i|^2.  Row direction (thresholds of
    // the lane's own row): fma(dsc, max_j a'_j, |r_i|^2) against them; "any" direction (column thresholds T'_j + E'_j):
    // a'_j - (T'_j + E'_j) / dsc >= -|r_i|^2 / dsc.  aux[0] and aux[1] are then in accumulator units, aux[2] (certain witness)
    // stays a squared distance.
    static constexpr bool ACC_INIT = Lane::ACC_INIT;
    const float* qnorm;
    const float* qthr;
    int64_t nq;
    float fc, rnmax_c;
    float* aux;                 // LDS [2][3][256] : |c_j|^2, T'_j + E'_j, T'_j - E'_j of the tile
    int32_t* col_count;
    uint2* wgq;
    int* qn;
    int qcap;
    uint2* ovq;
    int* ov_count;
    int ovcap;
    int* fail;
    float dsc, idsc;
    int64_t prow[NT];
    float xn[NT], xs[NT], xsn[NT], thi[NT], tlo[NT], e2[NT], m[NT];   // xsn (ACC_INIT): xs while the row still lacks its witness, +inf after
    float athr[NT];             // (not ACC_INIT) what an "any" margin is compared with: 0 while the row lacks its witness, -inf after
    bool rowok[NT], anyf[NT], covf[NT];
    float aux_n, aux_hi;
    const Lane& L;

    __device__ __forceinline__ CrossWideEpilogue(const Lane& l) : L(l) {}
    __device__ __forceinline__ void push(int64_t i, unsigned jflag) {
        const int slot = atomicAdd(qn, 1);
        if (slot < qcap) {
            wgq[slot] = make_uint2((unsigned)i, jflag);
        } else if (*reinterpret_cast<volatile int*>(fail) == 0) {   // (see CrossFastEpilogue::push)
            const unsigned s2 = atomicAdd(reinterpret_cast<unsigned*>(ov_count), 1u);
            if (s2 < (unsigned)ovcap) ovq[s2] = make_uint2((unsigned)i, jflag);
            else *fail = 1;
        }
    }
    // loads only: the values are first used in aux_commit, after the stage's MFMAs (an arithmetic use here would park
    // waves 0-3 on a full memory round trip at the start of the last stage of every tile)
    __device__ __forceinline__ void aux_issue(int, int64_t qtile) {
        if (L.tid < WTB) {
            const int64_t j = qtile * WTB + L.tid;
            const bool in = j < nq;
            aux_n = in ? qnorm[j] : INFINITY;               // a = +inf: never below anything
            aux_hi = in ? qthr[j] : -INFINITY;
        }
    }
    __device__ __forceinline__ void aux_commit(int t) {
        if (L.tid < WTB) {
            float* d = aux + (t & 1) * 3 * WTB + L.tid;
            const bool in = aux_hi > -INFINITY;               // thresholds are >= 0; -inf marks a column past the end
            const float e = fmaf(fc, aux_n, rnmax_c);
            d[0] = ACC_INIT ? aux_n * idsc : aux_n;                      // (past the end: +inf -> -inf, never the maximum)
            d[WTB] = in ? (ACC_INIT ? (aux_hi + e) * idsc : aux_hi + e) : (ACC_INIT ? INFINITY : -INFINITY);
            d[2 * WTB] = in ? aux_hi - e : -INFINITY;
        }
    }
    // NEED_ANY = false: every row of this block already has its "any" witness - the column-threshold test (one subtraction
    // and half a min3 per accumulator element, and the threshold loads) is not compiled in
    __device__ __forceinline__ f32x16 acc_init(int t, int mt) const {
        const float* a = aux + (t & 1) * 3 * WTB + L.wm * (MT * 32) + L.h * 4 + mt * 32;
        f32x16 c;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(a + g4 * 8);
            c[g4 * 4 + 0] = v.x; c[g4 * 4 + 1] = v.y; c[g4 * 4 + 2] = v.z; c[g4 * 4 + 3] = v.w;
        }
        return c;
    }
    template <bool WANT_MIN, bool NEED_ANY>
    __device__ __forceinline__ void finish_impl(int t, int64_t qtile, f32x16 (&acc)[MT][NT]) {
        const float* a = aux + (t & 1) * 3 * WTB + L.wm * (MT * 32) + L.h * 4;
        const int64_t jbase = qtile * WTB + L.wm * (MT * 32) + L.h * 4;
#ifdef AM_DEV_KNOBS
        if (g_wide_dbg & 2) return;                            // timing experiment: MFMA pipeline only
#endif
        if constexpr (ACC_INIT) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x4 ths[4];
                if constexpr (NEED_ANY) {
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) ths[g4] = *reinterpret_cast<const f32x4*>(a + WTB + mt * 32 + g4 * 8);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    // Gates as lane masks on the scalar unit (lanes_le / lanes_ge, pairwise_common.h).  No row-validity or
                    // "still needs a witness" masks beside them: a row past the end carries thresholds of -inf (thi, tlo, m)
                    // and xsn = +inf, a row that has its witness xsn = +inf - their compares fail by themselves (combining
                    // masks costs a chain of dependent scalar instructions per gate; measured, pairwise_common.h).
                    float amax4[4], wmax4[4];
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        float am = -INFINITY, wmx = -INFINITY;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            am = fmaxf(am, acc[mt][nt][g4 * 4 + e]);
                            if constexpr (NEED_ANY) wmx = fmaxf(wmx, acc[mt][nt][g4 * 4 + e] - ths[g4][e]);
                        }
                        amax4[g4] = am;
                        wmax4[g4] = wmx;
                    }
                    const float tmin = fmaf(dsc, fmaxf(fmaxf(amax4[0], amax4[1]), fmaxf(amax4[2], amax4[3])), xn[nt]);
                    if constexpr (WANT_MIN) m[nt] = fminf(m[nt], fmaxf(tmin, 0.f));
                    const float prow_thr = WANT_MIN ? fmaxf(thi[nt], m[nt] + e2[nt]) : thi[nt];
                    {
                        unsigned long long hit = lanes_le(tmin, prow_thr);
                        if constexpr (NEED_ANY) hit |= lanes_ge(fmaxf(fmaxf(wmax4[0], wmax4[1]), fmaxf(wmax4[2], wmax4[3])), xsn[nt]);
                        if (hit == 0ull) continue;
#ifdef AM_DEV_KNOBS
                        if (g_wide_dbg & 16) continue;
#endif
                    }
                    const float* alo = a + 2 * WTB + mt * 32;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const bool row_hit = lanes_le(fmaf(dsc, amax4[g4], xn[nt]), prow_thr) != 0ull;
                        bool any_hit = false;
                        if constexpr (NEED_ANY) any_hit = lanes_ge(wmax4[g4], xsn[nt]) != 0ull;
#ifdef AM_DEV_KNOBS
                        if (g_wide_dbg & 8) any_hit = false;
#endif
                        if (!(row_hit || any_hit)) continue;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int reg = g4 * 4 + e;
                            const float u = fmaf(dsc, acc[mt][nt][reg], xn[nt]);
                            const int64_t j = jbase + mt * 32 + g4 * 8 + e;
                            bool sure = false, want = false;
                            if (row_hit) {
                                const unsigned long long mask = lanes_lt(u, tlo[nt]);
                                sure = u < tlo[nt];
                                if (mask != 0ull && L.lane == 0) {
                                    const int lo = __popcll(mask & 0xffffffffull);
                                    const int hi = __popcll(mask >> 32);
                                    if (lo) atomicAdd(col_count + j - L.h * 4, lo);
                                    if (hi) atomicAdd(col_count + j - L.h * 4 + 4, hi);
                                }
                                covf[nt] = covf[nt] || sure;
                                want = !sure && u <= thi[nt];
                                if constexpr (WANT_MIN) want = want || u <= m[nt] + e2[nt];
                            }
                            if constexpr (NEED_ANY) {
                                if (acc[mt][nt][reg] - ths[g4][e] >= xsn[nt]) {
                                    if (u < alo[g4 * 8 + e]) {                      // certain witness
                                        anyf[nt] = true;
                                        xsn[nt] = INFINITY;
                                    } else {
                                        want = true;                                // ambiguous "any"
                                    }
                                }
                            }
                            if (want) push(prow[nt], (unsigned)j | (sure ? FAST_COUNTED : 0u));
                        }
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x4 yn[4], th[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                yn[g4] = *reinterpret_cast<const f32x4*>(a + mt * 32 + g4 * 8);
                if constexpr (NEED_ANY) th[g4] = *reinterpret_cast<const f32x4*>(a + WTB + mt * 32 + g4 * 8);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                // Fast path: per group of four accumulator registers (four columns), the smallest value and the smallest margin
                // against the column thresholds.  The row's own norm is added to the minima, not to every element (the extra
                // rounding is one of those fast_c's 2^-19 term pays for).  (Comparing a group's minimum with the LARGEST of its
                // four column thresholds - one subtraction per group - was measured: the thresholds of neighbouring columns
                // differ by more than the distance distribution allows this far out in its tail, four in five groups pass
                // such a gate, 10.6 -> 13.0 ms.)
                float tmin4[4], marg4[4];
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    float tm = INFINITY, mg = INFINITY;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = fmaf(dsc, acc[mt][nt][g4 * 4 + e], yn[g4][e]);
                        tm = fminf(tm, t);
                        if constexpr (NEED_ANY) mg = fminf(mg, t - th[g4][e]);
                    }
                    tmin4[g4] = tm + xn[nt];
                    if constexpr (NEED_ANY) marg4[g4] = mg + xn[nt];
                }
                const float tmin = fminf(fminf(tmin4[0], tmin4[1]), fminf(tmin4[2], tmin4[3]));
                if constexpr (WANT_MIN) m[nt] = fminf(m[nt], fmaxf(tmin, 0.f));
                const float prow_thr = WANT_MIN ? fmaxf(thi[nt], m[nt] + e2[nt]) : thi[nt];
                // Gate 1, one wave-uniform branch per accumulator tile (1024 pairs).  The row direction (column counts,
                // coverage, row minimum) has a candidate in 0.2 % of the 32 x 32 tiles of the bench problem, the "any"
                // direction of the rows that still lack a witness in 16 % - the other tiles used to pay four group gates each.
                {
                    // (lane masks on the scalar unit; a row past the end or with its witness fails through its thresholds:
                    // thi, tlo, m = -inf, athr = -inf - see the ACC_INIT form above)
                    unsigned long long hit = lanes_le(tmin, prow_thr);
                    if constexpr (NEED_ANY) hit |= lanes_le(fminf(fminf(marg4[0], marg4[1]), fminf(marg4[2], marg4[3])), athr[nt]);
                    if (hit == 0ull) continue;
#ifdef AM_DEV_KNOBS
                    if (g_wide_dbg & 16) continue;                 // timing experiment: fast path and gate only, no detail path
#endif
                }
                // Detail path, per register group and direction, behind wave-uniform gates - those groups run without ballots
                // and counts.  (One loop with switches, not two loops: a second unrolled copy pushes the epilogue over the
                // compiler's unroll budget and the accumulator array into scratch.)
                const float* alo = a + 2 * WTB + mt * 32;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const bool row_hit = lanes_le(tmin4[g4], prow_thr) != 0ull;
                    bool any_hit = false;
                    if constexpr (NEED_ANY) any_hit = lanes_le(marg4[g4], athr[nt]) != 0ull;
#ifdef AM_DEV_KNOBS
                    if (g_wide_dbg & 8) any_hit = false;               // timing experiment: the gate's cost without its loop
#endif
                    if (!(row_hit || any_hit)) continue;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int reg = g4 * 4 + e;
                        const float u = fmaf(dsc, acc[mt][nt][reg], xn[nt] + yn[g4][e]);
                        const int64_t j = jbase + mt * 32 + g4 * 8 + e;
                        bool sure = false, want = false;
                        if (row_hit) {
                            sure = u < tlo[nt];
                            const unsigned long long mask = lanes_lt(u, tlo[nt]);
                            if (mask != 0ull && L.lane == 0) {               // lanes 0-31: column j, lanes 32-63: column j + 4
                                const int lo = __popcll(mask & 0xffffffffull);
                                const int hi = __popcll(mask >> 32);
                                if (lo) atomicAdd(col_count + j - L.h * 4, lo);
                                if (hi) atomicAdd(col_count + j - L.h * 4 + 4, hi);
                            }
                            covf[nt] = covf[nt] || sure;
                            want = !sure && u <= thi[nt];
                            if constexpr (WANT_MIN) want = want || u <= m[nt] + e2[nt];
                        }
                        if constexpr (NEED_ANY) {
                            if (!anyf[nt] && u <= th[g4][e]) {
                                if (u < alo[g4 * 8 + e]) {                      // certain witness
                                    anyf[nt] = true;
                                    athr[nt] = -INFINITY;
                                } else {
                                    want = true;                                // ambiguous "any"
                                }
                            }
                        }
                        if (want) push(prow[nt], (unsigned)j | (sure ? FAST_COUNTED : 0u));
                    }
                }
            }
        }
    }
};

template <class Lane, bool WANT_MIN, bool NEED_ANY>
struct CrossWideShim {
    CrossWideEpilogue<Lane>& e;
    __device__ __forceinline__ void aux_issue(int t, int64_t q) { e.aux_issue(t, q); }
    __device__ __forceinline__ void aux_commit(int t) { e.aux_commit(t); }
    __device__ __forceinline__ void finish(int t, int64_t q, f32x16 (&acc)[Lane::MT][Lane::NT]) { e.template finish_impl<WANT_MIN, NEED_ANY>(t, q, acc); }
    static constexpr bool ACC_INIT = Lane::ACC_INIT;
    __device__ __forceinline__ f32x16 acc_init(int t, int mt) const { return e.acc_init(t, mt); }
};

struct WideTiles {
    int64_t q0;
    __device__ __forceinline__ int64_t operator()(int t) const { return q0 + t; }
};

// work item = (256-row block, column chunk), XCD-grouped: block b runs on XCD b % 8; the 32 workgroups resident on an
// XCD (one per CU) form a group of grp_rows row blocks x 32 / grp_rows chunks, so a group keeps grp_rows P blocks
// (256 KB each) in the 4 MB L2 and fetches each Q tile once.
struct WideWork {
    int64_t rb, qtile0;
    int ntiles;
};
inline int64_t wide_grouped_blocks_impl(int64_t row_blocks, int nchunks, int grp_rows) {
    const int grp_chunks = 32 / grp_rows;
    const int64_t groups = ceil_div(row_blocks, grp_rows) * ceil_div(nchunks, grp_chunks);
    return ceil_div(groups, 8) * 8 * 32;
}
__device__ __forceinline__ WideWork wide_work(int64_t q_tiles, int nchunks, int64_t row_blocks, int grp_rows) {
    const int grp_chunks = 32 / grp_rows;
    const int64_t cgroups = (nchunks + grp_chunks - 1) / grp_chunks;
    const int xcd = blockIdx.x & 7;
    const int64_t seq = blockIdx.x >> 3;
    const int64_t g = (seq >> 5) * 8 + xcd;
    const int within = (int)(seq & 31);
    WideWork w;
    w.rb = (g / cgroups) * grp_rows + within / grp_chunks;
    const int chunk = (int)((g % cgroups) * grp_chunks + within % grp_chunks);
    w.qtile0 = 0;
    w.ntiles = 0;
    if (w.rb < row_blocks && chunk < nchunks) {
        w.qtile0 = q_tiles * chunk / nchunks;
        w.ntiles = (int)(q_tiles * (chunk + 1) / nchunks - w.qtile0);
    }
    return w;
}

template <class Eng, bool WANT_MIN>
__device__ __forceinline__ void
cross_wide_body(const float* __restrict__ Rb, int64_t Nr, int64_t ldr, const float* __restrict__ rnorm,
                  const float* __restrict__ rthr, const float* __restrict__ Cb, int64_t Nc, int64_t ldc,
                  const float* __restrict__ cnorm, const float* __restrict__ cthr, int Dh, int nchunks, int grp_rows,
                  const unsigned* __restrict__ maxn, unsigned* __restrict__ rmin_approx, unsigned* __restrict__ row_any,
                  unsigned* __restrict__ row_cover, int32_t* __restrict__ col_count, uint2* __restrict__ wgq, int qcap,
                  int* __restrict__ wgq_count, uint2* __restrict__ items, uint2* __restrict__ ovq, int* __restrict__ ov_count,
                  int ovcap, int* __restrict__ fail, float fc) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using Lane = typename Eng::Lane;
    Lane L;
    const WideWork w = wide_work((Nc + WTB - 1) / WTB, nchunks, (Nr + WTB - 1) / WTB, grp_rows);
    // (fail already raised - by an earlier workgroup whose entries no longer fit the overflow queue's budget, see
    // cross_fast_decide_kernel: the exact kernel will redo the whole call, the rest of this grid has nothing to add)
    if (w.ntiles == 0 || *reinterpret_cast<volatile int*>(fail) != 0) {
        if (L.tid == 0) wgq_count[blockIdx.x] = 0;
        return;
    }
    int* qn = reinterpret_cast<int*>(lds + Eng::LDS_WORDS + WIDE_AUX_WORDS + WIDE_RAW_WORDS);
    if (L.tid == 0) *qn = 0;
    const float gmax = fmaxf(__uint_as_float(maxn[0]), __uint_as_float(maxn[1]));
    CrossWideEpilogue<Lane> epi(L);
    epi.fc = fc;
    epi.qnorm = cnorm;
    epi.qthr = cthr;
    epi.nq = Nc;
    epi.rnmax_c = fc * gmax;
    epi.aux = lds + Eng::LDS_WORDS;
    epi.col_count = col_count;
    epi.wgq = wgq + (int64_t)blockIdx.x * qcap;
    epi.qn = qn;
    epi.qcap = qcap;
    epi.ovq = ovq;
    epi.ov_count = ov_count;
    epi.ovcap = ovcap;
    epi.fail = fail;
    epi.dsc = half_unscale(maxn[2], maxn[3]);
    epi.idsc = 1.f / epi.dsc;
    // (ACC_INIT: norms and thresholds travel in accumulator units, |.| <= a few G / |dsc| - two sets whose magnitudes lie so far
    // apart that this leaves f32 go to the exact kernel like operands that cannot be scaled)
    const bool units_ok = !Lane::ACC_INIT || gmax * -epi.idsc < 1.0e36f;
    if (blockIdx.x == 0 && L.tid == 0 && !(half_scale_ok(maxn[2]) && half_scale_ok(maxn[3]) && units_ok)) *fail = 1;
    const int64_t prow0 = w.rb * WTB;
#pragma unroll
    for (int nt = 0; nt < Lane::NT; ++nt) {
        const int64_t i = prow0 + L.prow(nt);
        const bool ok = i < Nr;
        epi.prow[nt] = i;
        epi.rowok[nt] = ok;
        epi.xn[nt] = ok ? rnorm[i] : 0.f;
        epi.xs[nt] = -epi.xn[nt] * epi.idsc;
        const float e = fc * ((ok ? rnorm[i] : 0.f) + gmax);
        epi.thi[nt] = ok ? rthr[i] + e : -INFINITY;
        epi.tlo[nt] = ok ? rthr[i] - e : -INFINITY;
        epi.e2[nt] = 2.f * e;
        // (a row past the end takes part in no test through its thresholds alone - thi, tlo = -inf above, the running minimum
        // -inf so that m + e2 is, and no witness wanted)
        epi.m[nt] = (WANT_MIN && ok) ? __uint_as_float(rmin_approx[i]) : -INFINITY;
        epi.anyf[nt] = ok ? (row_any[i] != 0u) : true;
        epi.xsn[nt] = epi.anyf[nt] ? INFINITY : epi.xs[nt];
        epi.athr[nt] = epi.anyf[nt] ? -INFINITY : 0.f;
        epi.covf[nt] = false;
    }
#ifdef AM_DEV_KNOBS
    if (g_wide_dbg & 4) {                                      // timing experiment: no block needs the "any" test
        CrossWideShim<Lane, WANT_MIN, false> quick{epi};
        Eng::run(Cb, Nc, ldc, WideTiles{w.qtile0}, Rb, Nr, ldr, prow0, w.ntiles, Dh, lds, L, quick);
    } else
#endif
    {
        CrossWideShim<Lane, WANT_MIN, true> shim{epi};
        Eng::run(Cb, Nc, ldc, WideTiles{w.qtile0}, Rb, Nr, ldr, prow0, w.ntiles, Dh, lds, L, shim);
    }
#pragma unroll
    for (int nt = 0; nt < Lane::NT; ++nt) {
        const float mn = fminf(epi.m[nt], __shfl_xor(epi.m[nt], 32));
        const int other = __shfl_xor((int)epi.anyf[nt], 32);
        const int other_c = __shfl_xor((int)epi.covf[nt], 32);
        const bool any = epi.anyf[nt] || other != 0;
        const bool cov = epi.covf[nt] || other_c != 0;
        if (L.h == 0 && epi.rowok[nt]) {
            if constexpr (WANT_MIN) atomicMin(rmin_approx + epi.prow[nt], __float_as_uint(mn));
            if (any) atomicOr(row_any + epi.prow[nt], 1u);
            if (cov) atomicOr(row_cover + epi.prow[nt], 1u);
        }
    }
    __syncthreads();
    // the region's entries in batches of 64, appended to the list of work items of cross_verify_regions_kernel
    // (ov_count[2] = number of items; the order of the list does not matter: the verification only feeds integer atomics)
    if (L.tid == 0) {
        const int n = *qn < qcap ? *qn : qcap, nb = (n + 63) / 64;
        wgq_count[blockIdx.x] = n;
        qn[0] = n;
        qn[1] = nb > 0 ? atomicAdd(ov_count + 2, nb) : 0;
    }
    __syncthreads();
    const int n = qn[0], base = qn[1];
    for (int t = L.tid; t * 64 < n; t += Lane::WAVES * 64) items[base + t] = make_uint2(blockIdx.x, (unsigned)(t * 64));
}



// The same sweep on the 256 x 256 f16 engine (wide_engine.h): row blocks and column tiles of 256 rows, 512 threads.
template <class Eng> constexpr size_t knn_wide_lds_bytes() { return (Eng::LDS_WORDS + 8 * WTB) * sizeof(float) + 16; }   // + aux [2][2][256] + raw [2][2][256]

template <int KCAP, class Lane>
using KnnWideEpilogue = KnnFastEpilogue<KCAP, Lane, WTB, Lane::MT>;

template <class Eng, int KCAP>
__device__ __forceinline__ void
knn_wide_body(const float* __restrict__ Xb, int64_t N, int64_t ldh, const float* __restrict__ xnorm, float* thr, int Dh,
                int win_tiles, int nwin, int per_win, int k1, const unsigned* __restrict__ maxn, float* __restrict__ partial,
                int* __restrict__ cnt, int cap, uint2* __restrict__ wgq, float* __restrict__ wgv, int qcap,
                int* __restrict__ wgq_count, int part, int nparts, float fc, uint2* __restrict__ ovq, float* __restrict__ ovv,
                unsigned long long* __restrict__ ovn, int ovcap, const int* __restrict__ skip, int* __restrict__ region_counter) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using Lane = typename Eng::Lane;
    Lane L;
    const int64_t T = (N + WTB - 1) / WTB;
    const SymWork sw = sym_work(T, win_tiles, nwin, per_win, part, nparts);
    // The queue region of a 256-row workgroup is NW = 8 private sub-regions + a shared part (what the host sizes, zeroes and
    // scatters: pairwise_fast.h).  An engine of four waves (pstat64) gives each wave SPW = 2 adjacent sub-regions as ONE private
    // part and reports its fill as two counts.
    constexpr int NW = KnnWideEpilogue<KCAP, Lane>::NWAVES;
    constexpr int SPW = NW / Lane::WAVES;
    static_assert(SPW * Lane::WAVES == NW, "sub-regions per wave");
    if (sw.ntiles == 0 || (skip != nullptr && *skip != 0)) {      // (skip: the data-dependent fallback took over, pairwise_fast.h)
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
        int* slot = reinterpret_cast<int*>(lds + Eng::LDS_WORDS + 8 * WTB) + 1;
        if (L.tid == 0) *slot = atomicAdd(region_counter, 1);
        __syncthreads();
        region = __builtin_amdgcn_readfirstlane(*slot);
    }
    const float nmax = __uint_as_float(maxn[0]);
    // 2 fc through the exponent field: an integer add on the scalar unit (a float multiply of a wave-uniform value lands in a
    // vector register, and the k <= 10 instantiation at 512 columns has none to spare)
    const float fc2 = __uint_as_float(__float_as_uint(fc) + (1u << 23));
    KnnWideEpilogue<KCAP, Lane> epi(L);
    epi.qnorm = xnorm;
    epi.thr = thr;
    epi.n = N;
    epi.pblock = sw.pb;
    epi.aux = lds + Eng::LDS_WORDS;
    const int wave = __builtin_amdgcn_readfirstlane(L.wave);
    const int wc1 = qcap / (2 * NW);                  // private sub-regions: half of the workgroup's region in all
    epi.wcap = SPW * wc1;
    epi.wgq = wgq + region * qcap + wave * epi.wcap;
    epi.wgv = wgv + region * qcap + wave * epi.wcap;
    epi.wq = 0;
    epi.shcap = qcap - NW * wc1;                      // the shared part behind them
    epi.shq = wgq + region * qcap + NW * wc1;
    epi.shv = wgv + region * qcap + NW * wc1;
    epi.qn = reinterpret_cast<int*>(lds + Eng::LDS_WORDS + 8 * WTB);
    if (L.tid == 0) *epi.qn = 0;                    // visible after the pipeline's first barrier
    epi.ovq = ovq;
    epi.ovv = ovv;
    epi.ovn = ovn;
    epi.ovcap = ovcap;
    epi.cnt = cnt;
    epi.cap = cap;
    epi.dsc = half_unscale(maxn[2], maxn[2]);
    epi.idsc = 1.f / epi.dsc;
#pragma unroll
    for (int nt = 0; nt < Lane::NT; ++nt) {
        const int64_t i = sw.pb * WTB + L.prow(nt);
        epi.prow[nt] = (unsigned)i;
        epi.xn[nt] = i < N ? xnorm[i] : INFINITY;
        epi.xs[nt] = -epi.xn[nt] * epi.idsc;
        epi.e2c = fc2;
        epi.e2n = fc2 * nmax;
        epi.flt[nt] = i < N ? __hip_atomic_load(thr + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -INFINITY;
#pragma unroll
        for (int s = 0; s < KCAP; ++s) epi.best[nt][s] = s < KCAP - k1 ? -INFINITY : INFINITY;
    }
    Eng::run(Xb, N, ldh, WideTiles{sw.qa}, Xb, N, ldh, sw.pb * WTB, sw.ntiles, Dh, lds, L, epi);
    float* mg = lds;                                   // [256][LISTS][KCAP]: the engine's buffers are free now
    constexpr int LISTS = Lane::LISTS;
#pragma unroll
    for (int nt = 0; nt < Lane::NT; ++nt) {
        float* dst = mg + (L.prow(nt) * LISTS + L.list_slot()) * KCAP;
#pragma unroll
        for (int s = 0; s < KCAP; ++s) dst[s] = epi.best[nt][s];
    }
    __syncthreads();
    if (L.lane == 0) {
#pragma unroll
        for (int sub = 0; sub < SPW; ++sub) wgq_count[region * (NW + 1) + wave * SPW + sub] = max(0, min(min(epi.wq, epi.wcap) - sub * wc1, wc1));
    }
    if (L.tid == 0) wgq_count[region * (NW + 1) + NW] = min(*epi.qn, epi.shcap);
    if (L.tid < WTB) {
        const int64_t i = sw.pb * WTB + L.tid;
        if (i < N) {
            const float* src = mg + L.tid * LISTS * KCAP;
            float m[KCAP];
#pragma unroll
            for (int s = 0; s < KCAP; ++s) m[s] = src[s];
            for (int s = KCAP; s < LISTS * KCAP; ++s)
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
            const float bound = kthv + fc2 * (xnorm[i] + nmax);
            // (a row taken out of the sweep keeps its -inf: knn_fast_mask_flat_kernel; nobody else writes thr[i])
            if (epi_row_in_sweep(thr, i)) atomicMin(reinterpret_cast<unsigned*>(thr) + i, __float_as_uint(bound));
        }
    }
}



}  // namespace am
