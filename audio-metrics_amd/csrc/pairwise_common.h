// Helpers shared by the two translation units of the PRDC path: pairwise.hip (exact f32 kernels, the 128-row f16 filter
// kernels and every host entry point) and pairwise_wide.hip (the 256 x 256 f16 filter kernels of wide_engine.h, built
// separately so that the two compile in parallel).
#pragma once
#include "am_common.h"
#include "tile_engine.h"
#include <stdlib.h>

namespace am {

constexpr int EV_DEFAULT = EV_RSRC | EV_FRAGDB | EV_EARLY;   // the production schedule of the tile engine
constexpr int WIDE_TILE_ROWS = 256;                          // tile rows of the wide engine (wide_engine.h: WTB)

// Development knobs exist only in the A/B build (-DAM_DEV_KNOBS -> libaudio_metrics_hip_dev.so, loaded by the tools and
// by the tests that force fallback paths).  In the shipped library every knob is its default, a compile-time constant:
// no getenv, path selection is a pure function of the shapes, and the older engine schedules are not instantiated.
#ifdef AM_DEV_KNOBS
static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}
static int engine_variant() {
    static const int v = env_int("AM_ENGINE_VARIANT", EV_DEFAULT);
    return v;
}
#else
static constexpr int env_int(const char*, int dflt) { return dflt; }
static constexpr int engine_variant() { return EV_DEFAULT; }
#endif


// Scaled f16 copies of the filter passes (pairwise_fast.h): a matrix whose largest |element| has the f32 bit pattern
// `maxabs_bits` is multiplied by 2^half_scale_exp so that the largest element lands in [2^13, 2^14) - far from
// f16 overflow (65504), and small elements far from f16's subnormal range.  Powers of two: the scaling is exact.
__device__ __forceinline__ int half_scale_exp(unsigned maxabs_bits) {
    if (maxabs_bits == 0u) return 0;
    const int e = (int)((maxabs_bits >> 23) & 255u) - 127;          // floor(log2(max |x|)) for a normal maximum
    const int ex = 13 - e;
    return ex < -60 ? -60 : (ex > 60 ? 60 : ex);
}
__device__ __forceinline__ bool half_scale_ok(unsigned maxabs_bits) {   // finite, and the exponent was not clamped
    if (maxabs_bits == 0u) return true;
    const int e = (int)((maxabs_bits >> 23) & 255u) - 127;
    return e != 128 && e != -127 && 13 - e >= -60 && 13 - e <= 60;
}
// the factor that turns the dot product of two scaled copies into -2 <x, y>:  -2 * 2^-(ex + ey)   (exact)
__device__ __forceinline__ float half_unscale(unsigned maxabs_bits_x, unsigned maxabs_bits_y) {
    const int s = half_scale_exp(maxabs_bits_x) + half_scale_exp(maxabs_bits_y);
    return -2.f * __uint_as_float((unsigned)(127 - s) << 23);
}


// Work item of the symmetric sweep (shared by the exact kernel and the f16 filter kernel of pairwise_fast.h).
struct SymWork {
    int W;              // column-tile window
    int64_t pb;         // row block
    int64_t qa;         // first Q tile of this block inside the window
    int ntiles;         // 0: nothing of this window belongs to this block / this rank
};

// tiles of window W that belong to row block pb (cyclic half-range pairing; see sym_work)
__host__ __device__ __forceinline__ SymWork sym_item(int64_t T, int win_tiles, int W, int64_t pb, int part, int nparts) {
    const int64_t q0 = (int64_t)W * win_tiles;
    const int64_t q1 = (q0 + win_tiles < T) ? q0 + win_tiles : T;
    // offsets 0 .. T/2; for even T the antipodal offset belongs to the lower-numbered block only
    int64_t noff = T / 2 + 1;
    if ((T % 2) == 0 && pb >= T / 2) noff = T / 2;
    // tiles q of the window with (q - pb) mod T < noff form one contiguous piece (window << T/2)
    int64_t qa = pb > q0 ? pb : q0, qb = (pb + noff < q1) ? pb + noff : q1;          // q >= pb
    if (qa >= qb) {                                                                   // wrapped: q < pb
        qa = q0;
        qb = (pb + noff - T < q1) ? pb + noff - T : q1;
    }
    const int ntiles = qb > qa ? (int)(qb - qa) : 0;
    // multi-GPU: rank `part` of `nparts` owns the CONTIGUOUS range of row blocks with floor(pb*nparts/T) == part
    // (see am_knn_sym_part_f32).  Not pb mod nparts: consecutive blockIdx map to consecutive pb, and ownership
    // by residue would put every owned workgroup of a window on the same XCD (blockIdx % 8).
    SymWork w;
    w.W = W;
    w.pb = pb;
    w.qa = qa;
    w.ntiles = (ntiles == 0 || (int)(pb * nparts / T) != part) ? 0 : ntiles;
    return w;
}

// (window, slot e of the window) -> row block: within a window the row blocks that own ALL of its tiles come first (equal
// work items that walk the window's Q tiles in lockstep), the partial ones last
__host__ __device__ __forceinline__ int64_t sym_block_of(int64_t T, int win_tiles, int W, int e) {
    const int64_t q0 = (int64_t)W * win_tiles;
    const int64_t q1 = (q0 + win_tiles < T) ? q0 + win_tiles : T;
    const int wlen = (int)(q1 - q0);
    const int full_lo = wlen - 1, full_hi = (int)(T / 2);          // r in [full_lo, full_hi]: pb <= q0, pb+noff >= q1
    const int nfull = full_hi >= full_lo ? full_hi - full_lo + 1 : 0;
    int r;
    if (e < nfull) r = full_lo + e;
    else if (e - nfull < full_lo) r = e - nfull;
    else r = full_hi + 1 + (e - nfull - full_lo);
    int64_t pb = (q1 - 1 - r) % T;
    if (pb < 0) pb += T;
    return pb;
}

// workgroups of the sweep that have tiles to multiply for rank `part` (host side: sizes the queue regions of a partitioned run)
static inline int64_t sym_active_items(int64_t T, int win_tiles, int nwin, int per_win, int part, int nparts) {
    int64_t n = 0;
    for (int W = 0; W < nwin; ++W)
        for (int e = 0; e < per_win; ++e) n += sym_item(T, win_tiles, W, sym_block_of(T, win_tiles, W, e), part, nparts).ntiles > 0;
    return n;
}

__device__ __forceinline__ SymWork sym_work(int64_t T, int win_tiles, int nwin, int per_win, int part, int nparts) {
    // Work item = (column-tile WINDOW, row block) over the CYCLIC HALF-RANGE pairing: block pb owns the tile
    // pairs (pb, q) with (q - pb) mod T in 0 .. T/2.  All workgroups in flight stream the same window of Q
    // tiles (L2 / Infinity-Cache reuse) with different row blocks.  Windows are swept in DESCENDING order: the
    // lane-local tiles of block pb lie in the windows from the one holding pb upwards (plus, for the upper
    // half of the blocks, a wrapped piece at the bottom), so when the window holding pb is reached - the one
    // in which other blocks generate the mirrored candidates for pb's rows - pb has already published a
    // bound over most of its half-range.
    const int W = nwin - 1 - (int)(blockIdx.x / per_win);
    const int64_t pb = sym_block_of(T, win_tiles, W, (int)(blockIdx.x % per_win));
    return sym_item(T, win_tiles, W, pb, part, nparts);
}

// (An XCD-aware order of these items for the 256-row engine - the 32 workgroups resident on an XCD = 8 consecutive row
// blocks x 4 consecutive windows, 3 MB of operands per L2 instead of 8 MB - was built and measured at 100k x 512: same
// bits, 15 % more queue entries because four windows are in flight per row block, 6.55 -> 6.63 ms per launch.  With
// every operand read confined to 2 MB the sweep gains 8 % at most (AM_WIDE_DBG=1): the kernel is bound by its
// per-stage issue pattern and by the clock its power draw allows, not by L2 misses.)


#ifdef AM_DEV_KNOBS
// timing experiments of the A/B build (AM_WIDE_DBG, set by the launchers of pairwise_wide.hip): 1 = every tile of the wide
// engine reads the first eight 256-row blocks (results meaningless), 2 = tile epilogues skipped (matrix pipeline only)
static __constant__ int g_wide_dbg;
// AM_WIDE_TRACE: s_memtime stamps of wave 0 / wave 4 of the first 64 workgroups, 96 stages x 6 points each (dev tool)
static __constant__ unsigned long long* g_wide_trace;
static __constant__ int g_wide_trace_b0;                      // AM_WIDE_TRACE_B0: first traced workgroup
#endif

// false for a row that knn_fast_mask_flat_kernel took out of the sweep (bound -inf, set before the sweep starts)
__device__ __forceinline__ bool epi_row_in_sweep(const float* thr, int64_t i) {
    return __hip_atomic_load(thr + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > -INFINITY;
}

constexpr unsigned FAST_COUNTED = 0x80000000u;                               // membership queue entry: the pair was already counted as certain
constexpr unsigned FAST_BOTH = 0x80000000u;
constexpr int FAST_ROW_OVERFLOW = 0x40000000;                                 // OR-ed into a row's entry count: > any cap, and the
                                                                              // later +1's of the scatter cannot wrap it
constexpr unsigned FAST_HOLE = 0xffffffffu;                                  // pair-list slot left unwritten (list full)

// Wave-wide float compares as LANE MASKS in scalar registers (v_cmp_*_f32 sN, a, b).  `__any(a <= b || ...)` materialises the
// bool as 0 / 1 in a vector register and compares that again (v_cndmask, v_and, v_cmp_ne: five vector instructions per gate in
// the filter epilogues, where every vector pass is paid in full - pstat_engine.h); masks combine on the scalar unit and a gate
// is `mask == 0`.
__device__ __forceinline__ unsigned long long lanes_le(float a, float b) { return __builtin_amdgcn_fcmpf(a, b, 5); }   // ordered <=
__device__ __forceinline__ unsigned long long lanes_ge(float a, float b) { return __builtin_amdgcn_fcmpf(a, b, 3); }   // ordered >=
__device__ __forceinline__ unsigned long long lanes_lt(float a, float b) { return __builtin_amdgcn_fcmpf(a, b, 4); }   // ordered <

// Lane / TBX / MT: geometry of the engine underneath - LaneInfo, 128, 2 (tile_engine.h) or WLane, 256, 4 (wide_engine.h)
//
// Queue protocol (round 3).  A workgroup's region of `qcap` entries is NWAVES private sub-regions of wcap = qcap / (2 NWAVES)
// entries followed by one shared part of the remaining half.  Every WAVE counts the entries of its private sub-region in a
// scalar register: the tests that select an entry are wave-wide anyway (ballots), so a slot is `count + (number of
// selected lanes below this one)` - no LDS atomic, no round trip in the epilogue's critical path.  (Round 2 reserved slots
// with one returning LDS atomic per lane and accumulator tile, and looked at every register of a hit group twice: once
// for the bit masks, once to store.)  Only a wave whose private part is full - a lane holding an outlier row with a loose
// bound sends all of that row's entries through ONE wave - takes slots of the shared part from an LDS counter; with
// private parts alone 1M-row sets overflowed regions that the whole workgroup would have absorbed.
template <int KCAP, class Lane = LaneInfo, int TBX = TB, int MT = 2>
struct KnnFastEpilogue {
    static constexpr int NWAVES = TBX == WIDE_TILE_ROWS ? 8 : 4;   // waves per workgroup
    static constexpr int NT = Lane::NT;                             // 32-row P tiles per wave
    // ACC_INIT (pstat_engine.h): the accumulators start at c0_j = |x_j|^2 / dsc (dsc = -2 / scales: c0_j = -|x_j|^2 / 2 in the
    // units of the scaled dot product), so an element holds a' = <x_i, x_j>' + c0_j and the approximate squared distance is
    // dsc a' + |x_i|^2 - a DEcreasing function of a', exact up to the scaling by a power of two.  The tests of the fast path
    // then run on the accumulator itself:
    //   own row     min_j (dsc a'_j + |x_i|^2) <= bound_i        <=>  fma(dsc, max_j a'_j, |x_i|^2) <= bound_i      (max3 only)
    //   column j    dsc a'_j + |x_i|^2 <= thr_j                    <=>  a'_j - thr_j / dsc >= -|x_i|^2 / dsc          (sub, max3)
    // two vector passes per accumulator element instead of three (fma with the column norm, subtract, two half min3).
    // Error bound: c0_j is one more term of the matrix core's accumulation (pairwise_fast.h, fast_c: D + 1 terms).
    static constexpr bool ACC_INIT = Lane::ACC_INIT;
    const float* qnorm;
    const float* thr;
    int64_t n, pblock;
    float* aux;                 // LDS [2][2][TBX] : |x_j|^2 and thr[j] of the tile
    uint2* wgq;                 // this WAVE's private sub-region
    float* wgv;                 // approximate value of each queued pair (pruning, knn_fast_prune_kernel)
    int wq;                     // entries this wave has queued (wave-uniform, lives in a scalar register)
    int wcap;                   // capacity of the private sub-region
    uint2* shq;                 // the workgroup's shared part (slots from the LDS counter `qn`)
    float* shv;
    int* qn;
    int shcap;
    uint2* ovq;                 // global spill queue for entries that do not fit their region
    float* ovv;
    unsigned long long* ovn;
    int ovcap;
    int* cnt;
    int cap;
    float dsc;                  // -2 / (operand scale)^2
    float idsc;                 // 1 / dsc (a power of two: exact)
    float xs[NT];               // ACC_INIT: -|x_i|^2 / dsc, what a column-direction margin is compared with
    unsigned prow[NT];           // (row indices fit 32 bits: the filter path is limited to < 2^31 rows)
    float xn[NT], flt[NT];        // xn = +inf for rows past the end: every approximate value is +inf and passes no test
    float e2c, e2n;             // 2 E_i = e2c * |x_i|^2 + e2n  (recomputed per tile group: two registers less than keeping it)
    float best[NT][KCAP];        // ascending; the first KCAP - (k+1) slots are -inf pads, so best[KCAP-1] is the (k+1)-th smallest
    float aux_n, aux_t;
    const Lane& L;

    __device__ __forceinline__ KnnFastEpilogue(const Lane& l) : L(l) {}
    __device__ __forceinline__ void store_entry(int slot, unsigned a, unsigned b, bool both, float val) {
        if (slot < wcap) {
            wgq[slot] = make_uint2(a | (both ? FAST_BOTH : 0u), b);
            wgv[slot] = val;
            return;
        }
        const int s1 = atomicAdd(qn, 1);                 // private part full: the workgroup's shared part
        if (s1 < shcap) {
            shq[s1] = make_uint2(a | (both ? FAST_BOTH : 0u), b);
            shv[s1] = val;
        } else {
            // Region full (rare: the regions hold 8x the expected survivors).  The 128-row short-list instantiation keeps a
            // global spill queue; everywhere else the row(s) go straight to the exact fix-up kernel - this store is
            // instantiated 128 times per tile, and a larger body here pushes the unrolled epilogue over the compiler's
            // full-unroll budget (the accumulator array is then indexed dynamically and lands in scratch).
            if constexpr (KCAP <= 6 && TBX != WIDE_TILE_ROWS) {
                // (inputs the bound cannot decide send hundreds of millions of pairs here: neither the counter nor the
                // per-row markers may wrap)
                const unsigned long long s2 = atomicAdd(ovn, 1ull);          // 64-bit: cannot wrap
                if (s2 < (unsigned long long)ovcap) {
                    ovq[s2] = make_uint2(a | (both ? FAST_BOTH : 0u), b);
                    ovv[s2] = val;
                    return;
                }
            }
            atomicOr(cnt + a, FAST_ROW_OVERFLOW);
            if (both) atomicOr(cnt + b, FAST_ROW_OVERFLOW);
        }
    }
    __device__ __forceinline__ void aux_issue(int, int64_t qtile) {
        if (L.tid < TBX) {
            const int64_t j = qtile * TBX + L.tid;
            aux_n = j < n ? qnorm[j] : INFINITY;
            aux_t = j < n ? __hip_atomic_load(thr + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -INFINITY;
        }
    }
    __device__ __forceinline__ void aux_commit(int t) {
        if (L.tid < TBX) {
            // ACC_INIT: both in the units of the accumulator (+-inf keep their meaning: a column past the end starts at -inf
            // and can never be the maximum; its bound -inf becomes +inf and no margin reaches it)
            aux[(t & 1) * 2 * TBX + L.tid] = ACC_INIT ? aux_n * idsc : aux_n;
            aux[(t & 1) * 2 * TBX + TBX + L.tid] = ACC_INIT ? aux_t * idsc : aux_t;
        }
    }
    // start value of accumulator tile mt of the half tile L.wm: register 4 g + e <-> column mt * 32 + g * 8 + h * 4 + e
    __device__ __forceinline__ f32x16 acc_init(int t, int mt) const {
        const float* a = aux + (t & 1) * 2 * TBX + L.wm * (MT * 32) + L.h * 4 + mt * 32;
        f32x16 c;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(a + g4 * 8);
            c[g4 * 4 + 0] = v.x; c[g4 * 4 + 1] = v.y; c[g4 * 4 + 2] = v.z; c[g4 * 4 + 3] = v.w;
        }
        return c;
    }
    __device__ __forceinline__ void finish(int t, int64_t qtile, f32x16 (&acc)[MT][NT]) {
        const float* a = aux + (t & 1) * 2 * TBX + L.wm * (MT * 32) + L.h * 4;
        const bool mirror = qtile != pblock;                // the diagonal tile holds both directions itself
#ifdef AM_DEV_KNOBS
        if constexpr (TBX == WIDE_TILE_ROWS) {
            if (g_wide_dbg & 2) return;                     // timing experiment: MFMA pipeline only
        }
#endif
        const unsigned jbase = (unsigned)(qtile * TBX) + L.wm * (MT * 32) + L.h * 4;
        if constexpr (ACC_INIT) {
            float xsm[NT];                                    // the diagonal tile holds both directions itself: no column-direction hit
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) xsm[nt] = mirror ? xs[nt] : INFINITY;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x4 tqs[4];
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) tqs[g4] = *reinterpret_cast<const f32x4*>(a + TBX + mt * 32 + g4 * 8);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    float amax4[4], wmax4[4];
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        float am = -INFINITY, wm = -INFINITY;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            am = fmaxf(am, acc[mt][nt][g4 * 4 + e]);
                            wm = fmaxf(wm, acc[mt][nt][g4 * 4 + e] - tqs[g4][e]);
                        }
                        amax4[g4] = am;
                        wmax4[g4] = wm;
                    }
                    const float amax = fmaxf(fmaxf(amax4[0], amax4[1]), fmaxf(amax4[2], amax4[3]));
                    const float wmax = fmaxf(fmaxf(wmax4[0], wmax4[1]), fmaxf(wmax4[2], wmax4[3]));
                    const float tmin = fmaf(dsc, amax, xn[nt]);         // the smallest approximate value of the tile's 16 elements
                    const float pl = fminf(flt[nt], best[nt][KCAP - 1] + fmaf(e2c, xn[nt], e2n));
                    if ((lanes_le(tmin, pl) | lanes_ge(wmax, xsm[nt])) == 0ull) continue;    // gate 1 (see below)
#ifdef AM_DEV_KNOBS
                    if constexpr (TBX == WIDE_TILE_ROWS) {
                        if (g_wide_dbg & 16) continue;
                    }
#endif
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        if ((lanes_le(fmaf(dsc, amax4[g4], xn[nt]), pl) | lanes_ge(wmax4[g4], xsm[nt])) == 0ull) continue;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int reg = g4 * 4 + e;
                            const float u = fmaf(dsc, acc[mt][nt][reg], xn[nt]);
                            const float w = acc[mt][nt][reg] - tqs[g4][e];
                            const unsigned long long sel = lanes_le(u, pl) | lanes_ge(w, xsm[nt]);
                            if (sel != 0ull) {
                                const bool own = u <= pl, mir = w >= xsm[nt];
                                const int below = __builtin_amdgcn_mbcnt_hi((unsigned)(sel >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)sel, 0u));
                                const int slot = wq + below;
                                wq += __popcll(sel);
                                const unsigned j = jbase + mt * 32 + g4 * 8 + e;
#ifdef AM_DEV_KNOBS
                                if (TBX == WIDE_TILE_ROWS && (g_wide_dbg & 32)) continue;
#endif
                                if (own || mir) store_entry(slot, own ? prow[nt] : j, own ? j : prow[nt], own && mir, u);
                            }
                        }
                    }
                    const float vmin = tmin <= pl ? fmaxf(tmin, 0.f) : INFINITY;
#ifdef AM_DEV_KNOBS
                    if (TBX == WIDE_TILE_ROWS && (g_wide_dbg & 64)) continue;
#endif
                    if (lanes_lt(vmin, best[nt][KCAP - 1]) != 0ull) list_insert<KCAP>(best[nt], vmin);
                }
            }
            return;
        }
        const float mthr = mirror ? 0.f : -INFINITY;        // the diagonal tile holds both directions itself: no margin reaches -inf
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x4 yn[4], tq[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                yn[g4] = *reinterpret_cast<const f32x4*>(a + mt * 32 + g4 * 8);
                tq[g4] = *reinterpret_cast<const f32x4*>(a + TBX + mt * 32 + g4 * 8);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                // Fast path: per group of four accumulator registers (four columns) the smallest value and the smallest margin
                // against the columns' bounds.  The row's own norm is added to the minima, not to every element (the extra
                // rounding is one of those fast_c's 2^-19 term pays for).
                float tmin4[4], marg4[4];
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    float tm = INFINITY, mg = INFINITY;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = fmaf(dsc, acc[mt][nt][g4 * 4 + e], yn[g4][e]);
                        tm = fminf(tm, t);
                        mg = fminf(mg, t - tq[g4][e]);
                    }
                    tmin4[g4] = tm + xn[nt];
                    marg4[g4] = mg + xn[nt];
                }
                const float tmin = fminf(fminf(tmin4[0], tmin4[1]), fminf(tmin4[2], tmin4[3]));
                const float mmin = fminf(fminf(marg4[0], marg4[1]), fminf(marg4[2], marg4[3]));
                // the own-row bound is frozen for the 16 elements of this accumulator tile (a looser filter is always
                // safe); the list - and with it the bound of the next tile - is updated behind the stores
                const float pl = fminf(flt[nt], best[nt][KCAP - 1] + fmaf(e2c, xn[nt], e2n));
                // Gate 1, one wave-uniform branch per accumulator tile (1024 pairs): nothing to look at.  (Round 2 evaluated
                // the four group gates of every tile and kept their outcomes as values for two later loops: ~100 VALU
                // instructions per accumulator tile of which ~50 were flag bookkeeping.)
                if ((lanes_le(tmin, pl) | lanes_le(mmin, mthr)) == 0ull) continue;      // (lane masks on the scalar unit: see lanes_le)
#ifdef AM_DEV_KNOBS
                if constexpr (TBX == WIDE_TILE_ROWS) {
                    if (g_wide_dbg & 16) continue;              // timing experiment: fast path and gate only, no detail path
                }
#endif
                // Gate 2 per register group, then ONE pass over the group's four registers: test, ballot, slot, store.
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    if ((lanes_le(tmin4[g4], pl) | lanes_le(marg4[g4], mthr)) == 0ull) continue;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int reg = g4 * 4 + e;
                        const float u = fmaf(dsc, acc[mt][nt][reg], xn[nt] + yn[g4][e]);
                        const bool own = u <= pl, mir = mirror && u <= tq[g4][e];
                        const unsigned long long sel = lanes_le(u, pl) | (mirror ? lanes_le(u, tq[g4][e]) : 0ull);
                        if (sel != 0ull) {                                                  // wave-uniform
                            const int below = __builtin_amdgcn_mbcnt_hi((unsigned)(sel >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)sel, 0u));
                            const int slot = wq + below;
                            wq += __popcll(sel);
                            const unsigned j = jbase + mt * 32 + g4 * 8 + e;
                            // filed under its own row, or (mirrored only) under row j
#ifdef AM_DEV_KNOBS
                            if (TBX == WIDE_TILE_ROWS && (g_wide_dbg & 32)) continue;     // timing experiment: entries found, not stored
#endif
                            if (own || mir) store_entry(slot, own ? prow[nt] : j, own ? j : prow[nt], own && mir, u);
                        }
                    }
                }
                // The lane-local list only steers the filter (its (k+1)-th smallest bounds the row's final value from
                // above).  It takes ONE value per accumulator tile - the smallest of the sixteen - instead of every
                // queued one: a list over a subset of the row's values still bounds from above, two of a row's k+1
                // nearest neighbours practically never share a tile, and sixteen conditional insertions per tile were
                // a large part of the epilogue while the bounds are loose.
                const float vmin = tmin <= pl ? fmaxf(tmin, 0.f) : INFINITY;
#ifdef AM_DEV_KNOBS
                if (TBX == WIDE_TILE_ROWS && (g_wide_dbg & 64)) continue;                 // timing experiment: no lane-local lists
#endif
                if (lanes_lt(vmin, best[nt][KCAP - 1]) != 0ull) list_insert<KCAP>(best[nt], vmin);
            }
        }
    }
};


// ---- launchers of the 256 x 256 f16 filter kernels (pairwise_wide.hip) ------------------------------------------------
constexpr int KNN_WIDE_MAX_KCAP = 11;                     // list registers beside 128 accumulators: k <= 10 on the wide engine
int64_t wide_grouped_blocks(int64_t row_blocks, int nchunks, int grp_rows);
int launch_cross_wide(bool want_min, unsigned blocks, const float* Rb, int64_t Nr, int64_t ldr, const float* rnorm, const float* rthr,
                      const float* Cb, int64_t Nc, int64_t ldc, const float* cnorm, const float* cthr, int Dh, int nchunks,
                      int grp_rows, const unsigned* maxn, unsigned* rmin_approx, unsigned* row_any, unsigned* row_cover,
                      int32_t* col_count, uint2* wgq, int qcap, int* wgq_count, uint2* items, uint2* ovq, int* ov_count, int ovcap,
                      int* fail, float fc, hipStream_t st);
int launch_cross_wide_sample(const float* Rb, int64_t Nr, int64_t ldr, const float* rnorm, const float* Cb, int64_t Nc, int64_t ldc,
                             const float* cnorm, const float* cthr, int Dh, int stride, int nchunks, const unsigned* maxn,
                             unsigned* row_any, float fc, hipStream_t st);
int launch_knn_wide_sample(int kcap, const float* Xb, int64_t N, int64_t ldh, const float* xnorm, int Dh, int stride, int nchunks,
                           const unsigned* maxn, float* partial, int64_t row0, int64_t nrows, hipStream_t st);
int launch_knn_wide(int kcap, unsigned nwg, const float* Xb, int64_t N, int64_t ldh, const float* xnorm, float* thr, int Dh,
                    int win_tiles, int nwin, int per_win, int k1, const unsigned* maxn, float* partial, int* cnt, int cap,
                    uint2* wgq, float* wgv, int qcap, int* wgq_count, int part, int nparts, float fc, uint2* ovq, float* ovv,
                    unsigned long long* ovn, int ovcap, const int* skip, int* region_counter, hipStream_t st);

// ---- the same two kernels on the operand-stationary engine (pairwise_pstat.hip, pstat_engine.h): rows of up to 512 f16
bool pstat_supported(int Dh);
bool pstat64_supported(int Dh);              // ... in its 256-thread form (rows of up to two slabs: D <= 128)
int launch_cross_pstat(bool want_min, unsigned blocks, const float* Rb, int64_t Nr, int64_t ldr, const float* rnorm, const float* rthr,
                       const float* Cb, int64_t Nc, int64_t ldc, const float* cnorm, const float* cthr, int Dh, int nchunks,
                       int grp_rows, const unsigned* maxn, unsigned* rmin_approx, unsigned* row_any, unsigned* row_cover,
                       int32_t* col_count, uint2* wgq, int qcap, int* wgq_count, uint2* items, uint2* ovq, int* ov_count, int ovcap,
                       int* fail, float fc, hipStream_t st);
int launch_knn_pstat(int kcap, unsigned nwg, const float* Xb, int64_t N, int64_t ldh, const float* xnorm, float* thr, int Dh,
                     int win_tiles, int nwin, int per_win, int k1, const unsigned* maxn, float* partial, int* cnt, int cap,
                     uint2* wgq, float* wgv, int qcap, int* wgq_count, int part, int nparts, float fc, uint2* ovq, float* ovv,
                     unsigned long long* ovn, int ovcap, const int* skip, int* region_counter, hipStream_t st);
// which of the two 256-row engines multiplies the tiles of the main filter passes: a pure function of the row length
// (AM_WIDE_STATIONARY=0 in the A/B build: wide_engine.h for every shape)
static inline bool wide_stationary(int Dh) {
    static const int on = env_int("AM_WIDE_STATIONARY", 1);
    return on != 0 && pstat_supported(Dh);
}

}  // namespace am
