// Phased schedule of the 256 x 256 f16 tile engine (included by wide_engine.h; the geometry, the LDS image and the
// operand roles are documented there).
//
// One k-slab (64 f16 of all 512 rows, 64 KB) is multiplied in FOUR phases, one quadrant of the wave's 128 x 64 tile each:
//     phase 1: Q rows 0-63 x P rows 0-31      reads 8 Q + 4 P fragments (ds_read_b128)
//     phase 2: Q rows 0-63 x P rows 32-63     reads 4 P fragments
//     phase 3: Q rows 64-127 x P rows 32-63   reads 8 Q fragments
//     phase 4: Q rows 64-127 x P rows 0-31    reads nothing (P rows 0-31 stayed in registers)
// A phase is  [fragment reads of this phase; LDS-DMA of one half-slab; s_waitcnt vmcnt(8)]  s_barrier  [8 MFMAs]  s_barrier.
// Waves 4-7 (the second wave of every SIMD) pass one extra barrier before their first phase and therefore run half a phase
// behind waves 0-3: while one wave of a SIMD multiplies (256 matrix-pipe cycles out of registers, nothing to wait for),
// its partner reads fragments and issues DMA, and vice versa.  With the one-barrier-per-slab schedule this replaces, both
// waves of a SIMD read, waited and multiplied at the same time and the matrix pipe idled after every barrier.
//
// Half-slabs (16 KB each, the unit of staging): QA / QB = Q rows {0-63, 128-191} / {64-127, 192-255} - the rows phase 1 /
// phase 3 read, of both wave rows; PA / PB = P rows 64 w + {0-31} / 64 w + {32-63} for the four wave columns w.  Slab
// g sits in buffer g & 1 at its natural row positions, so the fragment addresses are those of the old schedule.
// Staging order, one half-slab per phase: slab g phase 1 -> PB(g+1), phase 2 -> QB(g+1), phase 3 -> QA(g+2), phase 4 -> PA(g+2).
//   * WAR: a half-slab is overwritten two or more phases after the phase that read it (QA: read in phase 1, restaged in
//     phase 3; PA: 1 -> 4; PB: 2 -> 5; QB: 3 -> 6).  The reads of a phase are retired (lgkmcnt) right after that phase's
//     first barrier by either wave group, i.e. at most three barriers later in workgroup time, and the restaging DMA is
//     issued four or more barriers later.
//   * RAW: every half-slab is staged exactly five phases before the phase that reads it.  s_waitcnt vmcnt(8) at the end
//     of the read/issue part of EVERY phase leaves the DMAs of the last four phases in flight, so a half-slab staged in
//     phase x has landed (for the issuing wave) before the first barrier of phase x+4; a reader passes that barrier, or -
//     for the group that runs ahead - the next one, before the reads of phase x+5.  (An LDS-DMA is only ordered for a
//     ds_read by the issuing wave's vmcnt wait followed by a barrier.)  The DMAs are in flight for four phases
//     (>= 2048 matrix-pipe cycles), against half a slab with vmcnt(0) before every barrier in the old schedule.
// Extra DMAs in the stream (the per-tile side data, one instruction per wave and tile) only make vmcnt(8) stricter.
// The compiler is kept out of the protocol: raw s_barrier (a __syncthreads() would add a fence that waits vmcnt(0)),
// sched_barrier(0) around every barrier, explicit vmcnt counts; it inserts its own lgkmcnt waits in front of the MFMAs.
//
// Tile end.  Waves 0-3 run the epilogue after the closing barrier of their last phase, waves 4-7 BEFORE theirs: both
// epilogues then fall into the same inter-barrier interval (the partner's last eight MFMAs lead in) instead of one after
// the other.  Side data of the Q rows (norms, thresholds) reaches LDS by DMA too (an ordinary load beside LDS-DMAs makes
// the compiler wait vmcnt(0) at its first use): Epi::aux_dma(t, qtile, wave) issues at most one DMA instruction per wave
// into the raw buffer of tile parity t & 1 in the first phase of tile t-1, Epi::aux_cook(t, qtile) (threads 0-255: waves
// 0-3) turns it into the form finish() reads six phases later, finish(t, qtile, acc) is the epilogue proper.
#pragma once

namespace am {

// descriptor over elements j0 .. j0+255 of a float array of n elements (elements past the end read as zero)
__device__ __forceinline__ TileRsrc make_aux_rsrc(const float* base, int64_t n, int64_t j0) {
    int64_t valid = n - j0;
    valid = valid < 0 ? 0 : (valid > WTB ? WTB : valid);
    const float* p = base + (valid > 0 ? j0 : 0);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<uintptr_t>(p) & 0xffffffffu));
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<uintptr_t>(p) >> 32));
    const unsigned bytes = __builtin_amdgcn_readfirstlane((unsigned)(valid * 4));
    void* q = reinterpret_cast<void*>((static_cast<uintptr_t>(hi) << 32) | lo);
    TileRsrc r;
    r.rsrc = __builtin_amdgcn_make_buffer_rsrc(q, 0, (int)bytes, 0x00020000);
    return r;
}

// 64 consecutive floats of `r` starting at element e0 -> LDS dst[0..63], one per lane; COHERENT: read at device scope
// (values other workgroups update while this kernel runs)
template <bool COHERENT>
__device__ __forceinline__ void lds_direct_b32(const TileRsrc& r, float* dst, int lane, int e0) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r.rsrc, (__attribute__((address_space(3))) void*)dst, 4, lane * 4, e0 * 4, 0,
                                             COHERENT ? 16 : 0);
}

#define AM_WAIT_VMCNT(n) __builtin_amdgcn_s_waitcnt(0x0F70 | (n))      /* n < 16; expcnt / lgkmcnt untouched */
#define AM_PHASE_BARRIER()                       \
    do {                                         \
        __builtin_amdgcn_sched_barrier(0);       \
        __builtin_amdgcn_s_barrier();            \
        __builtin_amdgcn_sched_barrier(0);       \
    } while (0)

template <class TileMap, class Epi>
__device__ __forceinline__ void wide_pipeline(const float* __restrict__ Q, int64_t nq, int64_t ldq, const TileMap& tmap,
                                              const float* __restrict__ P, int64_t np, int64_t ldp, int64_t prow0,
                                              int ntiles, int Dh, float* __restrict__ lds, const WLane& L, Epi& epi) {
    const int nk = Dh / WROW;
    const int G = ntiles * nk;
    const int wave = __builtin_amdgcn_readfirstlane(L.wave);
    const bool second = wave >= 4;                                 // the group that runs half a phase behind
    // DMA source: lane l of a wave-instruction fetches 16 B for slot l & 7 of LDS row base + (l >> 3); every row base used
    // below is 8 (wave & 1) mod 16, so the swizzle key (row >> 1) & 7 of that row is 4 (wave & 1) + (l >> 4)
    const int key = 4 * (wave & 1) + (L.lane >> 4);
    const int chunk = (L.lane & 7) ^ key;
    const unsigned voq = (unsigned)(((int64_t)(L.lane >> 3) * ldq + chunk * 4) * 4);
    const unsigned vop = (unsigned)(((int64_t)(L.lane >> 3) * ldp + chunk * 4) * 4);
    const unsigned rq = (unsigned)(ldq * 4), rp = (unsigned)(ldp * 4);          // bytes per source row
    const TileRsrc prs = make_wide_rsrc(P, ldp, np, prow0);
    const int64_t q_tiles_total = (nq + WTB - 1) / WTB;
    auto qtile_of = [&](int t) -> int64_t { return t < ntiles ? tmap(t) : q_tiles_total; };   // past the end: empty descriptor
    int ft = 0, fkt = 0, fbuf = 0;                                 // (tile, k-slab, buffer) of the slab being staged
    TileRsrc qrs = make_wide_rsrc(Q, ldq, nq, qtile_of(0) * WTB);
    auto advance_fetch = [&]() {
        fbuf ^= 1;
        if (++fkt == nk) {
            fkt = 0;
            ++ft;
            qrs = make_wide_rsrc(Q, ldq, nq, qtile_of(ft) * WTB);
        }
    };
    // two DMA instructions per wave = one half-slab of the slab being staged
    auto stage_q = [&](int second_half) {
#ifdef AM_PH_NODMA
        if (fbuf >= 0) return;
#endif
        const unsigned so = (unsigned)(fkt * WROW * 4);
        float* s = lds + fbuf * WSTAGE_WORDS;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = i * 128 + second_half * 64 + wave * 8;
            lds_direct_b128(qrs, s + row * WROW, voq, so + (unsigned)row * rq);
        }
    };
    auto stage_p = [&](int second_half) {
#ifdef AM_PH_NODMA
        if (fbuf >= 0) return;
#endif
        const unsigned so = (unsigned)(fkt * WROW * 4);
        float* s = lds + fbuf * WSTAGE_WORDS + WTILE_WORDS;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int b = i * 8 + wave;
            const int row = (b >> 2) * 64 + second_half * 32 + (b & 3) * 8;
            lds_direct_b128(prs, s + row * WROW, vop, so + (unsigned)row * rp);
        }
    };
    // fragment addresses: logical 16-B chunk 2c+h of row r sits in slot (2c+h) ^ ((r >> 1) & 7)
    const int sw = (L.r >> 1) & 7;
    int coff[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) coff[c] = ((2 * c + L.h) ^ sw) * 4;
    const int qrow = (L.wm * 128 + L.r) * WROW;
    const int prow = WTILE_WORDS + (L.wn * 64 + L.r) * WROW;
    f32x4 fq[2][4], fp0[4], fp1[4];
    bool reads_on = true;
    auto read_q = [&](const float* st, int half) {
        if (!reads_on) return;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int m = 0; m < 2; ++m) fq[m][c] = *reinterpret_cast<const f32x4*>(st + qrow + (half * 2 + m) * 32 * WROW + coff[c]);
    };
    auto read_p = [&](const float* st, f32x4 (&f)[4], int half) {
        if (!reads_on) return;
#pragma unroll
        for (int c = 0; c < 4; ++c) f[c] = *reinterpret_cast<const f32x4*>(st + prow + half * 32 * WROW + coff[c]);
    };
    f32x16 acc[4][2];
    wide_zero(acc);
    // eight MFMAs of one quadrant: Q half `qh` (accumulator rows 2 qh, 2 qh + 1) x P half `n`; FIRST: C = 0 as an inline
    // constant in the first chunk (first slab of a tile: no accumulator clears)
    auto quadrant = [&](const f32x4 (&fp)[4], auto first, auto qh, auto n) {
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        constexpr int QH = decltype(qh)::value, N = decltype(n)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int m = 0; m < 2; ++m)
                acc[2 * QH + m][N] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fq[m][c]), __builtin_bit_cast(f16x8, fp[c]),
                                                                          (decltype(first)::value && c == 0) ? zero : acc[2 * QH + m][N], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using T = std::true_type;
    using F = std::false_type;

    // ---- prologue: side data of tile 0, slab 0 completely, the first two half-slabs of slab 1 -------------------------
    epi.aux_dma(0, qtile_of(0), wave);
    stage_q(0);
    stage_p(0);
    stage_p(1);
    stage_q(1);
    advance_fetch();
    stage_q(0);
    stage_p(0);
    AM_WAIT_VMCNT(0);
    __syncthreads();                                // also publishes the caller's LDS initialisation
    epi.aux_cook(0, qtile_of(0));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (second) AM_PHASE_BARRIER();

#ifdef AM_WIDE_STAMPS                               // (a third build: -DAM_DEV_KNOBS -DAM_WIDE_STAMPS, tools/wide_trace.py)
    // s_memtime stamps of waves 0 and 4 (lane 0) of the first 64 workgroups, 96 phases x 6 stamps, collected in LDS (bytes
    // 150 K .. 157.5 K: the A/B build launches with the full 160 KB) and copied out at the end: tools/wide_trace.py
    unsigned long long* tr = nullptr;
    if (g_wide_trace != nullptr && blockIdx.x < 64 && (wave == 0 || wave == 4) && L.lane == 0)
        tr = reinterpret_cast<unsigned long long*>(lds + 38400) + (wave >> 2) * 96 * 6;
    unsigned long long sv[5] = {0, 0, 0, 0, 0};
    int ph = 0;
#define WIDE_CLK0() sv[0] = __builtin_readcyclecounter()
#if AM_WIDE_STAMPS >= 6                             // all six stamps: ~300 cycles of instrumentation per phase
#define WIDE_CLK(k) sv[k] = __builtin_readcyclecounter()
#define WIDE_PHASE_END()                                                               \
    do {                                                                               \
        const unsigned long long s5 = __builtin_readcyclecounter();                    \
        if (tr != nullptr && ph < 96) {                                                \
            unsigned long long* d = tr + ph * 6;                                       \
            d[0] = sv[0]; d[1] = sv[1]; d[2] = sv[2]; d[3] = sv[3]; d[4] = sv[4]; d[5] = s5;          \
        }                                                                              \
        ++ph;                                                                          \
    } while (0)
#else                                               // the phase start and ONE more stamp (s1 .. s4 for AM_WIDE_STAMPS = 1 .. 4)
#define WIDE_CLK(k) do { if ((k) == AM_WIDE_STAMPS) sv[k] = __builtin_readcyclecounter(); } while (0)
#define WIDE_PHASE_END()                                                               \
    do {                                                                               \
        if (tr != nullptr && ph < 96) {                                                \
            tr[ph * 6] = sv[0];                                                          \
            tr[ph * 6 + 1] = sv[AM_WIDE_STAMPS];                                      \
        }                                                                              \
        ++ph;                                                                          \
    } while (0)
#endif
#else
#define WIDE_CLK0() do { } while (0)
#define WIDE_CLK(k) do { } while (0)
#define WIDE_PHASE_END() do { } while (0)
#endif
#ifdef AM_PH_NOREAD
    read_q(lds, 0);
    read_p(lds, fp0, 0);
    read_p(lds, fp1, 1);
    reads_on = false;
#endif
    int t = 0, kt = 0;
    int cook_t = 1, cook_g = 1;                     // tile t is cooked in phase 2 of slab (t - 1) nk + 1: >= 5 phases after its DMA
    for (int g = 0; g < G; ++g) {
        const float* st = lds + (g & 1) * WSTAGE_WORDS;
        const bool last_k = kt == nk - 1;
        // -- phase 1 --
        WIDE_CLK0();
        read_p(st, fp0, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_q(st, 0);
        stage_p(1);
        if (kt == 0) epi.aux_dma(t + 1, qtile_of(t + 1), wave);    // (past the last tile: an empty descriptor)
        WIDE_CLK(1);
        AM_WAIT_VMCNT(8);
        WIDE_CLK(2);
        AM_PHASE_BARRIER();
        WIDE_CLK(3);
        if (kt == 0) quadrant(fp0, T{}, I0{}, I0{});
        else quadrant(fp0, F{}, I0{}, I0{});
        WIDE_CLK(4);
        AM_PHASE_BARRIER();
        WIDE_PHASE_END();
        // -- phase 2 --
        WIDE_CLK0();
        read_p(st, fp1, 1);
        stage_q(1);
        if (g == cook_g) {
            if (cook_t < ntiles) {
                epi.aux_cook(cook_t, qtile_of(cook_t));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            ++cook_t;
            cook_g += nk;
        }
        WIDE_CLK(1);
        AM_WAIT_VMCNT(8);
        WIDE_CLK(2);
        AM_PHASE_BARRIER();
        WIDE_CLK(3);
        if (kt == 0) quadrant(fp1, T{}, I0{}, I1{});
        else quadrant(fp1, F{}, I0{}, I1{});
        WIDE_CLK(4);
        AM_PHASE_BARRIER();
        WIDE_PHASE_END();
        // -- phase 3 --
        WIDE_CLK0();
        read_q(st, 1);
        advance_fetch();
        stage_q(0);
        WIDE_CLK(1);
        AM_WAIT_VMCNT(8);
        WIDE_CLK(2);
        AM_PHASE_BARRIER();
        WIDE_CLK(3);
        if (kt == 0) quadrant(fp1, T{}, I1{}, I1{});
        else quadrant(fp1, F{}, I1{}, I1{});
        WIDE_CLK(4);
        AM_PHASE_BARRIER();
        WIDE_PHASE_END();
        // -- phase 4 --
        WIDE_CLK0();
        stage_p(0);
        WIDE_CLK(1);
        AM_WAIT_VMCNT(8);
        WIDE_CLK(2);
        AM_PHASE_BARRIER();
        WIDE_CLK(3);
        if (kt == 0) quadrant(fp0, T{}, I1{}, I0{});
        else quadrant(fp0, F{}, I1{}, I0{});
        WIDE_CLK(4);
        if (!(last_k && second)) AM_PHASE_BARRIER();
        if (last_k) epi.finish(t, qtile_of(t), acc);
        if (last_k && second) AM_PHASE_BARRIER();
        WIDE_PHASE_END();
        if (last_k) {
            kt = 0;
            ++t;
        } else {
            ++kt;
        }
    }
    // the staging stream ran two slabs past the end (empty Q descriptors, P rows nobody reads): drain it before the
    // caller reuses the buffers; waves 0-3 make up for the extra barrier waves 4-7 passed at the start
    AM_WAIT_VMCNT(0);
    if (!second) AM_PHASE_BARRIER();
    __syncthreads();
#ifdef AM_WIDE_STAMPS
    if (tr != nullptr) {
        unsigned long long* out = g_wide_trace + ((size_t)blockIdx.x * 2 + (wave >> 2)) * 96 * 6;
        for (int i = 0; i < 96 * 6; ++i) out[i] = (i < ph * 6 && (AM_WIDE_STAMPS >= 6 || i % 6 < 2)) ? tr[i] : 0ull;
    }
#endif
}

}  // namespace am
