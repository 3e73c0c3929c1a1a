// Operand-stationary form of the 256-row f16 filter engine (round 5): the workgroup's P block never moves again.
//
// wide_engine.h streams BOTH operands of a 256 x 256 tile through LDS: 64 KB of LDS-DMA and 24 fragment reads per wave for
// every 8.4 MFLOP of a CU, and its stage takes 2.18 us (L2-missing operands, the real kernels' regime) for 1.0 us of MFMA
// time - the fill path, not the instruction order, bounds it (rounds 2-4).  A filter kernel's work item pairs ONE
// 256-row P block with MANY Q tiles, so here the P block is read once, straight into registers, as MFMA B fragments:
//   * 512 threads = 8 wave64; wave w owns P rows 32 w .. 32 w + 31 for the whole inner dimension: KSLABS * 4 fragments of
//     16 B per lane (128 registers at D = 512), loaded with buffer loads at the start of the work item and never re-read;
//   * a stage is a 64-element slab of 128 Q rows (half a 256-row tile) = 16 KB, eight of them make a unit (half tile x D = 512);
//     a wave multiplies its 32 P rows with all 128 Q rows: 4 x 1 MFMA tiles, 64 accumulator registers, 16 MFMAs and 16
//     fragment reads per stage (Q only);
//   * the Q slabs travel through a ring of four 16-KB LDS slots by LDS-DMA, three stages ahead, two pieces per wave and
//     stage, with COUNTED vmcnt and raw s_barrier: the barrier at the end of stage g publishes stage g + 2, so the first
//     fragments of stage g + 1 are read under the last MFMAs of stage g and no DMA is ever drained inside a unit;
//   * one set of four Q fragments per wave: fragment m of the next k-step is read right behind the MFMA that used fragment m.
// Per 8.4 MFLOP the load path moves 32 KB instead of 64 KB, and the L2 of an XCD only has to hold the Q window its 32
// workgroups walk in lockstep - the 32 P blocks (8 MB against 4 MB of L2) that thrashed it are gone.
// tools/ubench/pstat.hip (random f16 operands, same unit): 1.55-1.60 us per 8.4 MFLOP against 1.73 (L2-resident source) /
// 2.19 (200 MB source) for wide_engine.h's adopted schedule on the same box (profiles/r5/ubench_pstat.txt).
//
// LDS image of a slot: 128 rows of 128 B, 16-B chunks XOR-swizzled by (row >> 1) & 7 (wide_engine.h).  MFMA roles, k order and
// therefore every accumulator value are those of wide_engine.h: P rows lane-local (MFMA column = lane & 31), Q rows in the
// registers, logical chunk 2c + h of slab kt in k-step 4 kt + c.
//
// Hazards (MI355X_MICROARCH.md, "LDS-DMA requests stay in flight across s_barrier"):
//   RAW  stage g + 2 is read (first: the pre-read at the end of stage g + 1) only after the barrier that ends stage g, before
//        which every wave has waited for its own pieces of that stage (vmcnt(PIECES): only the pieces of stage g + 3, issued
//        during stage g, may still be in flight);
//   WAR  the pieces of stage g + 3 are issued during stage g into the slot stage g - 1 was read from; every wave's reads of
//        stage g - 1 had returned (they fed its MFMAs) before it arrived at the barrier that ends stage g - 1.
#pragma once
#include "wide_engine.h"

namespace am {

constexpr int PQ_ROWS = 128;                          // Q rows of a stage: half a 256-row tile
constexpr int PSTAGE_WORDS = PQ_ROWS * WROW;          // 16 KB
constexpr int PRING = 4;
constexpr int PTHREADS = 512;
// Of a P row's KSLABS 64-element slabs the first PREG_SLABS live in registers (16 per slab); slabs 7 and 8 would leave
// the epilogues 48 registers beside 64 accumulators and the compiler spills P fragments into the main loop - they stay in LDS
// instead (32 KB per slab, each wave reading only its own 32 rows: no barrier involved), one extra ds_read_b128 per k-step.
constexpr int PREG_SLABS = 6;
constexpr int pstat_lds_slabs(int kslabs) { return kslabs > PREG_SLABS ? kslabs - PREG_SLABS : 0; }
constexpr int pstat_lds_words(int kslabs) { return PRING * PSTAGE_WORDS + pstat_lds_slabs(kslabs) * WTB * WROW; }   // 64 KB + 32 KB per LDS slab

// Lane geometry the epilogues see.  `wm` = which 128-row half of the Q tile the accumulators of the CURRENT finish() call belong
// to (the engine sets it; in wide_engine.h it is a property of the wave), NT = 32-row P tiles per wave.
struct PLane {
    static constexpr int NT = 1;
    static constexpr int MT = 4;                      // 32-row Q tiles of a unit (accumulator tiles per P tile)
    static constexpr int WAVES = 8;
    static constexpr int LISTS = 2;                   // partial per-row lists after the sweep: the two lane halves
    // The accumulators of a unit do not start at zero but at a per-COLUMN value the epilogue supplies (acc_init): the k-NN
    // sweep and the membership filter start them at -|y_j|^2 / 2 in the units of the scaled dot product, so that what comes out
    // is <x, y_j> - |y_j|^2 / 2 and the epilogue's fast path needs no fma per element to add the column norm.  The vector ALU
    // and the matrix pipe of a SIMD do not overlap here (profiles/r5/wide_bench_nomargin.txt: one vector pass per accumulator
    // element = 0.15 ms per sweep at 100 000 rows, at D = 128 and at D = 512 alike), and the start values are the same LDS
    // reads the epilogue did anyway.
    static constexpr bool ACC_INIT = true;
    int tid, lane, wave, wm, r, h;
    __device__ __forceinline__ PLane() {
        tid = threadIdx.x;
        lane = tid & 63;
        wave = tid >> 6;
        wm = 0;
        r = lane & 31;
        h = lane >> 5;
    }
    __device__ __forceinline__ int prow(int) const { return wave * 32 + r; }          // row of the P block this lane owns
    __device__ __forceinline__ int list_slot() const { return h; }
};

template <int N>
__device__ __forceinline__ void pstat_wait() {        // vmcnt(N) and lgkmcnt(0): the stage's own LDS writes (side data) are out too
    static_assert(N >= 0 && N < 64, "vmcnt");
    __builtin_amdgcn_s_waitcnt(0x0070 | (N & 15) | ((N >> 4) << 14));
}

// Q, P: f16 matrices viewed as f32 words (ld in words), rows of KSLABS * 32 words.  tmap(t) = index of the 256-row Q tile that
// local tile t multiplies; P block = rows prow0 .. prow0 + 255.  Epi:
//   aux_issue(t, qtile) / aux_commit(t)   per-TILE side data through LDS (as in wide_engine.h)
//   finish(t, qtile, acc[4][1])           once per HALF tile, L.wm telling which
//   acc_init(t, m)  (Epi::ACC_INIT)       start value of accumulator tile m (32 Q rows) of the half tile L.wm of local tile t
template <int KSLABS, class TileMap, class Epi>
__device__ __forceinline__ void pstat_pipeline(const float* __restrict__ Q, int64_t nq, int64_t ldq, const TileMap& tmap,
                                               const float* __restrict__ P, int64_t np, int64_t ldp, int64_t prow0, int ntiles,
                                               float* __restrict__ lds, PLane& L, Epi& epi) {
    static_assert(KSLABS >= 1 && KSLABS <= 8, "P fragments of at most 512 f16 per row fit the register file");
    constexpr int KREG = KSLABS < PREG_SLABS ? KSLABS : PREG_SLABS;      // slabs of the P rows held in registers
    constexpr int KLDS = KSLABS - KREG;                                   // ... and in LDS
    constexpr int KSTEPS = KREG * 4;
    constexpr int PIECES = 2;                         // 1-KB LDS-DMA pieces per wave and stage
    constexpr int DEPTH = 3;                          // stages in flight ahead of the one being multiplied
    const int wave = __builtin_amdgcn_readfirstlane(L.wave);

    // ---- the stationary operand
    f32x4 pf[KSTEPS];
    {
        const TileRsrc prs = make_wide_rsrc(P, ldp, np, prow0, 0);
        const unsigned vop = (unsigned)(((int64_t)(wave * 32 + L.r) * ldp + L.h * 4) * 4);
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
            pf[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs.rsrc, (int)vop, s * 32, 0));
        // the slabs kept in LDS: [slab][256 rows][128 B], swizzled like a Q slot; this wave's 32 rows = 4 pieces per slab
        const int lr8 = L.lane >> 3, s8 = L.lane & 7;
#pragma unroll
        for (int sl = 0; sl < KLDS; ++sl)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wave * 32 + i * 8 + lr8;
                const unsigned vo = (unsigned)(((int64_t)row * ldp + (s8 ^ ((row >> 1) & 7)) * 4) * 4);
                lds_direct_b128(prs, lds + PRING * PSTAGE_WORDS + sl * WTB * WROW + (wave * 32 + i * 8) * WROW, vo, (unsigned)((KREG + sl) * 128));
            }
    }
    const float* plds = lds + PRING * PSTAGE_WORDS + wave * 32 * WROW;     // this wave's rows of LDS slab 0

    // ---- Q slabs: piece p = 8 rows x 128 B; wave w fetches pieces 2 w, 2 w + 1 (rows 16 w .. 16 w + 15 of the half tile)
    const int lr = L.lane >> 3, slot8 = L.lane & 7;
    // even piece (rows lr of the wave's sixteen); the odd piece's rows lie 8 further down and their swizzle (row >> 1) & 7
    // differs in bit 2 of the chunk index = bit 6 of the byte offset (rows are multiples of 128 B): one register, one XOR
    const unsigned voq0 = (unsigned)(((int64_t)lr * ldq + (slot8 ^ ((lr >> 1) & 7)) * 4) * 4);
    const unsigned odd_soff = (unsigned)((int64_t)8 * ldq * 4);
    const unsigned wave_soff = (unsigned)((int64_t)wave * 16 * ldq * 4), half_soff = (unsigned)((int64_t)PQ_ROWS * ldq * 4);
    const int64_t q_tiles_total = (nq + WTB - 1) / WTB;
    auto qtile_of = [&](int t) -> int64_t { return t < ntiles ? tmap(t) : q_tiles_total; };   // past the end: empty descriptor
    int ft = 0, fh = 0, fk = 0, fslot = 0;            // (tile, half, slab, ring slot) of the stage being fetched
    TileRsrc qrs = make_wide_rsrc(Q, ldq, nq, qtile_of(0) * WTB);
    auto fetch_stage = [&](int i) {                   // piece i of the stage under the fetch cursor
        float* dst = lds + fslot * PSTAGE_WORDS + (wave * 2 + i) * 8 * WROW;
        lds_direct_b128(qrs, dst, i == 0 ? voq0 : (voq0 ^ 64u), (unsigned)(fk * 128) + (fh ? half_soff : 0u) + wave_soff + (i == 0 ? 0u : odd_soff));
    };
    auto advance_fetch = [&]() {
        fslot = (fslot + 1) & (PRING - 1);
        if (++fk == KSLABS) {
            fk = 0;
            fh ^= 1;
            if (fh == 0) {
                ++ft;
                qrs = make_wide_rsrc(Q, ldq, nq, qtile_of(ft) * WTB);
            }
        }
    };

    // fragment addresses: logical 16-B chunk 2c+h of row r sits in slot (2c+h) ^ ((r >> 1) & 7)
    const int sw = (L.r >> 1) & 7;
    int coff[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) coff[c] = L.r * WROW + ((2 * c + L.h) ^ sw) * 4;
    auto qfrag = [&](int ring_slot, int c, int m) -> f32x4 {
        return *reinterpret_cast<const f32x4*>(lds + ring_slot * PSTAGE_WORDS + m * 32 * WROW + coff[c]);
    };

    f32x16 acc[4][1];
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // prologue: DEPTH stages in flight, the first two landed and published
#pragma unroll
    for (int g = 0; g < DEPTH; ++g) {
        fetch_stage(0);
        fetch_stage(1);
        advance_fetch();
    }
    epi.aux_issue(0, qtile_of(0));
    __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0): side data (and the P fragments) are here
    epi.aux_commit(0);
    pstat_wait<0>();
    __builtin_amdgcn_s_barrier();

    // ONE fragment set: the fragment of row tile m for the next k-step is read right behind the MFMA that used this one
    // (ubench: as fast as two sets read a k-step ahead - 1.49-1.50 us per 8.4 MFLOP - and sixteen registers less)
    f32x4 q[4];
    f32x4 pl[2];                                      // P fragments of the LDS slabs, read one k-step ahead
    int slot = 0;                                     // ring slot of the stage being multiplied
#pragma unroll
    for (int m = 0; m < 4; ++m) q[m] = qfrag(0, 0, m);
    if constexpr (Epi::ACC_INIT) {
        L.wm = 0;
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m][0] = epi.acc_init(0, m);
    }
    const int units = 2 * ntiles;
    for (int u = 0; u < units; ++u) {
        const int t = u >> 1;
        const bool second = (u & 1) != 0;
#pragma unroll
        for (int ks = 0; ks < KSLABS; ++ks) {
            const bool last_stage = ks == KSLABS - 1;
            const int next_slot = (slot + 1) & (PRING - 1);
            if (last_stage && second && t + 1 < ntiles) epi.aux_issue(t + 1, qtile_of(t + 1));
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int step = ks * 4 + c;          // k-step of the unit; steps >= KSTEPS take their P fragment from LDS
                if (KLDS > 0 && step + 1 >= KSTEPS && step + 1 < KSLABS * 4)
                    pl[(step + 1) & 1] = *reinterpret_cast<const f32x4*>(plds + ((step + 1) / 4 - KREG) * WTB * WROW + coff[(step + 1) & 3]);
                if (KLDS > 0 && step == 0 && KSTEPS == 0) pl[0] = *reinterpret_cast<const f32x4*>(plds + coff[0]);
                const f32x4 pfrag = step < KSTEPS ? pf[step < KSTEPS ? step : 0] : pl[step & 1];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, q[m]), __builtin_bit_cast(f16x8, pfrag),
                                                                         (!Epi::ACC_INIT && ks == 0 && c == 0) ? zero : acc[m][0], 0, 0, 0);
                    if (c < 3) q[m] = qfrag(slot, c + 1, m);
                    else if (!last_stage) q[m] = qfrag(next_slot, 0, m);     // published by the barrier of the stage before
                    if (m == 1 && (c & 1) == 0) fetch_stage(c >> 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            advance_fetch();
            if (last_stage) {
                L.wm = second ? 1 : 0;
                epi.finish(t, qtile_of(t), acc);
                if (second && t + 1 < ntiles) epi.aux_commit(t + 1);
            }
            // this wave's pieces of stage g + 2 have landed; the barrier publishes that stage
            pstat_wait<PIECES>();
            __builtin_amdgcn_s_barrier();
            slot = next_slot;
            if (last_stage) {                         // a unit's first fragments are read behind its predecessor's epilogue:
#pragma unroll                                        // no fragment register is live across the epilogue
                for (int m = 0; m < 4; ++m) q[m] = qfrag(slot, 0, m);
                // ... and, ACC_INIT, its accumulators' start values (side data of a new tile: committed before the barrier above).
                // Reading them a unit earlier, in front of the barrier, was measured: the same time (profiles/r5/wide_bench_accinit_ab.txt).
                if constexpr (Epi::ACC_INIT) {
                    if (u + 1 < units) {
                        L.wm = second ? 0 : 1;
#pragma unroll
                        for (int m = 0; m < 4; ++m) acc[m][0] = epi.acc_init((u + 1) >> 1, m);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);               // the (empty) fetches past the end: nothing in flight when LDS is reused
    __syncthreads();
}

// ---- the stationary engine for NARROW rows (round 6): KSLABS <= P64_MAX_SLABS, i.e. D <= 128 (VGGish, n_pca = 64 ...) -------------
//
// At D = 128 a unit of the engine above is two stages long and its epilogue (vector ALU, matrix pipe idle) is 55 % of the k-NN
// sweep.  VERDICT r5 asked for the epilogue of one wave to run under the MFMAs of its SIMD partner.  What the
// microbenchmark found (tools/ubench/pskew.hip, profiles/r6/ubench_pskew_v*.txt):
//   * beside an OLDER wave that issues MFMAs a wave's vector instructions get about half their issue slots (s_setprio changes
//     nothing; with the vector wave the older one both run at their own pace), and the chip sits at its power limit while the
//     matrix pipe runs (1.5-1.65 GHz against 2.3 GHz in a vector-only phase): overlap buys cycles, the clock gives part back;
//   * skewing waves 0-3 against 4-7 on ONE Q stream with ONE barrier per stage is slower than lockstep (the wave in its
//     epilogue is alone on the vector ALU at ~5 cycles per instruction and every other wave waits for it at the barrier);
//     LDS-flag synchronisation instead of s_barrier costs more than it frees;
//   * what does pay where the registers allow it: TWO INDEPENDENT 256-thread workgroups per CU (own ring, own barriers - the two
//     waves of a SIMD drift apart by themselves) whose waves own 64 P rows = TWO row tiles each, against 64-row Q units:
//     every Q fragment read from LDS feeds two MFMAs (half the LDS reads per flop), the LDS-DMA bytes per flop stay those of
//     the 512-thread form, 64 accumulator registers, and the P fragments of two row tiles (32 registers per slab) fit beside
//     them up to two slabs.  Synthetic epilogue, us per 256 MFMAs of a CU: D = 128 3.17 -> 2.43 (epilogue of 4 passes), 3.75 ->
//     2.98 (6 passes); D = 64 5.15 -> 4.49.
// Same ring (four slots, three stages ahead, counted vmcnt, raw s_barrier), same LDS image and swizzle, same k order: the
// accumulator values are those of pstat_pipeline, bit for bit.  A stage is 64 Q rows x 64 elements = 8 KB (two pieces per wave,
// as above); a 256-row Q tile is four units (L.wm = 0 .. 3).
constexpr int P64_THREADS = 256;
constexpr int P64_QROWS = 64;
constexpr int P64_STAGE_WORDS = P64_QROWS * WROW;    // 8 KB
constexpr int P64_MAX_SLABS = 2;
constexpr int pstat64_lds_words() { return PRING * P64_STAGE_WORDS; }

struct P64Lane {
    static constexpr int NT = 2;                      // 32-row P tiles per wave
    static constexpr int MT = 2;                      // 32-row Q tiles of a unit
    static constexpr int WAVES = 4;
    static constexpr int LISTS = 2;
    static constexpr bool ACC_INIT = true;
    int tid, lane, wave, wm, r, h;
    __device__ __forceinline__ P64Lane() {
        tid = threadIdx.x;
        lane = tid & 63;
        wave = tid >> 6;
        wm = 0;
        r = lane & 31;
        h = lane >> 5;
    }
    __device__ __forceinline__ int prow(int nt) const { return wave * 64 + nt * 32 + r; }
    __device__ __forceinline__ int list_slot() const { return h; }
};

template <int KSLABS, class TileMap, class Epi>
__device__ __forceinline__ void pstat64_pipeline(const float* __restrict__ Q, int64_t nq, int64_t ldq, const TileMap& tmap,
                                                 const float* __restrict__ P, int64_t np, int64_t ldp, int64_t prow0, int ntiles,
                                                 float* __restrict__ lds, P64Lane& L, Epi& epi) {
    static_assert(KSLABS >= 1 && KSLABS <= P64_MAX_SLABS, "two row tiles of P fragments beside 64 accumulators");
    static_assert(Epi::ACC_INIT, "the epilogues of this engine start their accumulators at the column norm");
    constexpr int KSTEPS = KSLABS * 4;
    constexpr int PIECES = 2;
    constexpr int DEPTH = 3;
    const int wave = __builtin_amdgcn_readfirstlane(L.wave);

    // ---- the stationary operand: two row tiles
    f32x4 pf[2][KSTEPS];
    {
        const TileRsrc prs = make_wide_rsrc(P, ldp, np, prow0, 0);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const unsigned vop = (unsigned)(((int64_t)(wave * 64 + nt * 32 + L.r) * ldp + L.h * 4) * 4);
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s)
                pf[nt][s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs.rsrc, (int)vop, s * 32, 0));
        }
    }

    // ---- Q slabs: piece p = 8 rows x 128 B; wave w fetches pieces 2 w, 2 w + 1 (rows 16 w .. 16 w + 15 of the unit's 64)
    const int lr = L.lane >> 3, slot8 = L.lane & 7;
    const unsigned voq0 = (unsigned)(((int64_t)lr * ldq + (slot8 ^ ((lr >> 1) & 7)) * 4) * 4);
    const unsigned odd_soff = (unsigned)((int64_t)8 * ldq * 4);
    const unsigned wave_soff = (unsigned)((int64_t)wave * 16 * ldq * 4), unit_soff = (unsigned)((int64_t)P64_QROWS * ldq * 4);
    const int64_t q_tiles_total = (nq + WTB - 1) / WTB;
    auto qtile_of = [&](int t) -> int64_t { return t < ntiles ? tmap(t) : q_tiles_total; };   // past the end: empty descriptor
    int ft = 0, fq = 0, fk = 0, fslot = 0;            // (tile, quarter, slab, ring slot) of the stage being fetched
    TileRsrc qrs = make_wide_rsrc(Q, ldq, nq, qtile_of(0) * WTB);
    auto fetch_stage = [&](int i) {
        float* dst = lds + fslot * P64_STAGE_WORDS + (wave * 2 + i) * 8 * WROW;
        lds_direct_b128(qrs, dst, i == 0 ? voq0 : (voq0 ^ 64u), (unsigned)(fk * 128) + (unsigned)fq * unit_soff + wave_soff + (i == 0 ? 0u : odd_soff));
    };
    auto advance_fetch = [&]() {
        fslot = (fslot + 1) & (PRING - 1);
        if (++fk == KSLABS) {
            fk = 0;
            if (++fq == 4) {
                fq = 0;
                ++ft;
                qrs = make_wide_rsrc(Q, ldq, nq, qtile_of(ft) * WTB);
            }
        }
    };

    const int sw = (L.r >> 1) & 7;
    int coff[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) coff[c] = L.r * WROW + ((2 * c + L.h) ^ sw) * 4;
    auto qfrag = [&](int ring_slot, int c, int m) -> f32x4 {
        return *reinterpret_cast<const f32x4*>(lds + ring_slot * P64_STAGE_WORDS + m * 32 * WROW + coff[c]);
    };

    f32x16 acc[2][2];

#pragma unroll
    for (int g = 0; g < DEPTH; ++g) {
        fetch_stage(0);
        fetch_stage(1);
        advance_fetch();
    }
    epi.aux_issue(0, qtile_of(0));
    __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0): side data (and the P fragments) are here
    epi.aux_commit(0);
    pstat_wait<0>();
    __builtin_amdgcn_s_barrier();

    f32x4 q[2];
    int slot = 0;
#pragma unroll
    for (int m = 0; m < 2; ++m) q[m] = qfrag(0, 0, m);
    L.wm = 0;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const f32x16 c0 = epi.acc_init(0, m);
        acc[m][0] = c0;
        acc[m][1] = c0;
    }
    const int units = 4 * ntiles;
    for (int u = 0; u < units; ++u) {
        const int t = u >> 2, quarter = u & 3;
#pragma unroll
        for (int ks = 0; ks < KSLABS; ++ks) {
            const bool last_stage = ks == KSLABS - 1;
            const int next_slot = (slot + 1) & (PRING - 1);
            if (last_stage && quarter == 3 && t + 1 < ntiles) epi.aux_issue(t + 1, qtile_of(t + 1));
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int step = ks * 4 + c;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, q[m]), __builtin_bit_cast(f16x8, pf[nt][step]),
                                                                              acc[m][nt], 0, 0, 0);
                    if (c < 3) q[m] = qfrag(slot, c + 1, m);
                    else if (!last_stage) q[m] = qfrag(next_slot, 0, m);     // published by the barrier of the stage before
                    if (m == 1 && (c & 1) == 0) fetch_stage(c >> 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            advance_fetch();
            if (last_stage) {
                L.wm = quarter;
                epi.finish(t, qtile_of(t), acc);
                if (quarter == 3 && t + 1 < ntiles) epi.aux_commit(t + 1);
            }
            pstat_wait<PIECES>();
            __builtin_amdgcn_s_barrier();
            slot = next_slot;
            if (last_stage) {
#pragma unroll
                for (int m = 0; m < 2; ++m) q[m] = qfrag(slot, 0, m);
                if (u + 1 < units) {
                    L.wm = (u + 1) & 3;
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const f32x16 c0 = epi.acc_init((u + 1) >> 2, m);
                        acc[m][0] = c0;
                        acc[m][1] = c0;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);               // the (empty) fetches past the end: nothing in flight when LDS is reused
    __syncthreads();
}

}  // namespace am
