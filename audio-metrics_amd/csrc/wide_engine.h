// 256 x 256 f16 tile engine on the gfx950 matrix cores (v_mfma_f32_32x32x16_f16) for the FILTER passes of
// pairwise_fast.h.
//
// Why a second engine.  The f16 MFMA retires 16x the inner products of the f32 one per cycle, so per unit of time a
// filter kernel on the 128 x 128 engine of tile_engine.h asks 16x more of everything around the matrix core:
// measured on the membership filter (2 x 100k x 512), 38 % MFMA-busy with the LDS store path (ds_write_b128: 13
// cycles per wave-instruction, two SIMD halves) at ~80 % of its rate, a barrier every 512 MFMA cycles, and
// L2 -> LDS fill traffic of 160 GB per launch.  This engine changes the ratios instead of the schedule:
//   * workgroup tile 256 (Q) x 256 (P), 512 threads = 8 wave64 as 2 (Q halves) x 4 (P quarters); a wave owns
//     128 x 64 = 4 x 2 MFMA tiles (128 accumulator registers): fill bytes per flop are halved, LDS fragment reads
//     per MFMA drop from 1 to 0.75 (6 x ds_read_b128 per 8 MFMAs);
//   * operands go global -> LDS directly (buffer_load_dwordx4 ... lds): no staging registers, no ds_write;
//   * a stage is a 128-byte slab (64 f16) of all 512 rows = 64 KB, two stages = 128 KB of the CU's 160 KB LDS, one
//     workgroup per CU (two waves per SIMD); a wave issues 32 MFMAs (1024 MFMA cycles) per barrier; the fill of
//     stage g+1 is in flight while stage g multiplies.  (A four-stage ring of 64-byte slabs with the fills three
//     stages ahead was built and measured: bit-identical, 10 % SLOWER - twice the barriers, and the fill latency is
//     not what bounds the kernel.)
// LDS image: unpadded 128-B rows, 16-B chunks XOR-swizzled by ((row >> 1) & 7) - a wave's DMA instruction writes
// 64 x 16 B contiguously (8 whole rows) while the MFMA fragment reads (row = lane & 31, chunk = 2c + (lane >> 5)) stay
// bank-conflict free: a row starts at bank 32 (row & 1), so the 16 lanes of a ds_read_b128 group (rows
// {0-3, 12-15, 20-27} or {4-11, 16-19, 28-31}) must land on 16 different (row & 1, slot) pairs, which the swizzle by
// row >> 1 gives (a swizzle by row & 7 ties the slot's parity to the row's and is 2-way conflicted: measured,
// SQ_LDS_BANK_CONFLICT 9.4e8 -> 0).
// Operand roles as in tile_engine.h: P rows are lane-local (MFMA column = lane & 31), Q rows sit in the registers:
// register r of a lane belongs to Q row (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of its 32 x 32 tile.
// Results feed error-bounded filters only, so the accumulation order inside the instruction is irrelevant.
#pragma once
#include "tile_engine.h"
#include <type_traits>

namespace am {

constexpr int WTB = 256;                              // tile rows of either operand
constexpr int WTHREADS = 512;
constexpr int WROW = 32;                              // LDS row: 32 words = 128 B = 64 f16
constexpr int WTILE_WORDS = WTB * WROW;               // one operand slab, 32 KB
constexpr int WSTAGE_WORDS = 2 * WTILE_WORDS;         // Q slab then P slab
constexpr int WENGINE_LDS_WORDS = 2 * WSTAGE_WORDS;   // two stages, 128 KB

struct WLane {
    static constexpr int NT = 2;                      // 32-row P tiles per wave
    static constexpr int MT = 4;                      // 32-row Q tiles per wave
    static constexpr int WAVES = 8;
    static constexpr int LISTS = 4;                   // partial per-row lists after a sweep: Q half x lane half
    static constexpr bool ACC_INIT = false;
    int tid, lane, wave, wm, wn, r, h;
    __device__ __forceinline__ WLane() {
        tid = threadIdx.x;
        lane = tid & 63;
        wave = tid >> 6;
        wm = wave >> 2;      // which 128-row half of the Q tile
        wn = wave & 3;       // which 64-row quarter of the P tile
        r = lane & 31;
        h = lane >> 5;
    }
    __device__ __forceinline__ int prow(int nt) const { return wn * 64 + nt * 32 + r; }   // row of the P block behind acc[.][nt]
    __device__ __forceinline__ int list_slot() const { return wm * 2 + h; }
};

// descriptor over the valid rows of a 256-row tile starting at row0 (rows past the end read as zero)
__device__ __forceinline__ TileRsrc make_wide_rsrc(const float* base, int64_t ld, int64_t n_rows, int64_t row0, int dbg_bit = 1) {
    int64_t valid = n_rows - row0;
    valid = valid < 0 ? 0 : (valid > WTB ? WTB : valid);
#ifdef AM_DEV_KNOBS
    if ((g_wide_dbg & dbg_bit) && valid > 0) row0 = row0 % (4 * WTB);   // four blocks per operand: 2 MB in all, L2 resident
#endif
    const float* p = base + (valid > 0 ? row0 : 0) * ld;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<uintptr_t>(p) & 0xffffffffu));
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<uintptr_t>(p) >> 32));
    const unsigned bytes = __builtin_amdgcn_readfirstlane((unsigned)(valid * ld * 4));
    void* q = reinterpret_cast<void*>((static_cast<uintptr_t>(hi) << 32) | lo);
    TileRsrc r;
    r.rsrc = __builtin_amdgcn_make_buffer_rsrc(q, 0, (int)bytes, 0x00020000);
    return r;
}

__device__ __forceinline__ void wide_zero(f32x16 (&acc)[4][2]) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
}

// Q, P: f16 matrices viewed as f32 words (ld and Dh in words, Dh % 32 == 0).  tmap(t) = index of the 256-row Q tile
// that local tile t multiplies; P block = rows prow0 .. prow0 + 255.  Epi as in tile_engine.h:
//   aux_issue(t, qtile) / aux_commit(t) : per-tile side data through LDS;  finish(t, qtile, acc[4][2])
//
// Schedule of one stage (k-slab g of 64 f16, four chunks c0..c3 of 8 MFMAs per wave).  An LDS-DMA piece costs its wave
// ~60-180 issue cycles; eight of them in a row at the top of the stage - the obvious order - keep BOTH waves of every
// SIMD off the matrix pipe right after each barrier.  The pieces of stage g+1 are therefore slipped in one per two MFMAs
// during the first sixteen MFMAs of stage g (pinned with sched_barrier: the compiler would hoist them back to the top).
// Measured with tools/ubench/stage_sched.hip / stream_prefetch.hip on random operands, 2 x 100 MB, the membership
// kernel's sharing pattern: 1.88 -> ~1.6 us per stage.  (Carrying the last chunk of a stage across the barrier so that
// its MFMAs cover the first fragment reads gains a further 6 % in the microbenchmark; in the real kernels - both of
// them, also the membership filter at 225 registers - the two-path loop it needs (epilogue before / after the first
// fragment reads) makes the register allocator spill 180-600 registers.  Instead the first MFMA pair of a stage starts
// as soon as ITS three fragments are back and the reads of the second chunk go out behind it.)
// LDS safety: stage g+1 is written into the buffer that stage g-1 was read from; every wave has finished those reads
// (s_waitcnt lgkmcnt(0) of __syncthreads) before the barrier that ends stage g-1, and the pieces are issued after it.
struct WideSingleTile {            // tile map of a workgroup that multiplies exactly one Q tile
    int64_t q;
    __device__ __forceinline__ int64_t operator()(int) const { return q; }
};

// KMap: which 128-byte slab of a row stage kt fetches, per operand (byte offset inside the row).  PlainSlabs walks a row
// front to back; the kernel-distance kernel (kd.hip) lays two f16 planes side by side in a row and makes three stages
// of every 64-element slab - (hi, hi), (lo, hi), (hi, lo) - so that the same pipeline accumulates a split-f16 product.
struct PlainSlabs {
    __device__ __forceinline__ int count(int Dh) const { return Dh / WROW; }
    __device__ __forceinline__ unsigned q(int kt) const { return (unsigned)(kt * WROW * 4); }
    __device__ __forceinline__ unsigned p(int kt) const { return (unsigned)(kt * WROW * 4); }
};

template <class TileMap, class Epi, class KMap = PlainSlabs>
__device__ __forceinline__ void wide_pipeline(const float* __restrict__ Q, int64_t nq, int64_t ldq, const TileMap& tmap,
                                              const float* __restrict__ P, int64_t np, int64_t ldp, int64_t prow0,
                                              int ntiles, int Dh, float* __restrict__ lds, const WLane& L, Epi& epi,
                                              const KMap& kmap = KMap()) {
    const int nk = kmap.count(Dh);
    const int G = ntiles * nk;
    const int wave = __builtin_amdgcn_readfirstlane(L.wave);
    const int srow = L.tid >> 3;                                   // 0..63 (+64 j)
    const int chunk = (L.tid & 7) ^ ((srow >> 1) & 7);             // 16-B chunk of the row this thread fetches into slot tid & 7
    // byte offset of this thread's 16 B inside a 64-row group; the group's own offset (j * 64 rows) is wave-uniform and
    // travels in the instruction's scalar offset (two address registers instead of eight)
    const unsigned voq = (unsigned)(((int64_t)srow * ldq + chunk * 4) * 4), vop = (unsigned)(((int64_t)srow * ldp + chunk * 4) * 4);
    const unsigned gq = (unsigned)(64 * ldq * 4), gp = (unsigned)(64 * ldp * 4);
    const TileRsrc prs = make_wide_rsrc(P, ldp, np, prow0, 1 | 32);     // (bit 32: only the P operand L2-resident)
    const int64_t q_tiles_total = (nq + WTB - 1) / WTB;
    auto qtile_of = [&](int t) -> int64_t { return t < ntiles ? tmap(t) : q_tiles_total; };   // past the end: empty descriptor
    int ft = 0, fkt = 0;                                           // (tile, k-slab) of the stage being fetched
    TileRsrc qrs = make_wide_rsrc(Q, ldq, nq, qtile_of(0) * WTB);
    // piece j of the stage being fetched into buffer `buf`: even j -> 64 Q rows, odd j -> 64 P rows
    auto piece = [&](int buf, int j) {
        float* s = lds + buf * WSTAGE_WORDS + wave * 8 * WROW;
        if ((j & 1) == 0) lds_direct_b128(qrs, s + (j >> 1) * 64 * WROW, voq, kmap.q(fkt) + (unsigned)(j >> 1) * gq);
        else lds_direct_b128(prs, s + WTILE_WORDS + (j >> 1) * 64 * WROW, vop, kmap.p(fkt) + (unsigned)(j >> 1) * gp);
    };
    auto advance_fetch = [&]() {
        if (++fkt == nk) {
            fkt = 0;
            ++ft;
            qrs = make_wide_rsrc(Q, ldq, nq, qtile_of(ft) * WTB);
        }
    };
    // fragment addresses: logical 16-B chunk 2c+h of row r sits in slot (2c+h) ^ ((r >> 1) & 7)
    const int sw = (L.r >> 1) & 7;
    int coff[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) coff[c] = ((2 * c + L.h) ^ sw) * 4;
    const int qrow = (L.wm * 128 + L.r) * WROW;
    const int prow = WTILE_WORDS + (L.wn * 64 + L.r) * WROW;
    struct Frags {
        f32x4 q[4], p[2];
    };
    auto frags = [&](const float* st, int c) {
        Frags f;                                                    // P first: every MFMA of the chunk needs one of them
#pragma unroll
        for (int n = 0; n < 2; ++n) f.p[n] = *reinterpret_cast<const f32x4*>(st + prow + n * 32 * WROW + coff[c]);
#pragma unroll
        for (int m = 0; m < 4; ++m) f.q[m] = *reinterpret_cast<const f32x4*>(st + qrow + m * 32 * WROW + coff[c]);
        return f;
    };
    f32x16 acc[4][2];
    wide_zero(acc);
    // eight MFMAs of one chunk; FIRST: C = 0 as an inline constant (first chunk of a tile, no accumulator clears);
    // DMA: pieces j0 .. j0+3 of the next stage follow the MFMA pairs
    // (M_LO, M_HI are template arguments on purpose: with run-time bounds the accumulator array is indexed dynamically
    // until the loop is unrolled, which is too late for it to be promoted to registers cleanly)
    auto mm = [&](const Frags& f, bool first, bool dma, int buf, int j0, auto m_lo, auto m_hi) {
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = decltype(m_lo)::value; m < decltype(m_hi)::value; ++m) {
            const f16x8 a = __builtin_bit_cast(f16x8, f.q[m]);
#pragma unroll
            for (int n = 0; n < 2; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(f16x8, f.p[n]), first ? zero : acc[m][n], 0, 0, 0);
            if (dma) piece(buf, j0 + m);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

#pragma unroll
    for (int j = 0; j < 8; ++j) piece(0, j);
    advance_fetch();
    epi.aux_issue(0, qtile_of(0));
    __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0)
    epi.aux_commit(0);
    __syncthreads();

#ifdef AM_DEV_KNOBS
    unsigned long long* trace = nullptr;
    if (g_wide_trace != nullptr && (int)blockIdx.x >= g_wide_trace_b0 && (int)blockIdx.x < g_wide_trace_b0 + 64 && (wave == 0 || wave == 4) && L.lane == 0)
        trace = g_wide_trace + ((size_t)((int)blockIdx.x - g_wide_trace_b0) * 2 + (wave >> 2)) * 96 * 6;
#define WIDE_STAMP(k) do { if (trace != nullptr && g < 96) trace[g * 6 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define WIDE_STAMP(k) do { } while (0)
#endif
    // The pieces of "the next stage" are issued unconditionally - also in the last stage, where the Q descriptor is empty
    // (qtile_of past the end: zero records, no memory traffic) and the P slab lands in a buffer nobody reads before the
    // final vmcnt(0) + barrier: with a run-time "is there a next stage" flag every piece sat in its own basic block and
    // the waits at the block joins were conservative.  The only branch left in a stage is first slab of a tile (C = 0) /
    // later slab, taken BEFORE the first fragment reads, so that the first MFMA pair waits for exactly its three
    // fragments.  (Four straight-line stage variants - first x last - were tried: the accumulator live ranges across a
    // four-way join make the register allocator spill 130-450 registers.)
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I4 = std::integral_constant<int, 4>;
    int t = 0, kt = 0;
    for (int g = 0; g < G; ++g) {
        const float* st = lds + (g & 1) * WSTAGE_WORDS;
        const bool last_k = kt == nk - 1;
        const int nbuf = (g + 1) & 1;
        WIDE_STAMP(0);
        if (last_k && t + 1 < ntiles) epi.aux_issue(t + 1, qtile_of(t + 1));   // early: covered by this stage's vmcnt(0)
        Frags f0, f1;
        if (kt == 0) {
            f0 = frags(st, 0);
            __builtin_amdgcn_sched_barrier(0);
            mm(f0, true, true, nbuf, 0, I0{}, I1{});
            f1 = frags(st, 1);
            __builtin_amdgcn_sched_barrier(0);
            mm(f0, true, true, nbuf, 0, I1{}, I4{});
        } else {
            f0 = frags(st, 0);
            __builtin_amdgcn_sched_barrier(0);
            mm(f0, false, true, nbuf, 0, I0{}, I1{});
            f1 = frags(st, 1);
            __builtin_amdgcn_sched_barrier(0);
            mm(f0, false, true, nbuf, 0, I1{}, I4{});
        }
        WIDE_STAMP(1);
        f0 = frags(st, 2);
        __builtin_amdgcn_sched_barrier(0);
        mm(f1, false, true, nbuf, 4, I0{}, I4{});
        advance_fetch();
        WIDE_STAMP(2);
        f1 = frags(st, 3);
        __builtin_amdgcn_sched_barrier(0);
        mm(f0, false, false, 0, 0, I0{}, I4{});
        mm(f1, false, false, 0, 0, I0{}, I4{});
        WIDE_STAMP(3);
        if (last_k) {
            epi.finish(t, qtile_of(t), acc);
            if (t + 1 < ntiles) epi.aux_commit(t + 1);
        }
        WIDE_STAMP(4);
        __builtin_amdgcn_s_waitcnt(0x0F70);         // the slab of stage g+1 has landed in LDS
        WIDE_STAMP(5);
        __syncthreads();
        if (last_k) {
            kt = 0;
            ++t;
        } else {
            ++kt;
        }
    }
}


}  // namespace am
