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

namespace am {

constexpr int WTB = 256;                              // tile rows of either operand
constexpr int WTHREADS = 512;
constexpr int WROW = 32;                              // LDS row: 32 words = 128 B = 64 f16
constexpr int WTILE_WORDS = WTB * WROW;               // one operand slab, 32 KB
constexpr int WSTAGE_WORDS = 2 * WTILE_WORDS;         // Q slab then P slab
constexpr int WENGINE_LDS_WORDS = 2 * WSTAGE_WORDS;   // two stages, 128 KB

struct WLane {
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
};

// descriptor over the valid rows of a 256-row tile starting at row0 (rows past the end read as zero)
__device__ __forceinline__ TileRsrc make_wide_rsrc(const float* base, int64_t ld, int64_t n_rows, int64_t row0) {
    int64_t valid = n_rows - row0;
    valid = valid < 0 ? 0 : (valid > WTB ? WTB : valid);
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
template <class TileMap, class Epi>
__device__ __forceinline__ void wide_pipeline(const float* __restrict__ Q, int64_t nq, int64_t ldq, const TileMap& tmap,
                                              const float* __restrict__ P, int64_t np, int64_t ldp, int64_t prow0,
                                              int ntiles, int Dh, float* __restrict__ lds, const WLane& L, Epi& epi) {
    const int nk = Dh / WROW;
    const int G = ntiles * nk;
    const int wave = __builtin_amdgcn_readfirstlane(L.wave);
    const int srow = L.tid >> 3;                                   // 0..63 (+64 j)
    const int chunk = (L.tid & 7) ^ ((srow >> 1) & 7);             // 16-B chunk of the row this thread fetches into slot tid & 7
    unsigned voq[4], vop[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        voq[j] = (unsigned)(((int64_t)(j * 64 + srow) * ldq + chunk * 4) * 4);
        vop[j] = (unsigned)(((int64_t)(j * 64 + srow) * ldp + chunk * 4) * 4);
    }
    const TileRsrc prs = make_wide_rsrc(P, ldp, np, prow0);
    const int64_t q_tiles_total = (nq + WTB - 1) / WTB;
    auto qtile_of = [&](int t) -> int64_t { return t < ntiles ? tmap(t) : q_tiles_total; };   // past the end: empty descriptor
    int ft = 0, fkt = 0;                                           // (tile, k-slab) of the next fetch
    TileRsrc qrs = make_wide_rsrc(Q, ldq, nq, qtile_of(0) * WTB);
    auto issue = [&](int g) {
        const unsigned so = (unsigned)(fkt * WROW * 4);
        float* s = lds + (g & 1) * WSTAGE_WORDS + wave * 8 * WROW;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            lds_direct_b128(qrs, s + j * 64 * WROW, voq[j], so);
            lds_direct_b128(prs, s + WTILE_WORDS + j * 64 * WROW, vop[j], so);
        }
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
        Frags f;
#pragma unroll
        for (int m = 0; m < 4; ++m) f.q[m] = *reinterpret_cast<const f32x4*>(st + qrow + m * 32 * WROW + coff[c]);
#pragma unroll
        for (int n = 0; n < 2; ++n) f.p[n] = *reinterpret_cast<const f32x4*>(st + prow + n * 32 * WROW + coff[c]);
        return f;
    };
    f32x16 acc[4][2];
    wide_zero(acc);
    auto mm = [&](const Frags& f) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const f16x8 a = __builtin_bit_cast(f16x8, f.q[m]);
#pragma unroll
            for (int n = 0; n < 2; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(f16x8, f.p[n]), acc[m][n], 0, 0, 0);
        }
    };
    // first chunk of a tile: C = 0 as an inline constant instead of 128 register clears per tile
    auto mm_first = [&](const Frags& f) {
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const f16x8 a = __builtin_bit_cast(f16x8, f.q[m]);
#pragma unroll
            for (int n = 0; n < 2; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(f16x8, f.p[n]), zero, 0, 0, 0);
        }
    };

    issue(0);
    epi.aux_issue(0, qtile_of(0));
    __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0)
    epi.aux_commit(0);
    __syncthreads();

    int t = 0, kt = 0;
    for (int g = 0; g < G; ++g) {
        const bool last_k = (kt == nk - 1);
        const int nt_ = last_k ? t + 1 : t;
        const int nkt = last_k ? 0 : kt + 1;
        if (g + 1 < G) issue(g + 1);
        if (last_k) epi.aux_issue(nt_, qtile_of(nt_));
        const float* st = lds + (g & 1) * WSTAGE_WORDS;
        Frags f0 = frags(st, 0);
        Frags f1 = frags(st, 1);
        if (kt == 0) mm_first(f0);
        else mm(f0);
        f0 = frags(st, 2);
        mm(f1);
        f1 = frags(st, 3);
        mm(f0);
        mm(f1);
        if (last_k) {
            epi.finish(t, qtile_of(t), acc);
            epi.aux_commit(nt_);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);         // the slab of stage g+1 has landed in LDS
        __syncthreads();
        t = nt_;
        kt = nkt;
    }
}

}  // namespace am
