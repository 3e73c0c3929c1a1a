// 128x128 f32 tile engine on the gfx950 f32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Every dense contraction of the hot path that has two row-major operands with a
// shared, contiguous inner (feature) dimension goes through this engine:
//   * pairwise squared distances for the k-NN radii / PRDC counts (pairwise.hip),
//   * the three Gram blocks per subset of the kernel distance (kd.hip).
// A workgroup is 256 threads = 4 wave64 in a 2x2 arrangement; each wave owns a
// 64x64 sub-tile = 2x2 MFMA tiles of 32x32 (64 accumulator VGPRs).
//
// Operand roles.  "P" rows are the lane-local axis (MFMA n, column = lane & 31):
// after the K loop a lane holds 16 results that all belong to ONE P row, so
// per-P-row reductions (top-k lists, row min / any) need no cross-lane traffic.
// "Q" rows are the register axis (MFMA m): register r of a lane belongs to Q row
// (r&3) + 8*(r>>2) + 4*(lane>>5) of the 32x32 tile.
//
// LDS image: per stage a [128][BK=32] slab of each operand, row stride LDK = 36
// floats (144 B): ds_write_b128 by 8-lane row groups and ds_read_b128 by the
// MFMA fragment pattern (row = lane&31, 16-B column block = lane>>5) are both
// bank-conflict free with this stride (MI355X_MICROARCH.md, LDS table).
//
// K order.  A lane's float4 covers inner indices 8c+4h .. 8c+4h+3 (h = lane>>5);
// MFMA step s of chunk c therefore multiplies index 8c+s (lower half-wave) and
// 8c+4+s (upper half-wave).  The hardware accumulates the lower half-wave's
// product first, so every dot product is the f32 fmaf chain over the index order
//   8c+0, 8c+4, 8c+1, 8c+5, 8c+2, 8c+6, 8c+3, 8c+7   (c = 0, 1, ...)
// which oracle/exact_c/pairwise_exact.c reproduces bit for bit.
//
// Pipeline.  Two LDS stages; the global loads of stage g+1 are issued before the
// MFMAs of stage g and written to LDS after them (one barrier per stage), and
// the stage sequence runs on across consecutive Q tiles so the pipeline never
// drains inside a workgroup.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

namespace am {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TB = 128;                          // tile rows of either operand
constexpr int BK = 32;                           // inner-dimension slab per stage
constexpr int LDK = 36;                          // padded LDS row stride (floats)
constexpr int TILE_FLOATS = TB * LDK;            // one operand slab
constexpr int STAGE_FLOATS = 2 * TILE_FLOATS;    // Q slab then P slab
constexpr int ENGINE_LDS_FLOATS = 2 * STAGE_FLOATS;
constexpr int ENGINE_THREADS = 256;

struct LaneInfo {
    static constexpr int NT = 2;                 // 32-row P tiles per wave (pstat_engine.h: 1)
    static constexpr bool ACC_INIT = false;      // accumulators start at zero (pstat_engine.h: at the epilogue's per-column start value)
    int tid, lane, wm, wn, r, h;
    __device__ __forceinline__ LaneInfo() {
        tid = threadIdx.x;
        lane = tid & 63;
        const int wave = tid >> 6;
        wm = wave >> 1;      // which 64-row half of the Q tile
        wn = wave & 1;       // which 64-row half of the P tile
        r = lane & 31;
        h = lane >> 5;
    }
};

// One float4 of a row with the inner-dimension tail masked to zero.  Rows are
// 16-B aligned and ld % 4 == 0, so the load itself never leaves the row.
__device__ __forceinline__ f32x4 load_k4(const float* __restrict__ row, int k, int D) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row != nullptr && k < D) {
        v = *reinterpret_cast<const f32x4*>(row + k);
        if (k + 3 >= D) {
            if (k + 1 >= D) v.y = 0.f;
            if (k + 2 >= D) v.z = 0.f;
            if (k + 3 >= D) v.w = 0.f;
        }
    }
    return v;
}

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
}

// 64 MFMAs: one BK=32 slab of the wave's 64x64 sub-tile.
__device__ __forceinline__ void compute_stage(const float* __restrict__ sQ, const float* __restrict__ sP,
                                              const LaneInfo& L, f32x16 (&acc)[2][2]) {
    const float* q = sQ + (L.wm * 64 + L.r) * LDK + L.h * 4;
    const float* p = sP + (L.wn * 64 + L.r) * LDK + L.h * 4;
#pragma unroll
    for (int c = 0; c < BK / 8; ++c) {
        const f32x4 qa = *reinterpret_cast<const f32x4*>(q + c * 8);
        const f32x4 qb = *reinterpret_cast<const f32x4*>(q + 32 * LDK + c * 8);
        const f32x4 pa = *reinterpret_cast<const f32x4*>(p + c * 8);
        const f32x4 pb = *reinterpret_cast<const f32x4*>(p + 32 * LDK + c * 8);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[s], pa[s], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[s], pb[s], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(qb[s], pa[s], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(qb[s], pb[s], acc[1][1], 0, 0, 0);
        }
    }
}

// ---------------------------------------------------------------------------
// Buffer-resource staging (engine variant bit 0).  A dense 128-row operand tile
// is addressed through a raw buffer descriptor whose base is the tile's first
// row and whose num_records is the byte extent of its VALID rows: rows past the
// end of the matrix read as zero in hardware, so the staging loads need neither
// branches nor 64-bit per-lane address arithmetic - each thread keeps four
// constant 32-bit offsets (its rows inside the tile) and the stage advances a
// scalar offset.
struct TileRsrc {
    __amdgpu_buffer_rsrc_t rsrc;
};

__device__ __forceinline__ TileRsrc make_tile_rsrc(const float* base, int64_t ld, int64_t n_rows, int64_t row0) {
    int64_t valid = n_rows - row0;
    valid = valid < 0 ? 0 : (valid > TB ? TB : valid);
    const float* p = base + (valid > 0 ? row0 : 0) * ld;
    // wave-uniform by construction (kernel arguments and blockIdx only); readfirstlane makes that provable
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<uintptr_t>(p) & 0xffffffffu));
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<uintptr_t>(p) >> 32));
    const unsigned bytes = __builtin_amdgcn_readfirstlane((unsigned)(valid * ld * 4));
    void* q = reinterpret_cast<void*>((static_cast<uintptr_t>(hi) << 32) | lo);
    TileRsrc r;
    r.rsrc = __builtin_amdgcn_make_buffer_rsrc(q, 0, (int)bytes, 0x00020000);
    return r;
}

__device__ __forceinline__ f32x4 rsrc_load(const TileRsrc& r, unsigned voff, unsigned soff) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r.rsrc, (int)voff, (int)soff, 0);
    f32x4 f;
    f.x = __uint_as_float(v.x); f.y = __uint_as_float(v.y); f.z = __uint_as_float(v.z); f.w = __uint_as_float(v.w);
    return f;
}

// engine variant bits
constexpr int EV_RSRC = 1;       // buffer-resource staging loads (dense operands only)
constexpr int EV_FRAGDB = 2;     // double-buffered LDS fragments across the 8-wide k chunks
constexpr int EV_PRIO = 4;       // s_setprio rising through the stage: the wave further along wins the MFMA pipe
constexpr int EV_ABL_NOEPI = 8;  // ablation only (wrong results): skip the tile epilogue
constexpr int EV_ABL_NOLOAD = 16; // ablation only (wrong results): stage the first slab only

__device__ __forceinline__ void set_prio_level(int c) {   // s_setprio takes an immediate
    switch (c) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
    }
}

// 64 MFMAs of one BK=32 slab with the fragments of chunk c+1 read while chunk c multiplies.
template <int V>
__device__ __forceinline__ void compute_stage_v(const float* __restrict__ sQ, const float* __restrict__ sP,
                                                const LaneInfo& L, f32x16 (&acc)[2][2]) {
    if constexpr ((V & EV_FRAGDB) == 0) {
        if constexpr (V & EV_PRIO) __builtin_amdgcn_s_setprio(1);
        compute_stage(sQ, sP, L, acc);
        if constexpr (V & EV_PRIO) __builtin_amdgcn_s_setprio(0);
    } else {
        const float* q = sQ + (L.wm * 64 + L.r) * LDK + L.h * 4;
        const float* p = sP + (L.wn * 64 + L.r) * LDK + L.h * 4;
        f32x4 qa[2], qb[2], pa[2], pb[2];
        qa[0] = *reinterpret_cast<const f32x4*>(q);
        qb[0] = *reinterpret_cast<const f32x4*>(q + 32 * LDK);
        pa[0] = *reinterpret_cast<const f32x4*>(p);
        pb[0] = *reinterpret_cast<const f32x4*>(p + 32 * LDK);
#pragma unroll
        for (int c = 0; c < BK / 8; ++c) {
            const int cur = c & 1, nxt = cur ^ 1;
            if (c + 1 < BK / 8) {
                qa[nxt] = *reinterpret_cast<const f32x4*>(q + (c + 1) * 8);
                qb[nxt] = *reinterpret_cast<const f32x4*>(q + 32 * LDK + (c + 1) * 8);
                pa[nxt] = *reinterpret_cast<const f32x4*>(p + (c + 1) * 8);
                pb[nxt] = *reinterpret_cast<const f32x4*>(p + 32 * LDK + (c + 1) * 8);
            }
            if constexpr (V & EV_PRIO) set_prio_level(c);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[cur][s], pa[cur][s], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[cur][s], pb[cur][s], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(qb[cur][s], pa[cur][s], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(qb[cur][s], pb[cur][s], acc[1][1], 0, 0, 0);
            }
        }
        if constexpr (V & EV_PRIO) __builtin_amdgcn_s_setprio(0);
    }
}

// Dense-operand pipeline: Q tiles qtile0 .. qtile0+ntiles-1 of the row-major matrix (Q, nq, ldq) against the
// fixed 128-row P block starting at prow0 of (P, np, ldp).  Same stage order, LDS image and arithmetic as
// tile_pipeline; only the way the slabs are fetched differs.
template <int V, class Epi>
__device__ __forceinline__ void dense_pipeline(const float* __restrict__ Q, int64_t nq, int64_t ldq, int64_t qtile0,
                                               const float* __restrict__ P, int64_t np, int64_t ldp, int64_t prow0,
                                               int ntiles, int D, float* __restrict__ lds, const LaneInfo& L, Epi& epi) {
    const int nk = (D + BK - 1) / BK;
    const int G = ntiles * nk;
    const int srow = L.tid >> 3;
    const int scol = (L.tid & 7) * 4;
    const bool ktail = (D % BK) != 0;
    f32x4 rq[4], rp[4];
    f32x16 acc[2][2];
    zero_acc(acc);

    const TileRsrc prs = make_tile_rsrc(P, ldp, np, prow0);
    unsigned voq[4], vop[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        voq[q] = (unsigned)(((int64_t)(q * 32 + srow) * ldq + scol) * 4);
        vop[q] = (unsigned)(((int64_t)(q * 32 + srow) * ldp + scol) * 4);
    }
    TileRsrc qrs = make_tile_rsrc(Q, ldq, nq, qtile0 * TB);

    auto issue = [&](int kt) {
        const unsigned so = (unsigned)(kt * BK * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            rq[q] = rsrc_load(qrs, voq[q], so);
            rp[q] = rsrc_load(prs, vop[q], so);
        }
    };
    auto commit = [&](int g, int kt) {
        if (ktail && kt == nk - 1) {               // wave-uniform: zero the inner-dimension tail of the last slab
            const int k = kt * BK + scol;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (k + e >= D) { rq[q][e] = 0.f; rp[q][e] = 0.f; }
                }
        }
        float* s = lds + (g & 1) * STAGE_FLOATS + srow * LDK + scol;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            *reinterpret_cast<f32x4*>(s + q * 32 * LDK) = rq[q];
            *reinterpret_cast<f32x4*>(s + TILE_FLOATS + q * 32 * LDK) = rp[q];
        }
    };

    issue(0);
    epi.aux_issue(0, qtile0);
    commit(0, 0);
    epi.aux_commit(0);
    __syncthreads();

    int t = 0, kt = 0;
    for (int g = 0; g < G; ++g) {
        const bool more = (g + 1 < G);
        const bool last_k = (kt == nk - 1);
        const int nt_ = last_k ? t + 1 : t;
        const int nkt = last_k ? 0 : kt + 1;
        if (more) {
            if (last_k) qrs = make_tile_rsrc(Q, ldq, nq, (qtile0 + nt_) * TB);
            if constexpr ((V & EV_ABL_NOLOAD) == 0) issue(nkt);
            if (last_k) epi.aux_issue(nt_, qtile0 + nt_);
        }
        const float* s = lds + (g & 1) * STAGE_FLOATS;
        compute_stage_v<V>(s, s + TILE_FLOATS, L, acc);
        if (last_k) {
            if constexpr ((V & EV_ABL_NOEPI) == 0) {
                epi.finish(t, qtile0 + t, acc);
            } else {
                if (acc[0][0][0] == 123.456f && acc[1][1][7] == 3.f) epi.finish(t, qtile0 + t, acc);   // keep the MFMAs live
            }
            zero_acc(acc);
        }
        if (more) {
            if constexpr ((V & EV_ABL_NOLOAD) == 0) commit(g + 1, nkt);
            if (last_k) epi.aux_commit(nt_);
        }
        __syncthreads();
        t = nt_;
        kt = nkt;
    }
}

constexpr int EV_EARLY = 32;     // loads issued two stages ahead, LDS writes placed inside the MFMA stream
constexpr int EV_LDS = 64;       // operands go global -> LDS directly (buffer_load ... lds); needs D % 32 == 0
constexpr int EV_F16 = 128;      // operands are f16 pairs viewed as f32 words (filter passes only)

// LDS image of the LDS-direct variant: unpadded 128-B rows (a wave's buffer_load_dwordx4 ... lds writes 64 x 16 B
// contiguously = 8 whole rows), 16-B chunks XOR-swizzled by ((row >> 1) & 7) so the MFMA fragment reads stay conflict-free.
constexpr int LDR = 32;
constexpr int TILE_FLOATS_R = TB * LDR;
constexpr int STAGE_FLOATS_R = 2 * TILE_FLOATS_R;

// float offset inside a slab row that staging thread `tid` fetches (its LDS slot is always tid & 7)
template <bool DIRECT>
__device__ __forceinline__ int staging_col(int tid) {
    if constexpr (DIRECT) return (((tid & 7) ^ ((tid >> 4) & 7)) * 4);      // slot ^ ((row >> 1) & 7), row = tid >> 3 (+32 q)
    else return (tid & 7) * 4;
}

struct FragSet {
    f32x4 qa, qb, pa, pb;
};

__device__ __forceinline__ FragSet read_frags(const float* __restrict__ q, const float* __restrict__ p, int c) {
    FragSet f;
    f.qa = *reinterpret_cast<const f32x4*>(q + c * 8);
    f.qb = *reinterpret_cast<const f32x4*>(q + 32 * LDK + c * 8);
    f.pa = *reinterpret_cast<const f32x4*>(p + c * 8);
    f.pb = *reinterpret_cast<const f32x4*>(p + 32 * LDK + c * 8);
    return f;
}

// f16 operands (EV_F16): the same 16-B fragment is eight half-precision inner elements, consumed by ONE
// v_mfma_f32_32x32x16_f16 (lane half h supplies k = 8h .. 8h+7 of the 16-wide step), so a 128-B slab row is 64
// inner elements and a stage issues 16 MFMAs instead of 64.  Used only by the filter passes of pairwise_fast.h,
// whose results are re-verified in f32; the accumulation order inside the instruction is irrelevant there.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void mfma_chunk_f16(const FragSet& f, f32x16 (&acc)[2][2]) {
    const f16x8 qa = __builtin_bit_cast(f16x8, f.qa), qb = __builtin_bit_cast(f16x8, f.qb);
    const f16x8 pa = __builtin_bit_cast(f16x8, f.pa), pb = __builtin_bit_cast(f16x8, f.pb);
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qa, pa, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qa, pb, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qb, pa, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qb, pb, acc[1][1], 0, 0, 0);
}

__device__ __forceinline__ void mfma_chunk(const FragSet& f, f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.qa[s], f.pa[s], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.qa[s], f.pb[s], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.qb[s], f.pa[s], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.qb[s], f.pb[s], acc[1][1], 0, 0, 0);
    }
}

// Dense-operand pipeline, "early commit" schedule.  Stage g multiplies LDS buffer g&1 while
//   - the slab of stage g+2 is fetched into one of two register sets (issued at the top of stage g), and
//   - the slab of stage g+1 (fetched during stage g-1, long since landed) is written to buffer (g+1)&1
//     between the second and third 8-wide k chunk, i.e. in the shadow of the MFMA pipe,
// so that only the barrier itself separates two stages.  Same LDS image, stage order and arithmetic as
// dense_pipeline / tile_pipeline (bit-identical results).
// Column-tile sequences: local tile t of a workgroup -> 128-row tile index of Q.
struct LinearTiles {              // (q0 + t) * stride : consecutive tiles, or every stride-th tile (sampling pre-pass)
    int64_t q0;
    int64_t stride = 1;
    __device__ __forceinline__ int64_t operator()(int t) const { return (q0 + t) * stride; }
};
struct CyclicTiles {              // (start + t) mod T : the cyclic half-range used by the symmetric k-NN kernel
    int64_t start, T;
    __device__ __forceinline__ int64_t operator()(int t) const {
        const int64_t q = start + t;
        return q >= T ? q - T : q;
    }
};

// How the 4 staging rows of a thread are addressed for one 128-row operand tile: a buffer descriptor plus one
// 32-bit byte offset per row.  Dense tiles use a per-tile descriptor (rows past the end fall outside it and read
// as 0); gathered tiles (kernel distance subsets) use one descriptor for the whole matrix and per-row offsets
// through the index list, with 0xffffffff for padded rows.
struct TileAddr {
    TileRsrc rs;
    unsigned vo[4];
};

template <int V, bool KTAIL, class QAddrFn, class Epi>
__device__ __forceinline__ void addr_pipeline_early(const QAddrFn& qaddr, const TileAddr& paddr, int ntiles, int D,
                                                    int64_t q_tiles_total, float* __restrict__ lds, const LaneInfo& L,
                                                    Epi& epi);

template <bool F16, class QAddrFn, class Epi>
__device__ __forceinline__ void addr_pipeline_lds(const QAddrFn& qaddr, const TileAddr& paddr, int ntiles, int D,
                                                  float* __restrict__ lds, const LaneInfo& L, Epi& epi);

template <int V, bool KTAIL, class TileMap, class Epi>
__device__ __forceinline__ void dense_pipeline_early(const float* __restrict__ Q, int64_t nq, int64_t ldq,
                                                     const TileMap& tmap,
                                                     const float* __restrict__ P, int64_t np, int64_t ldp, int64_t prow0,
                                                     int ntiles, int D, float* __restrict__ lds, const LaneInfo& L,
                                                     Epi& epi) {
    constexpr bool DIRECT = (V & EV_LDS) != 0 && !KTAIL;
    const int srow = L.tid >> 3;
    const int scol = staging_col<DIRECT>(L.tid);
    TileAddr pa;
    pa.rs = make_tile_rsrc(P, ldp, np, prow0);
    unsigned voq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        voq[q] = (unsigned)(((int64_t)(q * 32 + srow) * ldq + scol) * 4);
        pa.vo[q] = (unsigned)(((int64_t)(q * 32 + srow) * ldp + scol) * 4);
    }
    // past this workgroup's last tile: row0 = nq -> zero valid rows -> every load returns 0
    auto qaddr = [&](int t) {
        TileAddr a;
        a.rs = make_tile_rsrc(Q, ldq, nq, t < ntiles ? tmap(t) * TB : nq);
#pragma unroll
        for (int q = 0; q < 4; ++q) a.vo[q] = voq[q];
        return a;
    };
    struct AbsTile {                 // the epilogue wants absolute Q tile indices
        const TileMap& m;
        int n;
        int64_t past;
        __device__ __forceinline__ int64_t operator()(int t) const { return t < n ? m(t) : past; }
    };
    const AbsTile abs_tile{tmap, ntiles, (nq + TB - 1) / TB};
    struct Shim {
        Epi& e;
        const AbsTile& at;
        __device__ __forceinline__ void aux_issue(int t, int64_t) { e.aux_issue(t, at(t)); }
        __device__ __forceinline__ void aux_commit(int t) { e.aux_commit(t); }
        __device__ __forceinline__ void finish(int t, int64_t, f32x16 (&acc)[2][2]) { e.finish(t, at(t), acc); }
    } shim{epi, abs_tile};
    if constexpr (DIRECT) addr_pipeline_lds<(V & EV_F16) != 0>(qaddr, pa, ntiles, D, lds, L, shim);
    else addr_pipeline_early<V, KTAIL>(qaddr, pa, ntiles, D, 0, lds, L, shim);
}

// LDS-direct schedule (EV_LDS).  Stage g multiplies LDS buffer g&1 while the slab of stage g+1 streams from
// global memory straight into buffer (g+1)&1 - no staging registers, no ds_write.  Each wave fills 8 rows per
// instruction (lane l -> row l>>3, 16-B slot l&7, fetching chunk (l&7)^(l>>3) of that row).  The loads must have
// landed before the barrier that ends the stage (explicit vmcnt(0)); the prefetch distance is therefore one stage,
// against two for the register-staged schedule.  Same slab order and arithmetic: bit-identical results.
__device__ __forceinline__ void lds_direct_b128(const TileRsrc& r, float* lds_wave_base, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r.rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voff,
                                             (int)soff, 0, 0);
}

template <bool F16, class QAddrFn, class Epi>
__device__ __forceinline__ void addr_pipeline_lds(const QAddrFn& qaddr, const TileAddr& paddr, int ntiles, int D,
                                                  float* __restrict__ lds, const LaneInfo& L, Epi& epi) {
    const int nk = D / BK;
    const int G = ntiles * nk;
    const int wave = __builtin_amdgcn_readfirstlane(L.tid >> 6);
    f32x16 acc[2][2];
    zero_acc(acc);

    int ft = 0, fkt = 0;                            // (tile, k-slab) of the next fetch
    TileAddr qa = qaddr(0);
    auto issue = [&](int g) {
        const unsigned so = (unsigned)(fkt * BK * 4);
        float* s = lds + (g & 1) * STAGE_FLOATS_R + wave * 8 * LDR;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            lds_direct_b128(qa.rs, s + q * 32 * LDR, qa.vo[q], so);
            lds_direct_b128(paddr.rs, s + TILE_FLOATS_R + q * 32 * LDR, paddr.vo[q], so);
        }
        if (++fkt == nk) {
            fkt = 0;
            ++ft;
            qa = qaddr(ft);
        }
    };
    // fragment addresses: logical 16-B chunk 2c+h of row r sits in slot (2c+h) ^ ((r >> 1) & 7)  (a swizzle by r & 7
    // ties the slot's parity to the row's and is 2-way bank-conflicted for the MFMA fragment pattern; see wide_engine.h)
    const int sw = (L.r >> 1) & 7;
    int coff[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) coff[c] = ((2 * c + L.h) ^ sw) * 4;
    const int qrow = (L.wm * 64 + L.r) * LDR;
    const int prow = TILE_FLOATS_R + (L.wn * 64 + L.r) * LDR;
    auto frags = [&](const float* st, int c) {
        FragSet f;
        f.qa = *reinterpret_cast<const f32x4*>(st + qrow + coff[c]);
        f.qb = *reinterpret_cast<const f32x4*>(st + qrow + 32 * LDR + coff[c]);
        f.pa = *reinterpret_cast<const f32x4*>(st + prow + coff[c]);
        f.pb = *reinterpret_cast<const f32x4*>(st + prow + 32 * LDR + coff[c]);
        return f;
    };

    issue(0);
    epi.aux_issue(0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0)
    epi.aux_commit(0);
    __syncthreads();

    int t = 0, kt = 0;
    for (int g = 0; g < G; ++g) {
        const bool last_k = (kt == nk - 1);
        const int nt_ = last_k ? t + 1 : t;
        const int nkt = last_k ? 0 : kt + 1;
        if (g + 1 < G) issue(g + 1);
        if (last_k) epi.aux_issue(nt_, nt_);
        const float* st = lds + (g & 1) * STAGE_FLOATS_R;
        auto mm = [&](const FragSet& f) {
            if constexpr (F16) mfma_chunk_f16(f, acc);
            else mfma_chunk(f, acc);
        };
        FragSet f0 = frags(st, 0);
        FragSet f1 = frags(st, 1);
        mm(f0);
        f0 = frags(st, 2);
        mm(f1);
        f1 = frags(st, 3);
        mm(f0);
        mm(f1);
        if (last_k) {
            epi.finish(t, t, acc);
            zero_acc(acc);
            epi.aux_commit(nt_);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);         // the slab of stage g+1 has landed in LDS
        __syncthreads();
        t = nt_;
        kt = nkt;
    }
}

template <int V, bool KTAIL, class QAddrFn, class Epi>
__device__ __forceinline__ void addr_pipeline_early(const QAddrFn& qaddr, const TileAddr& paddr, int ntiles, int D,
                                                    int64_t, float* __restrict__ lds, const LaneInfo& L, Epi& epi) {
    const int nk = (D + BK - 1) / BK;
    const int G = ntiles * nk;
    const int srow = L.tid >> 3;
    const int scol = (L.tid & 7) * 4;
    f32x16 acc[2][2];
    zero_acc(acc);

    // (tile, k-slab) of the stage being fetched; runs two stages ahead of the compute stage
    int ft = 0, fkt = 0;
    TileAddr qa = qaddr(0);
    auto fetch_advance = [&]() {
        if (++fkt == nk) {
            fkt = 0;
            ++ft;
            qa = qaddr(ft);
        }
    };
    auto issue = [&](f32x4 (&r)[8]) {
        const unsigned so = (unsigned)(fkt * BK * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            r[q] = rsrc_load(qa.rs, qa.vo[q], so);
            r[4 + q] = rsrc_load(paddr.rs, paddr.vo[q], so);
        }
        fetch_advance();
    };
    auto commit = [&](f32x4 (&r)[8], int g, int kt) {
        if constexpr (KTAIL) {                     // D % 32 != 0: zero the inner-dimension tail (branch-free select)
            const int k = kt * BK + scol;
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) r[q][e] = (k + e < D) ? r[q][e] : 0.f;
        }
        float* s = lds + (g & 1) * STAGE_FLOATS + srow * LDK + scol;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            *reinterpret_cast<f32x4*>(s + q * 32 * LDK) = r[q];
            *reinterpret_cast<f32x4*>(s + TILE_FLOATS + q * 32 * LDK) = r[4 + q];
        }
    };

    f32x4 ra[8], rb[8];
    issue(ra);                                     // stage 0
    issue(rb);                                     // stage 1 (an empty descriptor if there is none)
    epi.aux_issue(0, 0);
    commit(ra, 0, 0);
    epi.aux_commit(0);
    __syncthreads();

    int t = 0, kt = 0;
    // One compute stage; RI receives the fetch of stage g+2, RC holds the landed slab of stage g+1.
    // Fetch and commit are UNCONDITIONAL, so the stage is straight-line code and the compiler can wait for
    // the older register set with a counted vmcnt while the newer fetch is in flight.  Past the last stage
    // the fetch hits an empty buffer descriptor (hardware returns zeros, no memory traffic) and the commit
    // writes those zeros into the LDS buffer nobody reads any more.
    auto stage = [&](int g, f32x4 (&ri)[8], f32x4 (&rc)[8]) {
        const bool last_k = (kt == nk - 1);
        const int nt_ = last_k ? t + 1 : t;
        const int nkt = last_k ? 0 : kt + 1;
        issue(ri);
        if (last_k) epi.aux_issue(nt_, nt_);
        const float* sq = lds + (g & 1) * STAGE_FLOATS + (L.wm * 64 + L.r) * LDK + L.h * 4;
        const float* sp = lds + (g & 1) * STAGE_FLOATS + TILE_FLOATS + (L.wn * 64 + L.r) * LDK + L.h * 4;
        auto mm = [&](const FragSet& f) {
            if constexpr ((V & EV_F16) != 0) mfma_chunk_f16(f, acc);
            else mfma_chunk(f, acc);
        };
        FragSet f0 = read_frags(sq, sp, 0);
        FragSet f1 = read_frags(sq, sp, 1);
        mm(f0);
        f0 = read_frags(sq, sp, 2);
        mm(f1);
        commit(rc, g + 1, nkt);                    // LDS writes of the next slab, behind queued MFMAs
        f1 = read_frags(sq, sp, 3);
        mm(f0);
        mm(f1);
        if (last_k) {
            epi.finish(t, t, acc);
            zero_acc(acc);
            epi.aux_commit(nt_);
        }
        __syncthreads();
        t = nt_;
        kt = nkt;
    };
    int g = 0;
    for (; g + 1 < G; g += 2) {
        stage(g, ra, rb);
        stage(g + 1, rb, ra);
    }
    if (g < G) stage(g, ra, rb);
}

// Runs `ntiles` consecutive 128x128 tiles.  Src(t, row) returns the global row
// pointer feeding local row `row` of tile t (nullptr = zero row).  Epi provides
//   aux_issue(t, qtile) : start any per-tile side loads (all threads call it); qtile = Q tile index
//   aux_commit(t)       : write them to LDS (visible to finish(t) after a barrier)
//   finish(t,qtile,acc) : consume the finished accumulators of local tile t
template <class QSrc, class PSrc, class Epi>
__device__ __forceinline__ void tile_pipeline(const QSrc& qsrc, const PSrc& psrc, int ntiles, int D,
                                              float* __restrict__ lds, const LaneInfo& L, Epi& epi) {
    const int nk = (D + BK - 1) / BK;
    const int G = ntiles * nk;
    const int srow = L.tid >> 3;          // 0..31 (+32q)
    const int scol = (L.tid & 7) * 4;     // float offset inside the slab row
    f32x4 rq[4], rp[4];
    f32x16 acc[2][2];
    zero_acc(acc);

    auto issue = [&](int t, int kt) {
        const int k = kt * BK + scol;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            rq[q] = load_k4(qsrc(t, q * 32 + srow), k, D);
            rp[q] = load_k4(psrc(t, q * 32 + srow), k, D);
        }
    };
    auto commit = [&](int g) {
        float* s = lds + (g & 1) * STAGE_FLOATS + srow * LDK + scol;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            *reinterpret_cast<f32x4*>(s + q * 32 * LDK) = rq[q];
            *reinterpret_cast<f32x4*>(s + TILE_FLOATS + q * 32 * LDK) = rp[q];
        }
    };

    issue(0, 0);
    epi.aux_issue(0, 0);
    commit(0);
    epi.aux_commit(0);
    __syncthreads();

    int t = 0, kt = 0;
    for (int g = 0; g < G; ++g) {
        const bool more = (g + 1 < G);
        const bool last_k = (kt == nk - 1);
        const int nt_ = last_k ? t + 1 : t;
        const int nkt = last_k ? 0 : kt + 1;
        if (more) {
            issue(nt_, nkt);
            if (last_k) epi.aux_issue(nt_, nt_);
        }
        const float* s = lds + (g & 1) * STAGE_FLOATS;
        compute_stage(s, s + TILE_FLOATS, L, acc);
        if (last_k) {
            epi.finish(t, t, acc);
            zero_acc(acc);
        }
        if (more) {
            commit(g + 1);
            if (last_k) epi.aux_commit(nt_);
        }
        __syncthreads();
        t = nt_;
        kt = nkt;
    }
}

// Branch-free sorted insertion of x into an ascending list of CAP floats
// (keeps the CAP smallest values seen).  A value >= best[CAP-1] falls through.
template <int CAP>
__device__ __forceinline__ void list_insert(float (&best)[CAP], float x) {
#pragma unroll
    for (int i = 0; i < CAP; ++i) {
        const float lo = fminf(best[i], x);
        x = fmaxf(best[i], x);
        best[i] = lo;
    }
}

}  // namespace am
