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

// Runs `ntiles` consecutive 128x128 tiles.  Src(t, row) returns the global row
// pointer feeding local row `row` of tile t (nullptr = zero row).  Epi provides
//   aux_issue(t)  : start any per-tile side loads (all threads call it)
//   aux_commit(t) : write them to LDS (visible to finish(t) after a barrier)
//   finish(t,acc) : consume the finished accumulators of tile t
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
    epi.aux_issue(0);
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
            if (last_k) epi.aux_issue(nt_);
        }
        const float* s = lds + (g & 1) * STAGE_FLOATS;
        compute_stage(s, s + TILE_FLOATS, L, acc);
        if (last_k) {
            epi.finish(t, acc);
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
