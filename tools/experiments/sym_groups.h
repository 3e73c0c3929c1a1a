// XCD-grouped work order of the symmetric k-NN sweep on the 256-row engine (knn_wide_kernel, pairwise_wide.hip).
// Plain integer arithmetic, usable from host code as well: tests/test_sym_groups_cpu.py compiles this header with g++ and
// checks that the items of a grid cover every (row block, tile) pair of the cyclic half-range exactly once.
#pragma once
#include <stdint.h>
#ifndef __HIPCC__
#define AM_HD
#else
#define AM_HD __host__ __device__
#endif

namespace am {

// ---- XCD-grouped order of the symmetric sweep on the 256-row engine (round 4) -----------------------------------------
// The (window, row block) items of sym_work (pairwise_common.h) give an XCD - which receives every 8th workgroup - 32 different P blocks (8 MB of f16
// rows against 4 MB of L2): the P slabs, re-read for every tile, miss the L2 (hit rate 0.31, 28.6 GB per launch through the
// fabric at 100k x 512).  Here the 32 workgroups resident on an XCD form a GROUP of 8 consecutive row blocks x 4 chunks
// of their half-ranges, the membership filter's scheme (wide_work in pairwise_wide.hip): 2 MB of P slabs stay in the L2 and
// each of the four Q streams is fetched once for eight workgroups.  What makes that possible for the symmetric sweep:
//   * the chunk boundaries are common to the eight row blocks of a group, not to the whole grid: the tiles that lie in ALL
//     eight half-ranges, [p0 + 7, p0 + noff), are cut into C chunks which the eight blocks walk in lockstep; what is left of
//     a block's half-range - 7 - i tiles in front of the common range and i behind it for member i, seven tiles for every
//     member - is appended to its LAST chunk, whose share of the common range is seven tiles shorter: all items of a row
//     block have the same length to within one tile, and so have all items of the grid;
//   * groups are ordered phase-major (phase = chunk / 4): when phase f of a row block starts, its phases < f have long
//     finished (a whole pass over the row blocks lies between them) and have published their bounds, so the bounds
//     tighten C / 4 times per row block; the published list of chunk c is cumulative over the chain c, c - 4, c - 8, ...,
//     and the bound of a row is taken over its own chunk and the four chains of the previous phase - disjoint column sets.
struct GrpWork {
    int64_t pb;          // row block
    int chunk;           // index of this item's list (and of its chain: chunk & 3)
    int ntiles;          // 0: nothing to do
    int64_t c0;          // first tile of the common part (unwrapped: taken mod T)
    int ncommon;
    int64_t h0;          // first tile of the part in front of the common range
    int nhead;
    int64_t t0;          // first tile of the part behind it (unwrapped)
};
constexpr int GRP_ROWS = 8, GRP_CHUNKS = 4;

AM_HD inline int64_t grp_first_block(int64_t T, int part, int nparts) { return ((int64_t)part * T + nparts - 1) / nparts; }
AM_HD inline int64_t grp_noff(int64_t T, int64_t pb) { return T / 2 + 1 - (((T % 2) == 0 && pb >= T / 2) ? 1 : 0); }
// workgroups of the grouped sweep of rank `part`: phases x row groups x 32, rounded up to whole rounds of the 8 XCDs
AM_HD inline int64_t grp_grid(int64_t T, int chunks, int part, int nparts) {
    const int64_t lo = grp_first_block(T, part, nparts), hi = grp_first_block(T, part + 1, nparts);
    const int64_t groups = (int64_t)(chunks / GRP_CHUNKS) * ((hi - lo + GRP_ROWS - 1) / GRP_ROWS);
    return (groups + 7) / 8 * 8 * (GRP_ROWS * GRP_CHUNKS);
}

// the item of workgroup `block` (= blockIdx.x: workgroup b runs on XCD b % 8)
// order 0: phase-major (all row groups of phase 0, then of phase 1, ...).  order 1: by DESCENDING position of a group's
// first column tile (row group + phase x span of a phase, taken modulo the number of row groups - for every phase a
// bijection of the row groups, so every (row group, phase) appears exactly once): the order of the windowed sweep.  When
// a group generates mirrored candidates for the rows of column tile j, row block j itself - whose own items all start at
// positions >= j - has already been through (nearly) its whole half-range and published its bound; in phase-major order
// every row is only as far as the phase that is running.  The chain of a row then runs from the highest phase down.
AM_HD inline int grp_prev_phase(int phase, int chunks, int order) {
    const int p = order == 1 ? phase + 1 : phase - 1;
    return (p >= 0 && p < chunks / GRP_CHUNKS) ? p : -1;
}
AM_HD inline GrpWork grp_item(int64_t T, int chunks, int part, int nparts, int64_t block, int order = 0) {
    const int64_t lo = grp_first_block(T, part, nparts), hi = grp_first_block(T, part + 1, nparts);
    const int64_t row_groups = (hi - lo + GRP_ROWS - 1) / GRP_ROWS;
    const int xcd = (int)(block & 7);
    const int64_t seq = block >> 3;
    const int64_t g = (seq >> 5) * 8 + xcd;                       // group: all 32 members on one XCD
    const int within = (int)(seq & 31);
    GrpWork w;
    w.pb = 0;
    w.chunk = 0;
    w.ntiles = 0;
    w.c0 = w.h0 = w.t0 = 0;
    w.ncommon = w.nhead = 0;
    if (row_groups == 0) return w;                                // (this rank owns no row block)
    int phase;
    int64_t rg;
    if (order == 1) {
        const int phases = chunks / GRP_CHUNKS;
        const int64_t slot = g / phases;
        phase = (int)(g % phases);
        if (slot >= row_groups) return w;
        const int64_t span = (T / 2 + 1) * GRP_CHUNKS / chunks;                  // column tiles of one phase
        const int64_t off = (phase * span / GRP_ROWS) % row_groups;
        rg = ((row_groups - 1 - slot - off) % row_groups + row_groups) % row_groups;
    } else {
        phase = (int)(g / row_groups);
        rg = g % row_groups;
    }
    const int member = within / GRP_CHUNKS;
    w.chunk = phase * GRP_CHUNKS + within % GRP_CHUNKS;
    const int64_t p0 = lo + rg * GRP_ROWS;
    w.pb = p0 + member;
    if (w.chunk >= chunks || w.pb >= hi) return w;
    const int m = (int)((hi - p0 < GRP_ROWS) ? hi - p0 : GRP_ROWS);  // members of this row group
    int64_t ce = p0 + grp_noff(T, p0);
    for (int i = 1; i < m; ++i) {
        const int64_t e = p0 + i + grp_noff(T, p0 + i);
        ce = e < ce ? e : ce;
    }
    const int64_t cs = p0 + m - 1;                                  // tiles [cs, ce) lie in every member's half-range
    const int64_t own_end = w.pb + grp_noff(T, w.pb);
    if (ce < cs) ce = cs;                                           // (half-ranges shorter than the group: no common part)
    const int64_t lc = ce - cs;
    // virtual tiles 0 .. lc + (m - 1): the common ones, then the member's own; chunk c takes [total c / C, total (c + 1) / C),
    // the last chunk everything from its start
    const int64_t total = lc + (m - 1);
    int64_t b0 = total * w.chunk / chunks, b1 = total * (w.chunk + 1) / chunks;
    b0 = b0 < lc ? b0 : lc;
    b1 = b1 < lc ? b1 : lc;
    const bool last = w.chunk == chunks - 1;
    if (last) b1 = lc;
    w.c0 = cs + b0;
    w.ncommon = (int)(b1 - b0);
    w.h0 = w.pb;
    w.nhead = 0;
    w.t0 = ce;
    int ntail = 0;
    if (last) {
        const int64_t head_end = cs < own_end ? cs : own_end;
        w.nhead = (int)(head_end > w.pb ? head_end - w.pb : 0);
        ntail = (int)(own_end > ce ? own_end - ce : 0);
    }
    w.ntiles = w.ncommon + w.nhead + ntail;
    return w;
}

// tile map of a grouped item: common part (walked in lockstep with the other members of the group), then the member's own
struct GrpTiles {
    int c0, e1, e2, T;          // first common tile; jump at the start of the own tiles in front; jump at the start of those behind
    int ncommon, nown;          // tiles of the common part; ... + tiles in front (tile indices fit 32 bits: < 2^31 rows)
    AM_HD inline GrpTiles(int64_t c0_, int64_t h0, int64_t t0, int64_t T_, int ncommon_, int nhead)
        : c0((int)c0_), e1((int)(h0 - ncommon_ - c0_)), e2((int)(t0 - nhead - h0)), T((int)T_), ncommon(ncommon_), nown(ncommon_ + nhead) {}
    AM_HD inline int64_t operator()(int t) const {
        // q = c0 + t in the common part, h0 + (t - ncommon) in front, t0 + (t - ncommon - nhead) behind - written as two
        // additive jumps: a select over three BASES was compiled into a table on the stack (28 bytes of scratch per lane)
        const int q = c0 + t + (t >= ncommon ? e1 : 0) + (t >= nown ? e2 : 0);
        return q >= T ? q - T : q;
    }
};

}  // namespace am
