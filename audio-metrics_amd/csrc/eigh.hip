// PCA projection support (reference projection.py:6-46 delegates to scikit-learn's IncrementalPCA):
//   am_eigh_sym_f64   eigen-decomposition of the symmetric D x D Gram matrix of the stacked, centred batch
//   am_project_f64    (x - mean) . components^T for an N x D f32 matrix
//
// Eigensolver: one-sided BLOCK Jacobi (Hestenes) in f64.  W starts as A (rows = columns, A is symmetric), V as the
// identity; W = V A throughout.  At convergence the rows of W are mutually orthogonal: rows of V are the eigenvectors, and
// the eigenvalue of row i is the Rayleigh quotient W_i . V_i (signed, unlike |W_i|).
// Rows are grouped in blocks of 16.  A round pairs the blocks by the round-robin tournament (mb - 1 rounds per sweep) and
// makes the 32 rows of every pair mutually orthogonal in one go:
//   G = R R^T (32 x 32 Gram matrix of the pair's W rows, on the f64 matrix cores), a two-sided cyclic Jacobi of G in LDS that
//   accumulates the rotations into J (G is tiny: 16 x 16 rotations per visit, no row-length work), then R <- J^T R for
//   the pair's rows of W and of V (f64 matrix cores again).
// Round 2 rotated ONE row pair per workgroup and needed D - 1 = 511 launches of a latency-bound kernel per sweep (42 ms
// at D = 512); here a sweep is 31 rounds of three launches (Gram / inner solve / row update: the first and the last are
// spread over 8 workgroups per pair).  The rounds of several sweeps are enqueued ahead: a kernel of sweep s returns at once
// when sweep s - 1 applied no rotation (counter on the device), so the host synchronises once per block of sweeps -
// normally once per solve - instead of once per sweep.  All reductions run in a fixed order (deterministic).
#include "am_common.h"
#include <math.h>
#include <stdlib.h>
#include <algorithm>

namespace am {

typedef double f64x4e __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double block_sum3(double& a, double& b, double& c, double (*red)[3]) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        a += __shfl_xor(a, off);
        b += __shfl_xor(b, off);
        c += __shfl_xor(c, off);
    }
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6][0] = a;
        red[threadIdx.x >> 6][1] = b;
        red[threadIdx.x >> 6][2] = c;
    }
    __syncthreads();
    a = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
    b = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    c = (red[0][2] + red[1][2]) + (red[2][2] + red[3][2]);
    __syncthreads();
    return a;
}

// rank[i] = position of row i when the rows of A are ordered by decreasing diagonal entry (ties by index).  One-sided
// Jacobi converges in fewer sweeps when rows of similar norm sit together and the larger ones come first (de Rijk's
// ordering); for the Gram matrices of a PCA fit the diagonal is a good proxy of the final row norms.
__global__ void __launch_bounds__(256) jacobi_rank_kernel(const double* __restrict__ A, int n, int* __restrict__ rank) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double di = fabs(A[(int64_t)i * n + i]);
    int before = 0;
    for (int j = 0; j < n; ++j) {
        const double dj = fabs(A[(int64_t)j * n + j]);
        before += (dj > di || (dj == di && j < i)) ? 1 : 0;
    }
    rank[i] = before;
}

// W = P A, V = P (row rank[i] of W is row i of A)
__global__ void __launch_bounds__(256) jacobi_init_kernel(const double* __restrict__ A, int n, const int* __restrict__ rank,
                                                          double* __restrict__ W, double* __restrict__ V, double* __restrict__ scale) {
    const int64_t total = (int64_t)n * n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int row = (int)(i / n), col = (int)(i % n);
        const int64_t o = (int64_t)rank[row] * n + col;
        W[o] = A[i];
        V[o] = row == col ? 1.0 : 0.0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {        // trace A >= largest eigenvalue (A is positive semi-definite), fixed order
        double t = 0.0;
        for (int i = 0; i < n; ++i) t += fabs(A[(int64_t)i * n + i]);
        *scale = t;
    }
}

// 1 / sqrt(y) and 1 / y for positive normal y: hardware estimate + two Newton steps (quadratic: 2^-26 -> 2^-52)
__device__ __forceinline__ double rsqrt_f64(double y) {
    double r = __builtin_amdgcn_rsq(y);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double e = fma(-y * r, r, 1.0);          // 1 - y r^2
        r = fma(r * 0.5, e, r);
    }
    return r;
}
__device__ __forceinline__ double rcp_f64(double y) {
    double r = __builtin_amdgcn_rcp(y);
#pragma unroll
    for (int it = 0; it < 2; ++it) r = fma(fma(-y, r, 1.0), r, r);
    return r;
}

constexpr int JB = 16;            // rows per block
constexpr int JP = 2 * JB;        // rows a block pair orthogonalises
constexpr int JLD = JP + 1;       // LDS row stride of the 32 x 32 matrices (doubles)
constexpr int GRAM_SLICES = 8;    // workgroups that share the Gram matrix of one block pair (column slices)
constexpr int APPLY_SLICES = 8;   // workgroups that share the row update of one block pair (16-column strips dealt round-robin)
constexpr int APPLY_STRIPS = 2;   // strips a wave updates at a time

// Block pair `slot` of round `round`: (player(slot), player(mb - 1 - slot)) of the round-robin tournament over mb = number of
// blocks rounded up to even (player(0) = mb - 1 fixed, the others rotate); a block index >= nb is a bye.
__device__ __forceinline__ int tournament_player(int slot, int m, int r) { return slot == 0 ? m - 1 : (slot - 1 + r) % (m - 1); }

struct BlockPair {
    int bp, bq, n, nb;
    __device__ BlockPair(int slot, int mb, int round, int n_, int nb_) : n(n_), nb(nb_) {
        bp = tournament_player(slot, mb, round);
        bq = tournament_player(mb - 1 - slot, mb, round);
        if (bp > bq) { const int t = bp; bp = bq; bq = t; }
    }
    __device__ bool bye() const { return bp >= nb; }
    // local row r of the pair -> global row (or -1)
    __device__ int row(int r) const {
        const int b = r < JB ? bp : bq;
        const int g = b * JB + (r & (JB - 1));
        return (b < nb && g < n) ? g : -1;
    }
};

// A round of a sweep is three launches (round 3 had one workgroup per block pair do all of it - 16 CUs of 256 busy, 42 us per
// round at D = 512 of which 12 us were f64 MFMA time of ONE CU and as much again its load latency):
//   1. jacobi_gram_kernel   partial Gram matrices of the pair's 32 rows of W, one per column slice      (pairs x 8 workgroups)
//   2. jacobi_inner_kernel  G = sum of the partials; two-sided Jacobi of G in LDS, rotations accumulated in J      (pairs)
//   3. jacobi_apply_kernel  rows <- J^T rows for W and V, the 16-column strips dealt over the workgroups  (pairs x 8 workgroups)
// Every kernel of sweep s returns at once when sweep s - 1 applied no rotation.

// ---- 1. Gpart[pair][slice] = R[:, slice] R[:, slice]^T on v_mfma_f64_16x16x4_f64: wave w owns the 16 x 16 tile (w >> 1, w & 1)
__global__ void __launch_bounds__(256) jacobi_gram_kernel(const double* __restrict__ W, int n, int nb, int mb, int round, int sweep,
                                                          const unsigned* __restrict__ rotations, double* __restrict__ Gpart) {
    if (sweep > 0 && rotations[sweep - 1] == 0u) return;           // the previous sweep found every pair orthogonal: done
    const BlockPair bpair(blockIdx.x, mb, round, n, nb);
    if (bpair.bye()) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int ti = wave >> 1, tj = wave & 1;
    const int ra = bpair.row(ti * JB + l15), rb = bpair.row(tj * JB + l15);
    const double* wa = W + (int64_t)(ra < 0 ? 0 : ra) * n;
    const double* wb = W + (int64_t)(rb < 0 ? 0 : rb) * n;
    const int kslice = ((n + GRAM_SLICES - 1) / GRAM_SLICES + 15) / 16 * 16;
    const int kbeg = blockIdx.y * kslice, kend = min(n, kbeg + kslice);
    f64x4e acc = {0, 0, 0, 0};
    // The sum over k may run in any order as long as both operands use the same one: lane (l15, l4) takes the four
    // consecutive elements k0 + 4 l4 .. + 3 of its row per 16-element chunk (one 32-byte load per operand, 128 B per row
    // and instruction), eight chunks in flight - a load-then-MFMA loop is one exposed L2 round trip per MFMA.
    const bool vec = (n % 4) == 0;
    for (int k0 = kbeg; k0 < kend; k0 += 128) {
        double av[8][4], bw[8][4];
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) {
            const int k = k0 + ch * 16 + l4 * 4;
            if (vec && k + 3 < kend) {
                const f64x4e a4 = ra >= 0 ? *reinterpret_cast<const f64x4e*>(wa + k) : f64x4e{0, 0, 0, 0};
                const f64x4e b4 = rb >= 0 ? *reinterpret_cast<const f64x4e*>(wb + k) : f64x4e{0, 0, 0, 0};
#pragma unroll
                for (int j = 0; j < 4; ++j) { av[ch][j] = a4[j]; bw[ch][j] = b4[j]; }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    av[ch][j] = (ra >= 0 && k + j < kend) ? wa[k + j] : 0.0;
                    bw[ch][j] = (rb >= 0 && k + j < kend) ? wb[k + j] : 0.0;
                }
            }
        }
#pragma unroll
        for (int ch = 0; ch < 8; ++ch)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ch][j], bw[ch][j], acc, 0, 0, 0);
    }
    double* out = Gpart + ((int64_t)blockIdx.x * GRAM_SLICES + blockIdx.y) * (JP * JP);
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(ti * JB + l4 + 4 * r) * JP + tj * JB + l15] = acc[r];   // C layout: row = (lane >> 4) + 4 reg, column = lane & 15
}

// ---- 2. two-sided cyclic Jacobi of the pair's Gram matrix in LDS, rotations accumulated in J (written to Jout; applied[pair] = 1
//         when any rotation was applied).
// An inner round rotates 16 disjoint index pairs g = (p_g, q_g).  Thread (a, b) = (tid >> 4, tid & 15) owns the 2 x 2 block of G
// with rows {p_a, q_a} and columns {p_b, q_b}: G_ab <- R_a^T G_ab R_b, and the same block of J: J_ab <- J_ab R_b.  Every thread
// derives the two rotations it needs itself from the six diagonal-block entries (the same arithmetic on the same inputs in
// every thread that needs a rotation: all of them agree), and the matrices ping-pong between two LDS copies - so a round is ONE
// LDS round trip, one rsqrt chain and one barrier (round 3 rotated rows, barrier, then columns, barrier, with the pair's sixteen
// threads waiting for the chain in between: 0.87 us per inner round, 14 of the 42 us of a visit).
// ONE inner sweep per visit: the pair's rows meet again in the next outer sweep, and a full diagonalisation of G here (5-8 inner
// sweeps while the off-diagonal mass is large) costs more than the outer sweeps it saves - measured 54 ms against 28.
__global__ void __launch_bounds__(256) jacobi_inner_kernel(const double* __restrict__ Gpart, int n, int nb, int mb, int round, int sweep,
                                                           double tol, const double* __restrict__ scale, unsigned* __restrict__ rotations,
                                                           double* __restrict__ Jout, int* __restrict__ applied_out) {
    if (sweep > 0 && rotations[sweep - 1] == 0u) return;
    const BlockPair bpair(blockIdx.x, mb, round, n, nb);
    if (bpair.bye()) return;
    // rows whose squared norm is below (1e-14 trace A)^2 are numerically zero (null directions of a rank-deficient A: their
    // content is rounding noise, whose mutual angles never settle - 3x the sweeps on a Gram matrix of D/3 rows)
    const double floor2 = (1e-14 * *scale) * (1e-14 * *scale), tol2 = tol * tol;
    __shared__ double G[2][JP][JLD], J[2][JP][JLD];
    __shared__ int applied;
    const int tid = threadIdx.x;
    const double* gp = Gpart + (int64_t)blockIdx.x * GRAM_SLICES * (JP * JP);
    for (int e = tid; e < JP * JP; e += 256) {
        double g = 0.0;
#pragma unroll
        for (int sl = 0; sl < GRAM_SLICES; ++sl) g += gp[sl * (JP * JP) + e];          // fixed order: deterministic
        G[0][e / JP][e % JP] = g;
        J[0][e / JP][e % JP] = (e / JP == e % JP) ? 1.0 : 0.0;
    }
    if (tid == 0) applied = 0;
    __syncthreads();
    // Index pairs of a visit: in round 0 of a sweep (every block is in exactly one pair there) all 32 * 31 / 2 pairs of the
    // 32 rows, 31 inner rounds; in the other rounds only the 16 x 16 pairs ACROSS the two blocks (16 inner rounds: row g
    // of the first block with row (g + ir) mod 16 of the second) - the pairs inside a block have met in round 0, and
    // rotating them again in each of the 31 visits of a sweep buys nothing (19 -> 16 ms at D = 512, one sweep more).
    const bool all_pairs = round == 0;
    const int inner_rounds = all_pairs ? JP - 1 : JB;
    const int ga = tid >> 4, gb = tid & 15;
    auto index_pair = [&](int g, int ir, int& p, int& q) {
        if (all_pairs) {
            p = tournament_player(g, JP, ir);
            q = tournament_player(JP - 1 - g, JP, ir);
            if (p > q) { const int t = p; p = q; q = t; }
        } else {
            p = g;
            q = JB + ((g + ir) & (JB - 1));
        }
    };
    // tan(theta) = sign(d) 2 g_pq / (|d| + h), h = sqrt(d^2 + 4 g_pq^2), d = g_qq - g_pp, in the form that needs two reciprocal
    // square roots and no division: c^2 = (1 + |d| / h) / 2, c = c^2 / sqrt(c^2), s = sign(d) g_pq / (h sqrt(c^2)) - and
    // c^2 + s^2 = ((1 + x)^2 + (1 - x^2)) / (2 (1 + x)) = 1 for x = |d| / h.  The library sqrt and divide are ~200-cycle
    // sequences each and this chain is the critical path of an inner round: hardware estimates with two Newton steps instead
    // (full f64 accuracy: c^2 + s^2 = 1 to 1e-16, which is what keeps V orthonormal).
    auto rotation = [&](double gpp, double gqq, double gpq, double& c, double& sn) {
        c = 1.0;
        sn = 0.0;
        if (gpq * gpq > tol2 * (gpp * gqq) && fmin(gpp, gqq) > floor2) {      // not yet orthogonal, neither row (numerically) zero
            const double d = gqq - gpp, g2 = 2.0 * gpq;
            const double rh = rsqrt_f64(fma(d, d, g2 * g2));
            const double c2 = fma(0.5 * fabs(d), rh, 0.5);
            const double rc = rsqrt_f64(c2);
            c = c2 * rc;
            sn = (d >= 0 ? gpq : -gpq) * rh * rc;
            return true;
        }
        return false;
    };
    bool mine = false;
    int cur = 0;
    for (int ir = 0; ir < inner_rounds; ++ir, cur ^= 1) {
        int pa, qa, pb, qb;
        index_pair(ga, ir, pa, qa);
        index_pair(gb, ir, pb, qb);
        const double (*Gc)[JLD] = G[cur];
        const double (*Jc)[JLD] = J[cur];
        double ca, sa, cb, sb;
        const bool ra = rotation(Gc[pa][pa], Gc[qa][qa], Gc[pa][qa], ca, sa);
        rotation(Gc[pb][pb], Gc[qb][qb], Gc[pb][qb], cb, sb);
        if (ga == gb && ra) mine = true;
        // rows {pa, qa} x columns {pb, qb}: rows first (R_a^T from the left: new row p = c p - s q, new row q = s p + c q), then columns
        const double g00 = Gc[pa][pb], g01 = Gc[pa][qb], g10 = Gc[qa][pb], g11 = Gc[qa][qb];
        const double r00 = ca * g00 - sa * g10, r01 = ca * g01 - sa * g11;
        const double r10 = sa * g00 + ca * g10, r11 = sa * g01 + ca * g11;
        double (*Gn)[JLD] = G[cur ^ 1];
        Gn[pa][pb] = cb * r00 - sb * r01;
        Gn[pa][qb] = sb * r00 + cb * r01;
        Gn[qa][pb] = cb * r10 - sb * r11;
        Gn[qa][qb] = sb * r10 + cb * r11;
        // J <- J R: columns {pb, qb} of rows {pa, qa}
        const double j00 = Jc[pa][pb], j01 = Jc[pa][qb], j10 = Jc[qa][pb], j11 = Jc[qa][qb];
        double (*Jn)[JLD] = J[cur ^ 1];
        Jn[pa][pb] = cb * j00 - sb * j01;
        Jn[pa][qb] = sb * j00 + cb * j01;
        Jn[qa][pb] = cb * j10 - sb * j11;
        Jn[qa][qb] = sb * j10 + cb * j11;
        __syncthreads();
    }
    if (mine) applied = 1;
    __syncthreads();
    const bool any = applied != 0;
    if (tid == 0) {
        applied_out[blockIdx.x] = any ? 1 : 0;
        if (any) atomicAdd(rotations + sweep, 1u);
    }
    if (!any) return;                                                // the 32 rows were orthogonal already: nothing to apply
    double* jo = Jout + (int64_t)blockIdx.x * (JP * JP);
    for (int e = tid; e < JP * JP; e += 256) jo[e] = J[cur][e / JP][e % JP];
}

// ---- 3. rows <- J^T rows for W and V: out[r'][c] = sum_r J[r][r'] in[r][c].  A wave takes APPLY_STRIPS 16-column strips at a time: it
//         loads the strips' 32 x 16 blocks (eight k-steps of four rows), forms both 16-row output tiles of each, stores them
//         back (16 loads in flight and four independent MFMA chains; 128 workgroups per round at D = 512).
__global__ void __launch_bounds__(256) jacobi_apply_kernel(double* __restrict__ W, double* __restrict__ V, int n, int nb, int mb,
                                                           int round, int sweep, const unsigned* __restrict__ rotations,
                                                           const double* __restrict__ Jin, const int* __restrict__ applied_in) {
    if (sweep > 0 && rotations[sweep - 1] == 0u) return;
    const BlockPair bpair(blockIdx.x, mb, round, n, nb);
    if (bpair.bye() || applied_in[blockIdx.x] == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const double* Jm = Jin + (int64_t)blockIdx.x * (JP * JP);
    double j0[8], j1[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        j0[ks] = Jm[(ks * 4 + l4) * JP + l15];
        j1[ks] = Jm[(ks * 4 + l4) * JP + JB + l15];
    }
    const int strips = (n + 15) / 16;
    for (int st0 = (blockIdx.y * 4 + wave) * APPLY_STRIPS; st0 < 2 * strips; st0 += APPLY_SLICES * 4 * APPLY_STRIPS) {
        double bv[APPLY_STRIPS][8];
#pragma unroll
        for (int u = 0; u < APPLY_STRIPS; ++u) {
            const int st = st0 + u;
            const double* M = st < strips ? W : V;
            const int c = (st % strips) * 16 + l15;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int g = bpair.row(ks * 4 + l4);
                bv[u][ks] = (st < 2 * strips && g >= 0 && c < n) ? M[(int64_t)g * n + c] : 0.0;
            }
        }
        f64x4e acc[APPLY_STRIPS][2];
#pragma unroll
        for (int u = 0; u < APPLY_STRIPS; ++u)
#pragma unroll
            for (int to = 0; to < 2; ++to) acc[u][to] = f64x4e{0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
            for (int u = 0; u < APPLY_STRIPS; ++u) {
                acc[u][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(j0[ks], bv[u][ks], acc[u][0], 0, 0, 0);
                acc[u][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(j1[ks], bv[u][ks], acc[u][1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < APPLY_STRIPS; ++u) {
            const int st = st0 + u;
            if (st >= 2 * strips) continue;
            double* M = st < strips ? W : V;
            const int c = (st % strips) * 16 + l15;
#pragma unroll
            for (int to = 0; to < 2; ++to)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int g = bpair.row(to * JB + l4 + 4 * r);
                    if (g >= 0 && c < n) M[(int64_t)g * n + c] = acc[u][to][r];
                }
        }
    }
}

// lambda_i = W_i . V_i
__global__ void __launch_bounds__(256) jacobi_values_kernel(const double* __restrict__ W, const double* __restrict__ V, int n,
                                                            double* __restrict__ lambda) {
    __shared__ double red[4][3];
    const int i = blockIdx.x;
    double a = 0, b = 0, c = 0;
    for (int k = threadIdx.x; k < n; k += 256) a = fma(W[(int64_t)i * n + k], V[(int64_t)i * n + k], a);
    block_sum3(a, b, c, red);
    if (threadIdx.x == 0) lambda[i] = a;
}

// descending order: position of eigenpair i = number of pairs that come before it (ties by index)
__global__ void __launch_bounds__(256) jacobi_sort_kernel(const double* __restrict__ lambda, const double* __restrict__ V, int n,
                                                          double* __restrict__ evals, double* __restrict__ evecs) {
    const int i = blockIdx.x;
    __shared__ int pos_s;
    __shared__ int cnt[4];
    const double li = lambda[i];
    int before = 0;
    for (int j = threadIdx.x; j < n; j += 256) {
        const double lj = lambda[j];
        before += (lj > li || (lj == li && j < i)) ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) before += __shfl_xor(before, off);
    if ((threadIdx.x & 63) == 0) cnt[threadIdx.x >> 6] = before;
    __syncthreads();
    if (threadIdx.x == 0) pos_s = cnt[0] + cnt[1] + cnt[2] + cnt[3];
    __syncthreads();
    const int pos = pos_s;
    if (threadIdx.x == 0) evals[pos] = li;
    for (int k = threadIdx.x; k < n; k += 256) evecs[(int64_t)pos * n + k] = V[(int64_t)i * n + k];
}

// out[row][j] = sum_d (X[row][d] - mean[d]) * C[j][d]   on v_mfma_f64_16x16x4_f64: a workgroup takes 64 rows x 16
// components, each wave 16 rows; lane (l15, l4) feeds element d = d0 + l4 of row / component l15
template <class TX>
__global__ void __launch_bounds__(256) project_kernel(const TX* __restrict__ X, int64_t N, int64_t ld, int D,
                                                      const double* __restrict__ mean, const double* __restrict__ C, int p,
                                                      double* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t row = (int64_t)blockIdx.x * 64 + wave * 16 + l15;
    const int comp = blockIdx.y * 16 + l15;
    const bool row_ok = row < N, comp_ok = comp < p;
    const TX* x = X + (row_ok ? row : 0) * ld;
    const double* c = C + (int64_t)(comp_ok ? comp : 0) * D;
    f64x4e acc = {0, 0, 0, 0};
    for (int d0 = 0; d0 < D; d0 += 4) {
        const int d = d0 + l4;
        const bool in = d < D;
        const double a = (row_ok && in) ? (double)x[d] - mean[d] : 0.0;
        const double b = (comp_ok && in) ? c[d] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    // C/D layout: column (component) = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t orow = (int64_t)blockIdx.x * 64 + wave * 16 + l4 + 4 * r;
        if (orow < N && comp_ok) out[orow * p + comp] = acc[r];
    }
}

}  // namespace am

using namespace am;

extern "C" size_t am_eigh_workspace_bytes(int D) {
    if (D < 1) return 0;
    Carver c(nullptr, 0);
    c.take<double>((size_t)D * D);     // W
    c.take<double>((size_t)D * D);     // V
    c.take<double>((size_t)D);         // lambda (unsorted)
    c.take<unsigned>(64);              // rotation counters, one per sweep
    c.take<double>(1);                 // trace of A
    c.take<int>(D);                    // initial row order
    const size_t pairs = (size_t)((D + JB - 1) / JB + 1) / 2;
    c.take<double>(pairs * GRAM_SLICES * JP * JP);   // partial Gram matrices of a round
    c.take<double>(pairs * JP * JP);                 // accumulated rotations of a round
    c.take<int>(pairs);                              // block pairs that rotated in a round
    return c.off;
}

extern "C" int am_eigh_sym_f64(const double* A, int D, double* evals, double* evecs, int max_sweeps, void* ws, size_t ws_bytes,
                               am_stream_t stream) {
    AM_REQUIRE(A && evals && evecs, AM_ERR_BAD_ARG, "null pointer");
    AM_REQUIRE(D >= 1, AM_ERR_BAD_SHAPE, "D=%d", D);
    if (max_sweeps <= 0 || max_sweeps > 64) max_sweeps = 40;
    hipStream_t st = static_cast<hipStream_t>(stream);
    Carver c(ws, ws_bytes);
    double* W = c.take<double>((size_t)D * D);
    double* V = c.take<double>((size_t)D * D);
    double* lambda = c.take<double>((size_t)D);
    unsigned* rotations = c.take<unsigned>(64);
    double* scale = c.take<double>(1);
    int* rank = c.take<int>(D);
    const size_t pairs = (size_t)((D + JB - 1) / JB + 1) / 2;
    double* Gpart = c.take<double>(pairs * GRAM_SLICES * JP * JP);
    double* Jbuf = c.take<double>(pairs * JP * JP);
    int* applied = c.take<int>(pairs);
    AM_REQUIRE(c.ok(), AM_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", c.off, ws_bytes);
    hipLaunchKernelGGL(jacobi_rank_kernel, dim3((unsigned)ceil_div(D, 256)), dim3(256), 0, st, A, D, rank);
    hipLaunchKernelGGL(jacobi_init_kernel, dim3((unsigned)std::min<int64_t>(1024, ceil_div((int64_t)D * D, 256))), dim3(256), 0, st, A, D,
                       rank, W, V, scale);
    AM_LAUNCH_CHECK();
    AM_HIP_TRY(hipMemsetAsync(rotations, 0, 64 * sizeof(unsigned), st));
    const int nb = (D + JB - 1) / JB, mb = (nb + 1) / 2 * 2;
    const double tol = 1e-14 * sqrt((double)D);       // |W_p . W_q| <= tol |W_p| |W_q|: the rounding level of a D-term f64 dot product
    // Sweeps are enqueued in blocks; a round kernel returns at once when the sweep before it applied no rotation, so
    // running ahead costs only empty launches.  One read-back per block: normally one per solve.
    bool converged = false;
    unsigned counts[64];
    for (int done = 0; done < max_sweeps && !converged;) {
        const int block = std::min(done == 0 ? 12 : 6, max_sweeps - done);
        for (int sweep = done; sweep < done + block; ++sweep)
            for (int round = 0; round < mb - 1; ++round) {
                hipLaunchKernelGGL(jacobi_gram_kernel, dim3(mb / 2, GRAM_SLICES), dim3(256), 0, st, W, D, nb, mb, round, sweep, rotations, Gpart);
                hipLaunchKernelGGL(jacobi_inner_kernel, dim3(mb / 2), dim3(256), 0, st, Gpart, D, nb, mb, round, sweep, tol, scale, rotations,
                                   Jbuf, applied);
                hipLaunchKernelGGL(jacobi_apply_kernel, dim3(mb / 2, APPLY_SLICES), dim3(256), 0, st, W, V, D, nb, mb, round, sweep, rotations,
                                   Jbuf, applied);
            }
        AM_LAUNCH_CHECK();
        done += block;
        AM_HIP_TRY(hipMemcpyAsync(counts, rotations, (size_t)done * sizeof(unsigned), hipMemcpyDeviceToHost, st));
        AM_HIP_TRY(hipStreamSynchronize(st));
        for (int sweep = 0; sweep < done; ++sweep) converged = converged || counts[sweep] == 0u;
#ifdef AM_DEV_KNOBS
        if (getenv("AM_EIGH_DEBUG")) {                 // block pairs that rotated, per sweep (development aid)
            fprintf(stderr, "[am eigh] D=%d sweeps enqueued %d:", D, done);
            for (int sweep = 0; sweep < done; ++sweep) fprintf(stderr, " %u", counts[sweep]);
            fprintf(stderr, "\n");
        }
#endif
    }
    AM_REQUIRE(converged, AM_ERR_NO_CONVERGENCE, "Jacobi eigensolver: rows still not orthogonal after %d sweeps", max_sweeps);
    hipLaunchKernelGGL(jacobi_values_kernel, dim3(D), dim3(256), 0, st, W, V, D, lambda);
    hipLaunchKernelGGL(jacobi_sort_kernel, dim3(D), dim3(256), 0, st, lambda, V, D, evals, evecs);
    AM_LAUNCH_CHECK();
    return AM_OK;
}

extern "C" int am_project_f64(const float* X, int64_t N, int64_t ld, int D, const double* mean, const double* components, int p,
                              double* out, am_stream_t stream) {
    AM_REQUIRE(X && mean && components && out, AM_ERR_BAD_ARG, "null pointer");
    AM_REQUIRE(N >= 1 && D >= 1 && p >= 1 && ld >= D, AM_ERR_BAD_SHAPE, "N=%lld D=%d p=%d ld=%lld", (long long)N, D, p, (long long)ld);
    hipLaunchKernelGGL(project_kernel<float>, dim3((unsigned)ceil_div(N, 64), (unsigned)ceil_div(p, 16)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), X, N, ld, D, mean, components, p, out);
    AM_LAUNCH_CHECK();
    return AM_OK;
}

extern "C" int am_project_rows_f64(const double* X, int64_t N, int64_t ld, int D, const double* mean, const double* components, int p,
                                   double* out, am_stream_t stream) {
    AM_REQUIRE(X && mean && components && out, AM_ERR_BAD_ARG, "null pointer");
    AM_REQUIRE(N >= 1 && D >= 1 && p >= 1 && ld >= D, AM_ERR_BAD_SHAPE, "N=%lld D=%d p=%d ld=%lld", (long long)N, D, p, (long long)ld);
    hipLaunchKernelGGL(project_kernel<double>, dim3((unsigned)ceil_div(N, 64), (unsigned)ceil_div(p, 16)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), X, N, ld, D, mean, components, p, out);
    AM_LAUNCH_CHECK();
    return AM_OK;
}
