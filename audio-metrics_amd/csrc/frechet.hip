// Frechet distance between two Gaussians (reference fad.py:16-31):
//     fd = |mu_x - mu_y|^2 + tr(Cx) + tr(Cy) - 2 tr sqrt(Cx Cy)
// The reference takes LAPACK's general eigenvalues of Cx*Cy.  Here tr sqrt(A),
// A = Cx*Cy, comes from the coupled Newton-Schulz iteration in f64 on the f64
// matrix cores (v_mfma_f64_16x16x4_f64):
//     Y0 = A / |A|_F,  Z0 = I;   T = (3I - Z Y)/2;   Y <- Y T;   Z <- T Z
//     Y -> sqrt(A/|A|_F),   tr sqrt(A) = sqrt(|A|_F) * tr(Y)
// Every eigen-component of Y grows monotonically towards its limit, so tr(Y) is
// non-decreasing in exact arithmetic.  Rank-deficient products (N < D) carry
// rounding-noise eigenvalues of either sign in their null space; the negative
// ones eventually blow up.  Stopping rule (validated against eigvals on
// well-conditioned, decaying-spectrum and rank-deficient inputs):
//     stop when |I - Z Y|_F < tol*sqrt(D)            (converged), or
//     when tr(Y) decreases / turns non-finite        (noise took over: keep previous trace), or
//     at max_iter.
//
// The solve is ONE stream-ordered chain without host involvement: a block of iterations is enqueued up front, the
// stopping rule is evaluated on the device by every workgroup of the update kernel (same inputs, same fixed
// summation order, hence the same decision - no inter-workgroup hand-off), kernels behind the stopping point return
// at once, and a final kernel leaves {fd, tr_sqrt, iterations, residual, stop code} in device memory.  The host reads
// those five doubles once; only a solve that needs more than the first block (ill-conditioned products) continues
// with further blocks.  An iteration is two launches:
//     ns_t_kernel      T = 1.5 I - 0.5 Z Y; per-tile sums of (I - Z Y)^2 and of diag(Y)
//     ns_update_kernel stopping rule from those sums -> next state; Y' = Y T and Z' = T Z (gridDim.z = 2)
// The three D x D x D products of an iteration are latency-bound (8e8 flop at D = 512): 32 x 32 output tiles give
// 256 / 512 workgroups, and inside a workgroup the four waves split the inner dimension (each streams its own operand
// slices global -> registers, two 16-deep slabs in flight) and their partial tiles are added in a fixed order
// through LDS.  D = 512: ~25 us per iteration against 70 us for the 64 x 64-tile form with a separate check kernel.
#include "am_common.h"
#include <math.h>
#include <algorithm>
#include <mutex>

namespace am {

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int GT = 32;            // output tile (GT x GT), every wave computes all of it over its share of k
constexpr int GKC = 16;           // inner slab: lane (l15, l4) holds k = slab + 4 l4 + s, s = 0..3
constexpr int PST = GT + 1;       // row stride of the partial tiles in LDS

struct NsState {
    double prev_trace;            // last accepted tr(Y)
    double resid;                 // |I - ZY|_F at the last check
    double norm;                  // |A|_F
    int iters;
    int done;                     // 0 running, 1 converged, 2 trace stalled (noise), 3 zero matrix, 4 non-finite input
};

enum { MODE_PLAIN = 0, MODE_NS_T = 1 };

struct GemmJob {
    const double* A;
    const double* B;
    double* C;
};

// sum of `count` doubles by the whole workgroup in a fixed order (thread-strided partials, xor butterfly, 4 waves)
__device__ __forceinline__ double block_sum(double v, double* red) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
    __syncthreads();
    return s;
}

// 32 x 32 tile of A*B at (row0, col0); result of the four waves' k-shares combined in `part` (LDS, [4][GT * PST]).
// After the call thread t owns elements e = t, t + 256, t + 512, t + 768 (row e / 32, column e % 32) in out[4].
__device__ __forceinline__ void tile_product(const GemmJob& job, int n, int row0, int col0, double* part, double (&out)[4]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    f64x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f64x4{0, 0, 0, 0};
    const int nslab = (n + GKC - 1) / GKC;
    const bool vec = (n % 4 == 0) && ((reinterpret_cast<uintptr_t>(job.A) & 31u) == 0);
    double ra0[2][4], rb0[2][4], ra1[2][4], rb1[2][4];      // two register buffers, [tile half][s]
    auto fetch = [&](double (&ra)[2][4], double (&rb)[2][4], int slab) {
        const int k0 = slab * GKC + 4 * l4;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int gr = row0 + t * 16 + l15;
            if (vec && gr < n && k0 + 3 < n) {
                const f64x4 v = *reinterpret_cast<const f64x4*>(job.A + (int64_t)gr * n + k0);
#pragma unroll
                for (int s = 0; s < 4; ++s) ra[t][s] = v[s];
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) ra[t][s] = (gr < n && k0 + s < n) ? job.A[(int64_t)gr * n + k0 + s] : 0.0;
            }
            const int gc = col0 + t * 16 + l15;
#pragma unroll
            for (int s = 0; s < 4; ++s) rb[t][s] = (gc < n && k0 + s < n) ? job.B[(int64_t)(k0 + s) * n + gc] : 0.0;
        }
    };
    auto multiply = [&](const double (&ra)[2][4], const double (&rb)[2][4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[mt][s], rb[nt][s], acc[mt][nt], 0, 0, 0);
    };
    // wave w takes slabs w, w + 4, ...; the next slab is in flight while this one multiplies
    int slab = wave;
    if (slab < nslab) fetch(ra0, rb0, slab);
    while (slab < nslab) {
        if (slab + 4 < nslab) fetch(ra1, rb1, slab + 4);
        multiply(ra0, rb0);
        slab += 4;
        if (slab >= nslab) break;
        if (slab + 4 < nslab) fetch(ra0, rb0, slab + 4);
        multiply(ra1, rb1);
        slab += 4;
    }
    // f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
    double* mine = part + wave * GT * PST;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[(mt * 16 + l4 + 4 * r) * PST + nt * 16 + l15] = acc[mt][nt][r];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = tid + 256 * q;
        const int o = (e >> 5) * PST + (e & 31);
        out[q] = (part[o] + part[GT * PST + o]) + (part[2 * GT * PST + o] + part[3 * GT * PST + o]);
    }
    __syncthreads();
}

// A = Cx * Cy and per-tile sums of A^2 (for |A|_F)
__global__ void __launch_bounds__(256) ns_product_kernel(GemmJob job, int n, double* __restrict__ tile_sums) {
    __shared__ __attribute__((aligned(16))) double part[4 * GT * PST];
    __shared__ double red[4];
    const int row0 = blockIdx.y * GT, col0 = blockIdx.x * GT;
    double v[4];
    tile_product(job, n, row0, col0, part, v);
    double sq = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = threadIdx.x + 256 * q;
        const int gr = row0 + (e >> 5), gc = col0 + (e & 31);
        if (gr < n && gc < n) {
            job.C[(int64_t)gr * n + gc] = v[q];
            sq += v[q] * v[q];
        }
    }
    sq = block_sum(sq, red);
    if (threadIdx.x == 0) tile_sums[blockIdx.y * gridDim.x + blockIdx.x] = sq;
}

// norm = sqrt(sum tile_sums); Y = A/norm; Z = I; state init.  Every workgroup recomputes the (tiny) norm sum in
// the same fixed order; block 0 writes the state.
__global__ void __launch_bounds__(256) ns_init_kernel(const double* __restrict__ A, const double* __restrict__ tile_sums,
                                                      int ntiles, int n, double* __restrict__ Y, double* __restrict__ Z,
                                                      NsState* __restrict__ state) {
    __shared__ double red[4];
    double v = 0;
    for (int i = threadIdx.x; i < ntiles; i += blockDim.x) v += tile_sums[i];
    const double nrm = sqrt(block_sum(v, red));
    const bool bad = !(nrm == nrm) || isinf(nrm);
    const double inv = (nrm > 0.0 && !bad) ? 1.0 / nrm : 0.0;
    const int64_t total = (int64_t)n * n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        Y[i] = A[i] * inv;
        Z[i] = (i / n == i % n) ? 1.0 : 0.0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        state->prev_trace = (nrm > 0.0 && !bad) ? -INFINITY : 0.0;
        state->resid = 0.0;
        state->norm = bad ? 0.0 : nrm;
        state->iters = 0;
        state->done = bad ? 4 : (nrm > 0.0 ? 0 : 3);
    }
}

// T = 1.5 I - 0.5 Z Y;  resid_sums[tile] = sum (I - Z Y)^2 over the tile;  trace_sums[diagonal tile] = sum of diag(Y)
__global__ void __launch_bounds__(256) ns_t_kernel(GemmJob job /* A = Z, B = Y, C = T */, int n, const NsState* __restrict__ state,
                                                   double* __restrict__ resid_sums, double* __restrict__ trace_sums) {
    if (state->done) return;
    __shared__ __attribute__((aligned(16))) double part[4 * GT * PST];
    __shared__ double red[4];
    const int row0 = blockIdx.y * GT, col0 = blockIdx.x * GT;
    double v[4];
    tile_product(job, n, row0, col0, part, v);
    double sq = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = threadIdx.x + 256 * q;
        const int gr = row0 + (e >> 5), gc = col0 + (e & 31);
        if (gr < n && gc < n) {
            const double eye = (gr == gc) ? 1.0 : 0.0;
            const double d = eye - v[q];
            job.C[(int64_t)gr * n + gc] = 1.5 * eye - 0.5 * v[q];
            sq += d * d;
        }
    }
    sq = block_sum(sq, red);
    if (threadIdx.x == 0) resid_sums[blockIdx.y * gridDim.x + blockIdx.x] = sq;
    if (blockIdx.x == blockIdx.y) {
        const int i = row0 + (int)threadIdx.x;
        const double t = (threadIdx.x < GT && i < n) ? job.B[(int64_t)i * n + i] : 0.0;
        const double ts = block_sum(t, red);
        if (threadIdx.x == 0) trace_sums[blockIdx.x] = ts;
    }
}

// The stopping rule (see file header) as a pure function of the previous state and the sums of ns_t_kernel: every
// workgroup evaluates it and gets the same answer; only one of them writes it down.
__device__ __forceinline__ NsState ns_next_state(const NsState& s, const double* __restrict__ resid_sums, int ntiles,
                                                 const double* __restrict__ trace_sums, int g, int n, double tol, double* red) {
    double r = 0, t = 0;
    for (int i = threadIdx.x; i < ntiles; i += blockDim.x) r += resid_sums[i];
    for (int i = threadIdx.x; i < g; i += blockDim.x) t += trace_sums[i];
    r = sqrt(block_sum(r, red));
    t = block_sum(t, red);
    NsState o = s;
    const bool finite = (t == t) && !isinf(t) && (r == r);
    if (!finite || t < s.prev_trace * (1.0 - 1e-14) - 1e-300) {
        o.done = isinf(s.prev_trace) ? 4 : 2;               // keep the previous trace
    } else {
        o.prev_trace = t;
        o.resid = r;
        o.iters = s.iters + 1;
        if (r < tol * sqrt((double)n)) o.done = 1;
    }
    return o;
}

// state_out = rule(state_in); unless stopped: Y' = Y T (z = 0), Z' = T Z (z = 1).  products == 0: rule only.
__global__ void __launch_bounds__(256) ns_update_kernel(GemmJob jy, GemmJob jz, int n, const NsState* __restrict__ state_in,
                                                        NsState* __restrict__ state_out, const double* __restrict__ resid_sums,
                                                        const double* __restrict__ trace_sums, int g, double tol, int products) {
    __shared__ __attribute__((aligned(16))) double part[4 * GT * PST];
    __shared__ double red[4];
    const NsState s = *state_in;
    const bool writer = blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0;
    if (s.done) {
        if (writer) *state_out = s;
        return;
    }
    const NsState o = ns_next_state(s, resid_sums, g * g, trace_sums, g, n, tol, red);
    if (writer) *state_out = o;
    if (o.done || !products) return;
    const GemmJob job = blockIdx.z == 0 ? jy : jz;
    const int row0 = blockIdx.y * GT, col0 = blockIdx.x * GT;
    double v[4];
    tile_product(job, n, row0, col0, part, v);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = threadIdx.x + 256 * q;
        const int gr = row0 + (e >> 5), gc = col0 + (e & 31);
        if (gr < n && gc < n) job.C[(int64_t)gr * n + gc] = v[q];
    }
}

// out = { fd, tr_sqrt, iterations, residual, stop code }
__global__ void __launch_bounds__(256) fd_finish_kernel(const double* __restrict__ mu_x, const double* __restrict__ cov_x,
                                                        const double* __restrict__ mu_y, const double* __restrict__ cov_y,
                                                        int n, const NsState* __restrict__ state,
                                                        double* __restrict__ out) {
    __shared__ double red[4];
    double a = 0, b = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double d = mu_x[i] - mu_y[i];
        a += d * d;
        b += cov_x[(int64_t)i * n + i] + cov_y[(int64_t)i * n + i];
    }
    a = block_sum(a, red);
    b = block_sum(b, red);
    if (threadIdx.x == 0) {
        const double tr = (state->done == 3) ? 0.0 : state->prev_trace * sqrt(state->norm);
        out[0] = a + b - 2.0 * tr;
        out[1] = tr;
        out[2] = (double)state->iters;
        out[3] = state->resid;
        out[4] = (double)state->done;
    }
}

struct NsBuffers {
    double *A, *Y[2], *Z[2], *T, *resid_sums, *trace_sums;
    NsState* state;               // [2]: ping-pong, iteration `it` reads state[it & 1] and writes state[(it + 1) & 1]
    int g;
};

static bool carve_ns(Carver& c, int D, NsBuffers& b) {
    const size_t dd = (size_t)D * D;
    b.g = (int)ceil_div(D, GT);
    b.A = c.take<double>(dd);
    b.Y[0] = c.take<double>(dd);
    b.Y[1] = c.take<double>(dd);
    b.Z[0] = c.take<double>(dd);
    b.Z[1] = c.take<double>(dd);
    b.T = c.take<double>(dd);
    b.resid_sums = c.take<double>((size_t)b.g * b.g);
    b.trace_sums = c.take<double>((size_t)b.g);
    b.state = c.take<NsState>(2);
    return c.ok();
}

// iterations [first_iter, first_iter + n_iter) of the chain; `last_block`: the iteration budget ends with this block, so
// the update of its last iteration is accounted for by one more evaluation of the rule
static int enqueue_ns(const double* mu_x, const double* cov_x, const double* mu_y, const double* cov_y, int D, int first_iter,
                      int n_iter, bool last_block, double tol, const NsBuffers& b, double* out_dev, hipStream_t st) {
    const int g = b.g;
    const dim3 grid1(g, g, 1), grid2(g, g, 2), blk(256);
    if (first_iter == 0) {
        hipLaunchKernelGGL(ns_product_kernel, grid1, blk, 0, st, GemmJob{cov_x, cov_y, b.A}, D, b.resid_sums);
        AM_LAUNCH_CHECK();
        hipLaunchKernelGGL(ns_init_kernel, dim3((unsigned)std::min<int64_t>(256, ceil_div((int64_t)D * D, 1024))), blk, 0, st, b.A,
                           b.resid_sums, g * g, D, b.Y[0], b.Z[0], b.state);
        AM_LAUNCH_CHECK();
    }
    for (int it = first_iter; it < first_iter + n_iter; ++it) {
        const int cur = it & 1;
        hipLaunchKernelGGL(ns_t_kernel, grid1, blk, 0, st, GemmJob{b.Z[cur], b.Y[cur], b.T}, D, (const NsState*)(b.state + cur),
                           b.resid_sums, b.trace_sums);
        hipLaunchKernelGGL(ns_update_kernel, grid2, blk, 0, st, GemmJob{b.Y[cur], b.T, b.Y[cur ^ 1]},
                           GemmJob{b.T, b.Z[cur], b.Z[cur ^ 1]}, D, (const NsState*)(b.state + cur), b.state + (cur ^ 1),
                           (const double*)b.resid_sums, (const double*)b.trace_sums, g, tol, 1);
        AM_LAUNCH_CHECK();
    }
    int fin = (first_iter + n_iter) & 1;
    if (last_block) {
        hipLaunchKernelGGL(ns_t_kernel, grid1, blk, 0, st, GemmJob{b.Z[fin], b.Y[fin], b.T}, D, (const NsState*)(b.state + fin),
                           b.resid_sums, b.trace_sums);
        hipLaunchKernelGGL(ns_update_kernel, dim3(1, 1, 1), blk, 0, st, GemmJob{nullptr, nullptr, nullptr},
                           GemmJob{nullptr, nullptr, nullptr}, D, (const NsState*)(b.state + fin), b.state + (fin ^ 1),
                           (const double*)b.resid_sums, (const double*)b.trace_sums, g, tol, 0);
        AM_LAUNCH_CHECK();
        fin ^= 1;
    }
    hipLaunchKernelGGL(fd_finish_kernel, dim3(1), blk, 0, st, mu_x, cov_x, mu_y, cov_y, D, (const NsState*)(b.state + fin), out_dev);
    AM_LAUNCH_CHECK();
    return AM_OK;
}

}  // namespace am

using namespace am;

extern "C" size_t am_frechet_workspace_bytes(int D) {
    if (D < 1) return 0;
    Carver c(nullptr, 0);
    NsBuffers b;
    carve_ns(c, D, b);
    c.take<double>(8);
    return c.off;
}

extern "C" int am_frechet_enqueue_f64(const double* mu_x, const double* cov_x, const double* mu_y, const double* cov_y, int D,
                                      int first_iter, int n_iter, int max_iter, double tol, double* out_dev, void* ws,
                                      size_t ws_bytes, am_stream_t stream) {
    AM_REQUIRE(mu_x && cov_x && mu_y && cov_y && out_dev, AM_ERR_BAD_ARG, "null pointer");
    AM_REQUIRE(D >= 1, AM_ERR_BAD_SHAPE, "D=%d", D);
    if (max_iter <= 0) max_iter = 64;
    if (!(tol > 0)) tol = 1e-13;
    AM_REQUIRE(first_iter >= 0 && n_iter >= 1 && first_iter < max_iter, AM_ERR_BAD_ARG, "iterations [%d, %d) of %d", first_iter,
               first_iter + n_iter, max_iter);
    n_iter = std::min(n_iter, max_iter - first_iter);
    Carver c(ws, ws_bytes);
    NsBuffers b;
    AM_REQUIRE(carve_ns(c, D, b), AM_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", c.off, ws_bytes);
    return enqueue_ns(mu_x, cov_x, mu_y, cov_y, D, first_iter, n_iter, first_iter + n_iter >= max_iter, tol, b, out_dev,
                      static_cast<hipStream_t>(stream));
}

extern "C" int am_frechet_first_block(void) { return 16; }

extern "C" int am_frechet_f64(const double* mu_x, const double* cov_x, const double* mu_y, const double* cov_y, int D,
                              int max_iter, double tol, double* out_host, void* ws, size_t ws_bytes,
                              am_stream_t stream) {
    AM_REQUIRE(mu_x && cov_x && mu_y && cov_y && out_host, AM_ERR_BAD_ARG, "null pointer");
    AM_REQUIRE(D >= 1, AM_ERR_BAD_SHAPE, "D=%d", D);
    if (max_iter <= 0) max_iter = 64;
    if (!(tol > 0)) tol = 1e-13;
    hipStream_t st = static_cast<hipStream_t>(stream);
    Carver c(ws, ws_bytes);
    NsBuffers b;
    carve_ns(c, D, b);
    double* out_dev = c.take<double>(8);
    AM_REQUIRE(c.ok(), AM_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", c.off, ws_bytes);
    double out5[5] = {0, 0, 0, 0, 0};
    for (int first = 0; first < max_iter;) {
        const int n_iter = std::min(am_frechet_first_block(), max_iter - first);
        const int rc = enqueue_ns(mu_x, cov_x, mu_y, cov_y, D, first, n_iter, first + n_iter >= max_iter, tol, b, out_dev, st);
        if (rc != AM_OK) return rc;
        AM_HIP_TRY(hipMemcpyAsync(out5, out_dev, sizeof(out5), hipMemcpyDeviceToHost, st));
        AM_HIP_TRY(hipStreamSynchronize(st));
        first += n_iter;
        if ((int)out5[4] != 0) break;
    }
    for (int i = 0; i < 4; ++i) out_host[i] = out5[i];
    AM_REQUIRE((int)out5[4] != 4, AM_ERR_NO_CONVERGENCE, "non-finite covariance product or trace in Newton-Schulz");
    return AM_OK;
}
