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
// The iteration state lives on the device; kernels launched after convergence
// exit immediately, and the host only polls the state every few iterations.
#include "am_common.h"
#include <math.h>
#include <algorithm>

namespace am {

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int GB = 64;            // block tile (GB x GB), 4 waves of 32 x 32
constexpr int GK = 16;            // inner slab
constexpr int LDA = GK + 2;       // A slab [64][18] doubles: conflict-free ds_read_b64 fragments
constexpr int LDB = GB + 16;      // B slab [16][80] doubles

struct NsState {
    double prev_trace;            // last accepted tr(Y)
    double resid;                 // |I - ZY|_F at the last check
    double norm;                  // |A|_F
    int iters;
    int done;                     // 1 converged, 2 trace stalled (noise), 3 zero matrix, 4 non-finite input
};

enum { MODE_PLAIN = 0, MODE_NS_T = 1 };

struct GemmJob {
    const double* A;
    const double* B;
    double* C;
};

// C = A*B (n x n, row-major, ld = n).  MODE_PLAIN also emits sum C^2 per block;
// MODE_NS_T stores T = 1.5 I - 0.5 A*B and emits sum (I - A*B)^2 per block.
template <int MODE>
__global__ void __launch_bounds__(256) gemm_f64_kernel(GemmJob j0, GemmJob j1, int n, const NsState* __restrict__ state,
                                                       double* __restrict__ block_sums) {
    if (state != nullptr && state->done) return;
    __shared__ __attribute__((aligned(16))) double sA[GB * LDA];
    __shared__ __attribute__((aligned(16))) double sB[GK * LDB];
    __shared__ double red[4];
    const GemmJob job = blockIdx.z == 0 ? j0 : j1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int row0 = blockIdx.y * GB, col0 = blockIdx.x * GB;
    const int l15 = lane & 15, l4 = lane >> 4;

    f64x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f64x4{0, 0, 0, 0};

    const int ar = tid >> 2, ac = (tid & 3) * 4;       // A slab: row ar, 4 doubles from column ac
    const int br = tid >> 4, bc = (tid & 15) * 4;      // B slab: row br, 4 doubles from column bc
    // The slab of step k+1 is fetched into registers while step k multiplies: the kernel is a chain of 32 short
    // steps on 64-128 workgroups, so without the prefetch every step pays a full L2 round trip (45 us per product).
    double ra[4], rb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int gr = row0 + ar, gc = k0 + ac + e;
            ra[e] = (gr < n && gc < n) ? job.A[(int64_t)gr * n + gc] : 0.0;
            const int hr = k0 + br, hc = col0 + bc + e;
            rb[e] = (hr < n && hc < n) ? job.B[(int64_t)hr * n + hc] : 0.0;
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < n; k0 += GK) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sA[ar * LDA + ac + e] = ra[e];
            sB[br * LDB + bc + e] = rb[e];
        }
        __syncthreads();
        if (k0 + GK < n) fetch(k0 + GK);
#pragma unroll
        for (int kk = 0; kk < GK / 4; ++kk) {
            double a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = sA[(wm * 32 + t * 16 + l15) * LDA + kk * 4 + l4];
                b[t] = sB[(kk * 4 + l4) * LDB + wn * 32 + t * 16 + l15];
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
        }
        __syncthreads();
    }
    // f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
    double part = 0.0;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gr = row0 + wm * 32 + mt * 16 + l4 + 4 * r;
                const int gc = col0 + wn * 32 + nt * 16 + l15;
                if (gr < n && gc < n) {
                    const double v = acc[mt][nt][r];
                    if (MODE == MODE_PLAIN) {
                        job.C[(int64_t)gr * n + gc] = v;
                        part += v * v;
                    } else {
                        const double eye = (gr == gc) ? 1.0 : 0.0;
                        const double d = eye - v;
                        job.C[(int64_t)gr * n + gc] = 1.5 * eye - 0.5 * v;
                        part += d * d;
                    }
                }
            }
    if (block_sums != nullptr) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off);
        if (lane == 0) red[wave] = part;
        __syncthreads();
        if (tid == 0) block_sums[blockIdx.y * gridDim.x + blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
    }
}

// single-block helpers -------------------------------------------------------
__device__ __forceinline__ double block_sum(double v, double* red) {
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
    __syncthreads();
    return s;
}

// norm = sqrt(sum block_sums); Y = A/norm; Z = I; state init.  Every workgroup recomputes the (tiny) norm sum in
// the same fixed order, so no inter-workgroup hand-off is needed; block 0 writes the state.
__global__ void __launch_bounds__(256) ns_init_kernel(const double* __restrict__ A, const double* __restrict__ block_sums,
                                                      int nblocks, int n, double* __restrict__ Y, double* __restrict__ Z,
                                                      NsState* __restrict__ state) {
    __shared__ double red[4];
    double v = 0;
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) v += block_sums[i];
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

// the stopping rule (see file header)
__global__ void __launch_bounds__(256) ns_check_kernel(const double* __restrict__ Y, const double* __restrict__ block_sums,
                                                       int nblocks, int n, double tol, NsState* __restrict__ state) {
    if (state->done) return;
    __shared__ double red[4];
    double t = 0, r = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) t += Y[(int64_t)i * n + i];
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) r += block_sums[i];
    t = block_sum(t, red);
    r = sqrt(block_sum(r, red));
    if (threadIdx.x == 0) {
        const double prev = state->prev_trace;
        const bool finite = (t == t) && !isinf(t) && (r == r);
        if (!finite || t < prev * (1.0 - 1e-14) - 1e-300) {
            state->done = isinf(prev) ? 4 : 2;               // keep the previous trace
        } else {
            state->prev_trace = t;
            state->resid = r;
            state->iters += 1;
            if (r < tol * sqrt((double)n)) state->done = 1;
        }
    }
}

// out = { fd, tr_sqrt, iterations, residual }
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

}  // namespace am

using namespace am;

extern "C" size_t am_frechet_workspace_bytes(int D) {
    if (D < 1) return 0;
    const int g = (int)ceil_div(D, GB);
    Carver c(nullptr, 0);
    for (int i = 0; i < 6; ++i) c.take<double>((size_t)D * D);   // A, Y0, Y1, Z0, Z1, T
    c.take<double>((size_t)g * g);
    c.take<double>(8);
    c.take<NsState>(1);
    return c.off;
}

extern "C" int am_frechet_f64(const double* mu_x, const double* cov_x, const double* mu_y, const double* cov_y, int D,
                              int max_iter, double tol, double* out_host, void* ws, size_t ws_bytes,
                              am_stream_t stream) {
    AM_REQUIRE(mu_x && cov_x && mu_y && cov_y && out_host, AM_ERR_BAD_ARG, "null pointer");
    AM_REQUIRE(D >= 1, AM_ERR_BAD_SHAPE, "D=%d", D);
    if (max_iter <= 0) max_iter = 64;
    if (!(tol > 0)) tol = 1e-13;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int g = (int)ceil_div(D, GB);
    Carver c(ws, ws_bytes);
    double* A = c.take<double>((size_t)D * D);
    double* Yb[2] = {c.take<double>((size_t)D * D), c.take<double>((size_t)D * D)};
    double* Zb[2] = {c.take<double>((size_t)D * D), c.take<double>((size_t)D * D)};
    double* T = c.take<double>((size_t)D * D);
    double* sums = c.take<double>((size_t)g * g);
    double* out_dev = c.take<double>(8);
    NsState* state = c.take<NsState>(1);
    AM_REQUIRE(c.ok(), AM_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", c.off, ws_bytes);

    const dim3 grid1(g, g, 1), grid2(g, g, 2), blk(256);
    const GemmJob none{nullptr, nullptr, nullptr};
    hipLaunchKernelGGL(gemm_f64_kernel<MODE_PLAIN>, grid1, blk, 0, st, GemmJob{cov_x, cov_y, A}, none, D,
                       (const NsState*)nullptr, sums);
    AM_LAUNCH_CHECK();
    hipLaunchKernelGGL(ns_init_kernel, dim3((unsigned)std::min<int64_t>(256, ceil_div((int64_t)D * D, 1024))), blk, 0, st, A, sums,
                       g * g, D, Yb[0], Zb[0], state);
    AM_LAUNCH_CHECK();
    int cur = 0;
    // read-backs go through a pinned per-thread buffer: a pageable destination makes the runtime pin and unpin the page
    // around every copy
    static thread_local void* pinned = nullptr;
    if (pinned == nullptr) AM_HIP_TRY(hipHostMalloc(&pinned, 256, hipHostMallocDefault));
    NsState& host_state = *static_cast<NsState*>(pinned);
    double* out5 = reinterpret_cast<double*>(static_cast<char*>(pinned) + 128);
    host_state.done = 0;
    for (int it = 0; it < max_iter; ++it) {
        hipLaunchKernelGGL(gemm_f64_kernel<MODE_NS_T>, grid1, blk, 0, st, GemmJob{Zb[cur], Yb[cur], T}, none, D,
                           (const NsState*)state, sums);
        hipLaunchKernelGGL(ns_check_kernel, dim3(1), blk, 0, st, Yb[cur], sums, g * g, D, tol, state);
        hipLaunchKernelGGL(gemm_f64_kernel<MODE_PLAIN>, grid2, blk, 0, st, GemmJob{Yb[cur], T, Yb[cur ^ 1]},
                           GemmJob{T, Zb[cur], Zb[cur ^ 1]}, D, (const NsState*)state, (double*)nullptr);
        AM_LAUNCH_CHECK();
        cur ^= 1;
        if ((it % 6) == 5) {                               // poll the device-side state now and then
            AM_HIP_TRY(hipMemcpyAsync(&host_state, state, sizeof(NsState), hipMemcpyDeviceToHost, st));
            AM_HIP_TRY(hipStreamSynchronize(st));
            if (host_state.done) break;
        }
    }
    if (!host_state.done) {
        // ran out of iterations: account for the last update (no-op when a later check already fired)
        hipLaunchKernelGGL(gemm_f64_kernel<MODE_NS_T>, grid1, blk, 0, st, GemmJob{Zb[cur], Yb[cur], T}, none, D,
                           (const NsState*)state, sums);
        hipLaunchKernelGGL(ns_check_kernel, dim3(1), blk, 0, st, Yb[cur], sums, g * g, D, tol, state);
        AM_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(fd_finish_kernel, dim3(1), blk, 0, st, mu_x, cov_x, mu_y, cov_y, D, (const NsState*)state, out_dev);
    AM_LAUNCH_CHECK();
    AM_HIP_TRY(hipMemcpyAsync(out5, out_dev, 5 * sizeof(double), hipMemcpyDeviceToHost, st));
    AM_HIP_TRY(hipStreamSynchronize(st));
    for (int i = 0; i < 4; ++i) out_host[i] = out5[i];
    AM_REQUIRE((int)out5[4] != 4, AM_ERR_NO_CONVERGENCE, "non-finite covariance product or trace in Newton-Schulz");
    return AM_OK;
}
