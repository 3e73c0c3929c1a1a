// PCA projection support (reference projection.py:6-46 delegates to scikit-learn's IncrementalPCA):
//   am_eigh_sym_f64   eigen-decomposition of the symmetric D x D Gram matrix of the stacked, centred batch
//   am_project_f64    (x - mean) . components^T for an N x D f32 matrix
//
// Eigensolver: one-sided Jacobi (Hestenes) in f64.  W starts as A (rows = columns, A is symmetric), V as the identity;
// a rotation of rows (p, q) makes W_p . W_q = 0 and is applied to V as well, so W = V A throughout.  At convergence the
// rows of W are mutually orthogonal: rows of V are the eigenvectors and lambda_i = W_i . V_i (a Rayleigh quotient -
// signed, unlike |W_i|).  Rounds follow the round-robin tournament: m/2 disjoint pairs per round, one workgroup per
// pair, m - 1 rounds per sweep, launched back to back; the host reads one "rotations applied" counter per sweep.
// D = 512: ~10 sweeps x 511 launches of 256 small workgroups, ~20 ms - the projection is fitted once per reference set,
// not per evaluate.  All reductions run in a fixed order (deterministic).
#include "am_common.h"
#include <math.h>
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

// W = A, V = I
__global__ void __launch_bounds__(256) jacobi_init_kernel(const double* __restrict__ A, int n, double* __restrict__ W,
                                                          double* __restrict__ V) {
    const int64_t total = (int64_t)n * n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        W[i] = A[i];
        V[i] = (i / n == i % n) ? 1.0 : 0.0;
    }
}

// round `round` of the tournament over m = n rounded up to even players: pair i = (perm(i), perm(m - 1 - i)) with
// perm(0) = m - 1 fixed and the others rotating; a player index >= n is a bye
__global__ void __launch_bounds__(256) jacobi_round_kernel(double* __restrict__ W, double* __restrict__ V, int n, int m, int round,
                                                           double tol, unsigned* __restrict__ rotations) {
    __shared__ double red[4][3];
    const int i = blockIdx.x;
    auto player = [&](int slot) { return slot == 0 ? m - 1 : (slot - 1 + round) % (m - 1); };
    int p = player(i), q = player(m - 1 - i);
    if (p > q) { const int t = p; p = q; q = t; }
    if (q >= n) return;
    double* wp = W + (int64_t)p * n;
    double* wq = W + (int64_t)q * n;
    double alpha = 0, beta = 0, gamma = 0;
    for (int k = threadIdx.x; k < n; k += 256) {
        const double a = wp[k], b = wq[k];
        alpha = fma(a, a, alpha);
        beta = fma(b, b, beta);
        gamma = fma(a, b, gamma);
    }
    block_sum3(alpha, beta, gamma, red);
    if (!(fabs(gamma) > tol * sqrt(alpha * beta))) return;          // already orthogonal (or a zero row)
    const double zeta = (beta - alpha) / (2.0 * gamma);
    const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
    const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
    double* vp = V + (int64_t)p * n;
    double* vq = V + (int64_t)q * n;
    for (int k = threadIdx.x; k < n; k += 256) {
        const double a = wp[k], b = wq[k];
        wp[k] = c * a - s * b;
        wq[k] = s * a + c * b;
        const double x = vp[k], y = vq[k];
        vp[k] = c * x - s * y;
        vq[k] = s * x + c * y;
    }
    if (threadIdx.x == 0) atomicAdd(rotations, 1u);
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
__global__ void __launch_bounds__(256) project_kernel(const float* __restrict__ X, int64_t N, int64_t ld, int D,
                                                      const double* __restrict__ mean, const double* __restrict__ C, int p,
                                                      double* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t row = (int64_t)blockIdx.x * 64 + wave * 16 + l15;
    const int comp = blockIdx.y * 16 + l15;
    const bool row_ok = row < N, comp_ok = comp < p;
    const float* x = X + (row_ok ? row : 0) * ld;
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
    AM_REQUIRE(c.ok(), AM_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", c.off, ws_bytes);
    hipLaunchKernelGGL(jacobi_init_kernel, dim3((unsigned)std::min<int64_t>(1024, ceil_div((int64_t)D * D, 256))), dim3(256), 0, st, A, D,
                       W, V);
    AM_LAUNCH_CHECK();
    AM_HIP_TRY(hipMemsetAsync(rotations, 0, 64 * sizeof(unsigned), st));
    const int m = (D + 1) / 2 * 2;
    const double tol = 1e-14 * sqrt((double)D);       // |W_p . W_q| <= tol |W_p| |W_q|: the rounding level of a D-term f64 dot product
    bool converged = D == 1;
    for (int sweep = 0; sweep < max_sweeps && !converged; ++sweep) {
        for (int round = 0; round < m - 1; ++round)
            hipLaunchKernelGGL(jacobi_round_kernel, dim3(m / 2), dim3(256), 0, st, W, V, D, m, round, tol, rotations + sweep);
        AM_LAUNCH_CHECK();
        unsigned applied = 0;
        AM_HIP_TRY(hipMemcpyAsync(&applied, rotations + sweep, sizeof(unsigned), hipMemcpyDeviceToHost, st));
        AM_HIP_TRY(hipStreamSynchronize(st));
        converged = applied == 0;
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
    hipLaunchKernelGGL(project_kernel, dim3((unsigned)ceil_div(N, 64), (unsigned)ceil_div(p, 16)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), X, N, ld, D, mean, components, p, out);
    AM_LAUNCH_CHECK();
    return AM_OK;
}
