/*
 * audio_metrics_hip.h - C ABI of the MI355X (gfx950) distribution-distance library.
 *
 * This is the drop-in boundary for the hot path of SonyCSLParis/audio-metrics
 * (reference v1.0.4): everything that happens between "N x D embedding matrix"
 * and "metric value".  The reference has no FFI for this path - its boundary
 * is the Python call surface of data.py and metrics/{fad,kd,prdc,apa}.py - so every entry point
 * below cites the reference function (file:line, relative to the reference
 * checkout) whose arithmetic it replaces.  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *  - plain C types only; no torch / HIP types in signatures (am_stream_t is a
 *    hipStream_t passed as void*; NULL = the default stream);
 *  - every data pointer is a DEVICE pointer owned by the caller unless the
 *    parameter is documented as "host"; matrices are row-major with an explicit
 *    leading dimension `ld` (elements), base pointers 16-byte aligned and
 *    ld % 4 == 0 for float matrices;
 *  - the library allocates nothing: scratch comes from a caller-provided
 *    workspace whose size is returned by the matching workspace-size query;
 *  - all work is enqueued on `stream`; functions return without synchronising
 *    unless documented otherwise; return value 0 = AM_OK, negative = am_status;
 *  - no exceptions cross the ABI; am_last_error() gives a thread-local message.
 */
#ifndef AUDIO_METRICS_HIP_H
#define AUDIO_METRICS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* am_stream_t;

typedef enum am_status {
    AM_OK = 0,
    AM_ERR_BAD_ARG = -1,          /* null pointer, misaligned base, ld % 4 != 0 ...           */
    AM_ERR_BAD_SHAPE = -2,        /* empty input, D < 1, k + 1 > N (torch.kthvalue would raise) */
    AM_ERR_UNSUPPORTED_K = -3,    /* nearest_k > AM_MAX_K in an entry point of the partitioned form */
    AM_ERR_WORKSPACE = -4,        /* workspace missing or too small                            */
    AM_ERR_NO_CONVERGENCE = -5,   /* Newton-Schulz produced a non-finite trace                 */
    AM_ERR_HIP = -6               /* a HIP runtime call failed (see am_last_error)             */
} am_status;

#define AM_MAX_K 31               /* largest nearest_k of the TILE kernels (k+1 <= 32 list slots); am_knn_radii_f32 takes any
                                     k < M and runs larger ones row by row (am_knn_path == 4)        */

const char* am_version(void);
const char* am_status_string(int status);
const char* am_last_error(void);

/* ---------------------------------------------------------------------------
 * A1/A2  per-set sufficient statistics        reference: data.py:37-47, 49-58
 *   mean[D]  = column means of X (f64 accumulation of the f32 inputs)
 *   cov[D*D] = unbiased covariance  sum (x-mean)(x-mean)^T / (N-1)   (f64 out;
 *              products on the f32 matrix cores in short chains, f64 across)
 *   N == 1 -> cov = 0 (data.py:40-42).  N == 0 -> AM_ERR_BAD_SHAPE.
 * am_colsum_f32 / am_scatter_f32 are the two halves (column sums; centred
 * scatter matrix, NOT divided) that a multi-GPU caller all-reduces in between.
 * ------------------------------------------------------------------------- */
size_t am_stats_workspace_bytes(int64_t N, int D);
int am_stats_f32(const float* X, int64_t N, int D, int64_t ld,
                 double* mean, double* cov,
                 void* ws, size_t ws_bytes, am_stream_t stream);
int am_colsum_f32(const float* X, int64_t N, int D, int64_t ld,
                  double* colsum, void* ws, size_t ws_bytes, am_stream_t stream);
int am_scatter_f32(const float* X, int64_t N, int D, int64_t ld, const double* mean,
                   double* scatter, void* ws, size_t ws_bytes, am_stream_t stream);
/* float64 rows: the reference computes an add()'s statistics in the dtype of the embeddings it is given (data.py:39-44; its
 * own test embedder and the output of its PCA projection are float64).  Same outputs, every step in f64 (column sums per row
 * block, centred scatter on the f64 matrix cores, fixed-order reductions).  ld in elements, no alignment requirement. */
size_t am_stats_f64_workspace_bytes(int64_t N, int D);
int am_stats_f64(const double* X, int64_t N, int D, int64_t ld,
                 double* mean, double* cov,
                 void* ws, size_t ws_bytes, am_stream_t stream);
/* the two halves of am_stats_f64 (column sums; centred scatter, NOT divided) for a multi-GPU caller, as am_colsum_f32 /
 * am_scatter_f32; same workspace query.  D <= 8192. */
int am_colsum_f64(const double* X, int64_t N, int D, int64_t ld,
                  double* colsum, void* ws, size_t ws_bytes, am_stream_t stream);
int am_scatter_f64(const double* X, int64_t N, int D, int64_t ld, const double* mean,
                   double* scatter, void* ws, size_t ws_bytes, am_stream_t stream);

/* ---------------------------------------------------------------------------
 * A3  Chan / pairwise merge of two (n, mean, cov) triples in f64
 *                                             reference: data.py:77-94
 *   out may alias the first operand (in-place update, as _update_stats does).
 * ------------------------------------------------------------------------- */
int am_stats_merge_f64(int64_t n1, const double* mean1, const double* cov1,
                       int64_t n2, const double* mean2, const double* cov2,
                       int D, double* out_mean, double* out_cov, am_stream_t stream);

/* ---------------------------------------------------------------------------
 * A1 + A3 + A4 fused for the streaming pipeline   reference: data.py:37-47, 68-72, 77-94
 *   One launch per AudioMetricsData.add(batch) for batches of up to
 *   am_stats_push_max_rows() rows (the embedding pipeline adds <= 32 rows at a
 *   time, embed.py:231-236): batch mean / covariance, Chan merge into the running
 *   (n_old, mean, cov) and the row append.
 *     mean_in  [D]   running mean (ignored when n_old == 0)
 *     mean_out [D]   new mean; must not alias mean_in (other workgroups still read it)
 *     cov      [D,D] running covariance, updated in place (written when n_old == 0)
 *     rows_out       where row n_old of the stored matrix lives (row stride ld_out), or NULL
 *   E needs no alignment (any ld >= D).  Numerics as am_stats_f32: f64 means, f32-centred
 *   values, products summed in f64; merge in f64 with the reference's association.
 * ------------------------------------------------------------------------- */
int am_stats_push_max_rows(void);
int am_stats_push_f32(const float* E, int64_t b, int D, int64_t ld, int64_t n_old,
                      const double* mean_in, double* mean_out, double* cov,
                      float* rows_out, int64_t ld_out, am_stream_t stream);

/* ---------------------------------------------------------------------------
 * A5  Frechet distance                         reference: fad.py:16-31
 *   fd = |mu_x-mu_y|^2 + tr(cov_x) + tr(cov_y) - 2 tr sqrt(cov_x cov_y)
 *   tr sqrt by a coupled Newton-Schulz iteration in f64 on the f64 matrix
 *   cores (the reference uses LAPACK eigvals); stops when |I - ZY|_F <
 *   tol*sqrt(D), when the trace stops growing (rank-deficient inputs) or at
 *   max_iter.  The stopping rule runs on the device; the solve is a stream-
 *   ordered chain of kernels with no host polling.
 *   am_frechet_f64          enqueues blocks of am_frechet_first_block() iterations until the device-side stop code
 *                           is set (one block for well-conditioned inputs) and SYNCHRONISES `stream` once per block.
 *                           out_host (HOST pointer, 4 doubles): { fd, tr_sqrt, iterations, final residual }.
 *   am_frechet_enqueue_f64  the asynchronous building block: enqueues iterations [first_iter, first_iter + n_iter)
 *                           (first_iter == 0 starts the solve; the state lives in `ws`, which the caller keeps alive
 *                           and passes again to continue) and a final kernel that writes out_dev (DEVICE pointer, 5
 *                           doubles): { fd, tr_sqrt, iterations, residual, stop code } - stop code 0 = still running
 *                           (enqueue the next block), 1 converged, 2 trace stalled, 3 zero product, 4 non-finite.
 *                           Never synchronises: a caller can put the solve on a side stream under other work and read
 *                           out_dev when it needs the value.
 * ------------------------------------------------------------------------- */
size_t am_frechet_workspace_bytes(int D);
int am_frechet_f64(const double* mu_x, const double* cov_x,
                   const double* mu_y, const double* cov_y, int D,
                   int max_iter, double tol, double* out_host,
                   void* ws, size_t ws_bytes, am_stream_t stream);
int am_frechet_first_block(void);
int am_frechet_enqueue_f64(const double* mu_x, const double* cov_x,
                           const double* mu_y, const double* cov_y, int D,
                           int first_iter, int n_iter, int max_iter, double tol, double* out_dev,
                           void* ws, size_t ws_bytes, am_stream_t stream);

/* A11 APA scalar combination (host arithmetic)  reference: apa.py:22-32 */
double am_apa_f64(double d_y_x, double d_y_xp, double d_x_xp);

/* ---------------------------------------------------------------------------
 * A6-A8  kernel distance, polynomial kernel     reference: kd.py:38-83,112-124,178-187
 *   For each subset s < S: rows idx1[s*m .. s*m+m) of X (features_1) and
 *   idx2[...] of Y (features_2) -> K = (x.y * gamma + coef0)^degree on the three
 *   m x m Gram blocks, unbiased MMD^2:
 *   out_mmd[s] = (sum offdiag Kxx + sum offdiag Kyy)/(m(m-1)) - 2 sum Kxy / m^2.
 *   Dot products in f32 on the matrix cores, kernel values and sums in f64.
 *   The index table is drawn by the caller (numpy PCG64, kd.py:176,185-186).
 * ------------------------------------------------------------------------- */
/*   Two forms, selected by the SHAPES alone (never by the buffer handed in):
 *     m >= 512, 128 <= D <= 8192, degree == 3: split-f16 form - every gathered row as two f16 planes, three f16 MFMA
 *       stages per 64-element slab on the 256 x 256 engine, error of a dot product <= ~3 * 2^-22 |x||y|;
 *     everything else (and the RBF kernel): f32 MFMA on the 128 x 128 engine.
 *   am_kd_poly_workspace_bytes(S, m, D) is right for every shape.
 *   DEPRECATED: am_kd_workspace_bytes(S, m) predates the split form and covers the f32 form only.  A caller that still sizes
 *   with it gets AM_ERR_WORKSPACE from am_kd_poly_f32 at the shapes of the split form (m >= 512, 128 <= D <= 8192,
 *   degree 3) - am_last_error() names the size am_kd_poly_workspace_bytes would have given - and never a silently different
 *   kernel: the form, and with it the bits, depend on the shapes alone.  Kept for the RBF entry point's callers of round 2;
 *   new code uses am_kd_poly_workspace_bytes / am_kd_rbf_workspace_bytes. */
size_t am_kd_workspace_bytes(int S, int m);
size_t am_kd_poly_workspace_bytes(int S, int m, int D);
int am_kd_poly_f32(const float* X, int64_t N1, int64_t ldx,
                   const float* Y, int64_t N2, int64_t ldy, int D,
                   const int64_t* idx1, const int64_t* idx2, int S, int m,
                   double gamma, double coef0, int degree,
                   double* out_mmd, void* ws, size_t ws_bytes, am_stream_t stream);

/* The subset index table of A8 (host arithmetic, host pointers): idx1[S][m], idx2[S][m] exactly as the reference draws
 * them - `rng = np.random.default_rng(seed)`; per subset `rng.choice(n1, m, replace=False)` then
 * `rng.choice(n2, m, replace=False)` (kd.py:176,185-186) - from the PCG64 state numpy seeds (state and increment as two
 * 64-bit halves each).  A restatement of numpy 2.x's Generator.choice (Floyd's algorithm / tail shuffle on Lemire-
 * bounded 32-bit draws); 200 draws take ~1 ms instead of ~10 ms of Python-level calls. */
int am_kd_draw_indices(uint64_t state_hi, uint64_t state_lo, uint64_t inc_hi, uint64_t inc_lo,
                       int64_t n1, int64_t n2, int S, int m, int64_t* idx1, int64_t* idx2);

/* RBF variant (reference kd.py:86-109, selected with kernel_type="rbf"): K = exp(-|x-y|^2 / (2 sigma^2)),
 * squared distances |x|^2 + |y|^2 - 2 x.y with f64 norms and the f32 matrix-core dot product. */
size_t am_kd_rbf_workspace_bytes(int S, int m);
int am_kd_rbf_f32(const float* X, int64_t N1, int64_t ldx,
                  const float* Y, int64_t N2, int64_t ldy, int D,
                  const int64_t* idx1, const int64_t* idx2, int S, int m, double sigma,
                  double* out_mmd, void* ws, size_t ws_bytes, am_stream_t stream);

/* ---------------------------------------------------------------------------
 * A9  k-NN radii                                reference: prdc.py:4-14, data.py:60-66
 *   out_r[i] = (k+1)-th smallest Euclidean distance from row i of X to the M
 *   rows of Y (Y == X for the reference's self-distance use; a multi-GPU
 *   caller passes its row shard as X and the gathered set as Y).
 *   Distances follow torch.cdist's matmul form  sqrt(max(|x|^2+|y|^2-2x.y, 0))
 *   in f32; no N x M matrix is materialised.  1 <= k, k+1 <= M (k > AM_MAX_K: one row at a time on the vector ALUs,
 *   same values; a correctness path - the reference's evaluate() caps k at 10).
 *   Y == X with >= 6144 rows (D >= 256) / 8192 rows (128 <= D < 256) / 12000 rows (32 <= D < 128) / 16384 rows (D < 32), D <= 4096, runs as a scaled-f16 MFMA FILTER sweep over half of the tile pairs
 *   followed by an f32 evaluation - with the arithmetic of the exact kernel - of the pairs its error bound cannot
 *   rule out (csrc/pairwise_fast.h): the radii are bit-identical to the exact kernels', which remain the path for
 *   the other shapes and the automatic fallback - per row (a row whose buffers overflowed), or for the whole call when
 *   device-side checks find that the f16 values cannot separate the rows' neighbours (tightly clustered data) or that the
 *   operands cannot be scaled into f16 (non-finite values).  A row with a non-finite element is nobody's neighbour (its
 *   distances count as +inf, where torch carries NaN); its own radius is +inf.
 * ------------------------------------------------------------------------- */
size_t am_knn_workspace_bytes(int64_t N, int64_t M, int D, int k);
int am_knn_radii_f32(const float* X, int64_t N, int64_t ldx,
                     const float* Y, int64_t M, int64_t ldy, int D, int k,
                     float* out_r, void* ws, size_t ws_bytes, am_stream_t stream);

/* ---------------------------------------------------------------------------
 * A9, partitioned form for one-process-per-GPU callers that all hold the FULL set X.
 * The self-distance matrix is bitwise symmetric, so only half of the tile pairs are multiplied
 * (csrc/pairwise.hip, knn_sym_kernel); rank `part` of `nparts` owns a contiguous range of the 128-row blocks.
 *   am_knn_sym_eligible     1 if this path applies to the shape (else use am_knn_radii_f32 on row shards)
 *   am_knn_bounds_f32       upper bounds (SQUARED distances) of the final values of rows [row0, row0+nrows)
 *                           from a column sample; ranks split the rows and all-gather the result
 *   am_knn_sym_part_f32     this rank's share: out_lists[N][am_knn_list_width(k)] = its smallest entries per row
 *                           (+inf padded; a NaN in slot 0 flags a row whose candidate buffer overflowed);
 *                           bounds_sq[N] is read (the exact form also tightens it in place)
 *   am_knn_lists_finish_f32 lists[nparts][N][width] (all-gathered) -> out_r[N]; flagged rows are recomputed exactly
 *                           (workspace: am_knn_lists_finish_workspace_bytes)
 * Shapes that take the f16 filter + exact verification form in am_knn_radii_f32 take it here too (csrc/pairwise_fast.h):
 * the bounds come from an f16 sample pass, each rank sweeps its row blocks on the f16 copy and evaluates its surviving
 * pairs exactly; out_lists then holds the rank's smallest EXACT values per row.
 * The result is bit-identical to am_knn_radii_f32(X, X).
 * ------------------------------------------------------------------------- */
int am_knn_sym_eligible(int64_t N, int D, int k);
int am_knn_list_width(int k);
size_t am_knn_part_workspace_bytes(int64_t N, int D, int k);
int am_knn_bounds_f32(const float* X, int64_t N, int64_t ld, int D, int k, int64_t row0, int64_t nrows,
                      float* out_bound_sq, void* ws, size_t ws_bytes, am_stream_t stream);
int am_knn_sym_part_f32(const float* X, int64_t N, int64_t ld, int D, int k, int part, int nparts,
                        float* bounds_sq, float* out_lists, void* ws, size_t ws_bytes, am_stream_t stream);
size_t am_knn_lists_finish_workspace_bytes(int64_t N, int D, int k);
int am_knn_lists_finish_f32(const float* lists, int nparts, const float* X, int64_t N, int64_t ld, int D, int k,
                            float* out_r, void* ws, size_t ws_bytes, am_stream_t stream);

/* ---------------------------------------------------------------------------
 * Prepared sets.  Every PRDC entry point first derives from each set its squared row norms, their maximum, the largest
 * |element| and - for the f16 filter forms - a scaled f16 copy (0.14 ms per 100k x 512 set and call).  An evaluate() calls
 * three to four entry points per set (more per rank in the partitioned multi-GPU form): am_prepare_set_f32 computes the
 * three pieces ONCE into caller-owned device buffers
 *     norms  float[N]                     squared row norms (the summation order of the k-NN kernels)
 *     stats4 uint32[4]                    { largest squared norm, 0, largest |element|, 0 } as f32 bit patterns
 *     half   uint16[N * am_prepared_half_ld(D)]   f16 copy scaled by an exact power of two, rows zero-padded
 * and the *_prepared_* variants below take them instead of recomputing (same results, bit for bit).  A ROW SHARD of a
 * prepared set is the same struct with `norms` and `half` advanced to the shard's first row (the statistics of the
 * whole set remain valid bounds for the shard).  The struct is a HOST struct of DEVICE pointers.
 * ------------------------------------------------------------------------- */
typedef struct am_prepared_set {
    const float* norms;
    const uint32_t* stats;
    const uint16_t* half;
} am_prepared_set;
int64_t am_prepared_half_ld(int D);
int am_prepare_set_f32(const float* X, int64_t N, int64_t ld, int D, float* norms, uint32_t* stats4, uint16_t* half,
                       am_stream_t stream);
int am_knn_radii_prepared_f32(const float* X, int64_t N, int64_t ldx, int D, const am_prepared_set* prepared, int k,
                              float* out_r, void* ws, size_t ws_bytes, am_stream_t stream);          /* Y == X */
int am_knn_bounds_prepared_f32(const float* X, int64_t N, int64_t ld, int D, const am_prepared_set* prepared, int k,
                               int64_t row0, int64_t nrows, float* out_bound_sq, void* ws, size_t ws_bytes, am_stream_t stream);
int am_knn_sym_part_prepared_f32(const float* X, int64_t N, int64_t ld, int D, const am_prepared_set* prepared, int k,
                                 int part, int nparts, float* bounds_sq, float* out_lists, void* ws, size_t ws_bytes,
                                 am_stream_t stream);
int am_prdc_counts_prepared_f32(const float* R, int64_t Nr, int64_t ldr, const am_prepared_set* prepared_r,
                                const float* C, int64_t Nc, int64_t ldc, const am_prepared_set* prepared_c, int D,
                                const float* r_ref, const float* r_cand,
                                int32_t* out_col_count, uint8_t* out_row_any, uint8_t* out_row_cover, float* out_row_min,
                                void* ws, size_t ws_bytes, am_stream_t stream);

/* ---------------------------------------------------------------------------
 * A10  hypersphere membership counts            reference: prdc.py:34-48
 *   With d(i,j) the distance between reference row i and candidate row j:
 *   out_col_count[j] = #{ i : d(i,j) < r_ref[i] }             (precision, density)
 *   out_row_any[i]   = any_j d(i,j) < r_cand[j]               (recall)
 *   out_row_cover[i] = any_j d(i,j) < r_ref[i]                (coverage: min_j d(i,j) < r_ref[i], prdc.py:45-47)
 *   out_row_min[i]   = min_j d(i,j)                           OPTIONAL (may be NULL): not needed by any metric
 *   Outputs are OVERWRITTEN.  am_prdc_reduce turns the first three into the four integer totals
 *   { #cols with count>0, #rows with any, sum of counts, #rows covered } (device int64[4]); the caller
 *   divides in f64.
 *   Large problems (Nr * Nc >= 2^24 pairs for D >= 256, 2^26 for 128 <= D < 256, 1e8 for 32 <= D < 128, 2^28 below; D <= 4096) run as a
 *   scaled-f16 MFMA FILTER pass that queues every pair whose membership its error bound
 *   cannot decide, followed by an f32 evaluation of exactly those pairs with the arithmetic of the exact
 *   kernel (csrc/pairwise_fast.h): the outputs are bit-identical to the exact kernel's, which remains the
 *   path for small problems and the automatic fallback.  Asking for out_row_min adds the candidates of the
 *   row minimum to the queue (slower).
 * ------------------------------------------------------------------------- */
size_t am_prdc_workspace_bytes(int64_t Nr, int64_t Nc, int D);
int am_prdc_counts_f32(const float* R, int64_t Nr, int64_t ldr,
                       const float* C, int64_t Nc, int64_t ldc, int D,
                       const float* r_ref, const float* r_cand,
                       int32_t* out_col_count, uint8_t* out_row_any, uint8_t* out_row_cover, float* out_row_min,
                       void* ws, size_t ws_bytes, am_stream_t stream);
int am_prdc_reduce(const int32_t* col_count, int64_t Nc,
                   const uint8_t* row_any, const uint8_t* row_cover, int64_t Nr,
                   int64_t* out4, am_stream_t stream);

/* ---------------------------------------------------------------------------
 * float64 rows: A6-A10 in the dtype of the embeddings.
 *   The reference computes every stage in the dtype of the rows it is given - torch.cdist / kthvalue (prdc.py:12-13,34),
 *   the comparisons against the radii (prdc.py:36-47), np.matmul and the kernel sums (kd.py:112-116, 56-81) - and float64
 *   rows are what its PCA projection hands on (projection.py:20-21: scikit-learn's float64 product; audio_metrics.py:163-182,
 *   the path of every reference test and of examples/2_musdb.py) and what its test embedder yields.  These entry points are
 *   the f64 forms of am_knn_radii_f32 / am_prdc_counts_f32 / am_kd_poly_f32 / am_kd_rbf_f32: same arguments with double
 *   rows, double radii and a double row minimum, every product and sum in f64 on the f64 matrix cores
 *   (v_mfma_f64_16x16x4_f64, csrc/pairwise_f64.hip):  d2 = max(fma(-2, <x, y>, |x|^2 + |y|^2), 0), radius = sqrt_rn of the
 *   (k+1)-th smallest d2, membership  sqrt_rn(d2) < r  decided exactly through d2 < T(r).  Rows need no alignment
 *   (ld >= D, in elements).  Any 1 <= k < M (k > 31: distance blocks + a radix select, a correctness path).
 *   Large problems - am_knn_radii_f64 of a set against itself (Y == X, k <= 10) and am_prdc_counts_f64 without a row minimum,
 *   from the row counts at which the float32 entry points switch to their f16 filter form - take that filter too: the sweep
 *   runs on a float32-rounded copy inside the workspace with an error band widened by the rounding, and every pair the band
 *   cannot decide is evaluated in f64 as a sum of squared differences against the f64 thresholds (csrc/pairwise_fast.h:
 *   knn_fast_select64_kernel, cross_verify_regions64_kernel).  Same results as the general kernels to rounding (counts equal;
 *   the sum of squared differences is the more accurate form on near-duplicate rows), about a tenth of their time at 100 000
 *   rows; data the filter cannot serve runs the general kernels, launched behind a device flag.  The workspace queries cover
 *   the route; with a smaller workspace the general kernels run.
 * ------------------------------------------------------------------------- */
size_t am_knn_f64_workspace_bytes(int64_t N, int64_t M, int D, int k);
int am_knn_radii_f64(const double* X, int64_t N, int64_t ldx,
                     const double* Y, int64_t M, int64_t ldy, int D, int k,
                     double* out_r, void* ws, size_t ws_bytes, am_stream_t stream);
size_t am_prdc_f64_workspace_bytes(int64_t Nr, int64_t Nc, int D);
int am_prdc_counts_f64(const double* R, int64_t Nr, int64_t ldr,
                       const double* C, int64_t Nc, int64_t ldc, int D,
                       const double* r_ref, const double* r_cand,
                       int32_t* out_col_count, uint8_t* out_row_any, uint8_t* out_row_cover, double* out_row_min,
                       void* ws, size_t ws_bytes, am_stream_t stream);
size_t am_kd_f64_workspace_bytes(int S, int m);
int am_kd_poly_f64(const double* X, int64_t N1, int64_t ldx,
                   const double* Y, int64_t N2, int64_t ldy, int D,
                   const int64_t* idx1, const int64_t* idx2, int S, int m,
                   double gamma, double coef0, int degree,
                   double* out_mmd, void* ws, size_t ws_bytes, am_stream_t stream);
int am_kd_rbf_f64(const double* X, int64_t N1, int64_t ldx,
                  const double* Y, int64_t N2, int64_t ldy, int D,
                  const int64_t* idx1, const int64_t* idx2, int S, int m, double sigma,
                  double* out_mmd, void* ws, size_t ws_bytes, am_stream_t stream);

/* ---------------------------------------------------------------------------
 * N1  PCA projection support                   reference: projection.py:6-46 (scikit-learn IncrementalPCA),
 *                                                           audio_metrics.py:163-209
 *   am_eigh_sym_f64  eigen-decomposition of a symmetric POSITIVE SEMI-DEFINITE D x D matrix (the Gram matrix of the
 *                    stacked, centred batch whose SVD scikit-learn takes): evals[D] in DESCENDING order, evecs[D][D]
 *                    with row i = eigenvector i (unit norm; sign not normalised - the caller applies scikit-learn's
 *                    svd_flip).  One-sided block Jacobi in f64; SYNCHRONISES `stream` once per BLOCK of enqueued
 *                    sweeps (12, then 6 at a time: normally once per solve - the kernels of a sweep return at once when
 *                    the sweep before it applied no rotation; fitting happens once per reference set, not per evaluate).  AM_ERR_NO_CONVERGENCE after max_sweeps (<= 0: 40).
 *   am_project_f64   out[N][p] (f64) = (X[n][:] - mean[:]) . components[j][:]  - IncrementalPCA.transform - on the f64
 *                    matrix cores; X is the N x D f32 embedding matrix (am_project_rows_f64: f64), mean f64[D], components f64[p][D].
 * ------------------------------------------------------------------------- */
size_t am_eigh_workspace_bytes(int D);
int am_eigh_sym_f64(const double* A, int D, double* evals, double* evecs, int max_sweeps,
                    void* ws, size_t ws_bytes, am_stream_t stream);
int am_project_f64(const float* X, int64_t N, int64_t ld, int D, const double* mean, const double* components, int p,
                   double* out, am_stream_t stream);
/* the same for float64 rows (a float64 embedder in front of the projection: scikit-learn then multiplies in f64 too) */
int am_project_rows_f64(const double* X, int64_t N, int64_t ld, int D, const double* mean, const double* components, int p,
                        double* out, am_stream_t stream);

/* ---------------------------------------------------------------------------
 * A13  one call = one evaluate()                reference: audio_metrics.py:254-274
 *   The FAD + KD + PRDC dispatch of AudioMetrics.evaluate for two embedding sets on ONE device as a single stream-ordered
 *   chain of the entry points above (statistics, Frechet solve on `side_stream` under the PRDC kernels, prepared sets,
 *   radii of both sets, membership counts and totals, kernel distance on the caller's index tables), with every workspace
 *   carved from one caller buffer and every result in ONE device buffer `out`:
 *     out[0..4]   fd, tr sqrt, iterations, residual, Frechet stop code (0: the product needs more than the 32 iterations
 *                 enqueued here - finish with am_frechet_f64 on the statistics; 4: non-finite; -1: FAD not requested)
 *     out[5..8]   #candidate columns with count > 0, #reference rows with a witness, sum of counts, #reference rows covered
 *                 (precision = out[5] / n_cand, recall = out[6] / n_ref, density = out[7] / (k n_cand), coverage = out[8] / n_ref)
 *     out[16 + s] unbiased MMD^2 of subset s (kernel_distance_mean / _std are numpy's mean / std of these, kd.py:189-192)
 *   `what` selects the metrics.  am_evaluate_side (optional, per set) hands in results the caller already holds - the
 *   reference keeps the reference set's statistics and radii between evaluates (data.py:60-66) - and/or names where the
 *   statistics / radii computed here are to be written so that the caller can keep them; NULL members are ignored.
 *   idx_cand / idx_ref: int64 [kd_subsets, kd_m] DEVICE tables (am_kd_draw_indices draws them on the host).
 * ------------------------------------------------------------------------- */
#define AM_EVAL_FAD 1u
#define AM_EVAL_KD 2u
#define AM_EVAL_PRDC 4u
#define AM_EVAL_HEAD 16
typedef struct am_evaluate_side {
    const double* mean;      /* in:  statistics already computed (both or neither) */
    const double* cov;
    const float* radii;      /* in:  radii for nearest_k already computed */
    double* mean_out;        /* out: where statistics computed by this call go (else workspace) */
    double* cov_out;
    float* radii_out;        /* out: where radii computed by this call go (else workspace) */
} am_evaluate_side;
size_t am_evaluate_workspace_bytes(int64_t n_ref, int64_t n_cand, int D, int nearest_k, int kd_subsets, int kd_m, unsigned what);
int am_evaluate_f32(const float* ref, int64_t n_ref, int64_t ld_ref, const float* cand, int64_t n_cand, int64_t ld_cand, int D,
                    unsigned what, int nearest_k, const int64_t* idx_cand, const int64_t* idx_ref, int kd_subsets, int kd_m,
                    double kd_gamma, double kd_coef0, int kd_degree, const am_evaluate_side* given_ref,
                    const am_evaluate_side* given_cand, double* out, void* ws, size_t ws_bytes, am_stream_t stream,
                    am_stream_t side_stream);

/* ---------------------------------------------------------------------------
 * One call = one RANK's share of a row-sharded evaluate: the exchange schedule of
 * audio-metrics_amd/distributed.py: evaluate_sharded behind the C ABI, for one-process-per-GPU hosts that are not Python.
 * The reference has no multi-GPU metrics path (its only multi-GPU code spreads the embedder, util/gpu_parallel.py:79-118).
 *
 *   am_collectives   the two collectives of the schedule as hooks: the library links no collective library.  Both enqueue
 *                    on the stream they are handed (ncclAllReduce / ncclAllGather semantics: stream-ordered, every rank
 *                    calls them in the same order) and return 0 on success.  csrc/rccl/am_rccl.cpp builds them over an
 *                    ncclComm_t (libaudio_metrics_rccl.so: am_rccl_collectives), the tests over torch.distributed.
 *     all_reduce_sum(ctx, buf, count, dtype, stream)        in place; dtype AM_COLL_F64 or AM_COLL_I32
 *     all_gather_v(ctx, send, recv, bytes_per_rank, stream) recv = the ranks' contributions in rank order, rank r's being
 *                    bytes_per_rank[r] bytes (a HOST array of `world` entries, valid during the call only; zero allowed);
 *                    send == recv + (bytes of the lower ranks): the call is always IN PLACE
 *   Every rank passes its row shards (row-major f32, ld % 4 == 0, 16-byte aligned; a shard may be empty - NULL - as long as
 *   each set has rows somewhere), the shard sizes of ALL ranks (host arrays [world], the same on every rank), the metrics
 *   and, for the kernel distance, the two int64 [kd_subsets, kd_m] DEVICE index tables (the same on every rank:
 *   am_kd_draw_indices).  Compute runs on `stream`, the collectives in issue order on `comm_stream` (fenced with events;
 *   pass `stream` itself, or NULL, for the serial form), the Frechet solve on `side_stream`.
 *   Results: the record of am_evaluate_f32 in `out` (device, 16 + kd_subsets doubles), identical on every rank.
 *   Wide, large sets with rows on every rank take the partitioned symmetric k-NN sweep (am_knn_sym_eligible), everything
 *   else the general kernel on the row shard and an all-gather of the radii.
 * ------------------------------------------------------------------------- */
#define AM_COLL_F64 0
#define AM_COLL_I32 1
typedef struct am_collectives {
    void* ctx;
    int rank, world;
    int (*all_reduce_sum)(void* ctx, void* buf, int64_t count, int dtype, am_stream_t stream);
    int (*all_gather_v)(void* ctx, const void* send, void* recv, const int64_t* bytes_per_rank, am_stream_t stream);
} am_collectives;
size_t am_evaluate_sharded_workspace_bytes(const int64_t* ref_counts, const int64_t* cand_counts, int rank, int world, int D,
                                           int nearest_k, int kd_subsets, int kd_m, unsigned what);
int am_evaluate_sharded_f32(const float* ref_local, int64_t ld_ref, const float* cand_local, int64_t ld_cand, int D,
                            const int64_t* ref_counts, const int64_t* cand_counts, const am_collectives* coll, unsigned what,
                            int nearest_k, const int64_t* idx_cand, const int64_t* idx_ref, int kd_subsets, int kd_m,
                            double kd_gamma, double kd_coef0, int kd_degree, double* out, void* ws, size_t ws_bytes,
                            am_stream_t stream, am_stream_t side_stream, am_stream_t comm_stream);

/* ---- optional kernel clock (benchmark support; bench.py's roofline) --------------------------------
 * When enabled, the library brackets every launch of the two tile kernels with a hipEvent pair recorded on
 * the caller's stream, so a benchmark can report the duration of exactly that kernel (the figure
 * `rocprofv3 --kernel-trace --stats` prints for it) rather than of the whole entry point.
 *   AM_KERNEL_KNN          the k-NN tile kernel: knn_wide_kernel / knn_fast_kernel (f16 filter sweep) where the filter path runs,
 *                          else knn_sym_kernel, else knn_partial_kernel's main pass (sampled pre-passes not counted)
 *   AM_KERNEL_PRDC_CROSS   the membership tile kernel: cross_wide_kernel / cross_fast_kernel (f16 filter) or prdc_cross_kernel
 *   AM_KERNEL_KNN_VERIFY   knn_fast_verify_kernel (exact f32 values of the queued pairs)
 *   AM_KERNEL_PRDC_VERIFY  cross_verify_kernel
 * am_kernel_clock_read waits for the recorded launches, returns their count and summed duration in
 * milliseconds, and resets that kernel's record.  Disabled by default; no cost when disabled.
 * am_knn_path / am_prdc_path tell which form the library picks for a shape (0 = exact general kernel,
 * 1 = exact symmetric kernel (k-NN only), 2 = f16 filter + exact verification on the 128 x 128 engine,
 * 3 = the same on the 256 x 256 f16 engine, 4 = row-at-a-time kernel for k > AM_MAX_K).  The choice is a pure function of
 * the shapes; inside forms 2 / 3 the DATA can still hand a call to the exact kernels on the device (rows the f16 values
 * cannot separate, queue overflow, operands that cannot be scaled into f16) - am_filter_stats_enable counts those. */
enum am_clocked_kernel { AM_KERNEL_KNN = 0, AM_KERNEL_PRDC_CROSS = 1, AM_KERNEL_KNN_VERIFY = 2, AM_KERNEL_PRDC_VERIFY = 3 };
int am_kernel_clock_enable(int on);
int am_kernel_clock_read(int kernel, int64_t* launches, double* total_ms);
int am_knn_path(int64_t N, int64_t M, int D, int k, int self);
int am_prdc_path(int64_t Nr, int64_t Nc, int D);
/* form 3 has three tile engines, chosen by the row length alone: 1 = operand-stationary (csrc/pstat_engine.h: the workgroup's
 * 256-row block held in registers, kernels knn_pstat_kernel / cross_pstat_kernel; rows of up to 512 elements),
 * 2 = the same as two independent 256-thread workgroups per CU whose waves own two row tiles (pstat64_pipeline: kernels
 * knn_pstat64_kernel / cross_pstat64_kernel; rows of up to 128 elements),
 * 0 = both operands streamed through LDS (csrc/wide_engine.h: knn_wide_kernel / cross_wide_kernel) */
int am_filter_engine(int D);

/* ---- optional filter statistics (benchmark support) -------------------------------------------------
 * What the f16 filter passes leave for the exact kernels is data dependent.  A caller hands the library a
 * zero-initialised DEVICE buffer of AM_FILTER_STATS_SLOTS int64 (on the current device; NULL switches the feature
 * off); every filter-form call of am_knn_radii_f32 / am_knn_sym_part_f32 / am_prdc_counts_f32 then ends with a one-
 * workgroup kernel that ADDS to it, stream-ordered, no synchronisation:
 *   [0] k-NN calls           [1] entries queued by the sweep (both directions)   [2] of those via the spill queue
 *   [3] pairs evaluated exactly (after pruning)           [4] rows recomputed by the exact fix-up kernel
 *   [5] membership calls     [6] pairs queued            [7] of those via the overflow queue
 *   [8] membership calls handed to the exact kernel (queues overflowed / operands not scalable)
 *   [9] the error bound of the f16 filter, MEASURED: max over every pair the k-NN verification evaluated of
 *       |f16 matrix-core value - exact f32 value| / (fast_c(D) (|x|^2 + G)), as the bit pattern of a float in the low half
 *       (the filter is sound while this stays <= 1; csrc/pairwise_fast.h derives the bound)     [10] pairs it was measured on
 * The caller reads and clears the buffer itself. */
#define AM_FILTER_STATS_SLOTS 16
int am_filter_stats_enable(int64_t* device_slots);

#ifdef __cplusplus
}
#endif
#endif /* AUDIO_METRICS_HIP_H */
