/*
 * pairwise_exact.c - plain-C model of the DEVICE kernels' f32 arithmetic for the
 * PRDC path (audio-metrics_amd/csrc/pairwise.hip), used to assert bit-exact
 * radii, thresholds and membership counts.  TEST INFRASTRUCTURE ONLY: nothing in
 * the product package links or calls this file.
 *
 * It restates the reference's algorithm (src/audio_metrics/metrics/prdc.py:4-14,
 * 34-48: Euclidean cdist in torch's matmul form, (k+1)-th smallest per row,
 * strict "<" membership tests) with the device's summation order spelled out:
 *   - |x|^2   : 64 interleaved fmaf chains (element 4*(64t+l)+c -> chain l) and a
 *               xor butterfly 32,16,...,1 of plain adds (row_sqnorm_kernel);
 *   - <x,y>   : one fmaf chain in the index order 8c+0, 8c+4, 8c+1, 8c+5, 8c+2,
 *               8c+6, 8c+3, 8c+7 (c = 0,1,...)  (v_mfma_f32_32x32x2_f32 feeds,
 *               tile_engine.h);
 *   - d2      : max(fmaf(-2, <x,y>, |x|^2 + |y|^2), 0), a NaN becoming +inf;
 *   - radius  : sqrtf of the (k+1)-th smallest d2 of the row;
 *   - "d < R" : d2 < T(R), T(R) = smallest float t with sqrtf(t) >= R.
 * Build: gcc -O3 -mavx2 -mfma -ffp-contract=off -fopenmp -shared -fPIC (see Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

void am_exact_sqnorm(const float* X, int64_t N, int64_t ld, int D, float* out) {
    for (int64_t i = 0; i < N; ++i) {
        const float* x = X + i * ld;
        float part[64];
        for (int l = 0; l < 64; ++l) {
            float acc = 0.f;
            for (int k = l * 4; k < D; k += 256)
                for (int c = 0; c < 4; ++c) {
                    const float v = (k + c < D) ? x[k + c] : 0.f;
                    acc = fmaf(v, v, acc);
                }
            part[l] = acc;
        }
        for (int off = 32; off >= 1; off >>= 1) {
            float nxt[64];
            for (int l = 0; l < 64; ++l) nxt[l] = part[l] + part[l ^ off];
            memcpy(part, nxt, sizeof(part));
        }
        out[i] = part[0];
    }
}

/* inner-index visiting order of the device dot product, padded to a multiple of 8 */
static int* chain_order(int D, int* n_out) {
    const int dp = (D + 7) / 8 * 8;
    int* ord = (int*)malloc(sizeof(int) * dp);
    int n = 0;
    for (int c = 0; c < dp / 8; ++c)
        for (int s = 0; s < 4; ++s) {
            ord[n++] = 8 * c + s;
            ord[n++] = 8 * c + 4 + s;
        }
    *n_out = n;
    return ord;
}

/* Yt[k][j] (k in chain order, zero rows for padded k) so that the j loop vectorises with independent chains */
static float* transpose_in_chain_order(const float* Y, int64_t M, int64_t ld, int D, const int* ord, int n) {
    float* t = (float*)calloc((size_t)n * M, sizeof(float));
    for (int q = 0; q < n; ++q) {
        const int k = ord[q];
        if (k >= D) continue;
        for (int64_t j = 0; j < M; ++j) t[(size_t)q * M + j] = Y[j * ld + k];
    }
    return t;
}

/* d2 of ROWS_AT_ONCE rows x_0 .. x_{R-1} (R <= ROWS_AT_ONCE) against all M columns: acc[r * M + j].  The rows share every load
   of Yt, and a block of COLS_AT_ONCE columns keeps its partial sums in the L1 cache over the whole chain - the arithmetic of each
   (row, column) pair is the same chain of fmaf in the same order as one row at a time (that form streamed all of Yt and the
   sums through the caches once per row and chain step: 45 s for 40 000 x 40 000 x 512 on 32 cores). */
#define ROWS_AT_ONCE 4
#define COLS_AT_ONCE 1024
static void d2_rows(const float* X, int64_t ldx, int R, const float* xn, int D, const float* Yt, const float* yn, int64_t M,
                    const int* ord, int n, float* acc) {
    float a[ROWS_AT_ONCE][512 + 8];
    for (int64_t j0 = 0; j0 < M; j0 += COLS_AT_ONCE) {
        const int64_t jn = M - j0 < COLS_AT_ONCE ? M - j0 : COLS_AT_ONCE;
        float s[ROWS_AT_ONCE][COLS_AT_ONCE];
        for (int r = 0; r < ROWS_AT_ONCE; ++r)
            for (int64_t j = 0; j < jn; ++j) s[r][j] = 0.f;
        for (int q0 = 0; q0 < n; q0 += 512) {                       /* (the row values of up to 512 chain steps at a time) */
            const int qn = n - q0 < 512 ? n - q0 : 512;
            for (int r = 0; r < ROWS_AT_ONCE; ++r)
                for (int q = 0; q < qn; ++q) {
                    const int k = ord[q0 + q];
                    a[r][q] = (r < R && k < D) ? X[r * ldx + k] : 0.f;
                }
            for (int q = 0; q < qn; ++q) {
                const float* y = Yt + (size_t)(q0 + q) * M + j0;
                const float a0 = a[0][q], a1 = a[1][q], a2 = a[2][q], a3 = a[3][q];
                float *s0 = s[0], *s1 = s[1], *s2 = s[2], *s3 = s[3];
                for (int64_t j = 0; j < jn; ++j) {
                    const float v = y[j];
                    s0[j] = fmaf(v, a0, s0[j]);
                    s1[j] = fmaf(v, a1, s1[j]);
                    s2[j] = fmaf(v, a2, s2[j]);
                    s3[j] = fmaf(v, a3, s3[j]);
                }
            }
        }
        for (int r = 0; r < R; ++r)
            for (int64_t j = 0; j < jn; ++j) {
                const float d2 = fmaf(-2.f, s[r][j], xn[r] + yn[j0 + j]);
                /* NaN (a non-finite row) -> +inf: like torch's NaN it never counts */
                acc[(size_t)r * M + j0 + j] = d2 != d2 ? INFINITY : (d2 < 0.f ? 0.f : d2);
            }
    }
}

static float threshold_of_radius(float R) {
    if (!(R > 0.f)) return 0.f;
    if (isinf(R)) return R;
    float c = R * R;
    for (int it = 0; it < 8; ++it) {
        const float p = nextafterf(c, 0.f);
        if (c > 0.f && sqrtf(p) >= R) c = p; else break;
    }
    for (int it = 0; it < 8; ++it) {
        if (sqrtf(c) < R) c = nextafterf(c, INFINITY); else break;
    }
    return c;
}

void am_exact_threshold(const float* R, int64_t n, float* T) {
    for (int64_t i = 0; i < n; ++i) T[i] = threshold_of_radius(R[i]);
}

/* out_r[i] = sqrtf((k+1)-th smallest d2(i, .)); optionally the squared value in out_r2 */
int am_exact_knn_radii(const float* X, int64_t N, int64_t ldx, const float* Y, int64_t M, int64_t ldy, int D, int k,
                       float* out_r, float* out_r2) {
    if (k + 1 > M || k < 1) return -2;
    int n;
    int* ord = chain_order(D, &n);
    float* xn = (float*)malloc(sizeof(float) * N);
    float* yn = (float*)malloc(sizeof(float) * M);
    am_exact_sqnorm(X, N, ldx, D, xn);
    am_exact_sqnorm(Y, M, ldy, D, yn);
    float* Yt = transpose_in_chain_order(Y, M, ldy, D, ord, n);
    const int k1 = k + 1;
#pragma omp parallel
    {
        float* acc = (float*)malloc(sizeof(float) * M * ROWS_AT_ONCE);
        float* best = (float*)malloc(sizeof(float) * k1);
#pragma omp for schedule(dynamic, 4)
        for (int64_t i0 = 0; i0 < N; i0 += ROWS_AT_ONCE) {
            const int R = N - i0 < ROWS_AT_ONCE ? (int)(N - i0) : ROWS_AT_ONCE;
            d2_rows(X + i0 * ldx, ldx, R, xn + i0, D, Yt, yn, M, ord, n, acc);
            for (int r = 0; r < R; ++r) {
                const float* row = acc + (size_t)r * M;
                for (int s = 0; s < k1; ++s) best[s] = INFINITY;
                for (int64_t j = 0; j < M; ++j) {
                    float v = row[j];
                    if (v < best[k1 - 1]) {
                        int s = k1 - 1;
                        while (s > 0 && best[s - 1] > v) { best[s] = best[s - 1]; --s; }
                        best[s] = v;
                    }
                }
                if (out_r2) out_r2[i0 + r] = best[k1 - 1];
                out_r[i0 + r] = sqrtf(best[k1 - 1]);
            }
        }
        free(acc);
        free(best);
    }
    free(Yt); free(xn); free(yn); free(ord);
    return 0;
}

int am_exact_prdc_counts(const float* R, int64_t Nr, int64_t ldr, const float* C, int64_t Nc, int64_t ldc, int D,
                         const float* r_ref, const float* r_cand, int32_t* col_count, uint8_t* row_any,
                         float* row_min) {
    int n;
    int* ord = chain_order(D, &n);
    float* rn = (float*)malloc(sizeof(float) * Nr);
    float* cn = (float*)malloc(sizeof(float) * Nc);
    float* tr = (float*)malloc(sizeof(float) * Nr);
    float* tc = (float*)malloc(sizeof(float) * Nc);
    am_exact_sqnorm(R, Nr, ldr, D, rn);
    am_exact_sqnorm(C, Nc, ldc, D, cn);
    am_exact_threshold(r_ref, Nr, tr);
    am_exact_threshold(r_cand, Nc, tc);
    float* Ct = transpose_in_chain_order(C, Nc, ldc, D, ord, n);
    memset(col_count, 0, sizeof(int32_t) * Nc);
#pragma omp parallel
    {
        float* acc = (float*)malloc(sizeof(float) * Nc * ROWS_AT_ONCE);
        int32_t* local = (int32_t*)calloc(Nc, sizeof(int32_t));
#pragma omp for schedule(dynamic, 4)
        for (int64_t i0 = 0; i0 < Nr; i0 += ROWS_AT_ONCE) {
            const int nr = Nr - i0 < ROWS_AT_ONCE ? (int)(Nr - i0) : ROWS_AT_ONCE;
            d2_rows(R + i0 * ldr, ldr, nr, rn + i0, D, Ct, cn, Nc, ord, n, acc);
            for (int r = 0; r < nr; ++r) {
                const float* row = acc + (size_t)r * Nc;
                const int64_t i = i0 + r;
                float mn = INFINITY;
                int any = 0;
                for (int64_t j = 0; j < Nc; ++j) {
                    const float d2 = row[j];
                    if (d2 < mn) mn = d2;
                    any |= d2 < tc[j];
                    local[j] += d2 < tr[i];
                }
                row_min[i] = sqrtf(mn);
                row_any[i] = (uint8_t)any;
            }
        }
#pragma omp critical
        for (int64_t j = 0; j < Nc; ++j) col_count[j] += local[j];
        free(acc);
        free(local);
    }
    free(Ct); free(rn); free(cn); free(tr); free(tc); free(ord);
    return 0;
}
