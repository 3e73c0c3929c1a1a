// One call = one evaluate(): the stream-ordered chain behind AudioMetrics.evaluate's FAD + KD + PRDC dispatch
// (reference src/audio_metrics/audio_metrics.py:254-274: one call, one dict) for two embedding sets resident on ONE
// device.  Every step is one of this library's own entry points, issued back to back on the caller's stream with the
// workspaces carved from one caller buffer; nothing is read back in between (path selection inside the entry points is a
// pure function of the shapes), and the results land in ONE small device buffer the caller copies once:
//
//   stats(ref), stats(cand)            am_stats_f32                       (skipped for a side whose statistics are handed in)
//   Frechet distance                   am_frechet_enqueue_f64 on `side_stream`, under the PRDC kernels
//   prepare(ref), prepare(cand)        am_prepare_set_f32
//   radii(ref), radii(cand)            am_knn_radii_prepared_f32          (skipped for a side whose radii are handed in)
//   membership counts + totals         am_prdc_counts_prepared_f32, am_prdc_reduce
//   kernel distance                    am_kd_poly_f32 on the caller's index tables
//   pack                               eval_pack_kernel: Frechet record + PRDC totals -> out[0..15]; MMD^2 per subset -> out[16..]
//
// The separate-launch form of the same chain (hip_ops.py / distributed.py) left 0.5-0.7 ms of GPU idle time per 31 ms
// step - three blocking read-backs at the end, a count read-back at the start - and ~25 glue kernels of the tensor
// library between the entry points (rocprofv3 timeline, profiles/r3).
#include "am_common.h"
#include <algorithm>

namespace am {

__global__ void eval_pack_kernel(const double* __restrict__ fad5, const long long* __restrict__ totals, double* __restrict__ out,
                                 unsigned what) {
    const int t = threadIdx.x;
    if (t < AM_EVAL_HEAD) {
        double v = 0.0;
        if (t < 5) v = (what & AM_EVAL_FAD) ? fad5[t] : (t == 4 ? -1.0 : 0.0);
        else if (t < 9) v = (what & AM_EVAL_PRDC) ? (double)totals[t - 5] : 0.0;
        out[t] = v;
    }
}

struct EvalLayout {
    size_t total = 0;
    size_t stats_ws = 0, knn_ws = 0, prdc_ws = 0, kd_ws = 0, fad_ws = 0;
    double *mean[2] = {nullptr, nullptr}, *cov[2] = {nullptr, nullptr};
    float *norms[2] = {nullptr, nullptr}, *radii[2] = {nullptr, nullptr};
    uint32_t* pstats[2] = {nullptr, nullptr};
    uint16_t* half[2] = {nullptr, nullptr};
    int32_t* col = nullptr;
    uint8_t *rany = nullptr, *rcov = nullptr;
    long long* totals = nullptr;
    double* fad_out = nullptr;
    void *ws_main = nullptr, *ws_fad = nullptr;      // ws_main is reused by the stream-ordered steps of the caller's stream
    size_t ws_main_bytes = 0;
};

static bool eval_layout(Carver& c, int64_t n_ref, int64_t n_cand, int D, int k, int S, int m, unsigned what, EvalLayout& L) {
    const int64_t n[2] = {n_ref, n_cand};
    for (int s = 0; s < 2; ++s) {
        L.mean[s] = c.take<double>(D);
        L.cov[s] = c.take<double>((size_t)D * D);
    }
    size_t main_bytes = 0;
    if (what & AM_EVAL_FAD) {
        main_bytes = std::max({main_bytes, am_stats_workspace_bytes(n_ref, D), am_stats_workspace_bytes(n_cand, D)});
        L.fad_ws = am_frechet_workspace_bytes(D);
        L.ws_fad = c.take<char>(L.fad_ws);
        L.fad_out = c.take<double>(8);
    }
    if (what & AM_EVAL_PRDC) {
        const int64_t ldh = am_prepared_half_ld(D);
        for (int s = 0; s < 2; ++s) {
            L.norms[s] = c.take<float>(n[s]);
            L.pstats[s] = c.take<uint32_t>(4);
            L.half[s] = c.take<uint16_t>((size_t)n[s] * ldh);
            L.radii[s] = c.take<float>(n[s]);
        }
        L.col = c.take<int32_t>(n_cand);
        L.rany = c.take<uint8_t>(n_ref);
        L.rcov = c.take<uint8_t>(n_ref);
        L.totals = c.take<long long>(4);
        main_bytes = std::max({main_bytes, am_knn_workspace_bytes(n_ref, n_ref, D, k), am_knn_workspace_bytes(n_cand, n_cand, D, k),
                               am_prdc_workspace_bytes(n_ref, n_cand, D)});
    }
    if (what & AM_EVAL_KD) main_bytes = std::max(main_bytes, am_kd_poly_workspace_bytes(S, m, D));
    L.ws_main_bytes = round_up(std::max<size_t>(main_bytes, 256), 256);
    L.ws_main = c.take<char>(L.ws_main_bytes);
    L.total = c.off;
    return c.ok();
}

}  // namespace am

using namespace am;

extern "C" size_t am_evaluate_workspace_bytes(int64_t n_ref, int64_t n_cand, int D, int nearest_k, int kd_subsets, int kd_m,
                                              unsigned what) {
    if (n_ref < 1 || n_cand < 1 || D < 1) return 0;
    Carver c(nullptr, 0);
    EvalLayout L;
    eval_layout(c, n_ref, n_cand, D, nearest_k, kd_subsets, kd_m, what, L);
    return L.total;
}

extern "C" int am_evaluate_f32(const float* ref, int64_t n_ref, int64_t ld_ref, const float* cand, int64_t n_cand,
                               int64_t ld_cand, int D, unsigned what, int nearest_k, const int64_t* idx_cand,
                               const int64_t* idx_ref, int kd_subsets, int kd_m, double kd_gamma, double kd_coef0, int kd_degree,
                               const am_evaluate_side* given_ref, const am_evaluate_side* given_cand, double* out, void* ws,
                               size_t ws_bytes, am_stream_t stream, am_stream_t side_stream) {
    AM_REQUIRE(ref && cand && out, AM_ERR_BAD_ARG, "null pointer");
    AM_REQUIRE(n_ref >= 1 && n_cand >= 1 && D >= 1, AM_ERR_BAD_SHAPE, "sets of %lld and %lld rows x %d", (long long)n_ref,
               (long long)n_cand, D);
    AM_REQUIRE((what & ~(unsigned)(AM_EVAL_FAD | AM_EVAL_KD | AM_EVAL_PRDC)) == 0 && what != 0, AM_ERR_BAD_ARG, "what = %u", what);
    AM_REQUIRE(!(what & AM_EVAL_KD) || (idx_cand && idx_ref && kd_subsets >= 1 && kd_m >= 1), AM_ERR_BAD_ARG,
               "kernel distance needs the two index tables");
    AM_REQUIRE(!(what & AM_EVAL_FAD) || side_stream != stream, AM_ERR_BAD_ARG,
               "the Frechet solve runs on side_stream, which must differ from stream");
    hipStream_t st = static_cast<hipStream_t>(stream), side = static_cast<hipStream_t>(side_stream);
    Carver c(ws, ws_bytes);
    EvalLayout L;
    AM_REQUIRE(eval_layout(c, n_ref, n_cand, D, nearest_k, kd_subsets, kd_m, what, L), AM_ERR_WORKSPACE,
               "workspace too small: need %zu bytes, have %zu", L.total, ws_bytes);
    const float* X[2] = {ref, cand};
    const int64_t n[2] = {n_ref, n_cand}, ld[2] = {ld_ref, ld_cand};
    const am_evaluate_side* given[2] = {given_ref, given_cand};
    int rc;

    // ---- statistics, then the Frechet solve on the side stream (stopping rule on the device: kernels behind the stopping
    //      point return at once).  Two blocks are enqueued; a product so ill-conditioned that it needs more leaves
    //      out[4] == 0 and the caller finishes it with am_frechet_f64 on the statistics.
    const double* mean[2];
    const double* cov[2];
    if (what & AM_EVAL_FAD) {
        for (int s = 0; s < 2; ++s) {
            if (given[s] && given[s]->mean && given[s]->cov) {
                mean[s] = given[s]->mean;
                cov[s] = given[s]->cov;
                continue;
            }
            double* m_out = given[s] && given[s]->mean_out ? given[s]->mean_out : L.mean[s];
            double* c_out = given[s] && given[s]->cov_out ? given[s]->cov_out : L.cov[s];
            if ((rc = am_stats_f32(X[s], n[s], D, ld[s], m_out, c_out, L.ws_main, L.ws_main_bytes, stream)) != AM_OK) return rc;
            mean[s] = m_out;
            cov[s] = c_out;
        }
    }
    // (events are per call: two host threads may evaluate on one device at the same time, each on its own streams)
    // (destroyed on every way out: an entry point below may return an error between creation and the waits)
    struct EventGuard {
        hipEvent_t e = nullptr;
        ~EventGuard() {
            if (e) (void)hipEventDestroy(e);                         // (released once the recorded work has completed)
        }
    } ev_stats, ev_fad;
    if (what & AM_EVAL_FAD) {
        AM_HIP_TRY(hipEventCreateWithFlags(&ev_stats.e, hipEventDisableTiming));
        AM_HIP_TRY(hipEventRecord(ev_stats.e, st));
    }

    // ---- PRDC on the caller's stream
    if (what & AM_EVAL_PRDC) {
        am_prepared_set prep[2];
        for (int s = 0; s < 2; ++s) {
            if ((rc = am_prepare_set_f32(X[s], n[s], ld[s], D, L.norms[s], L.pstats[s], L.half[s], stream)) != AM_OK) return rc;
            prep[s].norms = L.norms[s];
            prep[s].stats = L.pstats[s];
            prep[s].half = L.half[s];
        }
        const float* radii[2];
        for (int s = 0; s < 2; ++s) {
            if (given[s] && given[s]->radii) {
                radii[s] = given[s]->radii;
                continue;
            }
            float* r_out = given[s] && given[s]->radii_out ? given[s]->radii_out : L.radii[s];
            if ((rc = am_knn_radii_prepared_f32(X[s], n[s], ld[s], D, &prep[s], nearest_k, r_out, L.ws_main, L.ws_main_bytes,
                                                stream)) != AM_OK)
                return rc;
            radii[s] = r_out;
        }
        if ((rc = am_prdc_counts_prepared_f32(ref, n_ref, ld_ref, &prep[0], cand, n_cand, ld_cand, &prep[1], D, radii[0], radii[1],
                                              L.col, L.rany, L.rcov, nullptr, L.ws_main, L.ws_main_bytes, stream)) != AM_OK)
            return rc;
        if ((rc = am_prdc_reduce(L.col, n_cand, L.rany, L.rcov, n_ref, reinterpret_cast<int64_t*>(L.totals), stream)) != AM_OK)
            return rc;
    }

    // ---- kernel distance: per-subset values straight into the result buffer (features_1 = candidate, audio_metrics.py:260)
    if (what & AM_EVAL_KD) {
        if ((rc = am_kd_poly_f32(cand, n_cand, ld_cand, ref, n_ref, ld_ref, D, idx_cand, idx_ref, kd_subsets, kd_m, kd_gamma,
                                 kd_coef0, kd_degree, out + AM_EVAL_HEAD, L.ws_main, L.ws_main_bytes, stream)) != AM_OK)
            return rc;
    }

    // ---- Frechet blocks: enqueued LAST by the host (the caller's stream already holds the long kernels, so the GPU never
    //      waits for these launches), executed on the side stream as soon as the statistics exist
    if (what & AM_EVAL_FAD) {
        AM_HIP_TRY(hipStreamWaitEvent(side, ev_stats.e, 0));
        const int block = am_frechet_first_block(), max_iter = 64;
        for (int first = 0; first < 2 * block; first += block)
            if ((rc = am_frechet_enqueue_f64(mean[1], cov[1], mean[0], cov[0], D, first, block, max_iter, 1e-13, L.fad_out,
                                             L.ws_fad, L.fad_ws, side_stream)) != AM_OK)
                return rc;
        AM_HIP_TRY(hipEventCreateWithFlags(&ev_fad.e, hipEventDisableTiming));
        AM_HIP_TRY(hipEventRecord(ev_fad.e, side));
        AM_HIP_TRY(hipStreamWaitEvent(st, ev_fad.e, 0));
    }
    hipLaunchKernelGGL(eval_pack_kernel, dim3(1), dim3(64), 0, st, L.fad_out, L.totals, out, what);
    AM_LAUNCH_CHECK();
    return AM_OK;
}
