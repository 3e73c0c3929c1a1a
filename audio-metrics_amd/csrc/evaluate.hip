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
#include <vector>

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
    // Where the Frechet blocks run (A/B build only, AM_EVAL_FAD_PLACE; profiles/r6/ab_frechet_stream.txt): 0 = the side stream,
    // free to start once the statistics exist (shipped); 1 = the caller's stream, behind the kernel distance; 2 = the side
    // stream, held back until the first k-NN entry has finished
#ifdef AM_DEV_KNOBS
    static const int fad_place = getenv("AM_EVAL_FAD_PLACE") ? atoi(getenv("AM_EVAL_FAD_PLACE")) : 0;
#else
    constexpr int fad_place = 0;
#endif
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
    if ((what & AM_EVAL_FAD) && fad_place != 2) {
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
            if ((what & AM_EVAL_FAD) && fad_place == 2 && ev_stats.e == nullptr) {
                AM_HIP_TRY(hipEventCreateWithFlags(&ev_stats.e, hipEventDisableTiming));
                AM_HIP_TRY(hipEventRecord(ev_stats.e, st));
            }
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
        const bool serial = fad_place == 1;
        if (!serial) {
            if (ev_stats.e == nullptr) {                              // (place 2 without a k-NN entry in this call)
                AM_HIP_TRY(hipEventCreateWithFlags(&ev_stats.e, hipEventDisableTiming));
                AM_HIP_TRY(hipEventRecord(ev_stats.e, st));
            }
            AM_HIP_TRY(hipStreamWaitEvent(side, ev_stats.e, 0));
        }
        const int block = am_frechet_first_block(), max_iter = 64;
        for (int first = 0; first < 2 * block; first += block)
            if ((rc = am_frechet_enqueue_f64(mean[1], cov[1], mean[0], cov[0], D, first, block, max_iter, 1e-13, L.fad_out,
                                             L.ws_fad, L.fad_ws, serial ? stream : side_stream)) != AM_OK)
                return rc;
        if (!serial) {
            AM_HIP_TRY(hipEventCreateWithFlags(&ev_fad.e, hipEventDisableTiming));
            AM_HIP_TRY(hipEventRecord(ev_fad.e, side));
            AM_HIP_TRY(hipStreamWaitEvent(st, ev_fad.e, 0));
        }
    }
    hipLaunchKernelGGL(eval_pack_kernel, dim3(1), dim3(64), 0, st, L.fad_out, L.totals, out, what);
    AM_LAUNCH_CHECK();
    return AM_OK;
}

// =====================================================================================================================
// One call = one RANK's share of a row-sharded evaluate (VERDICT r4 "missing" item 4): the exchange schedule of
// audio-metrics_amd/distributed.py: evaluate_sharded behind the C ABI, for hosts that are not Python.  The library links no
// collective library: the caller hands in the two collectives of the schedule as stream-ordered hooks (am_collectives;
// audio-metrics_amd/csrc/rccl/am_rccl.cpp builds them over an ncclComm_t, tests build them over torch.distributed).
//
// Streams: compute on `stream`; every collective on `comm_stream` in ISSUE ORDER (one communicator = one queue), fenced with
// events - it waits for the event that marks its input ready, and the compute stream waits for its completion only where
// the result is first read; the Frechet solve on `side_stream`.  comm_stream == stream gives the serial form.
//   column sums (2 D f64, all-reduce)  FIRST: the centred scatters need the global means and nothing may queue behind 200 MB
//   reference rows (all-gather)        under the scatter kernels
//   scatters (2 D^2 f64, all-reduce)   only the Frechet solve waits for it
//   reference set: prepare, bounds of this rank's rows -> all-gather -> candidate rows (all-gather) issued right behind:
//                  they travel under the reference set's sweep -> this rank's share of the sweep -> lists all-gather -> finish
//   candidate set: the same;  membership counts of this rank's reference rows -> counts + row totals (all-reduce, i32)
//   kernel distance: subsets rank, rank + world, ... -> all-reduce of the S values
// Sets the partitioned sweep does not apply to (am_knn_sym_eligible, a rank without rows) take the general kernel on the row
// shard and an all-gather of the radii.  float32 rows; results in the record of am_evaluate_f32 on every rank.
namespace am {

__global__ void shard_scale_kernel(double* __restrict__ v, int64_t n, double f) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) v[i] *= f;
}
// local totals {.., rows any, .., rows covered} -> the two int32 slots behind the column counts
__global__ void shard_rows_kernel(const long long* __restrict__ totals, int32_t* __restrict__ packed_tail) {
    if (threadIdx.x == 0) {
        packed_tail[0] = (int32_t)totals[1];
        packed_tail[1] = (int32_t)totals[3];
    }
}
__global__ void shard_totals_kernel(const long long* __restrict__ col_totals, const int32_t* __restrict__ packed_tail,
                                    long long* __restrict__ out4) {
    if (threadIdx.x == 0) {
        out4[0] = col_totals[0];
        out4[1] = packed_tail[0];
        out4[2] = col_totals[2];
        out4[3] = packed_tail[1];
    }
}

struct ShardLayout {
    size_t total = 0;
    int64_t ldf = 0, width = 0;
    double *sums = nullptr, *scat = nullptr, *fad_out = nullptr, *mmds = nullptr, *mmds_own = nullptr;
    void* ws_fad = nullptr;
    size_t fad_ws = 0;
    float* full[2] = {nullptr, nullptr};
    float *norms[2] = {nullptr, nullptr}, *radii[2] = {nullptr, nullptr};
    uint32_t* pstats[2] = {nullptr, nullptr};
    uint16_t* half[2] = {nullptr, nullptr};
    float *bounds = nullptr, *lists_own = nullptr, *lists_all = nullptr;
    int32_t *packed = nullptr, *col = nullptr;
    uint8_t *rany = nullptr, *rcov = nullptr;
    long long *totals_local = nullptr, *totals_col = nullptr, *totals = nullptr;
    int64_t *idx_own[2] = {nullptr, nullptr};
    void* ws_main = nullptr;
    size_t ws_main_bytes = 0;
};

static int64_t shard_sum(const int64_t* c, int n) {
    int64_t s = 0;
    for (int i = 0; i < n; ++i) s += c[i];
    return s;
}

static bool shard_layout(Carver& c, const int64_t* ref_counts, const int64_t* cand_counts, int rank, int world, int D, int k, int S,
                         int m, unsigned what, ShardLayout& L) {
    const int64_t n[2] = {shard_sum(ref_counts, world), shard_sum(cand_counts, world)};
    const int64_t nl[2] = {ref_counts[rank], cand_counts[rank]};
    L.ldf = (int64_t)round_up((size_t)D, (size_t)4);
    size_t main_bytes = 256;
    if (what & AM_EVAL_FAD) {
        L.sums = c.take<double>((size_t)2 * D);
        L.scat = c.take<double>((size_t)2 * D * D);
        L.fad_ws = am_frechet_workspace_bytes(D);
        L.ws_fad = c.take<char>(L.fad_ws);
        L.fad_out = c.take<double>(8);
        main_bytes = std::max({main_bytes, am_stats_workspace_bytes(std::max<int64_t>(nl[0], 1), D),
                               am_stats_workspace_bytes(std::max<int64_t>(nl[1], 1), D)});
    }
    if (what & (AM_EVAL_KD | AM_EVAL_PRDC))
        for (int s = 0; s < 2; ++s) L.full[s] = c.take<float>((size_t)n[s] * L.ldf);
    if (what & AM_EVAL_PRDC) {
        const int64_t ldh = am_prepared_half_ld(D);
        L.width = am_knn_list_width(k);
        int64_t nmax = std::max(n[0], n[1]);
        for (int s = 0; s < 2; ++s) {
            L.norms[s] = c.take<float>(n[s]);
            L.pstats[s] = c.take<uint32_t>(4);
            L.half[s] = c.take<uint16_t>((size_t)n[s] * ldh);
            L.radii[s] = c.take<float>(n[s]);
            main_bytes = std::max(main_bytes, am_knn_workspace_bytes(std::max<int64_t>(nl[s], 1), n[s], D, k));
            if (world > 1 && am_knn_sym_eligible(n[s], D, k))
                main_bytes = std::max({main_bytes, am_knn_part_workspace_bytes(n[s], D, k), am_knn_lists_finish_workspace_bytes(n[s], D, k)});
        }
        L.bounds = c.take<float>(nmax);
        L.lists_own = c.take<float>((size_t)nmax * L.width);
        L.lists_all = c.take<float>((size_t)world * nmax * L.width);
        L.packed = c.take<int32_t>(n[1] + 2);
        L.col = c.take<int32_t>(n[1]);
        L.rany = c.take<uint8_t>(std::max<int64_t>(nl[0], 1));
        L.rcov = c.take<uint8_t>(std::max<int64_t>(nl[0], 1));
        L.totals_local = c.take<long long>(4);
        L.totals_col = c.take<long long>(4);
        L.totals = c.take<long long>(4);
        main_bytes = std::max(main_bytes, am_prdc_workspace_bytes(std::max<int64_t>(nl[0], 1), n[1], D));
    }
    if (what & AM_EVAL_KD) {
        const int own = rank < S ? (S - rank + world - 1) / world : 0;
        L.mmds = c.take<double>(S);
        L.mmds_own = c.take<double>(std::max(own, 1));
        for (int s = 0; s < 2; ++s) L.idx_own[s] = c.take<int64_t>((size_t)std::max(own, 1) * m);
        if (own > 0) main_bytes = std::max(main_bytes, am_kd_poly_workspace_bytes(own, m, D));
    }
    L.ws_main_bytes = round_up(main_bytes, (size_t)256);
    L.ws_main = c.take<char>(L.ws_main_bytes);
    L.total = c.off;
    return c.ok();
}

}  // namespace am

static bool shard_args_ok(const int64_t* ref_counts, const int64_t* cand_counts, int rank, int world) {
    if (!ref_counts || !cand_counts || world < 1 || rank < 0 || rank >= world) return false;
    for (int r = 0; r < world; ++r)
        if (ref_counts[r] < 0 || cand_counts[r] < 0) return false;
    return true;
}

extern "C" size_t am_evaluate_sharded_workspace_bytes(const int64_t* ref_counts, const int64_t* cand_counts, int rank, int world,
                                                      int D, int nearest_k, int kd_subsets, int kd_m, unsigned what) {
    if (!shard_args_ok(ref_counts, cand_counts, rank, world) || D < 1) return 0;
    if (shard_sum(ref_counts, world) < 1 || shard_sum(cand_counts, world) < 1) return 0;
    Carver c(nullptr, 0);
    ShardLayout L;
    shard_layout(c, ref_counts, cand_counts, rank, world, D, nearest_k, kd_subsets, kd_m, what, L);
    return L.total;
}

extern "C" int am_evaluate_sharded_f32(const float* ref_local, int64_t ld_ref, const float* cand_local, int64_t ld_cand, int D,
                                       const int64_t* ref_counts, const int64_t* cand_counts, const am_collectives* coll,
                                       unsigned what, int nearest_k, const int64_t* idx_cand, const int64_t* idx_ref, int kd_subsets,
                                       int kd_m, double kd_gamma, double kd_coef0, int kd_degree, double* out, void* ws,
                                       size_t ws_bytes, am_stream_t stream, am_stream_t side_stream, am_stream_t comm_stream) {
    AM_REQUIRE(coll && coll->all_reduce_sum && coll->all_gather_v && out, AM_ERR_BAD_ARG, "null pointer");
    const int rank = coll->rank, world = coll->world;
    AM_REQUIRE(shard_args_ok(ref_counts, cand_counts, rank, world), AM_ERR_BAD_ARG, "rank %d of %d, or negative shard sizes", rank, world);
    AM_REQUIRE(D >= 1, AM_ERR_BAD_SHAPE, "D = %d", D);
    AM_REQUIRE((what & ~(unsigned)(AM_EVAL_FAD | AM_EVAL_KD | AM_EVAL_PRDC)) == 0 && what != 0, AM_ERR_BAD_ARG, "what = %u", what);
    AM_REQUIRE(!(what & AM_EVAL_KD) || (idx_cand && idx_ref && kd_subsets >= 1 && kd_m >= 1), AM_ERR_BAD_ARG,
               "kernel distance needs the two index tables");
    AM_REQUIRE(!(what & AM_EVAL_FAD) || side_stream != stream, AM_ERR_BAD_ARG,
               "the Frechet solve runs on side_stream, which must differ from stream");
    const int64_t n[2] = {shard_sum(ref_counts, world), shard_sum(cand_counts, world)};
    // the same error on every rank: all of them hold the totals (distributed.py: evaluate_sharded)
    AM_REQUIRE(n[0] >= 1 && n[1] >= 1, AM_ERR_BAD_SHAPE, "empty embedding set: %lld reference and %lld candidate rows over %d ranks",
               (long long)n[0], (long long)n[1], world);
    const int64_t* counts[2] = {ref_counts, cand_counts};
    const int64_t nl[2] = {ref_counts[rank], cand_counts[rank]};
    int64_t lo[2] = {0, 0};
    for (int r = 0; r < rank; ++r) {
        lo[0] += ref_counts[r];
        lo[1] += cand_counts[r];
    }
    const float* X[2] = {ref_local, cand_local};
    const int64_t ld[2] = {ld_ref, ld_cand};
    for (int s = 0; s < 2; ++s) AM_REQUIRE(nl[s] == 0 || X[s] != nullptr, AM_ERR_BAD_ARG, "null pointer for a shard of %lld rows", (long long)nl[s]);
    hipStream_t st = static_cast<hipStream_t>(stream), side = static_cast<hipStream_t>(side_stream),
                cs = static_cast<hipStream_t>(comm_stream ? comm_stream : stream);
    am_stream_t comm = comm_stream ? comm_stream : stream;
    Carver c(ws, ws_bytes);
    ShardLayout L;
    AM_REQUIRE(shard_layout(c, ref_counts, cand_counts, rank, world, D, nearest_k, kd_subsets, kd_m, what, L), AM_ERR_WORKSPACE,
               "workspace too small: need %zu bytes, have %zu", L.total, ws_bytes);
    int rc;

    // events of this call (destroyed on every way out)
    struct Events {
        std::vector<hipEvent_t> all;
        ~Events() {
            for (hipEvent_t e : all) (void)hipEventDestroy(e);
        }
    } events;
    auto mark = [&](hipStream_t on, hipEvent_t* out_e) -> hipError_t {          // a new event recorded on `on`
        hipEvent_t e = nullptr;
        hipError_t err = hipEventCreateWithFlags(&e, hipEventDisableTiming);
        if (err != hipSuccess) return err;
        events.all.push_back(e);
        *out_e = e;
        return hipEventRecord(e, on);
    };
    // a collective: the communication stream waits for the compute stream's present position (its input), the hook
    // enqueues it there, `done` marks its end - waited for by whoever reads the result first
    auto collective = [&](hipEvent_t* done, auto&& call) -> int {
        if (cs != st) {
            hipEvent_t ready;
            AM_HIP_TRY(mark(st, &ready));
            AM_HIP_TRY(hipStreamWaitEvent(cs, ready, 0));
        }
        const int hook_rc = call();
        if (hook_rc != 0) {
            set_error("a collective hook returned %d", hook_rc);
            return AM_ERR_HIP;
        }
        AM_HIP_TRY(mark(cs, done));
        return AM_OK;
    };
    auto wait_on = [&](hipStream_t who, hipEvent_t e) -> hipError_t { return (who == cs) ? hipSuccess : hipStreamWaitEvent(who, e, 0); };
    std::vector<int64_t> bytes(world);
    // all-gather of per-row data of set s (`per_row` bytes each), in place: this rank's part already sits at its offset
    auto gather_rows = [&](int s, void* buf, int64_t per_row, hipEvent_t* done) -> int {
        return collective(done, [&]() {
            for (int r = 0; r < world; ++r) bytes[r] = counts[s][r] * per_row;
            return coll->all_gather_v(coll->ctx, static_cast<char*>(buf) + lo[s] * per_row, buf, bytes.data(), comm);
        });
    };

    // ---- 1. column sums first
    hipEvent_t ev_sums = nullptr, ev_scat = nullptr, ev_full[2] = {nullptr, nullptr};
    if (what & AM_EVAL_FAD) {
        for (int s = 0; s < 2; ++s) {
            if (nl[s] > 0) {
                if ((rc = am_colsum_f32(X[s], nl[s], D, ld[s], L.sums + (size_t)s * D, L.ws_main, L.ws_main_bytes, stream)) != AM_OK) return rc;
            } else {
                AM_HIP_TRY(hipMemsetAsync(L.sums + (size_t)s * D, 0, (size_t)D * sizeof(double), st));
            }
        }
        if ((rc = collective(&ev_sums, [&]() { return coll->all_reduce_sum(coll->ctx, L.sums, (int64_t)2 * D, AM_COLL_F64, comm); })) != AM_OK)
            return rc;
    }
    // ---- 2. the rows of this rank into their place of the gathered copies; the reference rows' gather starts
    const bool need_full = (what & (AM_EVAL_KD | AM_EVAL_PRDC)) != 0;
    auto start_gather = [&](int s) -> int {
        if (ev_full[s] != nullptr) return AM_OK;
        return gather_rows(s, L.full[s], L.ldf * (int64_t)sizeof(float), &ev_full[s]);
    };
    if (need_full) {
        for (int s = 0; s < 2; ++s) {
            if (nl[s] == 0) continue;
            // ld == ldf: one contiguous copy (a pitched device-to-device copy of 100 000 rows took 35 ms) of everything up to the
            // LAST row's D-th element - the header promises ld % 4 == 0 and 16-byte alignment, not a padded final row
            if (ld[s] == L.ldf)
                AM_HIP_TRY(hipMemcpyAsync(L.full[s] + lo[s] * L.ldf, X[s], ((size_t)(nl[s] - 1) * L.ldf + (size_t)D) * sizeof(float),
                                          hipMemcpyDeviceToDevice, st));
            else
                AM_HIP_TRY(hipMemcpy2DAsync(L.full[s] + lo[s] * L.ldf, (size_t)L.ldf * sizeof(float), X[s], (size_t)ld[s] * sizeof(float),
                                            (size_t)D * sizeof(float), (size_t)nl[s], hipMemcpyDeviceToDevice, st));
        }
        if (L.ldf != D)                                         // padding columns of the gathered copies: zero in BOTH forms (never
            for (int s = 0; s < 2; ++s)                         // read as data; the caller's padding is not propagated to other ranks)
                if (nl[s] > 0)
                    AM_HIP_TRY(hipMemset2DAsync(L.full[s] + lo[s] * L.ldf + D, (size_t)L.ldf * sizeof(float), 0,
                                                (size_t)(L.ldf - D) * sizeof(float), (size_t)nl[s], st));
        if ((rc = start_gather(0)) != AM_OK) return rc;
        if (!(what & AM_EVAL_PRDC) && (rc = start_gather(1)) != AM_OK) return rc;          // KD only: nothing to order it behind
    }
    // ---- 3. centred scatters around the global means; covariances; the Frechet solve is enqueued at the end
    if (what & AM_EVAL_FAD) {
        AM_HIP_TRY(wait_on(st, ev_sums));
        for (int s = 0; s < 2; ++s)
            hipLaunchKernelGGL(shard_scale_kernel, dim3(std::min<int64_t>(ceil_div((int64_t)D, (int64_t)256), 1024)), dim3(256), 0, st,
                               L.sums + (size_t)s * D, (int64_t)D, 1.0 / (double)n[s]);
        AM_LAUNCH_CHECK();
        for (int s = 0; s < 2; ++s) {
            double* sc = L.scat + (size_t)s * D * D;
            if (nl[s] > 0) {
                if ((rc = am_scatter_f32(X[s], nl[s], D, ld[s], L.sums + (size_t)s * D, sc, L.ws_main, L.ws_main_bytes, stream)) != AM_OK) return rc;
            } else {
                AM_HIP_TRY(hipMemsetAsync(sc, 0, (size_t)D * D * sizeof(double), st));
            }
        }
        if ((rc = collective(&ev_scat, [&]() { return coll->all_reduce_sum(coll->ctx, L.scat, (int64_t)2 * D * D, AM_COLL_F64, comm); })) != AM_OK)
            return rc;
    }

    // ---- 4. PRDC
    if (what & AM_EVAL_PRDC) {
        am_prepared_set prep[2];
        for (int s = 0; s < 2; ++s) {
            if ((rc = start_gather(s)) != AM_OK) return rc;                  // (the candidate rows: issued in the reference pass, below)
            AM_HIP_TRY(wait_on(st, ev_full[s]));
            if ((rc = am_prepare_set_f32(L.full[s], n[s], L.ldf, D, L.norms[s], L.pstats[s], L.half[s], stream)) != AM_OK) return rc;
            prep[s].norms = L.norms[s];
            prep[s].stats = L.pstats[s];
            prep[s].half = L.half[s];
            bool every_rank_has_rows = true;
            for (int r = 0; r < world; ++r) every_rank_has_rows = every_rank_has_rows && counts[s][r] > 0;
            hipEvent_t ev = nullptr;
            if (world > 1 && every_rank_has_rows && am_knn_sym_eligible(n[s], D, nearest_k)) {
                if ((rc = am_knn_bounds_prepared_f32(L.full[s], n[s], L.ldf, D, &prep[s], nearest_k, lo[s], nl[s], L.bounds + lo[s],
                                                     L.ws_main, L.ws_main_bytes, stream)) != AM_OK) return rc;
                if ((rc = gather_rows(s, L.bounds, sizeof(float), &ev)) != AM_OK) return rc;
                if (s == 0 && (rc = start_gather(1)) != AM_OK) return rc;     // right behind the 400 KB exchange: under the sweep
                AM_HIP_TRY(wait_on(st, ev));
                if ((rc = am_knn_sym_part_prepared_f32(L.full[s], n[s], L.ldf, D, &prep[s], nearest_k, rank, world, L.bounds, L.lists_own,
                                                       L.ws_main, L.ws_main_bytes, stream)) != AM_OK) return rc;
                const int64_t list_bytes = n[s] * L.width * (int64_t)sizeof(float);
                AM_HIP_TRY(hipMemcpyAsync(L.lists_all + (size_t)rank * n[s] * L.width, L.lists_own, (size_t)list_bytes, hipMemcpyDeviceToDevice, st));
                if ((rc = collective(&ev, [&]() {
                         for (int r = 0; r < world; ++r) bytes[r] = list_bytes;
                         return coll->all_gather_v(coll->ctx, reinterpret_cast<char*>(L.lists_all) + (size_t)rank * list_bytes, L.lists_all,
                                                   bytes.data(), comm);
                     })) != AM_OK) return rc;
                AM_HIP_TRY(wait_on(st, ev));
                if ((rc = am_knn_lists_finish_f32(L.lists_all, world, L.full[s], n[s], L.ldf, D, nearest_k, L.radii[s], L.ws_main,
                                                  L.ws_main_bytes, stream)) != AM_OK) return rc;
            } else {
                if (s == 0 && (rc = start_gather(1)) != AM_OK) return rc;
                if (world == 1) {
                    if ((rc = am_knn_radii_prepared_f32(L.full[s], n[s], L.ldf, D, &prep[s], nearest_k, L.radii[s], L.ws_main,
                                                        L.ws_main_bytes, stream)) != AM_OK) return rc;
                } else {
                    if (nl[s] > 0 && (rc = am_knn_radii_f32(L.full[s] + lo[s] * L.ldf, nl[s], L.ldf, L.full[s], n[s], L.ldf, D, nearest_k,
                                                            L.radii[s] + lo[s], L.ws_main, L.ws_main_bytes, stream)) != AM_OK) return rc;
                    if ((rc = gather_rows(s, L.radii[s], sizeof(float), &ev)) != AM_OK) return rc;
                    AM_HIP_TRY(wait_on(st, ev));
                }
            }
        }
        // membership counts of this rank's reference rows against all candidates
        AM_HIP_TRY(hipMemsetAsync(L.packed, 0, (size_t)(n[1] + 2) * sizeof(int32_t), st));
        if (nl[0] > 0) {
            am_prepared_set shard = prep[0];
            shard.norms = prep[0].norms + lo[0];
            shard.half = prep[0].half + lo[0] * am_prepared_half_ld(D);
            if ((rc = am_prdc_counts_prepared_f32(L.full[0] + lo[0] * L.ldf, nl[0], L.ldf, &shard, L.full[1], n[1], L.ldf, &prep[1], D,
                                                  L.radii[0] + lo[0], L.radii[1], L.col, L.rany, L.rcov, nullptr, L.ws_main, L.ws_main_bytes,
                                                  stream)) != AM_OK) return rc;
            if ((rc = am_prdc_reduce(L.col, n[1], L.rany, L.rcov, nl[0], reinterpret_cast<int64_t*>(L.totals_local), stream)) != AM_OK) return rc;
            AM_HIP_TRY(hipMemcpyAsync(L.packed, L.col, (size_t)n[1] * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
            hipLaunchKernelGGL(shard_rows_kernel, dim3(1), dim3(64), 0, st, L.totals_local, L.packed + n[1]);
            AM_LAUNCH_CHECK();
        }
        hipEvent_t ev = nullptr;
        if ((rc = collective(&ev, [&]() { return coll->all_reduce_sum(coll->ctx, L.packed, n[1] + 2, AM_COLL_I32, comm); })) != AM_OK) return rc;
        AM_HIP_TRY(wait_on(st, ev));
        if ((rc = am_prdc_reduce(L.packed, n[1], nullptr, nullptr, 0, reinterpret_cast<int64_t*>(L.totals_col), stream)) != AM_OK) return rc;
        hipLaunchKernelGGL(shard_totals_kernel, dim3(1), dim3(64), 0, st, L.totals_col, L.packed + n[1], L.totals);
        AM_LAUNCH_CHECK();
    }

    // ---- 5. kernel distance: the subsets rank, rank + world, ... (features_1 = candidate, audio_metrics.py:260)
    if (what & AM_EVAL_KD) {
        for (int s = 0; s < 2; ++s) {
            if ((rc = start_gather(s)) != AM_OK) return rc;
            AM_HIP_TRY(wait_on(st, ev_full[s]));
        }
        const int own = rank < kd_subsets ? (kd_subsets - rank + world - 1) / world : 0;
        AM_HIP_TRY(hipMemsetAsync(L.mmds, 0, (size_t)kd_subsets * sizeof(double), st));
        if (own > 0) {
            const size_t row_bytes = (size_t)kd_m * sizeof(int64_t);
            AM_HIP_TRY(hipMemcpy2DAsync(L.idx_own[0], row_bytes, idx_cand + (size_t)rank * kd_m, row_bytes * world, row_bytes, (size_t)own,
                                        hipMemcpyDeviceToDevice, st));
            AM_HIP_TRY(hipMemcpy2DAsync(L.idx_own[1], row_bytes, idx_ref + (size_t)rank * kd_m, row_bytes * world, row_bytes, (size_t)own,
                                        hipMemcpyDeviceToDevice, st));
            if ((rc = am_kd_poly_f32(L.full[1], n[1], L.ldf, L.full[0], n[0], L.ldf, D, L.idx_own[0], L.idx_own[1], own, kd_m, kd_gamma,
                                     kd_coef0, kd_degree, L.mmds_own, L.ws_main, L.ws_main_bytes, stream)) != AM_OK) return rc;
            AM_HIP_TRY(hipMemcpy2DAsync(L.mmds + rank, sizeof(double) * world, L.mmds_own, sizeof(double), sizeof(double), (size_t)own,
                                        hipMemcpyDeviceToDevice, st));
        }
        hipEvent_t ev = nullptr;
        if ((rc = collective(&ev, [&]() { return coll->all_reduce_sum(coll->ctx, L.mmds, (int64_t)kd_subsets, AM_COLL_F64, comm); })) != AM_OK) return rc;
        AM_HIP_TRY(wait_on(st, ev));
        AM_HIP_TRY(hipMemcpyAsync(out + AM_EVAL_HEAD, L.mmds, (size_t)kd_subsets * sizeof(double), hipMemcpyDeviceToDevice, st));
    }

    // ---- 6. Frechet blocks on the side stream, replicated on every rank (a 2 MB problem), behind the scatters' all-reduce
    if (what & AM_EVAL_FAD) {
        if (cs != side) AM_HIP_TRY(hipStreamWaitEvent(side, ev_scat, 0));
        for (int s = 0; s < 2; ++s)
            hipLaunchKernelGGL(shard_scale_kernel, dim3(std::min<int64_t>(ceil_div((int64_t)D * D, (int64_t)256), 1024)), dim3(256), 0, side,
                               L.scat + (size_t)s * D * D, (int64_t)D * D, 1.0 / (double)std::max<int64_t>(n[s] - 1, 1));
        AM_LAUNCH_CHECK();
        const int block = am_frechet_first_block(), max_iter = 64;
        for (int first = 0; first < 2 * block; first += block)
            if ((rc = am_frechet_enqueue_f64(L.sums + D, L.scat + (size_t)D * D, L.sums, L.scat, D, first, block, max_iter, 1e-13, L.fad_out,
                                             L.ws_fad, L.fad_ws, side_stream)) != AM_OK) return rc;
        hipEvent_t ev_fad;
        AM_HIP_TRY(mark(side, &ev_fad));
        AM_HIP_TRY(hipStreamWaitEvent(st, ev_fad, 0));
    }
    hipLaunchKernelGGL(eval_pack_kernel, dim3(1), dim3(64), 0, st, L.fad_out, L.totals, out, what);
    AM_LAUNCH_CHECK();
    return AM_OK;
}
