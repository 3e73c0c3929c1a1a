"""The PRODUCTION form of the PRDC kernels - the f16 filter sweeps on the 256 x 256 engine with exact f32 verification,
the path of every set of >= 6144 / 8192 rows (am_knn_path == am_prdc_path == 3) - against

  (a) outputs of the REFERENCE itself at sizes it can still run (tests/golden/prdc_large.npz, written by
      tests/golden/make_goldens.py prdc_large from /root/reference/src/audio_metrics/metrics/prdc.py:4-50): radii within
      f32 noise of torch.cdist's matmul form, membership flips <= 1e-4 of the inside pairs, the four values within 1e-4;
  (b) the plain-C model of the device arithmetic (oracle/exact_c, OpenMP): radii, integer counts and row flags BIT-EXACT;
  (c) oracle.prdc_blocked on bench.py's own 2 x 100k x 512 sets (tests/golden/bench_prdc.npz)."""
import numpy as np
import pytest
import torch

import inputs as gi

pytestmark = pytest.mark.gpu
REL = 1e-4


@pytest.fixture(scope="module")
def am():
    import audio_metrics_amd
    assert torch.cuda.is_available(), "these tests need the MI355X"
    audio_metrics_amd._lib.load()
    return audio_metrics_amd


def _run(am, ref, cand, k):
    dev = torch.device("cuda:0")
    a, b = am.AudioMetricsData(True), am.AudioMetricsData(True)
    a.add(torch.as_tensor(ref).to(dev))
    b.add(torch.as_tensor(cand).to(dev))
    res = am.prdc(a, b, k)
    col, rany, rcov = am.hip_ops.prdc_counts(a.embeddings, b.embeddings, a.get_radii(k), b.get_radii(k))
    return res, dict(r_ref=a.get_radii(k).cpu().numpy(), r_cand=b.get_radii(k).cpu().numpy(), col_count=col.cpu().numpy(),
                     row_any=rany.cpu().numpy().astype(bool), row_cover=rcov.cpu().numpy().astype(bool))


@pytest.mark.parametrize("name", list(gi.PRDC_LARGE_CASES))
def test_production_filter_kernels_vs_reference_and_c_model(am, golden, name):
    from oracle import exact
    g = golden("prdc_large")
    kind, seed, nr, nc, d, k = gi.PRDC_LARGE_CASES[name]
    ops = am.hip_ops
    # default path selection must put these sizes on the wide f16 filter kernels
    assert ops.knn_path(nr, nr, d, k) == 3 and ops.knn_path(nc, nc, d, k) == 3 and ops.prdc_path(nr, nc, d) == 3
    ref, cand = gi.pair(kind, seed, nr, nc, d)
    ops.filter_stats_enable("cuda:0", True)
    res, got = _run(am, ref, cand, k)
    stats = ops.filter_stats_read("cuda:0")
    ops.filter_stats_enable("cuda:0", False)
    assert stats["knn_calls"] == 2 and stats["prdc_calls"] == 2            # (am.prdc and the direct prdc_counts call)
    assert stats["prdc_fallback_calls"] == 0 and stats["knn_fallback_rows"] == 0, stats

    # (a) the reference's own outputs
    np.testing.assert_allclose(got["r_ref"], g[f"{name}/r_ref"], rtol=3e-5, atol=1e-6)
    np.testing.assert_allclose(got["r_cand"], g[f"{name}/r_cand"], rtol=3e-5, atol=1e-6)
    inside = int(g[f"{name}/col_count"].astype(np.int64).sum())
    flips = int(np.abs(got["col_count"].astype(np.int64) - g[f"{name}/col_count"]).sum())
    assert flips <= max(1, REL * inside), (flips, inside)
    assert int((got["row_any"] != g[f"{name}/row_any"]).sum()) <= max(1, REL * nr)
    assert int((got["row_cover"] != g[f"{name}/row_cover"]).sum()) <= max(1, REL * nr)
    for key in ("precision", "recall", "density", "coverage"):
        want = float(g[f"{name}/{key}"])
        assert abs(res[key] - want) <= max(REL * abs(want), 1.0 / min(nr, nc)), (key, res[key], want)

    # (b) bit-exact against the C model of the device arithmetic
    _, aux = exact.prdc(ref, cand, k)
    assert np.array_equal(got["r_ref"].view(np.uint32), aux["r_ref"].view(np.uint32))
    assert np.array_equal(got["r_cand"].view(np.uint32), aux["r_cand"].view(np.uint32))
    assert np.array_equal(got["col_count"], aux["col_count"])
    assert np.array_equal(got["row_any"], aux["row_any"].astype(bool))
    assert np.array_equal(got["row_cover"], aux["row_min"] < aux["r_ref"])


@pytest.mark.parametrize("kind,k", [("randn", 5), ("clap", 10)])
def test_bench_size_result_vs_blocked_oracle(am, golden, kind, k):
    """bench.py's workload (2 x 100k x 512, numpy-seeded) through the single-GPU evaluate; PRDC against the values
    oracle.prdc_blocked produced for the same sets in the build container."""
    from audio_metrics_amd.distributed import evaluate_sharded
    g = golden("bench_prdc")
    n, d = 100000, 512
    ref, cand = gi.bench_pair(kind, n, d)
    dev = torch.device("cuda:0")
    res = evaluate_sharded(torch.as_tensor(ref).to(dev), torch.as_tensor(cand).to(dev), metrics=("prdc",), nearest_k=k)
    for key in ("precision", "recall", "density", "coverage"):
        want = float(g[f"{kind}_k{k}/{key}"])
        assert abs(res[key] - want) <= max(REL * abs(want), 5.0 / n), (key, res[key], want)


def test_bench_size_float64_rows_vs_the_reference_in_float64(am, golden):
    """bench.py's variant `pca64_f64`: 2 x 100 000 float64 rows of width 64 (what n_pca = 64 hands on).  FAD and the kernel
    distance against the reference's own float64 outputs, PRDC against oracle.prdc_blocked in float64 (make_goldens.py
    gen_bench) - through the *_f64 entry points, the radii and the membership counts on the f16 filter routes."""
    from audio_metrics_amd import hip_ops as ops
    from audio_metrics_amd.distributed import evaluate_sharded
    g = golden("bench_prdc")
    n, d, k = 100000, 64, 5
    ref, cand = gi.pair64("randn", gi.BENCH_SEED, n, n, d)
    dev = torch.device("cuda:0")
    ops.filter_stats_enable(dev, True)
    ops.filter_stats_read(dev)
    res = evaluate_sharded(torch.as_tensor(ref).to(dev), torch.as_tensor(cand).to(dev), metrics=("fad", "kd", "prdc"), nearest_k=k)
    s = ops.filter_stats_read(dev)
    ops.filter_stats_enable(dev, False)
    assert s["knn_calls"] == 2 and s["prdc_calls"] == 1 and s["knn_fallback_rows"] == 0 and s["prdc_fallback_calls"] == 0, s
    for key in ("precision", "recall", "density", "coverage"):
        want = float(g[f"randn_d64_f64_k{k}/{key}"])
        assert abs(res[key] - want) <= 2.0 / n, (key, res[key], want)          # float64 on both sides: at most a pair or two on a tie
    assert abs(res["fad"] - float(g["randn_d64_f64/fad"])) <= 1e-7 * abs(res["fad"])
    assert abs(res["kernel_distance_mean"] - float(g["randn_d64_f64/kernel_distance_mean"])) <= 1e-10
    assert abs(res["kernel_distance_std"] - float(g["randn_d64_f64/kernel_distance_std"])) <= 1e-10
