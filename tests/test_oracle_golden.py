"""The oracle (CPU restatement) must reproduce the reference's own outputs.

Golden vectors were produced by tests/golden/make_goldens.py, which imports the
reference's hot-path modules (data.py, metrics/{fad,kd,prdc,apa}.py) unmodified.
Same torch/numpy versions here and there => the same library calls give the
same bits; tolerances below only allow for thread-count dependent BLAS blocking.
"""
import numpy as np
import pytest
import torch

import inputs as gi
import oracle

RT = dict(rtol=1e-9, atol=1e-12)


def _feed(x, splits, store=True):
    d = oracle.OracleData(store)
    t = torch.as_tensor(x)
    s = 0
    for b in splits:
        d.add(t[s:s + b])
        s += b
    return d


@pytest.mark.parametrize("name", list(gi.STATS_CASES))
def test_stats(golden, name):
    g = golden("stats")
    seed, d, splits = gi.STATS_CASES[name]
    x = gi.randn(seed, sum(splits), d, 1.3, 0.2)
    a = _feed(x, splits)
    assert a.n == int(g[f"{name}/n"])
    np.testing.assert_allclose(a.mean.numpy(), g[f"{name}/mean"], **RT)
    c = a.cov.numpy()
    if d <= 128:
        np.testing.assert_allclose(c, g[f"{name}/cov"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(np.trace(c), g[f"{name}/cov_trace"], rtol=1e-7)
    np.testing.assert_allclose(c[:16, :16], g[f"{name}/cov_block"], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(c[::37, ::41], g[f"{name}/cov_sample"], rtol=1e-6, atol=1e-8)
    a.recompute_stats()
    np.testing.assert_allclose(a.mean.numpy(), g[f"{name}/re_mean"], **RT)
    assert tuple(a.cov.shape) == tuple(g[f"{name}/re_cov_shape"])


def test_chan_merge_equals_oneshot():
    """Property pinned by the reference's tests/test_data.py:6-31 (1e-6)."""
    x = gi.randn(7, 1101, 8)
    a = _feed(x, [1, 100, 1000])
    b = _feed(x, [1101])
    np.testing.assert_allclose(a.mean.numpy(), b.mean.numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(a.cov.numpy(), b.cov.numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("name", list(gi.FAD_CASES))
def test_fad(golden, name):
    g = golden("fad")
    kind, seed, nr, nc, d = gi.FAD_CASES[name]
    ref, cand = gi.pair(kind, seed, nr, nc, d)
    a, b = _feed(cand, [nc], False), _feed(ref, [nr], False)
    fad = oracle.frechet_distance(a, b)
    scale = float(g[f"{name}/tr_sum"])
    # eigvals is LAPACK geev: allow thread-dependent rounding relative to the operands' magnitude
    assert abs(fad - float(g[f"{name}/fad"])) <= 1e-9 * scale + 1e-7 * abs(fad)
    assert abs((a.mean - b.mean).square().sum().item() - float(g[f"{name}/mean_sq"])) <= 1e-9 * scale


@pytest.mark.parametrize("name", list(gi.KD_CASES))
def test_kd(golden, name):
    g = golden("kd")
    kind, seed, n1, n2, d = gi.KD_CASES[name]
    f2, f1 = gi.pair(kind, seed, n2, n1, d)
    i1, i2 = oracle.draw_subsets(n1, n2)
    assert i1.shape == (100, int(g[f"{name}/m"]))
    np.testing.assert_array_equal(i1[0, :8], g[f"{name}/first_idx1"])
    np.testing.assert_array_equal(i2[0, :8], g[f"{name}/first_idx2"])
    res, mmds = oracle.kid_from_features(f1, f2, return_all=True)
    floor = 5e-7                                      # f32 noise floor of the reference itself
    np.testing.assert_allclose(mmds, g[f"{name}/mmds"], rtol=1e-5, atol=floor)
    assert abs(res["kernel_distance_mean"] - float(g[f"{name}/mean"])) <= 1e-5 * abs(float(g[f"{name}/mean"])) + floor
    assert abs(res["kernel_distance_std"] - float(g[f"{name}/std"])) <= 1e-4 * abs(float(g[f"{name}/std"])) + floor


@pytest.mark.parametrize("name", [n for n, c in gi.KD_CASES.items() if c[2] <= 3000])
def test_kd_rbf(golden, name):
    g = golden("kd")
    kind, seed, n1, n2, d = gi.KD_CASES[name]
    f2, f1 = gi.pair(kind, seed, n2, n1, d)
    for key in [k for k in g.files if k.startswith(f"{name}/rbf_") and k.endswith("/mean")]:
        sigma = float(key.split("rbf_")[1].split("/")[0])
        res = oracle.kid_from_features(f1, f2, kernel_type="rbf", sigma=sigma)
        assert abs(res["kernel_distance_mean"] - float(g[key])) <= 1e-9 + 1e-7 * abs(float(g[key]))
        assert abs(res["kernel_distance_std"] - float(g[key.replace("/mean", "/std")])) <= 1e-9 + 1e-6 * abs(float(g[key.replace("/mean", "/std")]))


def test_kd_first_draws_match_survey():
    """first draws of default_rng(1234) at n=100000."""
    i1, i2 = oracle.draw_subsets(100000, 100000, subsets=1)
    assert list(i1[0, :8]) == [57642, 95775, 28099, 5584, 88853, 71821, 71685, 94582]
    assert list(i2[0, :8]) == [54893, 10548, 28828, 19561, 15832, 25262, 27971, 29690]


@pytest.mark.parametrize("name", list(gi.PRDC_CASES))
def test_prdc(golden, name):
    g = golden("prdc")
    kind, seed, nr, nc, d, k = gi.PRDC_CASES[name]
    ref, cand = gi.pair(kind, seed, nr, nc, d)
    a, b = _feed(ref, [nr]), _feed(cand, [nc])
    res = oracle.prdc(a, b, k)
    np.testing.assert_allclose(a.radii[f"radii_{k}"].numpy(), g[f"{name}/r_ref"], rtol=2e-6)
    np.testing.assert_allclose(b.radii[f"radii_{k}"].numpy(), g[f"{name}/r_cand"], rtol=2e-6)
    for key in ("precision", "recall", "density", "coverage"):
        # one membership flip changes these by 1/N; allow two
        assert abs(res[key] - float(g[f"{name}/{key}"])) <= 2.0 / min(nr, nc) + 1e-12, key


def test_prdc_blocked_agrees():
    """Row blocks of the same torch calls against the N x N formulation: at most two membership flips (a flip moves a value
    by 1 / N; density by 1 / (k N))."""
    ref, cand = gi.pair("shifted", 42, 1000, 1000, 128)
    a, b = _feed(ref, [1000]), _feed(cand, [1000])
    full, fc = oracle.prdc_from_features(a.embeddings, b.embeddings, a.get_radii(5), b.get_radii(5), 5, return_counts=True)
    blk, bc = oracle.prdc_blocked(ref, cand, 5, block=256, return_counts=True)
    assert int((fc["col_count"] != bc["col_count"]).sum()) <= 2
    assert int((fc["row_any"] != bc["row_any"]).sum()) <= 2
    for key in full:
        assert abs(full[key] - blk[key]) <= 2.0 / 1000 + 1e-12


# Three of the six prdc_large cases - one per input pair, the k with more inside pairs where there is a choice (the whole CPU
# suite has to stay within minutes; all six ran with 0 flips when this test was written, and the GPU suite checks the device
# against all six): randn 33 000 x 128 k = 5, unit-norm 33 000 / 35 000 x 128 k = 10, unit-norm 40 000 x 512 k = 10 (1.4 M inside pairs)
@pytest.mark.parametrize("name", ["randn_33000_128_k5", "unit_33000_35000_128_k10", "unit_40000_512_k10"])
def test_prdc_blocked_vs_reference_large(golden, name):
    """Closes the fixture chain of the headline configuration: tests/golden/bench_prdc.npz - what
    bench.py's result_check and the 100k-row GPU tests compare with - is written by oracle.prdc_blocked, because the
    reference's N x N formulation needs 164 GB at 100 000 rows.  Here the SAME function runs on the inputs of
    prdc_large.npz, which the reference itself wrote (metrics/prdc.py:4-50 unmodified, 33 000 - 40 000 rows, the integer
    column counts and row flags stored): radii to f32 rounding, and the count of differing memberships asserted."""
    g = golden("prdc_large")
    kind, seed, nr, nc, d, k = gi.PRDC_LARGE_CASES[name]
    ref, cand = gi.pair(kind, seed, nr, nc, d)
    res, c = oracle.prdc_blocked(ref, cand, k, return_counts=True)
    np.testing.assert_allclose(c["r_ref"].numpy(), g[f"{name}/r_ref"], rtol=3e-6)
    np.testing.assert_allclose(c["r_cand"].numpy(), g[f"{name}/r_cand"], rtol=3e-6)
    want_col = g[f"{name}/col_count"].astype(np.int64)
    col_flips = int(np.abs(c["col_count"].numpy() - want_col).sum())
    any_flips = int((c["row_any"].numpy() != g[f"{name}/row_any"]).sum())
    cov_flips = int((c["row_cover"].numpy() != g[f"{name}/row_cover"]).sum())
    inside = int(want_col.sum())
    print(f"{name}: {inside} inside pairs, flips col {col_flips} any {any_flips} cover {cov_flips}")
    # a handful of last-bit flips among >= 1e5 inside pairs (cdist's blocking depends on the shape it is given)
    assert col_flips <= max(4, inside // 20000), (col_flips, inside)
    assert any_flips <= 2 and cov_flips <= 2
    for key in ("precision", "recall", "coverage"):
        assert abs(res[key] - float(g[f"{name}/{key}"])) <= 2.0 / min(nr, nc) + 1e-12, key
    assert abs(res["density"] - float(g[f"{name}/density"])) <= (col_flips + 0.5) / (k * nc), "density"


def test_apa(golden):
    g = golden("apa")
    for t, want in zip(g["table_in"], g["table_out"]):
        assert oracle.apa_from_distances(*t) == want
    d = 64
    ref, anti = gi.randn(51, 1500, d), gi.randn(52, 1500, d, 1.2, 0.3)
    for i in range(4):
        sc, sh = g[f"three_set_{i}/params"]
        cand = gi.randn(53 + i, 1200, d, sc, sh)
        a, b, c = (_feed(v, [len(v)], False) for v in (cand, ref, anti))
        assert abs(oracle.apa(a, b, c) - float(g[f"three_set_{i}/apa"])) <= 1e-7


# ---- float64 rows (round 5): the oracle's torch / numpy calls keep the dtype they are given, as the reference's do
@pytest.mark.parametrize("name", list(gi.F64_CASES))
def test_f64_rows(golden, name):
    g = golden("f64")
    kind, seed, nr, nc, d, n_pca, k = gi.F64_CASES[name]
    ref, cand = gi.pair64(kind, seed, nr, nc, d)
    assert ref.dtype == np.float64
    if n_pca:                               # project with the REFERENCE's fitted components (IncrementalPCA.transform in f64)
        comp, mean = g[f"{name}/components"], g[f"{name}/pca_mean"]
        ref, cand = (ref - mean) @ comp.T, (cand - mean) @ comp.T
        np.testing.assert_allclose(ref[:64], g[f"{name}/proj_ref_head"], rtol=0, atol=1e-12)
    a, b = _feed(ref, [nr]), _feed(cand, [nc])
    assert a.embeddings.dtype == torch.float64
    np.testing.assert_allclose(a.mean.numpy(), g[f"{name}/mean_ref"], rtol=0, atol=1e-12)
    res = oracle.prdc(a, b, k)
    assert a.radii[f"radii_{k}"].dtype == torch.float64
    np.testing.assert_allclose(a.radii[f"radii_{k}"].numpy(), g[f"{name}/r_ref"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(b.radii[f"radii_{k}"].numpy(), g[f"{name}/r_cand"], rtol=1e-9, atol=1e-9)
    for key in ("precision", "recall", "density", "coverage"):
        assert abs(res[key] - float(g[f"{name}/{key}"])) <= (0.0 if not n_pca else 1.0 / min(nr, nc)) + 1e-12, key
    kd, mmds = oracle.kid_from_features(cand, ref, return_all=True)
    np.testing.assert_allclose(mmds, g[f"{name}/mmds"], rtol=0, atol=1e-10)
    assert abs(kd["kernel_distance_mean"] - float(g[f"{name}/kd_mean"])) <= 1e-10
