"""float64 rows keep float64 arithmetic end to end (round 5): the reference computes every stage in the dtype of the rows it
is given (data.py:39-44, prdc.py:12-13,34-48, kd.py:112-116), and float64 rows are what its PCA projection hands on
(projection.py:20-21) - the path of every reference test and of examples/2_musdb.py.

Fixtures: tests/golden/f64.npz, written by make_goldens.py gen_f64() from the reference's own functions on float64
inputs, two of them 20 000-row sets behind the reference's own IncrementalPCA with n_pca = 8 / 64.
Tolerances: radii 1e-12 relative to the row norms' scale (the reference's BLAS and these kernels sum |x|^2 + |y|^2 - 2 x.y
in different orders: ~1e-16 of the NORMS, so the error relative to a small distance is larger by norm^2 / distance^2);
membership counts and row flags EXACTLY; kernel distance 1e-9 of the kernel scale."""
import warnings

import numpy as np
import pytest
import torch

import inputs as gi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def am():
    import audio_metrics_amd
    assert torch.cuda.is_available(), "these tests need the MI355X"
    audio_metrics_amd._lib.load()
    return audio_metrics_amd


def dev(x):
    return torch.as_tensor(x).to("cuda:0")


def projected(am, g, name, ref, cand):
    """The rows a case's metrics see: the inputs themselves, or their projection with the REFERENCE's fitted components
    (am_project_rows_f64) so that the kernels are compared on the reference's own rows up to the rounding of one product."""
    if f"{name}/components" not in g.files:
        return dev(ref), dev(cand)
    comp, mean = dev(g[f"{name}/components"]), dev(g[f"{name}/pca_mean"])
    ops = am.hip_ops
    yr, yc = ops.project(dev(ref), mean, comp), ops.project(dev(cand), mean, comp)
    assert yr.dtype == torch.float64 and yr.shape == (ref.shape[0], comp.shape[0])
    np.testing.assert_allclose(yr[:64].cpu().numpy(), g[f"{name}/proj_ref_head"], rtol=0, atol=1e-13 * float(np.abs(ref).max()) * ref.shape[1])
    return yr, yc


@pytest.mark.parametrize("name", list(gi.F64_CASES))
def test_f64_rows_vs_reference(am, golden, name):
    g = golden("f64")
    kind, seed, nr, nc, d, n_pca, k = gi.F64_CASES[name]
    ref, cand = gi.pair64(kind, seed, nr, nc, d)
    yr, yc = projected(am, g, name, ref, cand)
    with warnings.catch_warnings():
        warnings.simplefilter("error")                       # the float64 -> float32 narrowing warning is gone
        a, b = am.AudioMetricsData(True), am.AudioMetricsData(True)
        a.add(yr)
        b.add(yc)
    assert a.embeddings.dtype == torch.float64 and b.embeddings.dtype == torch.float64
    # statistics (unchanged since round 4: am_stats_f64)
    np.testing.assert_allclose(a.mean.cpu().numpy(), g[f"{name}/mean_ref"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(a.cov.cpu().numpy(), g[f"{name}/cov_ref"], rtol=1e-10, atol=1e-13)
    # radii: float64, within 1e-12 of the reference's torch.cdist / kthvalue in units of the rows' scale
    r_ref, r_cand = a.get_radii(k), b.get_radii(k)
    assert r_ref.dtype == torch.float64 and r_cand.dtype == torch.float64
    scale_r = float(torch.linalg.norm(yr, dim=1).max()), float(torch.linalg.norm(yc, dim=1).max())
    for mine, key, sc in ((r_ref, "r_ref", scale_r[0]), (r_cand, "r_cand", scale_r[1])):
        want = g[f"{name}/{key}"]
        err = np.abs(mine.cpu().numpy() - want)
        # d = sqrt(d2), d2 carries ~1e-16 norm^2 of summation-order noise: |delta d| ~ 1e-16 norm^2 / (2 d)
        bound = 1e-12 * sc + 4e-16 * sc * sc / np.maximum(want, 1e-300)
        assert (err <= bound).all(), (key, float(err.max()), float((err / bound).max()))
    # membership counts and flags: exactly the reference's
    ops = am.hip_ops
    col, rany, rcov, rmin = ops.prdc_counts(a.embeddings, b.embeddings, r_ref, r_cand, want_min=True)
    assert rmin.dtype == torch.float64
    assert np.array_equal(col.cpu().numpy(), g[f"{name}/col_count"])
    assert np.array_equal(rany.cpu().numpy().astype(bool), g[f"{name}/row_any"])
    assert np.array_equal(rcov.cpu().numpy().astype(bool), g[f"{name}/row_cover"])
    want_min = g[f"{name}/row_min_head"]
    err_min = np.abs(rmin[:len(want_min)].sqrt().cpu().numpy() - want_min)
    assert (err_min <= 1e-12 * scale_r[0] + 4e-16 * scale_r[0] ** 2 / np.maximum(want_min, 1e-300)).all(), float(err_min.max())
    res = am.prdc(a, b, k)
    for key in ("precision", "recall", "density", "coverage"):
        assert res[key] == float(g[f"{name}/{key}"]), (key, res[key], float(g[f"{name}/{key}"]))
    # kernel distance (candidate = features_1, audio_metrics.py:260): per-subset values, mean and std
    m = 1000 if 1000 < min(nr, nc) else max(1, min(nr, nc) // 2)
    from audio_metrics_amd.metrics.kd import device_subset_indices
    i1, i2 = device_subset_indices(nc, nr, 100, m, 1234, yc.device)
    mmds = ops.kd_poly(yc, yr, i1, i2, 1.0 / yr.shape[1], 1.0, 3).cpu().numpy()
    want = g[f"{name}/mmds"]
    kscale = float(((yr.square().sum(1).max() / yr.shape[1] + 1.0) ** 3).item())       # size of a kernel value
    np.testing.assert_allclose(mmds, want, rtol=0, atol=1e-12 * kscale)
    kd = am.kernel_distance(b, a)
    assert abs(kd["kernel_distance_mean"] - float(g[f"{name}/kd_mean"])) <= 1e-12 * kscale
    assert abs(kd["kernel_distance_std"] - float(g[f"{name}/kd_std"])) <= 1e-11 * kscale
    if f"{name}/rbf_mean" in g.files:
        rr = am.kid_features_to_metric(yc, yr, kernel_type="rbf", kid_sigma=3.0)
        assert abs(rr["kernel_distance_mean"] - float(g[f"{name}/rbf_mean"])) <= 1e-12
        assert abs(rr["kernel_distance_std"] - float(g[f"{name}/rbf_std"])) <= 1e-12
    fad = am.frechet_distance(b, a)
    want_fad = float(g[f"{name}/fad"])
    assert abs(fad - want_fad) <= 1e-7 * max(abs(want_fad), float(torch.trace(a.cov).item()))


@pytest.mark.parametrize("name", ["pca8_20000_k5", "pca64_20000_k10"])
def test_own_projection_then_f64_metrics(am, golden, name):
    """The whole n_pca chain in this build - own eigensolver, own projection, f64 k-NN / membership kernels - against the
    reference's values for the same inputs.  scikit-learn's SVD and the Jacobi solver agree on the components to rounding,
    so the projected rows, and with them radii and counts, agree far inside the north star's 1e-4."""
    g = golden("f64")
    kind, seed, nr, nc, d, n_pca, k = gi.F64_CASES[name]
    ref, cand = gi.pair64(kind, seed, nr, nc, d)
    pca = am.IncrementalPCA(n_components=n_pca)
    pca.partial_fit(dev(ref))
    comp = pca.components_.cpu().numpy()
    np.testing.assert_allclose(comp, g[f"{name}/components"], rtol=0, atol=1e-9)
    a, b = am.AudioMetricsData(True), am.AudioMetricsData(True)
    a.add(pca.transform(dev(ref)))
    b.add(pca.transform(dev(cand)))
    assert a.embeddings.dtype == torch.float64
    np.testing.assert_allclose(a.get_radii(k).cpu().numpy(), g[f"{name}/r_ref"], rtol=1e-9, atol=1e-12)
    res = am.prdc(a, b, k)
    for key in ("precision", "recall", "density", "coverage"):
        assert abs(res[key] - float(g[f"{name}/{key}"])) <= 2.0 / min(nr, nc), (key, res[key])
    kd = am.kernel_distance(b, a)
    assert abs(kd["kernel_distance_mean"] - float(g[f"{name}/kd_mean"])) <= 1e-8


def test_f64_store_promotion_state_and_mixed_sets(am):
    """A float32 store that receives float64 rows becomes float64 (torch.cat's promotion of the reference's store,
    data.py:68-72); state files keep the dtype; float64 against float32 sets promote like numpy / torch do."""
    rng = np.random.default_rng(5)
    x32, x64 = rng.standard_normal((700, 12)).astype(np.float32), rng.standard_normal((600, 12))
    d = am.AudioMetricsData(True)
    d.add(dev(x32))
    assert d.embeddings.dtype == torch.float32
    d.add(dev(x64))
    assert d.embeddings.dtype == torch.float64 and d.embeddings.shape == (1300, 12)
    np.testing.assert_array_equal(d.embeddings.cpu().numpy(), np.concatenate([x32.astype(np.float64), x64]))
    d.add(dev(x32[:5]))                                       # float32 rows into the float64 store: no narrowing back
    assert d.embeddings.dtype == torch.float64 and d.embeddings.shape[0] == 1305
    r = d.get_radii(3)
    assert r.dtype == torch.float64
    full = np.concatenate([x32.astype(np.float64), x64, x32[:5].astype(np.float64)])
    want = torch.kthvalue(torch.cdist(torch.as_tensor(full), torch.as_tensor(full)), 4, dim=-1)[0].numpy()
    # (duplicated rows: the self / duplicate distances are rounding noise of either side - compare above that level)
    np.testing.assert_allclose(r.cpu().numpy(), want, rtol=1e-9, atol=1e-6)
    state = d.serialize()
    assert state["embeddings"].dtype == torch.float64 and state["radii"]["radii_3"].dtype == torch.float64
    back = am.AudioMetricsData.deserialize(state)
    assert back.embeddings.dtype == torch.float64 and back.radii["radii_3"].dtype == torch.float64
    assert torch.equal(back.embeddings, d.embeddings)
    # mixed dtypes at the operator boundary
    ops = am.hip_ops
    r32 = ops.knn_radii(dev(x32), 3)
    assert r32.dtype == torch.float32
    col, rany, rcov = ops.prdc_counts(dev(x32), dev(x64), r32, ops.knn_radii(dev(x64), 3))
    dist = torch.cdist(torch.as_tensor(x32.astype(np.float64)), torch.as_tensor(x64))
    want_col = (dist < r32.cpu().double()[:, None]).sum(0).numpy()
    assert np.abs(col.cpu().numpy() - want_col).sum() <= 1


def test_f64_select_path_and_errors(am):
    """k + 1 > 32 (distance blocks + radix select) against torch, and the error behaviour of the f64 entry points."""
    rng = np.random.default_rng(9)
    x = rng.standard_normal((900, 17))
    ops = am.hip_ops
    for k in (32, 100, 898):
        r = ops.knn_radii(dev(x), k).cpu().numpy()
        want = torch.kthvalue(torch.cdist(torch.as_tensor(x), torch.as_tensor(x)), k + 1, dim=-1)[0].numpy()
        np.testing.assert_allclose(r, want, rtol=1e-12, atol=1e-12)
    with pytest.raises(am._lib.HipLibraryError):
        ops.knn_radii(dev(x), 900)                            # k + 1 > rows: torch.kthvalue would raise too
    with pytest.raises(ValueError):
        ops.knn_radii(dev(x), 3, columns=dev(rng.standard_normal((10, 16))))
    # a NaN row is nobody's neighbour and has an infinite radius (torch carries NaN; see am_common.h clamp0)
    y = x.copy()
    y[5, 3] = np.nan
    r = ops.knn_radii(dev(y), 4).cpu().numpy()
    assert np.isinf(r[5]) and np.isfinite(np.delete(r, 5)).all()


@pytest.mark.parametrize("rows,dim,k,kind", [(30000, 48, 5, "randn"), (20000, 8, 10, "randn"), (16500, 130, 3, "unit"),
                                              (30000, 48, 5, "block"), (20000, 64, 5, "tiny")])
def test_f64_radii_through_the_f16_filter_sweep(am, rows, dim, k, kind):
    """A large float64 set against itself takes the float32 path's f16 filter sweep for its CANDIDATES (on a rounded copy) and
    evaluates / selects them in f64 (csrc/pairwise_fast.h: knn_fast_select64_kernel); rows the sweep cannot serve - a block of
    identical rows, operands the f16 scaling cannot hold - send the call to the general f64 kernels behind a device flag.
    Against those general kernels (the same rows against a COPY of the set, which never takes the filter route): the same
    radii to the rounding of two different summation orders."""
    rng = np.random.default_rng(rows + dim)
    x = rng.standard_normal((rows, dim))
    if kind == "unit":
        x /= np.linalg.norm(x, axis=1, keepdims=True)
    elif kind == "block":
        x[5000:8000] = x[5000]                                # 3000 identical rows: taken out of the sweep -> fallback
    elif kind == "tiny":
        x *= 1e-30                                            # squares underflow in float32: the rounded copy cannot be scaled
    ops = am.hip_ops
    xd = dev(x)
    assert am._lib.load().am_knn_f64_workspace_bytes(rows, rows, dim, k) > am._lib.load().am_knn_f64_workspace_bytes(rows, rows + 1, dim, k)
    ops.filter_stats_enable("cuda:0", True)
    ops.filter_stats_read("cuda:0")
    r = ops.knn_radii(xd, k)
    s = ops.filter_stats_read("cuda:0")
    ops.filter_stats_enable("cuda:0", False)
    general = ops.knn_radii(xd, k, columns=xd.clone())
    assert r.dtype == torch.float64
    scale = float(np.linalg.norm(x, axis=1).max())
    err = (r - general).abs().cpu().numpy()
    bound = 1e-12 * scale + 4e-16 * scale * scale / np.maximum(general.cpu().numpy(), 1e-300)
    assert (err <= bound).all(), float((err / bound).max())
    assert s["knn_calls"] == 1                                # the filter route ran ...
    if kind in ("block", "tiny"):
        assert s["knn_fallback_rows"] == rows, s               # ... and handed the call to the general kernels
    else:
        assert s["knn_fallback_rows"] == 0 and s["knn_verified_pairs"] > rows * (k + 1), s


@pytest.mark.parametrize("nr,nc,dim,k,kind", [(30000, 28000, 48, 5, "randn"), (20000, 20000, 8, 10, "randn"), (16500, 16400, 130, 3, "unit"),
                                               (30000, 9000, 48, 5, "shared"), (20000, 20000, 64, 5, "tiny")])
def test_f64_membership_counts_through_the_f16_filter(am, nr, nc, dim, k, kind):
    """Large float64 problems: the float32 path's f16 filter pass on rounded copies decides what its (widened) band allows, the
    pairs inside the band are evaluated in f64 against the f64 thresholds (csrc/pairwise_fast.h: cross_verify_regions64_kernel);
    sets the filter cannot serve - shared tight clusters, operands that cannot be scaled - go to the general f64 kernel behind
    the route's fail flag.  Against the general kernel itself (asking for the row minimum keeps a call on it): counts and flags
    EXACTLY equal."""
    rng = np.random.default_rng(nr + dim)
    if kind == "shared":                                        # both sets around the same 40 tight clusters
        centres = rng.standard_normal((40, dim))
        x = centres[rng.integers(0, 40, nr)] + 1e-3 * rng.standard_normal((nr, dim))
        y = centres[rng.integers(0, 40, nc)] + 1e-3 * rng.standard_normal((nc, dim))
    else:
        x, y = rng.standard_normal((nr, dim)), rng.standard_normal((nc, dim)) * 1.05 + 0.05
    if kind == "unit":
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        y /= np.linalg.norm(y, axis=1, keepdims=True)
    elif kind == "tiny":
        x, y = x * 1e-30, y * 1e-30
    ops = am.hip_ops
    xd, yd = dev(x), dev(y)
    rx, ry = ops.knn_radii(xd, k), ops.knn_radii(yd, k)
    ops.filter_stats_enable("cuda:0", True)
    ops.filter_stats_read("cuda:0")
    col, rany, rcov = ops.prdc_counts(xd, yd, rx, ry)
    s = ops.filter_stats_read("cuda:0")
    ops.filter_stats_enable("cuda:0", False)
    want = ops.prdc_counts(xd, yd, rx, ry, want_min=True)
    assert torch.equal(col, want[0]) and torch.equal(rany, want[1]) and torch.equal(rcov, want[2])
    assert s["prdc_calls"] == 1                               # the filter route ran ...
    assert s["prdc_fallback_calls"] == (1 if kind in ("shared", "tiny") else 0), s
    print(f"{kind} {nr} x {nc} x {dim}: inside pairs {int(col.sum())}, rows with a witness {int(rany.sum())}, covered {int(rcov.sum())}, "
          f"queued {s['prdc_queued']}")


@pytest.mark.parametrize("bad", [float("nan"), float("inf"), 1e200, 1e-200])
def test_f64_filter_routes_with_values_float32_cannot_hold(am, bad):
    """A NaN / Inf / 1e200 element makes the float32-rounded copy unscalable: the filter routes hand the call to the general f64
    kernels on the device (a NaN row is nobody's neighbour and has an infinite radius); 1e-200 rounds to zero in the copy and is
    evaluated exactly in f64."""
    rng = np.random.default_rng(3)
    x, y = rng.standard_normal((20000, 32)), rng.standard_normal((18000, 32))
    x[5, 3] = bad
    y[7, 1] = bad
    ops = am.hip_ops
    xd, yd = dev(x), dev(y)
    r, general = ops.knn_radii(xd, 4), ops.knn_radii(xd, 4, columns=xd.clone())
    fin = torch.isfinite(general)
    assert torch.equal(torch.isfinite(r), fin) and bool(((r - general).abs()[fin] <= 1e-9 * general[fin]).all())
    r2 = ops.knn_radii(yd, 4)
    got, want = ops.prdc_counts(xd, yd, r, r2), ops.prdc_counts(xd, yd, r, r2, want_min=True)
    assert all(torch.equal(a, b) for a, b in zip(got, want[:3]))


def test_f64_thresholds_of_radii_whose_square_underflows(am):
    """ADVICE r5 (pairwise_f64.hip: threshold_of_radius64): the strict `sqrt(d2) < R` of the reference (prdc.py:34-48) becomes
    `d2 < T(R)`, T(R) = min{t : sqrt_rn(t) >= R}.  For R below ~1.5e-154 R * R is subnormal or zero: T(R) is then found by
    bisection over the bit patterns.  Rows on an axis at 0, 1e-160, 3e-160, 1e-170, 1e-150: squared distances to the origin
    1e-320 (subnormal), 9e-320, 0 (1e-340 underflows, as it does in torch), 1e-300 - tested against radii on both sides of
    every one of them, the expectation written down from the definition."""
    ops = am.hip_ops
    ref = dev(np.array([[0.0, 0.0]], dtype=np.float64))
    axis = np.array([0.0, 1e-160, 3e-160, 1e-170, 1e-150])
    cand = dev(np.stack([axis, np.zeros_like(axis)], axis=1))
    r_cand = dev(np.full(len(axis), 1e-3))
    for radius in (2e-160, 0.5e-160, 1e-160, 3.5e-160, 1e-165, 5e-151, 2e-150, 1e-200, 4.9e-324):
        d2 = axis * axis                                                     # what the device's |x|^2 + |y|^2 - 2 <x, y> gives here
        want = (np.sqrt(d2) < radius).astype(np.int32)
        col, rany, rcov = ops.prdc_counts(ref, cand, dev(np.array([radius])), r_cand)
        assert col.cpu().numpy().tolist() == want.tolist(), (radius, col.cpu().numpy(), want)
