"""The plain-C model of the device arithmetic (oracle/exact_c) against the
reference's golden PRDC outputs: same algorithm, different f32 summation order,
so radii agree to f32 rounding and membership counts to a handful of flips."""
import numpy as np
import pytest

import inputs as gi
from oracle import exact


@pytest.mark.parametrize("name", list(gi.PRDC_CASES))
def test_exact_model_matches_reference_goldens(golden, name):
    g = golden("prdc")
    kind, seed, nr, nc, d, k = gi.PRDC_CASES[name]
    ref, cand = gi.pair(kind, seed, nr, nc, d)
    res, aux = exact.prdc(ref, cand, k)
    np.testing.assert_allclose(aux["r_ref"], g[f"{name}/r_ref"], rtol=3e-5, atol=1e-6)
    np.testing.assert_allclose(aux["r_cand"], g[f"{name}/r_cand"], rtol=3e-5, atol=1e-6)
    flips = int(np.abs(aux["col_count"].astype(np.int64) - g[f"{name}/col_count"]).sum())
    inside = int(g[f"{name}/col_count"].sum())
    assert flips <= max(3, 2e-4 * inside), (flips, inside)
    assert int((aux["row_any"] != g[f"{name}/row_any"]).sum()) <= 2
    np.testing.assert_allclose(aux["row_min"], g[f"{name}/row_min"], rtol=3e-5, atol=1e-6)
    for key in ("precision", "recall", "density", "coverage"):
        assert abs(res[key] - float(g[f"{name}/{key}"])) <= max(1e-4 * abs(float(g[f"{name}/{key}"])), 2.0 / min(nr, nc)), key
    print(name, "flips", flips, "of", inside)


def test_threshold_is_exact_boundary():
    rng = np.random.default_rng(3)
    r = np.concatenate([rng.uniform(1e-3, 50, 4000).astype(np.float32),
                        np.array([0.0, 1.0, 2.0, 1e-20, 3.4e19, np.inf], dtype=np.float32)])
    t = exact.threshold(r)
    fin = np.isfinite(r) & (r > 0)
    below = np.nextafter(t[fin], np.float32(0))
    assert np.all(np.sqrt(t[fin]) >= r[fin])
    assert np.all(np.sqrt(below) < r[fin])
    assert t[r == 0].tolist() == [0.0] and np.isinf(t[-1])


def test_sqnorm_close_to_f64():
    x = gi.randn(5, 300, 515)          # D not a multiple of 4
    np.testing.assert_allclose(exact.sqnorm(x), (x.astype(np.float64) ** 2).sum(1), rtol=2e-6)
