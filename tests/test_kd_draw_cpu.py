"""The library's restatement of numpy's Generator.choice (am_kd_draw_indices, host arithmetic: runs without a GPU)
against numpy itself - the draw order is part of the kernel-distance result (reference kd.py:176,185-186)."""
import numpy as np
import pytest

import audio_metrics_amd as am
from audio_metrics_amd.metrics import kd


@pytest.mark.parametrize("n1,n2,subsets,m,seed", [
    (100000, 100000, 100, 1000, 1234),      # BASELINE shape: Floyd's algorithm on both sets
    (40000, 35000, 20, 1000, 1234),         # n > 10000 and m > n / 50: numpy's tail-shuffle branch
    (10000, 10001, 9, 150, 5),              # either side of the 10000 threshold
    (50049, 50050, 5, 1000, 3),             # either side of m > n // 50
    (300, 257, 11, 128, 99),                # the reference's small-sample shrink (m = n_min // 2)
    (2, 3, 4, 1, 9), (7, 7, 3, 7, 0),       # degenerate sizes; m == n
    (100000, 50, 3, 50, 1),
])
def test_native_draw_equals_numpy(n1, n2, subsets, m, seed):
    am._lib.load()
    a1, a2 = kd.subset_indices_native(n1, n2, subsets, m, seed)
    b1, b2 = kd.subset_indices_numpy(n1, n2, subsets, m, seed)
    assert a1.dtype == np.int64 and a1.shape == (subsets, m)
    assert np.array_equal(a1, b1) and np.array_equal(a2, b2)


def test_numpy_version_is_one_the_restatement_was_checked_against():
    """am_kd_draw_indices restates numpy's Generator.choice (Floyd / tail shuffle on Lemire-bounded 32-bit draws of PCG64) as
    implemented in numpy 1.17 - 2.x.  A major version beyond that range must be re-checked (the equality tests above do it)
    before this bound is raised; at run time a mismatch switches to numpy's own calls with one warning."""
    major = int(np.__version__.split(".")[0])
    assert 1 <= major <= 2, np.__version__


def test_a_mismatch_falls_back_to_numpy_with_one_warning(monkeypatch):
    """numpy's call is the reference's definition of the subsets (kd.py:176,185-186): a numpy whose draws the restatement
    does not reproduce must not break kernel distance - the table then comes from numpy itself."""
    monkeypatch.setattr(kd, "_NATIVE_DRAW_OK", None)
    real = kd.subset_indices_native
    monkeypatch.setattr(kd, "subset_indices_native", lambda *a: tuple(x + 1 for x in real(*a)))
    want = kd.subset_indices_numpy(5000, 5000, 2, 100, 1)
    with pytest.warns(RuntimeWarning, match="does not reproduce numpy"):
        got = kd.subset_indices(5000, 5000, 2, 100, 1)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")                      # the second call is silent
        again = kd.subset_indices(5000, 5000, 2, 100, 1)
    assert np.array_equal(again[0], want[0])
    monkeypatch.setattr(kd, "_NATIVE_DRAW_OK", None)


def test_dispatch_uses_the_native_draw_and_matches_the_golden_first_draws():
    i1, i2 = kd.subset_indices(100000, 100000, 2, 1000, 1234)
    # first draws of default_rng(1234) at n = 100000
    assert i1[0, :8].tolist() == [57642, 95775, 28099, 5584, 88853, 71821, 71685, 94582]
    assert i2[0, :8].tolist() == [54893, 10548, 28828, 19561, 15832, 25262, 27971, 29690]


def test_bad_arguments_are_rejected():
    with pytest.raises(am._lib.HipLibraryError):
        kd.subset_indices_native(10, 10, 2, 11, 0)          # m > n
