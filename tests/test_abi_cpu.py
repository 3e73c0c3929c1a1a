"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and
exports every function include/audio_metrics_hip.h declares (no compute calls -
there is no GPU here), the ctypes table matches the header, and the host logic
that does not touch the device behaves like the reference."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "audio_metrics_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(am_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import audio_metrics_amd as am
    am.build_library()
    lib = am._lib.load()
    names = header_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
    assert sorted(am._lib.SIGNATURES) == names
    assert lib.am_version().decode().startswith("audio_metrics_hip")
    assert lib.am_status_string(-3).decode() == "unsupported nearest_k"


def test_host_side_entry_points():
    import audio_metrics_amd as am
    lib = am._lib.load()
    # scalar APA combination (apa.py:22-32) is host arithmetic
    import oracle
    for t in [(1.0, 2.0, 3.0), (5.0, 1.0, 3.0), (-1.0, -1.0, -1.0), (2.0, 2.0, 0.0), (0.3, 0.9, 0.1)]:
        assert lib.am_apa_f64(*t) == oracle.apa_from_distances(*t)
    # workspace queries are pure host functions
    assert lib.am_knn_workspace_bytes(100000, 100000, 512, 5) > 0
    assert lib.am_knn_workspace_bytes(10, 10, 16, 99) == 0
    assert lib.am_stats_workspace_bytes(100000, 512) > 0
    assert lib.am_frechet_workspace_bytes(512) >= 6 * 512 * 512 * 8
    assert lib.am_kd_workspace_bytes(100, 1000) > 0
    assert lib.am_prdc_workspace_bytes(1000, 2000, 128) > 0


def test_deprecated_kd_workspace_query_is_answered_with_an_error_code():
    """a workspace sized with the D-less am_kd_workspace_bytes (deprecated) is too small for the split-f16 form
    that m >= 512, 128 <= D <= 8192, degree 3 take.  The entry point must answer AM_ERR_WORKSPACE - before any device work:
    the check is host arithmetic on the sizes, so it runs here - and say which size it wanted; at shapes of the f32 form the
    old query is still sufficient."""
    import ctypes
    import audio_metrics_amd as am
    lib = am._lib.load()
    S, m, D, n = 4, 1000, 512, 5000
    old, new = lib.am_kd_workspace_bytes(S, m), lib.am_kd_poly_workspace_bytes(S, m, D)
    assert 0 < old < new
    assert lib.am_kd_poly_workspace_bytes(S, 200, D) == lib.am_kd_workspace_bytes(S, 200)        # small subsets: the f32 form
    fake = ctypes.c_void_p(0x10000)                                   # 16-byte aligned, never dereferenced: the call stops at the size check
    rc = lib.am_kd_poly_f32(fake, n, D, fake, n, D, D, fake, fake, S, m, 1.0 / D, 1.0, 3, fake, fake, old, None)
    assert rc == -4 and lib.am_status_string(rc).decode() == "workspace too small"
    msg = lib.am_last_error().decode()
    assert str(new) in msg and "am_kd_poly_workspace_bytes" in msg, msg


def test_no_cpu_fallback():
    import audio_metrics_amd as am
    with pytest.raises(am._lib.HipLibraryError):
        am.hip_ops.knn_radii(torch.zeros(8, 4), 2)
    if not torch.cuda.is_available():
        with pytest.raises(am._lib.HipLibraryError):
            am.AudioMetricsData().add(np.zeros((4, 4), dtype=np.float32))


def test_kd_index_table_matches_reference_draws():
    from audio_metrics_amd.metrics.kd import subset_indices
    import oracle
    i1, i2 = subset_indices(100000, 100000, 3, 1000, 1234)
    o1, o2 = oracle.draw_subsets(100000, 100000, subsets=3)
    np.testing.assert_array_equal(i1, o1)
    np.testing.assert_array_equal(i2, o2)
    assert list(i1[0, :4]) == [57642, 95775, 28099, 5584]
