"""metrics/fad.py: the RuntimeWarning for sets with no more rows than dimensions (the build returns the
dust-free Frechet distance where the reference's float32 value is not defined to 1e-4 - reference fad.py:28-31 fed by the f32
torch.cov of data.py:44 - and must not do so silently).  The numbers the message quotes are in profiles/r6/fad_f32_probe.txt;
the -m gpu twin (test_gpu_parity.py::test_fad_warns_when_rows_do_not_exceed_dimensions) goes through frechet_distance."""
import warnings

import pytest
import torch

from audio_metrics_amd.metrics import fad


def test_warns_only_where_the_reference_is_undefined():
    with pytest.warns(RuntimeWarning, match="no more rows than dimensions"):
        fad.warn_if_rank_deficient(100, 4096, 512)
    with pytest.warns(RuntimeWarning, match=r"\(512 rows, 512 dimensions\)"):
        fad.warn_if_rank_deficient(5000, 512, 512, torch.float32)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        fad.warn_if_rank_deficient(513, 4096, 512)                       # full rank
        fad.warn_if_rank_deficient(100, 100, 512, torch.float64)         # float64 rows carry no dust
        fad.warn_if_rank_deficient(None, 100, 512)                       # statistics without a row count (raw mean / cov call)
        fad.warn_if_rank_deficient(1, 1, 8)                              # the reference's zero-covariance case
