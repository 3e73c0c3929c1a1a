"""The peak-based mix functions (host DSP in front of the embedder, reference mix_functions.py:209-250, registry :335-344)
against outputs of the reference's own functions (tests/golden/mix.npz, written by make_goldens.py mix): bit-identical,
f32 and f64, including the places where the reference's numpy scalars make it compute in f64."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def test_peak_mixers_bit_identical_to_the_reference():
    import make_goldens_inputs as mgi
    from audio_metrics_amd.mix_functions import MIX_FUNCTIONS
    g = np.load(os.path.join(HERE, "golden", "mix.npz"), allow_pickle=False)
    checked = 0
    for case, audio in mgi.mix_inputs():
        for name in ("PP", "P0", "P1", "P2"):
            key = f"{case}/{name}"
            if key not in g.files:
                continue
            got = MIX_FUNCTIONS[name](audio.copy(), sr=16000)
            assert got.dtype == g[key].dtype and np.array_equal(got, g[key]), key
            checked += 1
    assert checked >= 24


def test_loudness_mixers_are_rejected_at_construction():
    from audio_metrics_amd.mix_functions import resolve_mix_function
    with pytest.raises(ValueError):
        resolve_mix_function("L0", needed=True)
    with pytest.raises(ValueError):
        resolve_mix_function("nope")
    assert callable(resolve_mix_function("P0"))
    # the reference's default (None -> "L0") keeps AudioMetrics() constructible; the first mix raises the same message
    import numpy as np
    deferred = resolve_mix_function(None, needed=True)
    with pytest.raises(ValueError, match="BS.1770"):
        deferred(np.zeros((10, 2), dtype=np.float32), sr=16000)
