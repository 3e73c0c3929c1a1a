"""CPU oracle for the distribution-distance hot path (FAD / KD / PRDC / APA).

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped
product path: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and there only as the
checker / the timed CPU baseline.  The product package (``audio-metrics_amd``)
never imports this package and fails loudly when its HIP library is missing.

What it is: a restatement, in this repository's own words, of the reference's
algorithm for the hot path (SonyCSLParis/audio-metrics v1.0.4), using the same
third-party arithmetic the reference delegates to (torch CPU ops, numpy) and
the same dtype ladder, so that it reproduces the reference's values.  Each
function cites the reference ``file:line`` it follows (paths relative to the
reference checkout root).

Parity pin: the reference's own tests hold no known-answer vectors for
FAD/KD/PRDC/APA (only ``src/audio_metrics/tests/test_data.py:6-31`` pins the
Chan merge).  The oracle is therefore pinned against outputs of the reference
itself, generated in the build container by ``tests/golden/make_goldens.py``
(which imports the reference's hot-path modules) and committed as
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every one.

``oracle/exact_c/`` additionally holds a plain-C model of the *device* kernels'
f32 arithmetic order (k-ordered fmaf chains) used to assert bit-exact radii,
thresholds and membership counts for the PRDC kernels.
"""
from .stats import OracleData, batch_stats, chan_merge            # noqa: F401
from .fad import frechet_distance, frechet_from_stats             # noqa: F401
from .kd import kernel_distance, kid_from_features, draw_subsets  # noqa: F401
from .prdc import knn_radii, prdc, prdc_from_features, prdc_blocked  # noqa: F401
from .apa import apa, apa_from_distances                          # noqa: F401
