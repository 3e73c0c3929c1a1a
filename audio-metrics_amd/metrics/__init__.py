"""Metric functions of the hot path, one module per reference module
(src/audio_metrics/metrics/{fad,kd,prdc,apa}.py).  Submodules are kept importable by
name (``metrics.apa`` is the module, as in the reference), so nothing is re-exported here."""
from . import apa, fad, kd, prdc   # noqa: F401
