"""Accompaniment Prompt Adherence on the device.

Mirrors the public names of the reference's APA module (src/audio_metrics/metrics/apa.py:5-32):
``apa`` evaluates three Frechet distances with am_frechet_f64 and hands the three scalars to
am_apa_f64, which applies the reference's closed-form combination; ``apa_compute_d_x_xp`` lets the
caller cache the reference/anti-reference distance (audio_metrics.py:102-109 does so)."""
from .. import hip_ops as ops
from ..data import AudioMetricsData
from .fad import frechet_distance

__all__ = ["apa", "apa_compute_d_x_xp"]


def apa_compute_d_x_xp(reference: AudioMetricsData, anti_reference: AudioMetricsData):
    """Frechet distance between the matched and the mismatched reference sets."""
    return frechet_distance(reference, anti_reference)


def _apa(d_y_x, d_y_xp, d_x_xp):
    """Scalar combination of the three distances (computed by the library, f64)."""
    return ops.apa_scalar(d_y_x, d_y_xp, d_x_xp)


def apa(candidate: AudioMetricsData, reference: AudioMetricsData, anti_reference: AudioMetricsData,
        d_x_xp=None):
    """APA score of `candidate`; pass a cached `d_x_xp` to skip the third Frechet distance."""
    to_matched, to_mismatched = (frechet_distance(candidate, other) for other in (reference, anti_reference))
    between = apa_compute_d_x_xp(reference, anti_reference) if d_x_xp is None else d_x_xp
    return _apa(to_matched, to_mismatched, between)
